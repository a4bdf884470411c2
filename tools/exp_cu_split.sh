#!/bin/bash
# CU partitioning A/B (verdict r4 item 7): step time of the pipelined bench under hipExtStreamCreateWithCUMask splits.
run() { echo -n "$* : "; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-isolated "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run
run --opt cu_kp=4
run --opt cu_kp=8
run --opt cu_kp=8 --matcher-cus 8
run --opt cu_ss=28 --opt cu_kp=4
run --opt cu_ss=28 --opt cu_kp=4 --matcher-cus 4
run --opt cu_ss=24 --opt cu_kp=8 --matcher-cus 8
run --opt cu_ss=24 --opt cu_kp=8
run --matcher-cus 8
run --matcher-cus 16
run
