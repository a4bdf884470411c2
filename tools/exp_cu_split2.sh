#!/bin/bash
run() { echo -n "$* : "; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-isolated "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run --opt cu_kp=16
run --opt cu_kp=24
run --opt cu_kp=32
run --opt cu_ss=32
run --opt cu_ss=32 --opt cu_kp=32 --matcher-cus 32
run --opt cu_ss=20 --opt cu_kp=12 --matcher-cus 12
run --opt cu_ss=16 --opt cu_kp=16 --matcher-cus 16
run
