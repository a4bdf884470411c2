#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/latency_probe.py 2>&1 | tail -1
python tools/stress_parity.py --n 32 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "kitti_every or ragged or pathological or non_default or benchmark_mode" 2>&1 | tail -3
