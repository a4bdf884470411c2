#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export KSTAT_LINES=14
python tools/stress_parity.py --n 48 2>&1 | tail -2
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "kitti_every or ragged or pathological or full_hd or non_default" 2>&1 | tail -3
for v in "$@"; do
  set -- $v
  echo "=== $v"
  bash tools/kstat.sh "$@" 2>&1 | grep -E "det_stream|ms per|deriv_second"
done
