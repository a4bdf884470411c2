#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export KSTAT_LINES=16
python tools/stress_parity.py --n 24 2>&1 | tail -1
for v in "$@"; do
  set -- $v
  echo "=== $v"
  bash tools/kstat.sh "$@" 2>&1 | grep -E "stream|ms per|deriv_second|front"
done
