#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export KSTAT_LINES=40
python tools/stress_parity.py --n 24 2>&1 | tail -1
bash tools/kstat.sh "$@" 2>&1 | grep -E "describe|refine|ms per"
