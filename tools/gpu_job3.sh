#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export KSTAT_LINES=16
for v in "$@"; do
  set -- $v
  echo "=== $v"
  bash tools/kstat.sh "$@" 2>&1 | grep -E "front2<2, 3|ms per"
done
