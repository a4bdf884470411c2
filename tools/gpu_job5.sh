#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/stress_parity.py --n 16 2>&1 | tail -1
for i in 1 2; do
  python bench.py --no-extras --no-cpu-baseline --steps 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps',d['value'],'iso',d['scale_space_isolated']['frames_per_s'], d['phase_ms_per_step'])"
done
