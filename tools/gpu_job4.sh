#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/job4
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python bench.py --no-extras > $O/bench.log 2> $O/bench.err
echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/job4/bench.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('parity_checked'))
for e in d['roofline_top']: print(e['kernel'][:44], e['gpu_ms'], e['frac'], e.get('isolated'))
print(d['phase_ms_per_step'], d['scale_space_isolated']['frames_per_s'])
PY
