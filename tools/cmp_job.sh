R=${GRAFT_REPO_ROOT:-/root/repo}
O=/tmp/cmp; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o r -- python $R/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-isolated > $O/log.json 2> $O/err.txt
python $R/tools/rocpd_stats.py $O/r_results.db > $R/gpurun_out/cmp_kstat.txt
cp $O/log.json $R/gpurun_out/cmp_bench.json
head -12 $R/gpurun_out/cmp_kstat.txt | cut -c1-140
python -c "
import json
a=json.load(open('$O/log.json'))
print(a['value'])
for e in a['roofline_top'][:5]+[a['roofline_matcher']]:
    print(e['kernel'][:40], e['avg_launch_us'], e['launches'])
"
