#!/bin/bash
# Runs on the GPU box: throughput + per-kernel times of the batched two-view verification alone.
# usage: bash tools/verify_job.sh [tag] [bench_verify.py arguments...]  -> gpurun_out/verify_<tag>.{json,txt}
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-x}
shift
O=/tmp/verify_$T
rm -rf $O; mkdir -p $O $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_verify.py "$@" > $R/gpurun_out/verify_$T.json 2> $O/err.txt || tail -20 $O/err.txt
cat $R/gpurun_out/verify_$T.json
rocprofv3 --kernel-trace -d $O -o r -- python $R/tools/bench_verify.py --check 0 --reps 3 "$@" > $O/log.txt 2>&1
python $R/tools/rocpd_stats.py $O/r_results.db > $R/gpurun_out/verify_kstat_$T.txt
head -24 $R/gpurun_out/verify_kstat_$T.txt | cut -c1-140
