#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel statistics of bench.py's pipeline+register leg alone -> gpurun_out/register_kernel_stats.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=/tmp/prof_reg
rm -rf $O; mkdir -p $O $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-isolated --verify-steps 0 --extra-frames 4 --extra-hyp 64 --register-steps 4 --register-check 0 > $O/log.txt 2>&1
tail -c 300 $O/log.txt
python $R/tools/rocpd_stats.py $O/r_results.db > $R/gpurun_out/register_kernel_stats.txt
head -${KSTAT_LINES:-40} $R/gpurun_out/register_kernel_stats.txt | cut -c1-150
