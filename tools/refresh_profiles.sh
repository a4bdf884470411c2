#!/bin/bash
# Runs on the GPU box (via gpurun): regenerates the raw material of profiles/ under gpurun_out/refresh/.
# usage: bash tools/refresh_profiles.sh [round-tag, default r03] [micro-batch, default 256]
#   <tag>_bench_kernel_stats.txt  rocprofv3 --kernel-trace of `bench.py --steps 30 --warmup 2 --no-isolated --no-extras
#                                 --no-cpu-baseline` — every launch of that run is the pipelined workload's, so a kernel's
#                                 average duration here is what the same run's JSON line (<tag>_bench_profiled.json) reports as
#                                 roofline.avg_launch_us
#   <tag>_pmc_traffic.json        counter passes of `bench.py --pmc-run` (one pipeline step, nothing else)
#   <tag>_bench.json              the line of `python bench.py --gpus 1 --steps 20 --warmup 5` (what the driver runs)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r06}
MB=${2:-256}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --pmc-run --frames $MB --micro-batch $MB --steps 1 --warmup 0"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o $T -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o $T -- $B > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d $O/pmc_valu -o $T -- $B > $O/pmc_valu.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch/${T}_results.db $O/pmc_write/${T}_results.db $O/${T}_pmc_traffic.json $MB $O/pmc_valu/${T}_results.db $MB 1
cp $O/${T}_pmc_traffic.json profiles/${T}_pmc_traffic.json   # bench.py reads it for roofline.traffic / valu_frac
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o $T -- python $R/bench.py --steps 30 --warmup 2 --no-isolated --no-extras --no-cpu-baseline > $O/trace.log 2> $O/trace.err
cp $R/gpurun_out/bench_detail.json $O/${T}_bench_profiled.json       # the full report of that run (stdout carries the < 4 KB headline only)
cd $R
python tools/rocpd_pmc.py $O/pmc_fetch/${T}_results.db $O/${T}_pmc_fetch_size.txt > /dev/null
python tools/rocpd_pmc.py $O/pmc_write/${T}_results.db $O/${T}_pmc_write_size.txt > /dev/null
python tools/rocpd_stats.py $O/trace/${T}_results.db $O/${T}_bench_kernel_stats.txt > /dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err   # the command the driver runs
cp $R/gpurun_out/bench_detail.json $O/${T}_bench.json                # the full report ...
grep '^{' $O/bench.log | tail -1 > $O/${T}_bench_headline.json    # ... and the line the driver parses
tail -c 1500 $O/bench.log
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_valu $O/trace   # raw databases are large: only the summaries travel back
