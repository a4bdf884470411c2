#!/bin/bash
# Runs on the GPU box (via gpurun): regenerates the raw material of profiles/ under gpurun_out/refresh/.
# usage: bash tools/refresh_profiles.sh [round-tag, default r02] [micro-batch, default 256]
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r02}
MB=${2:-256}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --frames $MB --micro-batch $MB --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-pipeline"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o $T -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o $T -- $B > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace -o $T -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch/${T}_results.db $O/pmc_write/${T}_results.db $O/${T}_pmc_traffic.json $MB
python tools/rocpd_pmc.py $O/pmc_fetch/${T}_results.db $O/${T}_pmc_fetch_size.txt > /dev/null
python tools/rocpd_pmc.py $O/pmc_write/${T}_results.db $O/${T}_pmc_write_size.txt > /dev/null
python tools/rocpd_stats.py $O/trace/${T}_results.db $O/${T}_bench_kernel_stats.txt > /dev/null
cp $O/${T}_pmc_traffic.json profiles/${T}_pmc_traffic.json   # bench.py reads it for roofline.traffic
python bench.py > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log > $O/${T}_bench.json
tail -c 1500 $O/bench.log
rm -rf $O/pmc_fetch $O/pmc_write $O/trace   # raw databases are large: only the summaries travel back
