#!/bin/bash
# Runs on the GPU box (via gpurun): regenerates the raw material of profiles/ under gpurun_out/refresh/.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --frames 64 --micro-batch 64 --steps 1 --warmup 0 --no-cpu-baseline"
AKZ_PIPELINE=0 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o r01 -- $B > $O/pmc_fetch.log 2>&1
AKZ_PIPELINE=0 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o r01 -- $B > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace -o r01 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_fetch/r01_results.db $O/pmc_write/r01_results.db $O/r01_pmc_traffic.json
cp $O/r01_pmc_traffic.json profiles/r01_pmc_traffic.json   # bench.py reads it for roofline.traffic
python bench.py > $O/bench.log 2>&1
tail -1 $O/bench.log
