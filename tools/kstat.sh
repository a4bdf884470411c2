#!/bin/bash
# Runs on the GPU box: per-kernel times of the serial phase profile (each phase alone on the GPU).
# usage: bash tools/kstat.sh [tag] [phase_profile.py arguments...]   -> gpurun_out/kstat_<tag>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-x}
shift
O=/tmp/kstat_$T
rm -rf $O; mkdir -p $O $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o r -- python $R/tools/phase_profile.py --mb ${KSTAT_MB:-64} --reps 2 "$@" > $O/log.txt 2>&1
tail -1 $O/log.txt
python $R/tools/rocpd_stats.py $O/r_results.db > $R/gpurun_out/kstat_$T.txt
head -${KSTAT_LINES:-24} $R/gpurun_out/kstat_$T.txt | cut -c1-130
