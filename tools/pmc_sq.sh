#!/bin/bash
# Runs on the GPU box: the two SQ counter passes over the serial phase profile that tools/pmc_sq_summary.py reads.
# usage: bash tools/pmc_sq.sh  -> gpurun_out/pmc_sq_summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=/tmp/pmc_sq
rm -rf $O; mkdir -p $O $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/phase_profile.py --mb 64 --reps 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $O/a -o r -- $B > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE -d $O/b -o r -- $B > $O/b.log 2>&1
python $R/tools/pmc_sq_summary.py $O/a/r_results.db $O/b/r_results.db > $R/gpurun_out/pmc_sq_summary.txt
head -22 $R/gpurun_out/pmc_sq_summary.txt
