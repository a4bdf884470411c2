R=${GRAFT_REPO_ROOT:-/root/repo}
O=/tmp/pmc_match
rm -rf $O; mkdir -p $O $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
K=${1:-fp4}          # fp4 (wide, LDS-DMA) | fp4_regs | int8 | valu
B="python $R/tools/bench_match.py --reps 1 --kernel $K"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY -d $O/a -o r -- $B > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O/b -o r -- $B > $O/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL -d $O/c -o r -- $B > $O/c.log 2>&1
cd $R
for p in a b c; do python tools/rocpd_pmc.py $O/$p/r_results.db | grep -E "^k_knn_mfma|^kernel" ; done | tee $R/gpurun_out/pmc_matcher_$K.txt
