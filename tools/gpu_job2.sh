#!/bin/bash
# GPU job: quick parity subset + serial per-kernel profile.  Output under gpurun_out/job2/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/job2
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "${1:-kitti or ragged or pathological or full_hd or non_default or benchmark_mode or randomised or refuses}" > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
bash tools/kstat.sh ${2:-r02b} > $O/kstat.log 2>&1
tail -32 $O/kstat.log
