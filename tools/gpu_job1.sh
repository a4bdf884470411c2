#!/bin/bash
# GPU job: full gpu test suite, default bench line, serial per-kernel profile.  Output under gpurun_out/job1/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/job1
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -40 $O/pytest.log
timeout 600 python bench.py > $O/bench.log 2> $O/bench.err
echo "bench rc $?"
tail -c 6000 $O/bench.log
tail -5 $O/bench.err
bash tools/kstat.sh r02a > $O/kstat.log 2>&1
tail -30 $O/kstat.log
