#!/bin/bash
# full validation: gpu tests, smoke, SQ counters, profile refresh (counter passes, traced bench, the driver's bench command),
# serial phase profile, matcher counters, single-frame timeline.  usage: bash tools/validate_job.sh [round tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
T=${1:-r06}
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider > gpurun_out/gputest_$T.log 2>&1   # (the whole log: a flake must be nameable)
grep -E "passed|failed|FAILED" gpurun_out/gputest_$T.log | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/pmc_sq.sh > /dev/null 2>&1
cp gpurun_out/pmc_sq_summary.txt profiles/${T}_pmc_sq_summary.txt      # bench.py reads it for issue_counters
bash tools/refresh_profiles.sh $T 256 > gpurun_out/refresh.log 2>&1
tail -c 600 gpurun_out/refresh/bench.log
KSTAT_LINES=60 bash tools/kstat.sh final > /dev/null 2>&1
bash tools/pmc_match.sh fp4 > /dev/null 2>&1
bash tools/pmc_match.sh fp4_regs > /dev/null 2>&1
bash tools/lat_job.sh > /dev/null 2>&1
grep single gpurun_out/lat_plain.txt
KSTAT_LINES=70 bash tools/prof_register.sh > /dev/null 2>&1      # -> gpurun_out/register_kernel_stats.txt (tools/copy_profiles.sh)
