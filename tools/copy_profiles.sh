#!/bin/bash
# copies the summaries a tools/validate_job.sh run merged into gpurun_out/ to profiles/ (tracked)
T=${1:-r06}
cd "$(dirname "$0")/.."
cp gpurun_out/refresh/${T}_bench.json profiles/${T}_bench.json
cp gpurun_out/refresh/${T}_bench_headline.json profiles/${T}_bench_headline.json
cp gpurun_out/refresh/${T}_bench_profiled.json profiles/${T}_bench_profiled.json
cp gpurun_out/refresh/${T}_bench_kernel_stats.txt gpurun_out/refresh/${T}_pmc_traffic.json gpurun_out/refresh/${T}_pmc_fetch_size.txt gpurun_out/refresh/${T}_pmc_write_size.txt profiles/
cp gpurun_out/pmc_sq_summary.txt profiles/${T}_pmc_sq_summary.txt
(echo "# SQ counters of the k-NN launches of tools/bench_match.py (128 frame pairs, one repetition): the wide LDS-DMA kernel, then the register-staged one"; cat gpurun_out/pmc_matcher_fp4.txt gpurun_out/pmc_matcher_fp4_regs.txt) > profiles/${T}_pmc_matcher.txt
cp gpurun_out/kstat_final.txt profiles/${T}_kstat_serial_64frames.txt
(grep -i "single\|alone" gpurun_out/lat_plain.txt; cat gpurun_out/lat_trace.txt) > profiles/${T}_single_frame_timeline.txt
[ -f gpurun_out/register_kernel_stats.txt ] && cp gpurun_out/register_kernel_stats.txt profiles/${T}_register_kernel_stats.txt
[ -f gpurun_out/gputest_${T}.log ] && tail -12 gpurun_out/gputest_${T}.log > profiles/${T}_gputest.txt
[ -f gpurun_out/stress_${T}.txt ] && cp gpurun_out/stress_${T}.txt profiles/${T}_stress_parity.txt
ls -la profiles | grep ${T}_
