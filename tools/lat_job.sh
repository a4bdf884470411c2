#!/bin/bash
# single-frame latency: wall time of the host call and the per-kernel timeline of one frame
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/tools/latency_probe.py > $R/gpurun_out/lat_plain.txt 2>&1
python $R/tools/latency_probe.py det_side_stream=0 >> $R/gpurun_out/lat_plain.txt 2>&1
rm -rf /tmp/lat0; rocprofv3 --kernel-trace -d /tmp/lat0 -o r -- python $R/tools/latency_probe.py det_side_stream=0 > /tmp/lat0_log.txt 2>&1
python $R/tools/lat_trace.py /tmp/lat0/r_results.db > $R/gpurun_out/lat_trace_serial.txt 2>&1
rm -rf /tmp/lat; rocprofv3 --kernel-trace -d /tmp/lat -o r -- python $R/tools/latency_probe.py > /tmp/lat_log.txt 2>&1
python $R/tools/lat_trace.py /tmp/lat/r_results.db > $R/gpurun_out/lat_trace.txt 2>&1
grep single $R/gpurun_out/lat_plain.txt; cat $R/gpurun_out/lat_trace.txt
