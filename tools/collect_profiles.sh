#!/bin/bash
# Copy the summaries of the latest tools/profile_r02.sh run (merged back under gpurun_out/profile_r02)
# into profiles/r02/.  gpurun_out/ accumulates the files of earlier runs: the newest of each kind wins.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/profile_r02
D=profiles/r02
newest() { find "$1" -name "$2" -printf "%T@ %p\n" | sort -n | tail -1 | cut -d' ' -f2-; }
for d in serial pipelined lidar; do cp "$(newest $O/stats_$d '*kernel_stats.csv')" $D/${d}_kernel_stats.csv; done
cp $O/bench.json $D/bench_line.json
cp $O/pmc/pmc_summary.json $D/pmc_summary.json
T=$(dirname "$(newest $O/stats_pipelined '*kernel_trace.csv')")
mkdir -p /tmp/gv_tl && rm -rf /tmp/gv_tl/* && cp "$(newest $O/stats_pipelined '*kernel_trace.csv')" /tmp/gv_tl/
{ python3 tools/trace_timeline.py /tmp/gv_tl 28 0.5; echo; python3 tools/trace_overlap.py /tmp/gv_tl 0.15 0.6; } > $D/pipelined_timeline.txt
echo collected into $D
