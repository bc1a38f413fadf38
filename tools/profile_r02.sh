#!/bin/bash
# Round-2 profile set: full bench line, rocprofv3 kernel stats (serial + pipelined), PMC passes.
# Run on the GPU box from the repo root:  bash tools/profile_r02.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_r02
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
cd /tmp && export TMPDIR=/tmp
GV_PIPELINE=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --plain > $OUT/stats_serial.log 2>&1 || echo "serial stats failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pipelined -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --plain > $OUT/stats_pipelined.log 2>&1 || echo "pipelined stats failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lidar -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --plain --cloud lidar > $OUT/stats_lidar.log 2>&1 || echo "lidar stats failed"
bash $GRAFT_REPO_ROOT/tools/pmc_r02.sh $OUT/pmc > $OUT/pmc.log 2>&1 || echo "pmc failed"
for d in serial pipelined lidar; do echo "== $d"; find $OUT/stats_$d -name "*kernel_stats.csv" | head -1 | xargs cat | head -8; done
python3 $GRAFT_REPO_ROOT/tools/trace_timeline.py $OUT/stats_pipelined 24 0.5
