#!/bin/bash
# Round-1 profiles: bench line + rocprofv3 kernel stats + PMC traffic passes (separate runs).
# Run on the GPU box from the repo root:  bash tools/profile_r01.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_r01
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-hit-counts > $OUT/stats.log 2>&1 || echo "stats failed"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hit-counts > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch failed"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hit-counts > $OUT/pmc_write.log 2>&1 || echo "pmc write failed"
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_lds -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hit-counts > $OUT/pmc_lds.log 2>&1 || echo "pmc lds failed"
find $OUT -name "*.csv" | head -20
