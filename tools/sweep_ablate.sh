#!/bin/bash
# marginal cost of the sector kernel's phases at kernel level (results are wrong with any bit set: timing only)
cd $GRAFT_REPO_ROOT
for a in 0 4 2 6 256 32 16 8; do
  GV_ABLATE=$a GV_PIPELINE=0 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate=$a', 'sectors', round(d['stage_ms']['ray_march']*1000,1))"
done
