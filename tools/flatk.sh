#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in uniform lidar; do
  for k in 0 2 4 8 16 32; do
    for p in 1 0; do
    GV_FLAT_K=$k GV_PIPELINE=$p python bench.py --cloud $c --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flat_k=$k pipe=$p', '$c', round(d['value']), round(d['ms_per_step']*1000,1), round(d['stage_ms']['ray_march']*1000,1))"
    done
  done
done
