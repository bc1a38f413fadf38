#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for c in uniform lidar; do
  for a in 0 256; do
    for p in 1 0; do
    GV_ABLATE=$a GV_PIPELINE=$p python bench.py --cloud $c --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate=$a pipe=$p', '$c', round(d['value']), round(d['ms_per_step']*1000,1), round(d['stage_ms']['ray_march']*1000,1))"
    done
  done
done
done
