#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for c in uniform lidar; do
  for rc in 0 1; do
    for k in 0 8; do
    GV_RECTS_ON_C=$rc GV_FLAT_K=$k python bench.py --cloud $c --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rects_on_c=$rc flat_k=$k', '$c', round(d['value']), round(d['ms_per_step']*1000,1))"
    done
  done
done
done
