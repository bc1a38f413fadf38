#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for c in uniform lidar; do
  for n in 0 1; do
    GV_SPLIT_POINTS=$n python bench.py --cloud $c --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$n', '$c', round(d['value']), round(d['ms_per_step']*1000,1))"
  done
done
done
