#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in uniform lidar; do
  for p in 0 1 2 3; do
    GV_PRIO=$p python bench.py --cloud $c --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('prio=$p', '$c', round(d['value']), round(d['ms_per_step']*1000,1))"
  done
done
