#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in uniform lidar; do
  for n in 2 3 4; do
    GV_PIPE_SETS=$n python bench.py --cloud $c --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sets=$n', '$c', round(d['value']), round(d['ms_per_step']*1000,1))"
  done
  GV_PIPE_SETS=3 GV_PIPELINE=2 python bench.py --cloud $c --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sets=3 two-streams', '$c', round(d['value']), round(d['ms_per_step']*1000,1))"
done
