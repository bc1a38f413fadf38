#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (box-to-box variance is ~5 %):
#   A = grid-vision_amd/ab/lib_A.so (a saved build), B = the in-tree build.   usage: bash tools/ab.sh [rounds]
cd $GRAFT_REPO_ROOT
R=${1:-2}
for r in $(seq $R); do
  for c in uniform lidar; do
    for v in A B; do
      if [ $v = A ]; then export GV_LIB_AB=$GRAFT_REPO_ROOT/grid-vision_amd/ab/lib_A.so; else unset GV_LIB_AB; fi
      for m in 1 0; do
        GV_PIPELINE=$m python bench.py --cloud $c --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', '$c', 'pipe=$m', round(d['value']), round(d['ms_per_step']*1000,1), {k:round(x*1000,1) for k,x in d['stage_ms'].items()})"
      done
    done
  done
done
