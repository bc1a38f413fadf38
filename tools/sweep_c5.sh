#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in auto 8,8,8,8,9,9,7,7 9,9,9,9,10,10,8,8 10,10,10,10,11,11,9,9 7,7,7,7,8,8,6,6; do
  if [ "$v" = "auto" ]; then unset GV_LOG2S_OCT; else export GV_LOG2S_OCT=$v; fi
  python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), round(d['ms_per_step']*1000,1), {k:round(x*1000,1) for k,x in d['stage_ms'].items()})"
done
