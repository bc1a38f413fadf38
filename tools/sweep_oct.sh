#!/bin/bash
# sector-count sweep per octant (GPU experiment): GV_LOG2S_OCT variants (octant index = xmaj<<2 | smaj<<1 | smin),
# serial and pipelined frame time.  VARIANTS overrides the list; "" = the library's own choice.
cd $GRAFT_REPO_ROOT
VARIANTS=${VARIANTS:-"auto 7,7,7,7,7,7,7,7 6,6,6,6,7,7,5,5 6,6,6,6,7,7,6,6 6,6,6,6,7,7,4,4 7,7,7,7,7,7,5,5"}
for v in $VARIANTS; do
  for m in 0 1; do
    if [ "$v" = "auto" ]; then unset GV_LOG2S_OCT; else export GV_LOG2S_OCT=$v; fi
    GV_PIPELINE=$m python bench.py --cloud ${CLOUD:-uniform} --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', 'pipe=$m', round(d['value']), round(d['ms_per_step']*1000,1), 'sectors', round(d['stage_ms']['ray_march']*1000,1))"
  done
done
