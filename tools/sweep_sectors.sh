#!/bin/bash
# prints per-stage times (us) for several sector-kernel settings (experiments)
run() {
  echo "== $*"
  env "$@" timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fps %.0f'%d['value'], {k: round(v*1e3,1) for k,v in d['stage_ms'].items()})"
}
for a in "$@"; do run $a; done
