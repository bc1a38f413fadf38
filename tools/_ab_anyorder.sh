#!/bin/bash
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export GV_ANYORDER=1; else unset GV_ANYORDER; fi
  echo "== GV_ANYORDER=$v"
  python3 tools/lib_ab.py shipped || exit 1
done
unset GV_ANYORDER
echo "== lidar 0"; python3 tools/lib_ab.py lidar shipped
export GV_ANYORDER=1
echo "== lidar 1"; python3 tools/lib_ab.py lidar shipped
