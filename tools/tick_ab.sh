#!/bin/bash
# A/B of library builds on gv_tick's PCA branch: per variant the host-observed tick and the kernels' mean durations
# (rocprofv3 --kernel-trace --stats).  Variants are built into tools/_ab/ first (tools/build_ab.py).
# On the GPU box: bash tools/tick_ab.sh tools/_ab/a.so tools/_ab/b.so ... [-- kernel_name_filter]
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/tick_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FILTER="k_"
for lib in "$@"; do
  name=$(basename $lib .so)
  export GV_LIB_AB=$GRAFT_REPO_ROOT/$lib
  python3 $GRAFT_REPO_ROOT/tools/tick_run.py pca 40 > $OUT/$name.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/tick_run.py pca 20 > /dev/null 2>&1
  f=$(find $OUT/$name -name '*kernel_stats.csv' | head -1)
  echo "== $name: $(head -1 $OUT/$name.log)"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gv::k_" in r["Name"] and "k_hold" not in r["Name"]]
print("   " + "  ".join(f'{r["Name"].split("gv::")[1].split("(")[0].split("<")[0]} {float(r["AverageNs"])/1e3:.1f}' for r in rows[:9]))
PY
  rm -rf $OUT/$name
done
