#!/bin/bash
# Kernel stats of the PCA / kNN path at config-3 size (both clouds).  On the GPU box: bash tools/pca_profile.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pca_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in uniform lidar; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -- python3 $GRAFT_REPO_ROOT/tools/pca_run.py $c 12 > $OUT/$c.log 2>&1 || echo "$c failed"
  f=$(find $OUT/$c -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/pca_path_${c}_kernel_stats.csv
done
