#!/bin/bash
# Diagnostic build of the library (-DGV_DIAG: phase stamps, ablation switches, pipeline trace) into
# tools/_diag/libgv_diag.so; load it with GV_LIB_AB=tools/_diag/libgv_diag.so.  Never shipped.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_diag
cd grid-vision_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -Wall \
  -Wno-unused-function -Wno-bitwise-instead-of-logical -DGV_DIAG -o ../../tools/_diag/libgv_diag.so gv_api.hip gv_kernels.hip gv_binning.hip \
  gv_raysector.hip gv_shard.hip gv_knn_pca.hip gv_cloudops.hip -L/opt/rocm/lib -lrccl
echo built tools/_diag/libgv_diag.so
