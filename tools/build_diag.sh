#!/bin/bash
# Diagnostic build of the library (-DGV_DIAG: phase stamps, ablation switches, pipeline trace) into
# tools/_diag/libgv_diag.so; load it with GV_LIB_AB=tools/_diag/libgv_diag.so.  Never shipped.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_diag
python3 - <<'PY'
import os, sys
sys.path.insert(0, "grid-vision_amd")
from gvamd import build as b
root = os.getcwd()
print("built", b.build(lib=os.path.join(root, "tools/_diag/libgv_diag.so"), extra=["-DGV_DIAG"],
                       obj_dir=os.path.join(root, "tools/_diag/obj")))
PY
