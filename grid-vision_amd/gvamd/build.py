"""Build recipe of libgridvision_hip.so (hipcc, gfx950 only, in-tree)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libgridvision_hip.so")
SOURCES = ["gv_api.hip", "gv_kernels.hip", "gv_binning.hip", "gv_raysector.hip", "gv_shard.hip", "gv_knn_pca.hip",
           "gv_cloudops.hip"]
# -ffp-contract=off: cell indices must be bit-exact with the reference's separate
# multiply/add roundings; no fast-math anywhere.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    inc = os.path.join(os.path.dirname(PKG), "include", "gridvision_hip.h")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [inc]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False) -> str:
    if force or needs_build():
        srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
        extra = os.environ.get("GV_HIPCC_EXTRA", "").split()
        cmd = [hipcc(), *FLAGS, *extra, "-o", LIB, *srcs, *extra_link()]
        subprocess.check_call(cmd, cwd=CSRC)
    return LIB


def extra_link() -> list:
    # RCCL for the sharded multi-GPU frame (gv_comm_*)
    return ["-L/opt/rocm/lib", "-lrccl"]


if __name__ == "__main__":
    print(build(force=True))
