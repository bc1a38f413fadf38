"""Build recipe of libgridvision_hip.so (hipcc, gfx950 only, in-tree).

Every translation unit is compiled to an object of its own (in parallel, only when it or a header changed) and the
objects are linked into the shared library: a change to one kernel file rebuilds in seconds."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "libgridvision_hip.so")
SOURCES = ["gv_api.hip", "gv_kernels.hip", "gv_binning.hip", "gv_raysector.hip", "gv_shard.hip", "gv_knn_pca.hip",
           "gv_cloudops.hip"]
# -ffp-contract=off: cell indices must be bit-exact with the reference's separate
# multiply/add roundings; no fast-math anywhere.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC)")


def _headers() -> list:
    inc = os.path.join(os.path.dirname(PKG), "include", "gridvision_hip.h")
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))] + [inc]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, lib: str = LIB, extra: list | None = None, obj_dir: str = OBJ) -> str:
    """force: recompile every translation unit.  lib / extra / obj_dir: another build of the same library (the
    diagnostic -DGV_DIAG build of tools/build_diag.sh) with objects of its own."""
    if not (force or lib != LIB or needs_build()):
        return lib
    extra = list(extra or []) + os.environ.get("GV_HIPCC_EXTRA", "").split()
    os.makedirs(obj_dir, exist_ok=True)
    stamp = os.path.join(obj_dir, ".flags")
    flags_now = " ".join(FLAGS + extra)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cc = hipcc()

    def compile_one(src: str) -> str:
        path = os.path.join(CSRC, src)
        obj = os.path.join(obj_dir, src + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), hdr_t):
            subprocess.check_call([cc, *FLAGS, *extra, "-c", "-o", obj, path], cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, *extra_link()], cwd=CSRC)
    with open(stamp, "w") as f:
        f.write(flags_now)
    return lib


def extra_link() -> list:
    # RCCL for the sharded multi-GPU frame (gv_comm_*)
    return ["-L/opt/rocm/lib", "-lrccl"]


if __name__ == "__main__":
    print(build(force=True))
