"""How often does the upload stream land on a shared hardware queue, and does gv_create's probe catch it?  A number of
fresh processes, each creating `extra` other streams first (what a host application or framework would own), then a
handle with GV_VERBOSE=1; afterwards the streamed-ingest rate of that handle.  python3 tools/queue_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(%r, "grid-vision_amd"))
import numpy as np
hip = C.CDLL("libamdhip64.so")
extra = int(sys.argv[1])
streams = []
for _ in range(extra):
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; streams.append(s)
import gvamd
from gvamd import synth
config = 3
g = synth.CONFIGS[config]["grid"]; tfs = synth.transforms(True)
x, y, z, _ = synth.cloud_uniform(config)
h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
pins = [gvamd.PinnedF32(len(x)) for _ in range(3)]
for p, a in zip(pins, (x, y, z)): p.array[:] = a
h.set_detections(gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH)
def run(nf):
    t0 = time.perf_counter()
    for _ in range(nf):
        h.upload_xyz_async(pins[0].array, pins[1].array, pins[2].array)
        h.enqueue_frame()
    h.synchronize()
    return (time.perf_counter() - t0) / nf * 1e6
run(150)
print("extra streams %%d: streamed frame %%.0f us" %% (extra, min(run(150) for _ in range(3))), flush=True)
h.close()
''' % ROOT
for probe in ("1", "0"):
    for extra in (0, 1, 2, 3):
        env = dict(os.environ, GV_VERBOSE="1", GV_QUEUE_PROBE=probe)
        r = subprocess.run([sys.executable, "-c", CHILD, str(extra)], env=env, capture_output=True, text=True)
        print(f"probe {probe}:", r.stdout.strip(), "|", " ".join(l for l in r.stderr.splitlines() if "gridvision_hip" in l))
