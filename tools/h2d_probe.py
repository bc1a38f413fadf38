"""Streaming-ingest regime probe (round 3): per-chunk frame period of the streamed frame next to the
device's DPM state (sclk / mclk / fclk / socclk / pcie link), sampled from sysfs by a thread.

  python3 tools/h2d_probe.py [mode ...]
     modes: frames (default), copy (empty frame), pub (grid download too), idle<S> (sleep S seconds
     before the next phase), n<K> (frames per phase, default 1500), chunk<K> (default 25)
Prints one JSON object: phases[] with the per-chunk series and the DPM transitions seen."""
import glob, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grid-vision_amd"))
import numpy as np
import gvamd
from gvamd import synth

DPM = ["pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "pp_dpm_pcie", "current_link_speed",
       "current_link_width", "power_dpm_force_performance_level"]


def card_dir():
    """sysfs directory of HIP device 0 (the host's other GPUs are visible in sysfs too): matched by PCI bus id"""
    import ctypes as C
    try:
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            d = "/sys/bus/pci/devices/" + buf.value.decode().lower()
            if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                return d
    except OSError:
        pass
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
            return d
    return None


def read_state(d):
    st = {}
    for f in DPM:
        try:
            txt = open(os.path.join(d, f)).read()
        except OSError:
            continue
        if f.startswith("pp_dpm"):
            cur = [l.strip() for l in txt.splitlines() if l.strip().endswith("*")]
            st[f] = cur[0] if cur else txt.strip().replace("\n", " | ")
        else:
            st[f] = txt.strip()
    return st


class Sampler(threading.Thread):
    def __init__(self, d, period=0.004):
        super().__init__(daemon=True)
        self.d, self.period, self.log, self.stop = d, period, [], False
        self.t0 = time.perf_counter()

    def run(self):
        last = None
        while not self.stop:
            st = read_state(self.d) if self.d else {}
            if st != last:
                self.log.append((round(time.perf_counter() - self.t0, 4), st))
                last = st
            time.sleep(self.period)


def main():
    args = sys.argv[1:] or ["frames"]
    config = 3
    g = synth.CONFIGS[config]["grid"]
    tfs = synth.transforms(True)
    n_sets = 6
    pins, dets, blocks = [], [], []
    for f in range(n_sets):
        x, y, z, _ = synth.cloud_uniform(config, seed_extra=100 + f)
        n0 = len(x)
        blk = gvamd.PinnedF32(3 * n0)
        blk.array[:n0], blk.array[n0:2 * n0], blk.array[2 * n0:] = x, y, z
        blocks.append(blk)
        pins.append((blk.array[:n0], blk.array[n0:2 * n0], blk.array[2 * n0:]))
        dets.append((synth.detections(config, seed_extra=f), synth.lshape_poses(config, seed_extra=f)))
    h = gvamd.GridVisionHIP(g.grid_x, g.grid_y, g.resolution)
    h.set_transforms(tfs["cam_lidar"], tfs["base_cam"], tfs["base_lidar"])
    flags = gvamd.FRAME_BIN | gvamd.FRAME_RAYMARCH | gvamd.FRAME_BBOX_TEST
    pub = [gvamd.PinnedF32((h.G + 3) // 4) for _ in range(2)]
    d = card_dir()
    smp = Sampler(d)
    smp.start()
    frames, chunk = 1500, 25
    phases = []
    fno = 0
    for mode in args:
        if mode.startswith("idle"):
            time.sleep(float(mode[4:] or 1))
            phases.append({"mode": mode, "t": round(time.perf_counter() - smp.t0, 4)})
            continue
        if mode.startswith("chunk"):
            chunk = int(mode[5:])
            continue
        if mode[0] == "n" and mode[1:].isdigit():
            frames = int(mode[1:])
            continue
        per, done = [], 0
        t_begin = round(time.perf_counter() - smp.t0, 4)
        while done < frames:
            k = min(chunk, frames - done)
            t0 = time.perf_counter()
            for _ in range(k):
                px, py, pz = pins[fno % n_sets]
                h.upload_xyz_async(px, py, pz)
                if mode == "copy":
                    h.set_detections_async(0)
                else:
                    h.set_detections_async(flags, bboxes=dets[fno % n_sets][0], poses=dets[fno % n_sets][1])
                h.enqueue_frame()
                if mode == "pub":
                    h.to_occupancy_grid_async(pub[fno % 2].array.view("int8")[:h.G])
                fno += 1
            h.synchronize()
            per.append(round((time.perf_counter() - t0) / k * 1e6, 1))
            done += k
        phases.append({"mode": mode, "t_begin": t_begin, "t_end": round(time.perf_counter() - smp.t0, 4),
                       "frames": frames, "chunk": chunk, "us_per_frame": per})
    smp.stop = True
    smp.join()
    h.close()
    print(json.dumps({"sysfs": d, "phases": phases, "dpm_transitions": smp.log}))


if __name__ == "__main__":
    main()
