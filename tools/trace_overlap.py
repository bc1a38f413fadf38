"""Concurrency summary of a rocprofv3 kernel trace window: per-kernel mean duration, how much of the
window has 0 / 1 / 2 / 3+ kernels resident, and the frame period.
   trace_overlap.py DIR [from_fraction to_fraction]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
a = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
b = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
sel = rows[int(len(rows) * a):int(len(rows) * b)]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
ev = []
per = {}
for r in sel:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gv::", "").split("<")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    per.setdefault(n, []).append(e - s)
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = {}
cur, last = 0, t0
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += d; last = t
tot = sum(hist.values())
print(f"window {(t1 - t0) / 1e3:.1f} us, {len(sel)} kernels")
for k in sorted(hist): print(f"  {k} kernels resident: {100.0 * hist[k] / tot:5.1f} %")
for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:24s} n={len(v):4d} mean {sum(v) / len(v) / 1e3:6.1f} us  min {min(v) / 1e3:6.1f}  max {max(v) / 1e3:6.1f}")
nf = len(per.get("k_ray_sectors", [])) or 1
print(f"  period per frame: {(t1 - t0) / 1e3 / nf:.1f} us")
