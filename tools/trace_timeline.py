"""Print a window of the device timeline from a rocprofv3 kernel_trace.csv:
  trace_timeline.py DIR [N kernels] [position 0..1 of the window's end inside the trace; default 1 = the tail]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
nsel = int(sys.argv[2]) if len(sys.argv) > 2 else 40
end = max(nsel, int(len(rows) * (float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)))
sel = rows[end - nsel:end]
for r in sel:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gv::", "")[:28]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:10.1f} {e:10.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?'):>3}  {n}")

mf = glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True)
if mf:
    m = sorted(csv.DictReader(open(mf[0])), key=lambda r: int(r["Start_Timestamp"]))
    print("memory copies (last 12):")
    for r in m[-12:]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{s:10.1f} {e:10.1f} {e - s:8.1f}  {r['Direction']}")
