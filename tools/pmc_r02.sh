#!/bin/bash
# Round-2 PMC passes (one counter group per run, --pmc only: no trace domains), serial frame so that
# per-dispatch counters are not smeared by overlapping kernels.  Run on the GPU box from the repo root:
#   bash tools/pmc_r02.sh [outdir]
set -o pipefail
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc_r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GV_PIPELINE=0
k=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum"; do
  k=$((k+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --plain > $OUT/p$k.log 2>&1 || echo "pass $k ($grp) failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, collections, json, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "gv::" not in name: continue
            short = name.split("gv::")[1].split("(")[0].split("<")[0]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                res[k][c] = sum(v) / len(v)
for k, cs in res.items():
    if "FETCH_SIZE" in cs:   # KB; x2: gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM)
        cs["fetch_bytes_per_launch"] = cs["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in cs:
        cs["write_bytes_per_launch"] = cs["WRITE_SIZE"] * 1024
    if "fetch_bytes_per_launch" in cs and "write_bytes_per_launch" in cs:
        cs["hbm_bytes_per_launch"] = cs["fetch_bytes_per_launch"] + cs["write_bytes_per_launch"]
json.dump({"note": "rocprofv3 --pmc, one counter group per run, GV_PIPELINE=0 bench.py --steps 10 --warmup 3 --plain; averages per dispatch; FETCH_SIZE/WRITE_SIZE in KB, FETCH_SIZE doubled per MI355X_MICROARCH.md", "kernels": res}, open(out + "/pmc_summary.json", "w"), indent=1)
for k, cs in res.items():
    print(k, {c: round(v, 1) for c, v in cs.items()})
PY
