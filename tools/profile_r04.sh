#!/bin/bash
# Round-4 profile set: rocprofv3 kernel stats (serial + pipelined + lidar + config-5 working set), PMC passes for
# config 3 and for the beyond-L3 workload.  Run on the GPU box from the repo root:  bash tools/profile_r04.sh
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
GV_PIPELINE=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- python3 $B --steps 100 --warmup 10 --plain > $OUT/stats_serial.log 2>&1 || echo "serial stats failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pipelined -- python3 $B --steps 100 --warmup 10 --plain > $OUT/stats_pipelined.log 2>&1 || echo "pipelined stats failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lidar -- python3 $B --steps 100 --warmup 10 --plain --cloud lidar > $OUT/stats_lidar.log 2>&1 || echo "lidar stats failed"
GV_PIPELINE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5_serial -- python3 $B --config 5 --steps 20 --warmup 3 --plain > $OUT/stats_c5_serial.log 2>&1 || echo "config-5 stats failed"
for cfg in 3 5; do
  k=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
             "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU2"; do
    k=$((k+1))
    GV_PIPELINE=0 timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_c$cfg/p$k -- python3 $B --config $cfg --steps 8 --warmup 3 --plain > $OUT/pmc_c$cfg.p$k.log 2>&1 || echo "pmc c$cfg pass $k failed"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, os, collections, json, sys
out = sys.argv[1]
for cfg in (3, 5):
    res = collections.defaultdict(dict)
    for d in sorted(glob.glob(f"{out}/pmc_c{cfg}/p*/")):
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
        if "FETCH_SIZE" in cs: cs["fetch_bytes_per_launch"] = cs["FETCH_SIZE"] * 1024 * 2   # gfx950: 128-B requests tallied at 64 B
        if "WRITE_SIZE" in cs: cs["write_bytes_per_launch"] = cs["WRITE_SIZE"] * 1024
        if "fetch_bytes_per_launch" in cs and "write_bytes_per_launch" in cs:
            cs["hbm_bytes_per_launch"] = cs["fetch_bytes_per_launch"] + cs["write_bytes_per_launch"]
    json.dump({"note": f"rocprofv3 --pmc, one counter group per run, GV_PIPELINE=0 bench.py --config {cfg} --steps 8 --warmup 3 --plain; averages per dispatch; FETCH_SIZE/WRITE_SIZE in KB, FETCH_SIZE doubled per MI355X_MICROARCH.md", "kernels": res}, open(f"{out}/pmc_summary_c{cfg}.json", "w"), indent=1)
    print("config", cfg)
    for k, cs in res.items():
        print(" ", k, {c: round(v, 1) for c, v in cs.items() if c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "hbm_bytes_per_launch", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")})
PY
for d in serial pipelined lidar c5_serial; do echo "== $d"; find $OUT/stats_$d -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160 | head -7; done
