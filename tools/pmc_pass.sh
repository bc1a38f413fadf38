#!/bin/bash
# one rocprofv3 --pmc pass per argument group (counters separated by spaces inside quotes)
# usage on the GPU box: bash tools/pmc_pass.sh "VALUBusy MemUnitBusy" "WRITE_SIZE" ...
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
k=0
for grp in "$@"; do
  k=$((k+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/p$k.log 2>&1 || echo "pass $k ($grp) failed"
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc"
for d in sorted(glob.glob(out + "/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "gv::" not in k: continue
            print(os.path.basename(os.path.dirname(d)), k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
PY
