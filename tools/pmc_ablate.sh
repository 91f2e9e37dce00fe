#!/bin/bash
# Dynamic instruction mix of the headline's K1 by ABLATION (results INVALID, instruction counts only): a -DKCT_DEBUG_ENV build
# (make -C oxli_amd/csrc variant V=dbg EXTRA=-DKCT_DEBUG_ENV) run with KCT_ABLATE = 0 (all work), 1 (no ring append), 2 (no ring flush),
# 3 (neither) under rocprofv3 --pmc: the differences are what the append and the flush cost per window.  -> gpurun_out/pmc_ablate.json
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_ablate
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export KCT_LIB_PATH=/root/repo/oxli_amd/csrc/libkct_dbg.so
for ab in 0 1 2 3; do
  export KCT_ABLATE=$ab
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/ab$ab -- python3 /root/repo/bench.py --no-cpu-baseline --no-verify --no-second-process --configs none --steps 5 --warmup 2 --max-repeats 3 > $O/ab$ab.log 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
out = {}
for ab in range(4):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/ab{ab}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "partition_windows_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[f"ablate_{ab}"] = {c: sum(v) / len(v) * 64 / 1.51e8 for c, v in acc.items() if c.startswith("SQ_INSTS")}   # per window start
json.dump(out, open(f"{O}/../pmc_ablate.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $O/ab*/
