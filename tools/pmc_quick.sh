#!/bin/bash
# quick per-kernel PMC comparison of two bench paths on one box: tools/pmc_quick.sh <path> -> gpurun_out/pmcq_<path>/
P=${1:-auto}
O=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$P
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-second-process --configs none --path $P --steps 3 --warmup 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/lds -- $B > $O/lds.log 2>&1
python3 - "$O" <<'PY'
import csv,glob,sys,collections
O=sys.argv[1]
for sub in ("sq","lds"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if "aggregate" in k or "partition" in k:
            print(sub, k, {c: round(sum(x)/len(x)/1e6,2) for c,x in v.items()})
PY
