#!/bin/bash
# Round 6: how long is an LDS instruction of K1 in flight?  SQ_INST_LEVEL_LDS (instructions in flight, summed over cycles) / SQ_INSTS_LDS,
# FIFO-full and conflict cycles; KCT_K1_FLUSHERS = 0 (barrier-synchronised K1) and 4 (wave-specialised).  -> gpurun_out/k1_lds/
O=$GRAFT_REPO_ROOT/gpurun_out/k1_lds
rm -rf "$O"; mkdir -p "$O"
R=/root/repo
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-second-process --configs none --steps 10 --warmup 2 --max-repeats 3 --min-seconds 0.01"
for f in 0 4; do
export KCT_K1_FLUSHERS=$f
i=0
for set in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_BRANCH SQ_INSTS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/f${f}_pmc$i -- $B > $O/f${f}_pmc$i.log 2>&1
done
done
python3 - "$O" <<'PY'
import csv,glob,sys,collections,json
O=sys.argv[1]
out={}
for f in ("0","4"):
    acc=collections.defaultdict(list)
    for fn in glob.glob(f"{O}/f{f}_pmc*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "partition_windows" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out["flushers_"+f]={c:sum(v)/len(v) for c,v in acc.items()}
json.dump(out,open(f"{O}/summary.json","w"),indent=1)
keys=sorted(set(out["flushers_0"])|set(out["flushers_4"]))
for k in keys: print("%-24s %14.0f %14.0f" % (k, out["flushers_0"].get(k,0), out["flushers_4"].get(k,0)))
PY
tail -3 $O/f0_pmc1.log
rm -rf $O/f*_pmc1 $O/f*_pmc2
