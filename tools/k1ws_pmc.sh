#!/bin/bash
# Round 6: SQ counters of the wave-specialised K1 (KCT_K1_FLUSHERS=4) beside the barrier-synchronised one (0), C2 headline, and the
# ws kernel's own per-wave figures (libkct_stamps.so).   -> gpurun_out/k1ws_pmc/
O=$GRAFT_REPO_ROOT/gpurun_out/k1ws_pmc
rm -rf "$O"; mkdir -p "$O"
R=/root/repo
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-second-process --configs none --steps 10 --warmup 2 --max-repeats 3 --min-seconds 0.01"
KCT_K1_FLUSHERS=4 KCT_K1_STAMPS_OUT=$O/ws_stamps.jsonl KCT_LIB_PATH=$R/oxli_amd/csrc/libkct_stamps.so $B > /dev/null 2>&1
tail -2 $O/ws_stamps.jsonl
for f in 0 4; do
export KCT_K1_FLUSHERS=$f
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
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
for k in keys: print("%-24s %14.0f %14.0f  %.3f" % (k, out["flushers_0"].get(k,0), out["flushers_4"].get(k,0), out["flushers_4"].get(k,0)/max(1,out["flushers_0"].get(k,1))))
PY
rm -rf $O/f*_pmc1 $O/f*_pmc2
