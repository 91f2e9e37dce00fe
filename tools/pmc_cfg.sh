#!/bin/bash
# SQ counters of one big configuration (tools/run_config.py) per kernel, for an environment switch off / on, on one box:
#   gpurun -- 'bash tools/pmc_cfg.sh C3 KCT_NO_PK48'        (needs libkct_dbg.so: make -C oxli_amd/csrc variant V=dbg EXTRA=-DKCT_DEBUG_ENV)
C=${1:-C3}; SW=${2:-KCT_NO_PK48}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_cfg_$C
rm -rf "$O"; mkdir -p "$O"
export KCT_LIB_PATH=$GRAFT_REPO_ROOT/oxli_amd/csrc/libkct_dbg.so
cd /tmp && export TMPDIR=/tmp
for v in off on; do
  if [ $v = on ]; then export $SW=1; else unset $SW; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/$v -- python3 /root/repo/tools/run_config.py $C --paths auto --no-dump > $O/$v.log 2>&1
done
python3 - "$O" "$SW" <<'PY'
import csv,glob,sys,collections,json
O,SW=sys.argv[1],sys.argv[2]
out={}
for v in ("off","on"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob(f"{O}/{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("kct::","")[:70]
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    out[f"{SW}_{v}"]={k:{c:round(x/1e6,3) for c,x in d.items()} for k,d in acc.items() if "aggregate" in k or "partition" in k}
json.dump(out,open(f"{O}/summary.json","w"),indent=1)
for v,d in out.items():
    for k,c in d.items(): print(v,k,c)
PY
rm -rf $O/off $O/on
