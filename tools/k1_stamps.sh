#!/bin/bash
# Round 6: what do K1's waves wait for?  One box:
#   (1) libkct_stamps.so (make variant V=stamps EXTRA=-DKCT_K1_STAMPS): every wave's shader-clock cycles per phase of K1
#       (hashing / tile barrier / staging / the flush's three barriers and two work phases), C2 steady state, north star, C3, C5 shard
#   (2) libkct_nobar.so (make variant V=nobar EXTRA=-DKCT_ABLATE_FLUSH_BARRIERS, results INVALID): ring_flush without its barriers
#   (3) SQ counters of the shipped library's C2 headline: WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY split, GRBM_GUI_ACTIVE (real clock)
#   gpurun -- 'bash tools/k1_stamps.sh'   -> gpurun_out/k1_stamps/
O=$GRAFT_REPO_ROOT/gpurun_out/k1_stamps
rm -rf "$O"; mkdir -p "$O"
R=/root/repo
cd /tmp && export TMPDIR=/tmp
A="--no-cpu-baseline --no-second-process --configs none --steps 20 --warmup 5"
for lib in hip stamps nobar hip; do
  [ "$lib" = stamps ] && export KCT_K1_STAMPS_OUT=$O/stamps_C2.jsonl || unset KCT_K1_STAMPS_OUT
  NV=""; [ "$lib" = nobar ] && NV="--no-verify"
  KCT_LIB_PATH=$R/oxli_amd/csrc/libkct_$lib.so python3 $R/bench.py $A $NV 2>$O/bench_$lib.err | tail -1 > $O/bench_$lib.json
  python3 -c "
import json,sys
d=json.loads(open('$O/bench_$lib.json').read()); print('$lib C2 value %.4g ms/step %.4f' % (d['value'], d['ms_per_step']), d['roofline']['kernels_ms_per_step'])" | tee -a $O/summary.txt
done
for cfg in NS C3 C5; do
  KCT_K1_STAMPS_OUT=$O/stamps_$cfg.jsonl KCT_LIB_PATH=$R/oxli_amd/csrc/libkct_stamps.so python3 $R/tools/run_config.py $cfg --paths auto --no-dump > $O/run_stamps_$cfg.json 2>$O/run_stamps_$cfg.err
  KCT_LIB_PATH=$R/oxli_amd/csrc/libkct_hip.so python3 $R/tools/run_config.py $cfg --paths auto --no-dump > $O/run_hip_$cfg.json 2>$O/run_hip_$cfg.err
done
rocprofv3 -L > $O/counters_avail.txt 2>&1
B="python3 $R/bench.py --no-cpu-baseline --no-second-process --configs none --steps 10 --warmup 2 --max-repeats 3 --min-seconds 0.01"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- $B > $O/pmc$i.log 2>&1
done
python3 - "$O" <<'PY'
import csv,glob,sys,collections,json
O=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0].replace("void ","").replace("kct::","")
        if "partition_windows" in n or "aggregate_blocks32" in n:
            acc[n[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items()}
json.dump(out,open(f"{O}/pmc_summary.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
for f in $O/stamps_*.jsonl; do echo "== $f"; head -3 $f; done
