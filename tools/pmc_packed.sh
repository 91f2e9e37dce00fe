#!/bin/bash
# K1 (compact, k = 21) reading ASCII bytes against K1 reading packed base arrays: SQ counters per launch, one box.
#   gpurun -- 'bash tools/pmc_packed.sh'   -> gpurun_out/pmc_packed/summary.json
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_packed
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
A="python3 /root/repo/bench.py --no-cpu-baseline --no-second-process --configs none --steps 10 --warmup 2 --max-repeats 3 --min-seconds 0.01"
P="python3 /root/repo/bench.py --no-cpu-baseline --no-second-process --no-headline --configs packed_C2 --steps 10 --warmup 2"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/ascii_$tag -- $A > $O/ascii_$tag.log 2>&1
  rocprofv3 --pmc $set --output-format csv -d $O/packed_$tag -- $P > $O/packed_$tag.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ascii_trace -- $A > $O/ascii_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/packed_trace -- $P > $O/packed_trace.log 2>&1
python3 - "$O" <<'PY'
import csv,glob,sys,collections,json
O=sys.argv[1]
out={}
for v in ("ascii","packed"):
    acc=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob(f"{O}/{v}_SQ*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "partition_windows_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    out[v]={"per_launch":{c:round(x/n[c],1) for c,x in acc.items()},"launches":dict(n)}
    for f in glob.glob(f"{O}/{v}_trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "partition_windows_kernel" in r["Name"]: out[v]["trace_avg_ns"]=float(r["AverageNs"]); out[v]["trace_calls"]=int(r["Calls"])
json.dump(out,open(f"{O}/summary.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
rm -rf $O/ascii_SQ* $O/packed_SQ* $O/*_trace
