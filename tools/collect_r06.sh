#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py, per workload.
# Usage: tools/collect_r06.sh [tag]   -> writes gpurun_out/prof_<tag>/<workload>/{stats,pmc_fetch,pmc_write,pmc_sq[,pmc_tcc]}
# Every rocprofv3 line profiles `python3 bench.py` directly (no shell/env hop after `--`); --pmc passes never share a
# run with a trace.
set -u
TAG=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-verify --no-second-process"
run() {  # name, bench arguments...
  local name=$1; shift
  mkdir -p $O/$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name/stats -- $B "$@" > $O/$name/stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$name/pmc_fetch -- $B "$@" > $O/$name/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$name/pmc_write -- $B "$@" > $O/$name/pmc_write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/$name/pmc_sq -- $B "$@" > $O/$name/pmc_sq.log 2>&1
  echo "$name done: $(ls $O/$name)"
}
run C2 --configs none --steps 20 --warmup 5 --max-repeats 5
run C2_hashing --configs none --steps 20 --warmup 5 --max-repeats 5 --path partitioned
run cold_C2 --configs cold_C2 --no-headline
run north_star_k21 --configs north_star_k21 --no-headline
run C3 --configs C3 --no-headline
run C5_shard --configs C5_shard --no-headline
run C4_shard --configs C4_shard --no-headline
run C2_sub1pct --configs C2_sub1pct --no-headline
run packed_C2 --configs packed_C2 --no-headline --steps 20
# BASELINE configs[4] whole on one GPU (8 pieces into a 2^33-slot table): kernel stats only (its PMC passes would take four more 10 s runs each)
mkdir -p $O/C5_whole
rocprofv3 --kernel-trace --stats --output-format csv -d $O/C5_whole/stats -- $B --configs C5_whole --no-headline > $O/C5_whole/stats.log 2>&1
echo "C5_whole done"
# the early multi-GPU route's stages on this one GPU (tools/route_profile.py: the split for 8 owners, one owner's share of the job)
mkdir -p $O/route_C4
rocprofv3 --kernel-trace --stats --output-format csv -d $O/route_C4/stats -- python3 /root/repo/tools/route_profile.py NS --owners 8 --skip-loopback --skip-plain > $O/route_C4/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/route_C4/pmc_sq -- python3 /root/repo/tools/route_profile.py C4 --owners 8 --skip-loopback --skip-plain > $O/route_C4/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/route_C4/pmc_fetch -- python3 /root/repo/tools/route_profile.py C4 --owners 8 --skip-loopback --skip-plain > $O/route_C4/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/route_C4/pmc_write -- python3 /root/repo/tools/route_profile.py C4 --owners 8 --skip-loopback --skip-plain > $O/route_C4/pmc_write.log 2>&1
echo "route_C4 done"
# the direct path's L2-atomic counters (north_star: "L2-atomic counters")
mkdir -p $O/C2_direct
rocprofv3 --kernel-trace --stats --output-format csv -d $O/C2_direct/stats -- $B --configs none --steps 5 --warmup 2 --max-repeats 3 --path direct > $O/C2_direct/stats.log 2>&1
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/C2_direct/pmc_tcc -- $B --configs none --steps 5 --warmup 2 --max-repeats 3 --path direct > $O/C2_direct/pmc_tcc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/C2_direct/pmc_fetch -- $B --configs none --steps 5 --warmup 2 --max-repeats 3 --path direct > $O/C2_direct/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/C2_direct/pmc_write -- $B --configs none --steps 5 --warmup 2 --max-repeats 3 --path direct > $O/C2_direct/pmc_write.log 2>&1
# the bench line of the same build, un-profiled
python3 /root/repo/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
# keep what travels back small: the raw per-dispatch CSVs are summarised here
python3 /root/repo/tools/summarize_prof.py $TAG --on-box
du -sh $O; ls $O
