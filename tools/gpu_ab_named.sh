#!/bin/bash
# A/B of named builds (libkct_<name>.so) on one box: steady C2, then the given big configurations.
#   gpurun -- 'bash tools/gpu_ab_named.sh "hip half" NS C3'
libs="$1"; shift
mkdir -p gpurun_out/ab
for i in 1 2; do for lib in $libs; do
  KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C2 $lib', '%.4g'%d['value'], round(d['ms_per_step'],4), d['roofline']['kernels_ms_per_step'])" | tee -a gpurun_out/ab/ab_named.txt
done; done
for i in 1 2; do for c in "$@"; do for lib in $libs; do
  KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python tools/run_config.py $c --paths auto --no-dump 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['paths']['auto']; print('$c $lib', round(p['seconds']*1e3,2), 'ms', '%.4g'%p['kmers_per_s'], p['kernels_ms'])" | tee -a gpurun_out/ab/ab_named.txt
done; done; done
