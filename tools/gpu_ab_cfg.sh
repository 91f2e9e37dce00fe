#!/bin/bash
# A/B of two builds (libkct_base.so vs libkct_hip.so) on the big configurations, one box:  gpurun -- 'bash tools/gpu_ab_cfg.sh NS C3'
mkdir -p gpurun_out/ab
for i in 1 2; do
for c in "$@"; do
for lib in base hip; do
  KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python tools/run_config.py $c --paths auto --no-dump 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['paths']['auto']; print('$c $lib', round(p['seconds']*1e3,2), 'ms', '%.4g'%p['kmers_per_s'], p['kernels_ms'])" | tee -a gpurun_out/ab/ab_cfg.txt
done; done; done
