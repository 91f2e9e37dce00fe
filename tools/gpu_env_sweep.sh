#!/bin/bash
# one big configuration under several settings of ONE environment switch (a -DKCT_DEBUG_ENV build: libkct_dbg.so), one box:
#   gpurun -- 'bash tools/gpu_env_sweep.sh KCT_K1_SHARE "100 60 70 75 80" C3 C5'
VAR=$1; VALS=$2; shift 2
mkdir -p gpurun_out/ab
export KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_dbg.so
for c in "$@"; do
for v in $VALS; do
  env $VAR=$v python tools/run_config.py $c --paths auto --no-dump 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['paths']['auto']; print('$c $VAR=$v', round(p['seconds']*1e3,2), 'ms', '%.4g'%p['kmers_per_s'], p['kernels_ms'])" | tee -a gpurun_out/ab/env_sweep.txt
done; done
