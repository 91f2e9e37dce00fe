#!/bin/bash
# the big configurations under different environment settings, one box:  gpurun -- 'bash tools/gpu_env_cfg.sh "KCT_K1B_LINES=4" NS C3'
mkdir -p gpurun_out/ab
ENVSET=$1; shift
for i in 1 2; do
for c in "$@"; do
for e in "X=1" "$ENVSET"; do
  env $e python tools/run_config.py $c --paths auto --no-dump 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['paths']['auto']; print('$c $e', round(p['seconds']*1e3,2), 'ms', '%.4g'%p['kmers_per_s'], p['kernels_ms'])" | tee -a gpurun_out/ab/env_cfg.txt
done; done; done
