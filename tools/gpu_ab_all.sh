#!/bin/bash
# base vs hip builds: the headline (steady C2), then the big configurations
for i in 1 2; do for lib in base hip; do
  KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C2 $lib', '%.4g'%d['value'], round(d['ms_per_step'],4), d['roofline']['kernels_ms_per_step'])"
done; done
bash tools/gpu_ab_cfg.sh "$@"
