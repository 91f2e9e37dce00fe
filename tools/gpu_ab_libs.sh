#!/bin/bash
# several builds of the library on one box: headline (steady C2) and the hashing path.  gpurun -- 'bash tools/gpu_ab_libs.sh hip u2 u8 u16'
for i in 1 2; do for lib in "$@"; do
  for path in auto partitioned; do
  KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --configs none --path $path 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('C2 $lib $path', '%.4g'%d['value'], round(d['ms_per_step'],4), d['roofline']['kernels_ms_per_step'])"
  done
done; done
