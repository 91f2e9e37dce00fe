#!/bin/bash
# A/B of two builds of the library on ONE box: put the baseline build at oxli_amd/csrc/libkct_base.so
# (e.g. `git stash; make -C oxli_amd/csrc; cp libkct_hip.so libkct_base.so; git stash pop; make`), then
#   gpurun -- 'bash tools/ab_bench.sh'
for i in 1 2; do
for lib in base hip; do
  echo "== $lib"
  KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])
"
done; done
