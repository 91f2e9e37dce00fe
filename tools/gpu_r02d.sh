#!/bin/bash
mkdir -p gpurun_out/r02d
O=gpurun_out/r02d
timeout 1700 python -m pytest tests -x -q -m gpu --durations=8 > $O/suite.log 2>&1; echo "suite rc=$?"
KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_premul.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "partitioned or every_k or example or full_size" > $O/premul_parity.log 2>&1; echo "premul parity rc=$?"
for i in 1 2; do
for lib in hip nocare premul; do
  echo "== $lib steady" >> $O/ab.txt
  KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])
" >> $O/ab.txt
  echo "== $lib hashing path" >> $O/ab.txt
  KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --configs none --path partitioned 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])
" >> $O/ab.txt
done; done
for lib in hip premul; do
  echo "== $lib C5 x0.25 / C3 x0.25 partitioned" >> $O/ab.txt
  KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python tools/run_config.py C5 --scale 0.25 --paths auto --no-dump 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['paths']['auto']; print(p['kmers_per_s'], p['kernels_ms'])" >> $O/ab.txt
  KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python tools/run_config.py C3 --scale 0.25 --paths partitioned --no-dump 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['paths']['partitioned']; print(p['kmers_per_s'], p['kernels_ms'])" >> $O/ab.txt
done
tail -n 14 $O/suite.log; tail -n 3 $O/premul_parity.log; cat $O/ab.txt
