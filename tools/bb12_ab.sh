mkdir -p gpurun_out/ab
for i in 1 2; do
for c in C4 C5 C3; do
for lib in hip bb12a bb12; do
  KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python tools/run_config.py $c --paths auto --no-dump 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); p=d['paths']['auto']; print('$c $lib', round(p['seconds']*1e3,2), 'ms', '%.4g'%p['kmers_per_s'], p['kernels_ms'])
except Exception as e: print('$c $lib failed', e)" | tee -a gpurun_out/ab/bb12.txt
done; done; done
