#!/bin/bash
# headline (C2 steady state: value, K1 / K2 per step) and, with NS, the north-star run for two or more builds on one box:
#   gpurun -- 'bash tools/ab_quick.sh "base hip" [NS]'
for i in 1 2; do
for lib in $1; do
KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python bench.py --configs none --no-cpu-baseline --no-second-process --steps 20 --warmup 5 --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib C2 value %.4g ms/step %.4f' % (d['value'], d['ms_per_step']), d['roofline']['kernels_ms_per_step'])"
[ -n "$2" ] && KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python tools/run_config.py NS --paths auto --no-dump 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['paths']['auto']; print('$lib NS', round(p['seconds']*1e3,2), 'ms', '%.4g'%p['kmers_per_s'], p['kernels_ms'])"
done; done
