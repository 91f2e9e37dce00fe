#!/bin/bash
# ASCII headline and packed_C2 (K1 per step, overall rate) for two builds on one box: bash tools/packed_ab.sh base hip
for i in 1 2; do
for lib in "$@"; do
KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python bench.py --configs packed_C2 --no-cpu-baseline --no-second-process --steps 20 --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
p=d['configs']['packed_C2']; r=d['roofline']['kernels_ms_per_step']
k1=[v for k,v in p['kernels_ms'].items() if 'partition_windows' in k][0]/p['steps']
print('$lib ascii value %.4g K1/step %.4f | packed %.4g K1/step %.4f' % (d['value'], [v for k,v in r.items() if 'partition_windows' in k][0], p['kmers_per_s'], k1))"
done; done
