#!/bin/bash
# Round 6: the wave-specialised K1 (KCT_K1_FLUSHERS = 2 / 4) against the barrier-synchronised one (0) on ONE box: C2 headline twice
# each, and with NS the north-star run / C3 / C5's shard once each.    gpurun -- 'bash tools/k1ws_ab.sh [NS]'
for i in 1 2; do
for f in 0 4 2; do
KCT_K1_FLUSHERS=$f timeout 600 python bench.py --configs none --no-cpu-baseline --no-second-process --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('flushers=$f C2 value %.4g ms/step %.4f' % (d['value'], d['ms_per_step']), d['roofline']['kernels_ms_per_step'])"
done; done
if [ -n "$1" ]; then
for cfg in NS C3 C5 C4; do
for f in 0 4 2; do
KCT_K1_FLUSHERS=$f timeout 900 python tools/run_config.py $cfg --paths auto --no-dump 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['paths']['auto']; print('flushers=$f $cfg', round(p['seconds']*1e3,2), 'ms', '%.4g'%p['kmers_per_s'], p['kernels_ms'])"
done; done
fi
