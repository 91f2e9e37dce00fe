#!/bin/bash
# Round 6: the wave-specialised K1's flusher knobs on one box (libkct_dbg.so: make variant V=dbg EXTRA=-DKCT_DEBUG_ENV):
# KCT_ABLATE = (pause + 1) << 8 [| 1 << 16: flushers at priority 0], for 2 and 4 flusher waves; C2 headline, K1 / overflow merge per step.
export KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_dbg.so
run() {
KCT_K1_FLUSHERS=$1 KCT_ABLATE=$2 timeout 300 python bench.py --configs none --no-cpu-baseline --no-second-process --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['roofline']['kernels_ms_per_step']
print('flushers=$1 ablate=$2 C2 value %.4g ms/step %.4f' % (d['value'], d['ms_per_step']), {x: k[x] for x in k if 'partition' in x or 'overflow' in x or 'blocks32' in x})"
}
run 0 0
for f in 4 2; do
for ab in 256 1280 2304 4352 8448 16640 $((4352+65536)) $((256+65536)); do run $f $ab; done
done
run 0 0
