#!/bin/bash
# Timing experiments: KCT_ABLATE skips parts of the kernels (results are INVALID, times only).
#   K1: 1 = no ring append, 2 = no flush, 3 = hash only      K2: 4 = no count add, 16 = loads only, 64 = no streaming
for ab in ${ABLATE_LIST:-0 1 3 16 64}; do
  echo "== ablate $ab"
  KCT_ABLATE=$ab python bench.py --steps 30 --warmup 5 --no-cpu-baseline --configs none --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['ms_per_step'], d['roofline']['kernels_ms_per_step'])
"
done
