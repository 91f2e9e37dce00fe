#!/bin/bash
# standard partitioned path vs dedupe-first path on C2-shaped input, several k (one box)
for k in ${KS:-21 25 31 32}; do
for p in partitioned dedupe; do
  echo "== k $k $p"
  python bench.py --k $k --steps 30 --warmup 5 --no-cpu-baseline --configs none --path $p 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])
"
done; done
