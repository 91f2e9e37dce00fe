for k in 21 25 31 32 41 51 63; do
  echo "== k $k"
  python bench.py --k $k --steps 20 --warmup 3 --no-cpu-baseline --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])
"
done
