for lib in base hip; do
for cfg in "e2e_C2" "cold_C2,packed_C2,e2e_C2"; do
KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python bench.py --configs $cfg --no-cpu-baseline --no-second-process --steps 20 --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['configs']['e2e_C2']
v=d.get('variants')
print('$lib', '$cfg', {k:(round(x['kmers_per_s']/1e10,3), round(x['call_ms'],2)) for k,x in v.items()} if v else round(d['kmers_per_s']/1e10,3))"
done; done
