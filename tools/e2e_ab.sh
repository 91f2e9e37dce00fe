for i in 1 2; do
for lib in base hip; do
KCT_LIB_PATH=$PWD/oxli_amd/csrc/libkct_$lib.so python bench.py --no-headline --configs e2e_C2 --no-cpu-baseline --verbose 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['configs']['e2e_C2']
print('$lib', {k:(round(v['kmers_per_s']/1e10,3), round(v['call_ms'],2), round(v['host_side_ms'],2)) for k,v in d['variants'].items()}, d['gate'])"
done; done
