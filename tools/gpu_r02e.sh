#!/bin/bash
mkdir -p gpurun_out/r02e
O=gpurun_out/r02e
timeout 900 python -m pytest tests/test_gpu_robustness.py -x -q > $O/robust.log 2>&1; echo "robust rc=$?"
timeout 1700 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_robustness.py --durations=5 > $O/suite.log 2>&1; echo "suite rc=$?"
for c in NS C3 C4 C5; do
  KCT_DEBUG=1 timeout 600 python tools/run_config.py $c --paths auto --no-dump > $O/cfg_$c.json 2> $O/cfg_$c.err; echo "$c rc=$?"
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --configs cold_C2,e2e_C2 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
tail -n 12 $O/robust.log; tail -n 12 $O/suite.log
for c in NS C3 C4 C5; do grep -v amdgpu $O/cfg_$c.err | cut -c1-260; python -c "
import json;d=json.load(open('$O/cfg_$c.json'));p=d['paths']['auto'];print('$c', d['kmers'], round(p['seconds'],4), round(p['seconds_first_call'],3), '%.3g'%p['kmers_per_s'], p['kernels_ms'])"; done
python -c "
import json;d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['repeats'],d['roofline']['kernels_ms_per_step']);print(json.dumps(d['configs'])[:1500])"
