#!/bin/bash
mkdir -p gpurun_out/r02b
O=gpurun_out/r02b
timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -k "more_than_1024 or carry" > $O/scale_small.log 2>&1; echo "scale_small rc=$?"
timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_scale.py > $O/suite.log 2>&1; echo "suite rc=$?"
for c in C3 NS; do
  KCT_DEBUG=1 timeout 600 python tools/run_config.py $c --paths auto --no-dump > $O/cfg_$c.json 2> $O/cfg_$c.err; echo "$c rc=$?"
  KCT_K1B_LINES=1 timeout 600 python tools/run_config.py $c --paths auto --no-dump > $O/cfg_${c}_lines1.json 2> $O/cfg_${c}_lines1.err; echo "$c lines1 rc=$?"
done
KCT_DEBUG=1 timeout 600 python tools/run_config.py C4 --paths auto --no-dump > $O/cfg_C4.json 2> $O/cfg_C4.err; echo "C4 rc=$?"
KCT_DEBUG=1 timeout 600 python tools/run_config.py C2 --paths auto > $O/cfg_C2.json 2> $O/cfg_C2.err; echo "C2 rc=$?"
for f in $O/*.log; do echo "== $f"; tail -n 3 $f; done
