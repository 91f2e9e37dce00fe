#!/bin/bash
# round 2, first GPU session: the new tests, the whole suite, then the big configurations with pass-level debug output
mkdir -p gpurun_out/r02a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 900 python -m pytest tests/test_gpu_robustness.py -x -q > gpurun_out/r02a/robust.log 2>&1; echo "robust rc=$?"
timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -k "more_than_1024 or carry" > gpurun_out/r02a/scale_small.log 2>&1; echo "scale_small rc=$?"
timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_scale.py --deselect tests/test_gpu_robustness.py > gpurun_out/r02a/suite.log 2>&1; echo "suite rc=$?"
for c in C3 NS C4 C5; do
  KCT_DEBUG=1 timeout 600 python tools/run_config.py $c --paths auto --no-dump > gpurun_out/r02a/cfg_$c.json 2> gpurun_out/r02a/cfg_$c.err; echo "$c rc=$?"
done
tail -3 gpurun_out/r02a/*.log
