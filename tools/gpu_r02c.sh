#!/bin/bash
mkdir -p gpurun_out/r02c
O=gpurun_out/r02c
timeout 1700 python -m pytest tests -x -q -m gpu --durations=15 > $O/suite.log 2>&1; echo "suite rc=$?"
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -n 25 $O/suite.log
tail -n 30 $O/bench.err
