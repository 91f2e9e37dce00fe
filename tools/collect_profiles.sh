#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/collect_profiles.sh <tag>     -> writes gpurun_out/prof_<tag>/...
# Each rocprofv3 line profiles `python3 bench.py` directly (no shell/env hop after `--`).
set -u
TAG=${1:-r01}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 10 --warmup 2 > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- $B --steps 3 --warmup 1 > $O/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- $B --steps 3 --warmup 1 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lds -- $B --steps 3 --warmup 1 > $O/pmc_lds.log 2>&1
# the bench line of the same build, un-profiled
python3 /root/repo/bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
ls -R $O | head -40
