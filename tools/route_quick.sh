#!/bin/bash
# per-kernel times of the early route's stages on one GPU (tools/route_profile.py under rocprofv3 --stats): tools/route_quick.sh <tag>
T=${1:-q}
O=$GRAFT_REPO_ROOT/gpurun_out/route_$T
rm -rf "$O"; mkdir -p "$O"
python3 -m pytest /root/repo/tests/test_gpu_route.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 /root/repo/tools/route_profile.py NS --owners 8 --skip-loopback --skip-plain > $O/stats.log 2>&1
tail -4 $O/stats.log
python3 - "$O" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/stats/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e6,3), "ms avg", round(float(r["MinNs"])/1e6,3), round(float(r["MaxNs"])/1e6,3))
PY
