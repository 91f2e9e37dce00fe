#!/bin/bash
# where the sender's split kernel spends its time: tools/route_profile.py's split launches under rocprofv3 --stats with phases of
# split_superkmers_kernel skipped (KCT_ABLATE of a -DKCT_DEBUG_ENV build: 0x100 no emit, 0x200 cursors only, 0x400 emit but no flush)
O=$GRAFT_REPO_ROOT/gpurun_out/split_ablate
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export KCT_LIB_PATH=/root/repo/oxli_amd/csrc/libkct_dbg.so
for A in 0 256 512 1024; do
  export KCT_ABLATE=$A
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/a$A -- python3 /root/repo/tools/route_profile.py NS --owners 8 --skip-loopback --skip-plain --split-only > $O/a$A.log 2>&1
done
python3 - "$O" <<'PY'
import csv,glob,sys
for a in (0,256,512,1024):
    for f in glob.glob(f"{sys.argv[1]}/a{a}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "split_superkmers" in r["Name"] or "gather_units" in r["Name"]:
                print(a, r["Name"][:40], r["Calls"], "total ms", round(float(r["TotalDurationNs"])/1e6,2), "max", round(float(r["MaxNs"])/1e6,3))
PY
