#!/bin/bash
# GPU box: start / end times of the last kernels of a bench run (overlap between streams?)   bash tools/timeline.sh [env=val ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl
for kv in "$@"; do export "$kv"; done
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 4 > /tmp/tl.log 2>&1
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("/tmp/tl/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id","?"), r.get("Stream_Id","?")))
rows.sort()
t0=rows[-60][0]
for s,e,n,q,st in rows[-60:-36]:
    print("%9.1f %9.1f  dur %6.1f  q=%s s=%s %s" % ((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,q,st,n))
PY
