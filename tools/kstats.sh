#!/bin/bash
# GPU box: per-kernel rocprof averages of a python command:  bash tools/kstats.sh tools/prof_rd.py 4096 512 8 20 fused
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kstats
S=$1; shift
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python3 $R/$S "$@" > /tmp/kstats.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/kstats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "rsp::" in r["Name"]: print("  ", r["Name"][:64].ljust(64), r["Calls"].rjust(5), "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), "min_us %.1f" % (float(r["MinNs"]) / 1e3))
PY
