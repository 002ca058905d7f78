#!/bin/bash
# GPU box: per-kernel rocprof averages of every ablation build (tools/ablate_rd.sh); args = prof_rd.py args
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in $R/variants/abl/*.so; do
  echo "== $(basename $L)"
  rm -rf /tmp/abl_stats
  RSP_CHAIN_LIB=$L timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_stats -- python3 $R/tools/prof_rd.py "$@" > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/abl_stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "rsp::" in r["Name"]: print("  ", r["Name"][:48].ljust(48), r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), "min_us %.1f" % (float(r["MinNs"]) / 1e3))
PY
done
