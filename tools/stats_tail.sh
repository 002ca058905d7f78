#!/bin/bash
# rocprofv3 kernel durations of the quad tail and the per-cell tail on one box (3 alternating rounds).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02/stats_tail; rm -rf $O; mkdir -p $O
for r in 1 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/q$r -- python3 $R/tools/prof_chain.py 4096 4096 40 > $O/q$r.log 2>&1
  RSP_PROF_GENERIC_TAIL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g$r -- python3 $R/tools/prof_chain.py 4096 4096 40 > $O/g$r.log 2>&1
done
python3 - <<PY
import csv,glob
for d in sorted(glob.glob("$O/*/*/*kernel_stats.csv")):
    for r in csv.DictReader(open(d)):
        if 'chain1d' in r['Name']: print(d.split('/')[-3], r['Name'][:60], 'calls', r['Calls'], 'avg_ns', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
PY
