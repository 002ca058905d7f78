#!/bin/bash
# LDS conflict / activity counters of one chain kernel: tools/pmc_lds.sh <tag> <prof_chain.py args...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/pmc_lds_$TAG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O -- python3 $R/tools/prof_chain.py "$@" > $O/run.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$O/*/*counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if 'chain1d' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print("$TAG", {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
