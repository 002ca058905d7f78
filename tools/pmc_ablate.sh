# LDS conflict / activity counters per ablation mask (needs ab_ablate.so built with -DRSP_ABLATE)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_abl; rm -rf $O; mkdir -p $O
for m in 0 8 4 2 14; do
  RSP_ABLATE_MASK=$m RSP_CHAIN_LIB=$R/ab_ablate.so rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL --output-format csv -d $O/m$m -- python3 $R/tools/prof_chain.py 4096 4096 4 > $O/m$m.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$O/m*/*/*counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if 'chain1d' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d.split('/')[-3], {k: round(sum(v)/len(v)) for k,v in sorted(agg.items())})
PY
