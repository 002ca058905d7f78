cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc_insts -- python3 $R/tools/prof_chain.py 4096 4096 4 > $R/gpurun_out/pmc_insts.log 2>&1
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
for d in glob.glob(R+'/gpurun_out/pmc_insts/*/*counter_collection.csv'):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if 'chain1d' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    w=sum(agg['SQ_WAVES'])/len(agg['SQ_WAVES'])
    for k,v in sorted(agg.items()): print(k, round(sum(v)/len(v)/w,1))
PY
