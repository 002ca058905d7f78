cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_rd_$tag
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_rd_$tag -- python3 $R/tools/prof_rd.py ${@:-4096 512 8 3} > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/pmc_rd_*/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(d)):
        k=r['Kernel_Name']
        name='cfar2d' if 'cfar2d' in k else ('doppler' if 'doppler' in k else ('range' if 'range_fft' in k else None))
        if name: agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    for name,cs in agg.items():
        print(name, {k: round(sum(v)/len(v)) for k,v in cs.items()})
PY
