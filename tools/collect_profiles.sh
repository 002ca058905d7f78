#!/bin/bash
# Runs on the GPU box (via gpurun): the round's bench line, rocprofv3 kernel stats of the SAME command, separate
# --pmc passes (HBM traffic, SQ / LDS counters) of the headline kernel and of the 2-D chain, the 1-rank RCCL
# rehearsal of the multi-GPU (cfg5) line and the PCIe-inclusive rate.  Output under gpurun_out/<round>/ (gpurun MERGES
# it into the local copy: delete the local gpurun_out/<round>/collect first, or old runs' files are summarised too);
# tools/summarise_profiles.py turns it into profiles/ (tracked).
RND=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$RND/collect
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 60 --warmup 8 > $O/bench.json 2> $O/bench.err
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 60 --warmup 8 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -- python3 $R/tools/prof_chain.py 4096 4096 8 > $O/pmc_$tag.log 2>&1
  RSP_PROF_GENERIC_TAIL=1 timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcgen_$tag -- python3 $R/tools/prof_chain.py 4096 4096 8 > $O/pmcgen_$tag.log 2>&1
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcrd3_$tag -- python3 $R/tools/prof_rd.py 4096 512 8 4 fused > $O/pmcrd3_$tag.log 2>&1
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcrd5_$tag -- python3 $R/tools/prof_rd.py 8192 1024 8 3 fused > $O/pmcrd5_$tag.log 2>&1
done
RSP_BENCH_FORCE_DIST=1 python3 $R/bench.py --steps 8 --warmup 2 > $O/bench_cfg5_1rank_rccl.json 2> $O/bench_cfg5.err
python3 $R/tools/pcie_rate.py > $O/pcie.json 2> $O/pcie.err
echo done
