#!/bin/bash
# Runs on the GPU box (via gpurun): everything under profiles/ for one round, in ONE call.
#   bench line (+ the same command under rocprofv3 --kernel-trace --stats), ONE rocprofv3 --kernel-trace --stats run PER
#   SHAPE for the extras (so that profiles/ has one row per kernel and shape), separate --pmc passes (HBM traffic, SQ / LDS
#   counters, GRBM) of the headline kernel, the 2-D chain at both shapes, cfg 4 and the FIXED16 chain, the phase ablation of
#   the headline kernel (side library ab_ablate.so: tools/build_variant.sh ab_ablate.so -DRSP_ABLATE on the CPU box first),
#   the instruction-cost microbenchmark, the 1-rank RCCL rehearsal of the N > 1 line and the PCIe-inclusive host entry.
# Output under gpurun_out/<round>/collect (gpurun MERGES it into the local copy: delete the local directory first, or old
# runs' files are summarised too); tools/summarise_profiles.py turns it into profiles/ (tracked).
RND=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$RND/collect
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 60 --warmup 8 > $O/bench.json 2> $O/bench.err
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 60 --warmup 8 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
# one kernel-stats run per shape (program directly after --)
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg2 -- python3 $R/tools/prof_chain.py 4096 4096 200 > $O/stats_cfg2.log 2>&1
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg3 -- python3 $R/tools/prof_rd.py 4096 512 8 100 fused > $O/stats_cfg3.log 2>&1
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg5 -- python3 $R/tools/prof_rd.py 8192 1024 8 40 fused > $O/stats_cfg5.log 2>&1
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -- python3 $R/tools/prof_chain.py 8192 2048 100 f32 gos > $O/stats_cfg4.log 2>&1
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fx1024 -- python3 $R/tools/prof_chain.py 1024 16384 100 fx16 > $O/stats_fx1024.log 2>&1
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fx4096 -- python3 $R/tools/prof_chain.py 4096 4096 100 fx16 > $O/stats_fx4096.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -- python3 $R/tools/prof_chain.py 4096 4096 8 > $O/pmc_$tag.log 2>&1
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcrd3_$tag -- python3 $R/tools/prof_rd.py 4096 512 8 4 fused > $O/pmcrd3_$tag.log 2>&1
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcrd5_$tag -- python3 $R/tools/prof_rd.py 8192 1024 8 3 fused > $O/pmcrd5_$tag.log 2>&1
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcgos_$tag -- python3 $R/tools/prof_chain.py 8192 2048 8 f32 gos > $O/pmcgos_$tag.log 2>&1
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcfx1_$tag -- python3 $R/tools/prof_chain.py 1024 16384 8 fx16 > $O/pmcfx1_$tag.log 2>&1
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcfx4_$tag -- python3 $R/tools/prof_chain.py 4096 4096 8 fx16 > $O/pmcfx4_$tag.log 2>&1
done
cd $R
[ -f ab_ablate.so ] && bash tools/ablate.sh > $O/ablate.txt 2>&1
[ -f ab_ablate.so ] && bash tools/ablate.sh 4096 4096 100 fx16 > $O/ablate_fx4096.txt 2>&1
[ -f ab_ablate.so ] && bash tools/ablate.sh 8192 2048 100 f32 gos > $O/ablate_gos.txt 2>&1
[ -f ab_ablate.so ] && bash tools/ablate.sh 8192 2048 100 f32 > $O/ablate_ca8192.txt 2>&1
[ -x tools/valubench ] && timeout -k 5 120 tools/valubench > $O/valubench.txt 2>&1
cd /tmp
RSP_BENCH_FORCE_DIST=1 python3 $R/bench.py --steps 20 --warmup 4 > $O/bench_1rank_rccl.json 2> $O/bench_1rank_rccl.err
python3 $R/tools/pcie_rate.py > $O/pcie.json 2> $O/pcie.err
echo done
