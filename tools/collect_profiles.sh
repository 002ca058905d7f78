#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel stats + PMC traffic passes.
# Output under gpurun_out/r01/; tools/summarise_profiles.py turns it into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 60 --warmup 8 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 60 --warmup 8 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -- python3 $R/tools/prof_chain.py 4096 4096 8 > $O/pmc_$tag.log 2>&1
done
python3 $R/tools/pcie_rate.py > $O/pcie.json 2> $O/pcie.err
for w in cfg3 cfg4 cfg5; do python3 $R/bench.py --workload $w --steps 20 --warmup 4 > $O/bench_$w.json 2> $O/bench_$w.err; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg3 -- python3 $R/bench.py --workload cfg3 --steps 20 --warmup 4 > /dev/null 2> $O/stats_cfg3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -- python3 $R/bench.py --workload cfg4 --steps 20 --warmup 4 > /dev/null 2> $O/stats_cfg4.err
echo done
