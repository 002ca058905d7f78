#!/bin/bash
# Same-box A/B of an experimental chain kernel (side library of tools/build_experiment.sh) against the product kernel:
# tools/ab_experiment.sh <ab_name.so> [rounds reps]; also checks that the experiment's words are bit-identical.
LIB=$1; ROUNDS=${2:-3}; REPS=${3:-200}
RSP_CHAIN_LIB=$PWD/$LIB python3 -m pytest tools/experiments/test_gpu_experiments.py -x -q 2>&1 | tail -2
for round in $(seq $ROUNDS); do
  echo -n "experiment: "; RSP_CHAIN_LIB=$PWD/$LIB RSP_PROF_EXPERIMENT=1 python3 tools/prof_chain.py 4096 4096 $REPS 2>/dev/null | tail -1
  echo -n "product:    "; RSP_CHAIN_LIB=$PWD/$LIB python3 tools/prof_chain.py 4096 4096 $REPS 2>/dev/null | tail -1
done
