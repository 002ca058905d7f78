#!/bin/bash
# GPU box: same-box A/B of every variants/abl/*.so on the 1-D chain:  bash tools/ab_libs.sh <n> <frames> [dtype] [gos]
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for L in variants/abl/*.so; do echo -n "$(basename $L): "; RSP_CHAIN_LIB=$PWD/$L python3 tools/prof_chain.py $1 $2 30 ${3:-f32} $4 2>/dev/null | tail -1; done
done
