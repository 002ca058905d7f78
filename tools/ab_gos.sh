#!/bin/bash
# GPU box: same-box A/B of library builds on the OS-CFAR kernel (cfg 4 shape)
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for L in variants/abl/*.so; do echo -n "$(basename $L): "; RSP_CHAIN_LIB=$PWD/$L python3 tools/prof_chain.py ${1:-8192} ${2:-2048} 30 f32 gos 2>/dev/null | tail -1; done
done
