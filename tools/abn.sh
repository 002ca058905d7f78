#!/bin/bash
# Same-box comparison of N builds: tools/abn.sh "<lib1> <lib2> ..." [prof_chain.py args]
LIBS=$1; shift
for round in 1 2 3; do
  for L in $LIBS; do echo -n "$L: "; RSP_CHAIN_LIB=$PWD/$L python3 tools/prof_chain.py "$@" | tail -1; done
done
