#!/bin/bash
# Same-box comparison of library builds on the 2-D chain: tools/abrd.sh "<lib1> <lib2> ..." [prof_rd.py args]
LIBS=$1; shift
for round in 1 2 3; do
  for L in $LIBS; do echo -n "$L: "; RSP_CHAIN_LIB=$PWD/$L python3 tools/prof_rd.py "$@" 2>/dev/null | tail -1; done
done
