#!/bin/bash
# Same-box A/B of two builds of librspchain.so: tools/ab.sh <libA> <libB> [fft chirps reps]
A=$1; B=$2; shift 2
for round in 1 2 3; do
  for L in $A $B; do echo -n "$L: "; RSP_CHAIN_LIB=$PWD/$L python3 tools/prof_chain.py "$@" | tail -1; done
done
