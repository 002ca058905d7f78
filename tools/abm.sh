#!/bin/bash
# Careful same-box comparison: tools/abm.sh "<lib1> <lib2> ..." rounds [prof_chain.py args] -> mean / min per library
LIBS=$1; ROUNDS=$2; shift 2
declare -A ALL
for round in $(seq $ROUNDS); do
  for L in $LIBS; do
    v=$(RSP_CHAIN_LIB=$PWD/$L python3 tools/prof_chain.py "$@" 2>/dev/null | tail -1 | sed -E 's/.* ([0-9.]+) us\/launch.*/\1/')
    ALL[$L]="${ALL[$L]} $v"
  done
done
for L in $LIBS; do echo "$L:${ALL[$L]}" | python3 -c "
import sys
for line in sys.stdin:
    name, vals = line.split(':'); v = sorted(map(float, vals.split()))
    print(f'{name:16s} mean {sum(v)/len(v):6.2f}  median {v[len(v)//2]:6.2f}  min {v[0]:6.2f}  max {v[-1]:6.2f}  n={len(v)}')
"; done
