#!/bin/bash
# quad tail vs per-cell tail of the 1-D chain, alternating rounds on one box: tools/ab_tail.sh rounds [prof_chain.py args]
ROUNDS=${1:-5}; shift
for r in $(seq $ROUNDS); do
  echo -n "quad:    "; python3 tools/prof_chain.py "$@" | tail -1
  echo -n "generic: "; RSP_PROF_GENERIC_TAIL=1 python3 tools/prof_chain.py "$@" | tail -1
done
