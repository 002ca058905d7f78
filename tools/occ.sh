#!/bin/bash
for ex in 0 4096 16384 45000 90000; do echo -n "extra_lds=$ex: "; RSP_DEBUG_EXTRA_LDS=$ex python3 tools/prof_chain.py 4096 4096 30 | tail -1; done
