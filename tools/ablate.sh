#!/bin/bash
# phase ablation of chain1d (build with -DRSP_ABLATE): bit0 butterflies, bit1 CFAR cells, bit2 scan,
# bit3 LDS exchange traffic, bit4 magnitude math
for m in 0 1 2 4 8 16 3 9 25 27 31; do echo -n "mask=$m: "; RSP_ABLATE_MASK=$m RSP_CHAIN_LIB=$PWD/ab_ablate.so python3 tools/prof_chain.py 4096 4096 30 | tail -1; done
