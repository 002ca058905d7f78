#!/bin/bash
# Phase ablation of a chain kernel (side build: tools/build_variant.sh ab_ablate.so -DRSP_ABLATE); arguments = those of
# tools/prof_chain.py, default the fp32 4096-point quad kernel.  FIXED16: bits 0 / 3 / 4 act on the fixed-point FFT and
# magnitude; GOS kernel: bit 1 = the order-statistic stage.
# Mask bits (csrc/side_build.hpp): 1 butterflies, 2 CFAR cells, 4 scan, 8 FFT exchanges, 16 magnitude,
# 32 loads from 64 L2-resident frames, 64 no word stores.  96 = the kernel's compute alone; 96 + x = compute without x.
for m in 0 32 64 96 97 98 100 104 112 99 105 127; do echo -n "mask=$m: "; RSP_ABLATE_MASK=$m RSP_CHAIN_LIB=$PWD/ab_ablate.so python3 tools/prof_chain.py ${@:-4096 4096 100} 2>/dev/null | tail -1; done
