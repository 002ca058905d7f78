#!/bin/bash
# Per-phase cycle counters of chain1d_wave_kernel: side build with -DRSP_WAVE_PROF (run on the CPU box), then
#   gpurun -- 'RSP_CHAIN_LIB=$PWD/ab_wprof.so python tools/prof_chain.py 4096 4096 2'
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/rsp-chains_amd/csrc; T=$(mktemp -d /tmp/v.XXXX)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DRSP_WAVE_PROF "$@" -c $C/chain1d_wave.hip -o $T/w.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab_wprof.so $C/build/chain1d.o $T/w.o $C/build/rd2d.o $C/build/stimulus.o $C/build/small.o $C/build/rspchain_api.o
rm -rf $T; echo built ab_wprof.so
