#!/bin/bash
# tools/build_variant.sh <out.so> [extra hipcc flags for chain1d.hip]: a side build of the library
# with a differently-compiled chain1d.hip (other objects reused), for same-box A/B via tools/ab.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/rsp-chains_amd/csrc; OUT=$1; shift
T=$(mktemp -d /tmp/rspvar.XXXX)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $C/chain1d.hip -o $T/chain1d.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/$OUT $T/chain1d.o $C/build/chain1d_fx0.o $C/build/chain1d_fx1.o $C/build/chain1d_fx2.o $C/build/compact.o $C/build/rd2d.o $C/build/stimulus.o $C/build/small.o $C/build/rspchain_api.o
rm -rf $T; echo built $OUT
