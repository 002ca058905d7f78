#!/bin/bash
# side build with a differently compiled rd2d.hip: bv_rd.sh out.so flags...
set -e
R=/root/repo; C=$R/rsp-chains_amd/csrc; OUT=$1; shift
T=$(mktemp -d /tmp/rspvar.XXXX)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $C/rd2d.hip -o $T/rd2d.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/$OUT $C/build/chain1d.o $C/build/chain1d_fx0.o $C/build/chain1d_fx1.o $C/build/chain1d_fx2.o $C/build/compact.o $T/rd2d.o $C/build/stimulus.o $C/build/small.o $C/build/rspchain_api.o
rm -rf $T; echo built $OUT
