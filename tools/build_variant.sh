#!/bin/bash
# tools/build_variant.sh <out.so> [extra hipcc flags]: a side build of the library with chain1d.hip (the fp32 kernels)
# and rspchain_api.cpp compiled with the extra flags (-DRSP_ABLATE, -DRSP_STAMP, ...: csrc/side_build.hpp), the other
# objects reused, for same-box A/B runs (tools/abm.sh, tools/ablate.sh)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/rsp-chains_amd/csrc; OUT=$1; shift
T=$(mktemp -d /tmp/rspvar.XXXX)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $C/chain1d.hip -o $T/chain1d.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -x hip -c $C/rspchain_api.cpp -o $T/rspchain_api.o &
# the FIXED16 kernels of the default (convergent) trim too
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DRSP_PART_FX=0 "$@" -c $C/chain1d.hip -o $T/chain1d_fx0.o &
wait
OBJ=$(ls $C/build/*.o | grep -v -e '/chain1d.o' -e '/chain1d_fx0.o' -e '/rspchain_api.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/$OUT $T/chain1d.o $T/chain1d_fx0.o $T/rspchain_api.o $OBJ
rm -rf $T; echo built $OUT
