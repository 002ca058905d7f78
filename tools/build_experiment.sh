#!/bin/bash
# tools/build_experiment.sh <name> [hipcc flags]: side library ab_<name>.so = the product objects + the experimental
# kernel tools/experiments/chain1d_<name>.hip (ldsdma | regprefetch), which defines the weak hook rsp_experiment_chain1d
# that launch_chain1d takes when RSP_OPT_EXPERIMENT is set.  Run on the CPU box; then tools/ab_experiment.sh on the GPU box.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/rsp-chains_amd/csrc; NAME=$1; shift
T=$(mktemp -d /tmp/rspexp.XXXX)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $R/tools/experiments/chain1d_$NAME.hip -o $T/exp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab_$NAME.so $T/exp.o $C/build/*.o
rm -rf $T; echo built ab_$NAME.so
