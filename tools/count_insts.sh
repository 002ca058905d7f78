#!/bin/bash
# Static instruction mix of ONE path (CA, no grouping, JPL) of the two 4096-point fp32 chain kernels:
# the kernels are straight-line code, so with -DRSP_COUNT_PATH static counts = dynamic counts per wave
# (1024 cells).  Runs on the CPU box (hipcc cross-compiles).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d /tmp/rspcnt.XXXX); cd $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -DRSP_COUNT_PATH --cuda-device-only -S \
  $R/rsp-chains_amd/csrc/chain1d.hip -o k.s 2>/dev/null
for K in _ZN3rsp14chain1d_kernelILi12ELb0E _ZN3rsp19chain1d_quad_kernelILi12ELb0E; do
  awk -v k="^$K.*:" '$0 ~ k {f=1} f{print} f&&/s_endpgm/{exit}' k.s > body.s
  echo "== $K"
  python3 - <<PY
import re,collections
c=collections.Counter()
for l in open("body.s"):
    m=re.match(r"\s+([a-z_0-9]+)",l)
    if not m: continue
    op=m.group(1)
    if op.startswith("v_"): c["VALU"]+=1
    elif op.startswith("ds_"): c["LDS"]+=1; c[op]+=1
    elif op.startswith(("global_","buffer_")): c["VMEM"]+=1; c[op]+=1
    elif op.startswith("s_waitcnt"): c["s_waitcnt"]+=1
    elif op.startswith("s_barrier"): c["s_barrier"]+=1
    elif op.startswith("s_"): c["SALU"]+=1
tot=sum(c[k] for k in ("VALU","LDS","VMEM","SALU","s_waitcnt","s_barrier"))
print(" total", tot, {k:c[k] for k in ("VALU","LDS","VMEM","SALU","s_waitcnt","s_barrier")})
print(" ", {k:v for k,v in c.items() if k.startswith(("ds_","global_","buffer_"))})
PY
  grep -A30 "amdhsa_kernel $K" k.s | grep -E "next_free_vgpr|group_segment"
done
rm -rf $T
