#!/usr/bin/env python3
"""Resource table (VGPRs, scratch bytes, LDS, SGPRs) of every kernel in a gfx950 assembly file:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S x.hip -o x.s && tools/kres.py x.s
Used to check that a refactor leaves the product kernels' allocations unchanged."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(rf"\.amdhsa_{k} (\d+)", body) or [0, "?"])[1]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("rsp::", "").replace("void ", "")
    print(f"{dem:60s} vgpr {g('next_free_vgpr'):>4s} sgpr {g('next_free_sgpr'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
