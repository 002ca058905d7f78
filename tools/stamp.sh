#!/bin/bash
# Phase time stamps of the quad kernel under load: side build with -DRSP_STAMP (run tools/build_variant.sh on the CPU box
# first: tools/build_variant.sh ab_stamp.so -DRSP_STAMP), then on the GPU box: tools/stamp.sh
RSP_CHAIN_LIB=$PWD/ab_stamp.so python3 tools/prof_chain.py 4096 4096 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r02/stamp_raw.log
python3 - <<'PY'
import re,collections
rows=[l for l in open("gpurun_out/r02/stamp_raw.log") if l.startswith("stamp")]
keys=["load","p0","x1","p1","x2","p2mag","magw","scan","cells","store","total"]
acc=collections.defaultdict(list)
for l in rows:
    for k in keys:
        m=re.search(rf" {k} (\d+)",l)
        acc[k].append(int(m.group(1)))
n=len(rows)
print(f"{n} wave samples; s_memtime ticks (100 MHz = 10 ns each on gfx9 unless the counter runs at the shader clock)")
for k in keys:
    v=sorted(acc[k]); print(f"{k:7s} mean {sum(v)/len(v):9.1f}  median {v[len(v)//2]:7d}  min {v[0]:7d}  max {v[-1]:7d}")
PY
tail -2 gpurun_out/r02/stamp_raw.log
