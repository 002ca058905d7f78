#!/usr/bin/env python3
"""Minimal driver for rocprofv3: runs the cfg-2 chain kernel (and nothing else from torch)
a few times on rotating buffers.  Usage: rocprofv3 ... -- python3 tools/prof_chain.py [fft chirps reps dtype]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsp_chains_amd as R  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dtype = R.F32 if (len(sys.argv) <= 4 or sys.argv[4] == "f32") else R.FIXED16
gos = len(sys.argv) > 5 and sys.argv[5] == "gos"
params = R.FftMagCfarVanillaParameters(fftParams=R.FFTParams.fixed(numPoints=n), magParams=R.MAGParams.fixed(),
                                       cfarParams=R.CFARParams(fftSize=n, CFARAlgorithm=R.GOSCFARType if gos else R.CACFARType), dtype=dtype)
rt = (R.RunTimeRspChainParams(fftSize=n, CFARMode="Greatest Of", refWindowSize=32, guardWindowSize=4, divSum=None, indexLagg=24, indexLead=24)
      if gos else R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging", refWindowSize=32, guardWindowSize=4, divSum=5))
dut = R.FftMagCfarChainVanilla(params)
dut.configure(rt)
if os.environ.get("RSP_PROF_GENERIC_TAIL"):   # A/B: the per-cell tail instead of the quad tail
    dut.set_option(dut.FORCE_GENERIC_TAIL, 1)
if os.environ.get("RSP_PROF_EXPERIMENT"):     # A/B: the experimental kernel of a side library (tools/build_experiment.sh)
    dut.set_option(dut.EXPERIMENT, 1)
sets = 4
if dtype == R.F32:
    x = R.stimulus.chirp_frames(64, n, seed=1234)
    x = np.tile(x, (frames // 64, 1))
else:
    x = np.tile(R.stimulus.formAXI4StreamComplexData(
        R.stimulus.getComplexTones(n, 0.125, 0.25, 0.5, shiftRangeFactor=12)), (frames, 1))
ins, outs = [], []
for s in range(sets):
    b = R.DeviceBuffer(x.nbytes); b.upload(x); ins.append(b)
    outs.append(R.DeviceBuffer(frames * n * 4))
for i in range(reps):
    dut.process_device(ins[i % sets].ptr, frames, outs[i % sets].ptr)
dut.synchronize()
dut.timer_start()
for i in range(reps):
    dut.process_device(ins[i % sets].ptr, frames, outs[i % sets].ptr)
ms = dut.timer_stop() / reps
cells = n * frames
bpc = (12 if dtype == R.F32 else 8)
print(f"chain1d n={n} frames={frames} {'gos' if gos else 'ca'} {'f32' if dtype == R.F32 else 'fx16'} {ms*1e3:.1f} us/launch  {cells/ms/1e6:.1f} Gcells/s  {cells*bpc/ms/1e9:.2f} TB/s algorithmic")
