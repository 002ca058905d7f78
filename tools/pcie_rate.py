#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point rsp_chain_process (never the headline)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsp_chains_amd as R
n, frames = 4096, 4096
params = R.FftMagCfarVanillaParameters(fftParams=R.FFTParams.fixed(numPoints=n), magParams=R.MAGParams.fixed(),
                                       cfarParams=R.CFARParams(fftSize=n), dtype=R.F32)
dut = R.FftMagCfarChainVanilla(params)
dut.configure(R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging"))
x = np.tile(R.stimulus.chirp_frames(64, n, seed=1), (frames // 64, 1))
dut.stream(x)
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    dut.stream(x)
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"entry": "rsp_chain_process (pageable host buffers, H2D + kernel + D2H, synchronous)",
                  "cells_per_s": n * frames / dt, "ms_per_batch": dt * 1e3,
                  "host_bytes_per_batch": x.nbytes + 4 * n * frames}))
