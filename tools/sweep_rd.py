#!/usr/bin/env python3
"""Times the 2-D chain (8 x 4096 x 512) over its CFAR code paths: strip walker (compile-time windows 8/2 x 8/2) in
CA / GO / SO, the tiled kernel for run-time windows, both data types.  tools/sweep_rd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rsp_chains_amd as R
nr, nd, n_ch = 4096, 512, 8
rng = np.random.default_rng(2345)
xf = np.tile((0.05 * (rng.standard_normal((nd, nr)) + 1j * rng.standard_normal((nd, nr)))).astype(np.complex64), (n_ch, 1, 1))
xq = ((np.rint(xf.real * 2e4).astype(np.int64) & 0xffff) << 16 | (np.rint(xf.imag * 2e4).astype(np.int64) & 0xffff)).astype(np.uint32)
for dtype, x, name in ((R.F32, xf, "fp32"), (R.FIXED16, xq, "FIXED16")):
    for mode in ("Cell Averaging", "Greatest Of", "Smallest Of"):
        for (rr, gr, rd, gd, tag) in ((8, 2, 8, 2, "walker"), (4, 1, 6, 1, "tiled kernel")):
            params = R.FftMagCfarVanillaParameters(
                fftParams=R.FFTParams.fixed(numPoints=nr), magParams=R.MAGParams.fixed(),
                cfarParams=R.CFARParams(fftSize=nr, leadLaggWindowSize=16), dtype=dtype, dopplerPoints=nd, refDoppler=rd, guardDoppler=gd)
            rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode=mode, refWindowSize=rr, guardWindowSize=gr,
                                         divSum=8 if dtype == R.FIXED16 else 4, thresholdScaler=6.0)
            dut = R.FftMagCfarChainVanilla(params); dut.configure(rt)
            b = R.DeviceBuffer(x.nbytes); b.upload(x); o = R.DeviceBuffer(x.size * 4)
            lst, cnt = R.DeviceBuffer((1 << 16) * 16), R.DeviceBuffer(8)
            for _ in range(3): dut.process_detect_device(b.ptr, n_ch, o.ptr, lst.ptr, 1 << 16, cnt.ptr)
            dut.synchronize(); dut.timer_start()
            for _ in range(20): dut.process_detect_device(b.ptr, n_ch, o.ptr, lst.ptr, 1 << 16, cnt.ptr)
            us = dut.timer_stop() / 20 * 1e3
            print(f"{name:8s} {mode:15s} {tag:13s}: {us:7.1f} us per batch, detections {cnt.download(np.uint32, 2).tolist()}", flush=True)
            del dut, b, o, lst, cnt
