#!/usr/bin/env python3
"""Times the 1-D chain over the modes that have their own code paths (per 16.7 M cells): a sweep for anomalies.
tools/sweep_modes.py [f32|fixed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rsp_chains_amd as R
dtype = R.FIXED16 if (len(sys.argv) > 1 and sys.argv[1] != "f32") else R.F32
CASES = [  # name, n, params kw, runtime kw
    ("CA R32 G4 (quad)", 4096, dict(), dict(CFARMode="Cell Averaging", refWindowSize=32, guardWindowSize=4, divSum=5)),
    ("GO R32 G4 grouping", 4096, dict(), dict(CFARMode="Greatest Of", refWindowSize=32, guardWindowSize=4, divSum=5, peakGrouping=1)),
    ("CA R8 G4 (short windows)", 4096, dict(), dict(CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=4, divSum=3)),
    ("CA R32 G3 (per-cell tail)", 4096, dict(), dict(CFARMode="Cell Averaging", refWindowSize=32, guardWindowSize=3, divSum=5)),
    ("CASH R32 sub 8", 4096, dict(includeCASH=True), dict(CFARMode="CASH", refWindowSize=32, guardWindowSize=2, subWindowSize=8, divSum=3)),
    ("CA wrap edges, log mag", 4096, dict(edgeMode="wrap"), dict(CFARMode="Cell Averaging", refWindowSize=32, guardWindowSize=4, divSum=5, magMode=1, logOrLinearMode=0)),
    ("CA sendCut", 4096, dict(sendCut=True), dict(CFARMode="Cell Averaging", refWindowSize=32, guardWindowSize=4, divSum=5)),
    ("GOS R32 k24", 4096, dict(alg=R.GOSCFARType), dict(CFARMode="Greatest Of", refWindowSize=32, guardWindowSize=4, divSum=None, indexLagg=24, indexLead=24)),
    ("GOS R32 k20/9", 4096, dict(alg=R.GOSCFARType), dict(CFARMode="Greatest Of", refWindowSize=32, guardWindowSize=4, divSum=None, indexLagg=20, indexLead=9)),
    ("GOS R64 k48", 4096, dict(alg=R.GOSCFARType), dict(CFARMode="Greatest Of", refWindowSize=64, guardWindowSize=4, divSum=None, indexLagg=48, indexLead=48)),
    ("GOS R16 k12", 4096, dict(alg=R.GOSCFARType), dict(CFARMode="Greatest Of", refWindowSize=16, guardWindowSize=4, divSum=None, indexLagg=12, indexLead=12)),
    ("GOS R8 k6", 1024, dict(alg=R.GOSCFARType), dict(CFARMode="Greatest Of", refWindowSize=8, guardWindowSize=2, divSum=None, indexLagg=6, indexLead=6)),
    ("GOS R32 k24, 512 pts", 512, dict(alg=R.GOSCFARType), dict(CFARMode="Greatest Of", refWindowSize=32, guardWindowSize=4, divSum=None, indexLagg=24, indexLead=24)),
    ("GOS R64 k48, 512 pts", 512, dict(alg=R.GOSCFARType), dict(CFARMode="Greatest Of", refWindowSize=64, guardWindowSize=4, divSum=None, indexLagg=48, indexLead=48)),
    ("CA R4 G1, 64 pts (small kernel)", 64, dict(), dict(CFARMode="Cell Averaging", refWindowSize=4, guardWindowSize=1, divSum=2)),
]
for name, n, pk, rk in CASES:
    alg = pk.pop("alg", R.CACFARType)
    params = R.FftMagCfarVanillaParameters(fftParams=R.FFTParams.fixed(numPoints=n), magParams=R.MAGParams.fixed(),
                                           cfarParams=R.CFARParams(fftSize=n, CFARAlgorithm=alg, leadLaggWindowSize=64, guardWindowSize=8, **pk), dtype=dtype)
    frames = (1 << 24) // n
    try:
        dut = R.FftMagCfarChainVanilla(params); dut.configure(R.RunTimeRspChainParams(fftSize=n, **rk))
        if dtype == R.F32:
            x = np.tile(R.stimulus.chirp_frames(64, n, seed=1), (frames // 64, 1))
        else:
            x = np.tile(R.stimulus.formAXI4StreamComplexData(R.stimulus.getComplexTones(n, 0.125, 0.25, 0.5, shiftRangeFactor=12)), (frames, 1))
        b = R.DeviceBuffer(x.nbytes); b.upload(x)
        o = R.DeviceBuffer(frames * n * (8 if params.cfarParams.sendCut else 4))
        for _ in range(3): dut.process_device(b.ptr, frames, o.ptr)
        dut.synchronize(); dut.timer_start()
        reps = 30
        for _ in range(reps): dut.process_device(b.ptr, frames, o.ptr)
        us = dut.timer_stop() / reps * 1e3
        print(f"{name:36s} n={n:5d}: {us:7.1f} us per 16.7 M cells", flush=True)
        dut.close() if hasattr(dut, "close") else None
        del b, o, dut
    except Exception as e:
        print(f"{name:36s} n={n:5d}: {type(e).__name__}: {str(e)[:80]}", flush=True)
