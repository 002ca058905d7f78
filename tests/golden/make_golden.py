#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle with fixed seeds.

These are BUILD-GENERATED fixtures: the reference commits no vectors, pins no numbers and
cannot run in this pipeline, so they freeze this repo's build-defined
arithmetic spec (oracle/rsp_oracle.c) -- they let a change of the spec or of a kernel be
noticed, they do not prove parity with the Chisel simulation.  Inputs are data only.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import rsp_chains_amd as R  # noqa: E402
from oracle import oracle as O  # noqa: E402
from helpers import make_params, oracle_cfg, oracle_fcfg, random_beats, tone_beats  # noqa: E402


def main():
    # 1. the reference tester's configuration (FftMagCfarChainVanillaSpec) on its stimulus
    n = 1024
    params, rt = make_params(n), R.RunTimeRspChainParams()
    beats = tone_beats(2, n, 1234)
    np.savez_compressed(os.path.join(HERE, "fixed_n1024_tester.npz"), beats=beats,
                        words=O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(2, n))
    # 2. CFAR modes / edge / grouping on random data, N = 256
    n = 256
    beats = random_beats(3, n, 2024)
    out = {}
    for mode in ("Cell Averaging", "Greatest Of", "Smallest Of"):
        for edge in ("zero", "wrap"):
            p = make_params(n, edge=edge)
            r = R.RunTimeRspChainParams(fftSize=n, CFARMode=mode, refWindowSize=16, divSum=4, peakGrouping=1)
            out[f"{mode[0]}{mode.split()[-1][0]}_{edge}"] = O.chain_fixed(beats, oracle_cfg(p, r)).reshape(3, n)
    np.savez_compressed(os.path.join(HERE, "fixed_n256_modes.npz"), beats=beats, **out)
    # 3. fp32 cfg-2 shape (4096 points, CA-CFAR R = 32, G = 4), two chirps
    n = 4096
    params = make_params(n, dtype=R.F32)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging")
    x = R.stimulus.chirp_frames(2, n, seed=1234)
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(params, rt), want_mag=True)
    np.savez_compressed(os.path.join(HERE, "f32_n4096_cfg2.npz"), x=x, thr=thr, peak=peak, margin=margin,
                        mag_max=np.abs(mag).max(axis=1))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
