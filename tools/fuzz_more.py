#!/usr/bin/env python3
"""One-off wider fuzz: the generators of tests/test_gpu_fuzz.py on seeds beyond the ones the suite pins.
tools/fuzz_more.py [first last]  (GPU box; prints every disagreement, exit code = their number)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rsp_chains_amd as R
from oracle import oracle as O
from helpers import compare_f32, oracle_cfg, oracle_fcfg, random_beats, tone_beats
import test_gpu_fuzz as F

first = int(sys.argv[1]) if len(sys.argv) > 1 else 40
last = int(sys.argv[2]) if len(sys.argv) > 2 else 240
bad = 0
for seed in range(first, last):
    rng = np.random.default_rng(9000 + seed)
    n, params, rt, frames = F.random_case(rng, R.FIXED16)
    amp = int(rng.choice([300, 5000, 32767]))
    beats = random_beats(frames, n, seed, amp=amp)
    if n >= 256:
        beats[0] = tone_beats(1, n, seed, bp=min(params.fftParams.binPoint, 12))[0]
    try:
        with R.FftMagCfarChainVanilla(params) as dut:
            dut.configure(rt)
            got = dut.stream(beats)
        ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
        if not np.array_equal(got, ref):
            bad += 1
            print("FIXED mismatch seed", seed, n, rt, params.cfarParams, params.fftParams.trimType, flush=True)
    except (ValueError, NotImplementedError) as e:
        print("FIXED seed", seed, "rejected:", str(e)[:100], flush=True)
    rng = np.random.default_rng(7000 + seed)
    n, params, rt, frames = F.random_case(rng, R.F32)
    x = R.stimulus.chirp_frames(min(frames, 6), n, seed=seed, n_targets=2)
    try:
        with R.FftMagCfarChainVanilla(params) as dut:
            dut.configure(rt)
            words = dut.stream(x)
        thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(params, rt), want_mag=True)
        try:
            compare_f32(words, thr, peak, margin, mag, atol=2.0 ** -9 if rt.magMode == 1 else 0.0)
        except AssertionError as e:
            bad += 1
            t = R.unpack_output_f32(words)[0].astype(np.float64).reshape(thr.shape)
            rel_peak = float(np.max(np.abs(t - thr) / np.abs(mag).max(axis=-1, keepdims=True)))
            print("F32 mismatch seed", seed, n, rt.CFARAlgorithm, rt.CFARMode, "R", rt.refWindowSize, "G", rt.guardWindowSize, "mag", rt.magMode,
                  str(e)[:60], "| max error / frame peak = %.2e" % rel_peak, flush=True)
    except (ValueError, NotImplementedError) as e:
        print("F32 seed", seed, "rejected:", str(e)[:100], flush=True)
    if seed % 20 == 0:
        print("seed", seed, "ok so far, bad =", bad, flush=True)
print("done: bad =", bad)
sys.exit(min(bad, 100))
