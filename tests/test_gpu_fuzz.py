"""Randomised configurations: FIXED16 must stay bit-exact against the oracle whatever the combination
of frame size, trim type, magnitude mode, CFAR algorithm / mode, window sizes, edge policy, binary
points, scaler and frame count (ragged against frames-per-workgroup)."""
import numpy as np
import pytest

import rsp_chains_amd as R
from oracle import oracle as O
from helpers import compare_f32, make_params, oracle_cfg, oracle_fcfg, random_beats, tone_beats

pytestmark = pytest.mark.gpu


def random_case(rng, dtype):
    n = int(rng.choice([16, 64, 128, 256, 512, 1024, 2048, 4096, 8192]))
    alg = str(rng.choice([R.CACFARType, R.CACFARType, R.GOSCFARType, R.GOSCACFARType]))
    cash = alg == R.CACFARType and rng.random() < 0.3
    max_ref = min(64, n // 4)
    ref = int(2 ** rng.integers(2 if n >= 64 else 1, int(np.log2(max_ref)) + 1))
    guard = int(rng.integers(1, min(ref, 5)))
    gos = alg == R.GOSCFARType or (alg == R.GOSCACFARType and rng.random() < 0.5)
    if gos and n >= 256:
        ref = max(ref, 4)
        guard = min(guard, ref - 1)
    mode = "CASH" if cash else str(rng.choice(["Cell Averaging", "Greatest Of", "Smallest Of"]))
    bp = int(rng.choice([0, 8, 12]))
    mag = int(rng.choice([0, 1, 2, 2, 2]))
    linear = 0 if (mag == 1 and rng.random() < 0.7) else 1
    idx = (int(rng.integers(0, ref)), int(rng.integers(0, ref)))
    params = make_params(n if rng.random() < 0.5 else max(n, 1024), dtype=dtype, bp=bp, alg=alg,
                         edge=str(rng.choice(["zero", "wrap"])), trim=str(rng.choice(["RoundDown", "RoundHalfUp", "Convergent"])),
                         leadLagg=64, guard=8, includeCASH=cash)
    rt = R.RunTimeRspChainParams(
        CFARAlgorithm="GOS" if gos else "CA", CFARMode=mode, refWindowSize=ref, guardWindowSize=guard,
        subWindowSize=max(1, ref // int(rng.choice([2, 4]))) if cash else None, fftSize=n,
        thresholdScaler=float(rng.choice([0.75, 1.5, 3.5, 6.25])),
        divSum=None if alg == R.GOSCFARType else int(np.log2(ref) + rng.integers(-1, 2)) % 8,
        peakGrouping=int(rng.integers(0, 2)), indexLagg=idx[0] if alg != R.CACFARType else None,
        indexLead=idx[1] if alg != R.CACFARType else None, magMode=mag, logOrLinearMode=linear)
    frames = int(rng.integers(1, 40)) if n <= 1024 else int(rng.integers(1, 6))
    return n, params, rt, frames


@pytest.mark.parametrize("seed", range(40))
def test_fixed_random_configuration(gpu, seed):
    rng = np.random.default_rng(9000 + seed)
    n, params, rt, frames = random_case(rng, R.FIXED16)
    amp = int(rng.choice([300, 5000, 32767]))
    beats = random_beats(frames, n, seed, amp=amp)
    if n >= 256:
        beats[0] = tone_beats(1, n, seed, bp=min(params.fftParams.binPoint, 12))[0]
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        got = dut.stream(beats)
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref), (n, rt, params.cfarParams, params.fftParams.trimType)


@pytest.mark.parametrize("seed", range(16))
def test_f32_random_configuration(gpu, seed):
    rng = np.random.default_rng(7000 + seed)
    n, params, rt, frames = random_case(rng, R.F32)
    x = R.stimulus.chirp_frames(min(frames, 6), n, seed=seed, n_targets=2)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(params, rt), want_mag=True)
    short = rt.CFARMode == "CASH" or rt.refWindowSize <= 8     # short windows: prefix cancellation (see test_cash_mode)
    compare_f32(words, thr, peak, margin, mag, rtol=5e-5 if short else 2e-5, atol=2.0 ** -9 if rt.magMode == 1 else 0.0)
