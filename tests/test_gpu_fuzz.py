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
    compare_f32(words, thr, peak, margin, mag, atol=2.0 ** -9 if rt.magMode == 1 else 0.0)   # one tolerance for every window size


# ------------------------------------------------------------------ 2-D chain

def random_rd_case(rng):
    nr = int(rng.choice([256, 512, 1024, 2048, 8192]))
    nd = int(rng.choice([256, 512] if nr >= 2048 else [256, 512, 1024]))
    rr = int(rng.choice([2, 4, 8, 16]))
    gr = int(rng.integers(1, min(rr, 4)))          # refWindowSize > guardWindowSize > 0 (RspChainVanillaTester.scala:51-52)
    rd, gd = int(rng.integers(1, 12)), int(rng.integers(0, 4))
    if rng.random() < 0.3:
        rr, gr, rd, gd = 8, 2, 8, 2                  # the compile-time windows of the strip walker
    mode = str(rng.choice(["Cell Averaging", "Greatest Of", "Smallest Of"]))
    edge = str(rng.choice(["zero", "wrap"]))
    win = [None, "hann", "hamming", "blackman"]
    return dict(nr=nr, nd=nd, rr=rr, gr=gr, rd=rd, gd=gd, mode=mode, edge=edge, n_ch=int(rng.integers(1, 4)),
                window=win[int(rng.integers(0, 4))] if rng.random() < 0.4 else None,
                windowDoppler=win[int(rng.integers(0, 4))] if rng.random() < 0.4 else None)


@pytest.mark.parametrize("seed", range(10))
def test_rd2d_fixed_random_configuration(gpu, seed):
    """2-D chain, FIXED16: bit-exact against orc_rd_fixed for random sizes, windows, modes, trims, magnitude modes."""
    import test_gpu_rd2d as T
    from helpers import WINDOWS
    rng = np.random.default_rng(7100 + seed)
    c = random_rd_case(rng)
    mag = int(rng.choice([0, 1, 2, 2]))
    params = T.rd_params_fx(c["nr"], c["nd"], ref=c["rd"], guard=c["gd"], edge=c["edge"], window=c["window"],
                            windowDoppler=c["windowDoppler"], trim=str(rng.choice(["RoundDown", "RoundHalfUp", "Convergent"])),
                            bp=int(rng.choice([8, 12])))
    rt = R.RunTimeRspChainParams(fftSize=c["nr"], CFARMode=c["mode"], refWindowSize=c["rr"], guardWindowSize=c["gr"],
                                 divSum=int(rng.integers(4, 10)), thresholdScaler=float(rng.choice([1.5, 3.0, 6.25])),
                                 magMode=mag, logOrLinearMode=0 if (mag == 1 and rng.random() < 0.7) else 1)
    beats, _ = T.targets_fx(c["n_ch"], c["nd"], c["nr"], seed=seed, noise=int(rng.choice([50, 800])), amp=int(rng.choice([3000, 20000])))
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        got = dut.stream(beats)
    ref = T.fx_oracle(params, rt, beats, c["nd"], c["rd"], c["gd"], window_d=c["windowDoppler"])
    assert np.array_equal(got, ref), c


@pytest.mark.parametrize("seed", range(10))
def test_rd2d_f32_random_configuration(gpu, seed):
    import test_gpu_rd2d as T
    from helpers import WINDOWS
    rng = np.random.default_rng(7300 + seed)
    c = random_rd_case(rng)
    params = T.rd_params(c["nr"], c["nd"], ref=c["rd"], guard=c["gd"], edge=c["edge"], window=c["window"],
                         windowDoppler=c["windowDoppler"])
    rt = R.RunTimeRspChainParams(fftSize=c["nr"], CFARMode=c["mode"], refWindowSize=c["rr"], guardWindowSize=c["gr"], divSum=4,
                                 thresholdScaler=4.0)
    x, where = T.targets(c["n_ch"], c["nd"], c["nr"], seed=seed)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    cfg = O.OrcRdCfg(log2nr=R.log2Up(c["nr"]), log2nd=R.log2Up(c["nd"]), mag_mode=O.MAG_JPL, scaler=4.0, ref_r=c["rr"], ref_d=c["rd"],
                     guard_r=c["gr"], guard_d=c["gd"], edge=1 if c["edge"] == "wrap" else 0, window_r=WINDOWS[c["window"]],
                     window_d=WINDOWS[c["windowDoppler"]],
                     cfar_mode={"Cell Averaging": O.CFAR_CA, "Greatest Of": O.CFAR_GO, "Smallest Of": O.CFAR_SO}[c["mode"]])
    thr, peak, margin, mag = O.rd_f32(x, cfg, n_threads=4, want_mag=True)
    n_ch = c["n_ch"]
    compare_f32(words.reshape(n_ch, -1), thr.reshape(n_ch, -1), peak.reshape(n_ch, -1), margin.reshape(n_ch, -1), mag.reshape(n_ch, -1))
