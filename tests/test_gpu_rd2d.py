"""2-D range-Doppler chain (BASELINE.json configs 3/5; no reference counterpart):
range FFT -> Doppler FFT -> JPL magnitude -> 2-D CA-CFAR, GPU vs the float64 oracle."""
import numpy as np
import pytest

import rsp_chains_amd as R
from oracle import oracle as O
from helpers import compare_f32

pytestmark = pytest.mark.gpu


def rd_params(nr, nd, ref=8, guard=2, edge="zero", window=None, windowDoppler=None):
    return R.FftMagCfarVanillaParameters(
        fftParams=R.FFTParams.fixed(numPoints=nr), magParams=R.MAGParams.fixed(),
        cfarParams=R.CFARParams(fftSize=nr, leadLaggWindowSize=16, guardWindowSize=4, edgeMode=edge),
        dtype=R.F32, dopplerPoints=nd, refDoppler=ref, guardDoppler=guard, window=window, windowDoppler=windowDoppler)


def targets(n_ch, nd, nr, seed, k=4, sigma=0.05):
    """Point targets = complex exponentials in fast and slow time + complex white noise."""
    rng = np.random.default_rng(seed)
    x = sigma * (rng.standard_normal((n_ch, nd, nr)) + 1j * rng.standard_normal((n_ch, nd, nr)))
    tr, td = np.arange(nr), np.arange(nd)
    where = []
    for ch in range(n_ch):
        for j in range(k):
            rb, db = int(rng.integers(0, nr)), int(rng.integers(0, nd))
            amp = (0.4, 0.2, 0.1, 0.3)[j % 4]
            x[ch] += amp * np.exp(2j * np.pi * db * td / nd)[:, None] * np.exp(2j * np.pi * rb * tr / nr)[None, :]
            where.append((ch, db, rb))
    return x.astype(np.complex64), where


@pytest.mark.parametrize("nr,nd,edge", [(256, 256, "zero"), (1024, 512, "zero"), (512, 256, "wrap"), (4096, 512, "zero"),
                                        (2048, 1024, "zero"), (4096, 256, "wrap"), (8192, 256, "zero")])
def test_rd2d_against_oracle(gpu, nr, nd, edge):
    n_ch = 2
    params = rd_params(nr, nd, edge=edge)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2, divSum=4,
                                 thresholdScaler=4.0)
    x, where = targets(n_ch, nd, nr, seed=2345 + nr)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    assert words.shape == (n_ch, nd, nr)
    cfg = O.OrcRdCfg(log2nr=R.log2Up(nr), log2nd=R.log2Up(nd), mag_mode=O.MAG_JPL, scaler=4.0, ref_r=8, ref_d=8,
                     guard_r=2, guard_d=2, edge=1 if edge == "wrap" else 0)
    thr, peak, margin, mag = O.rd_f32(x, cfg, want_mag=True)
    compare_f32(words.reshape(n_ch, -1), thr.reshape(n_ch, -1), peak.reshape(n_ch, -1), margin.reshape(n_ch, -1),
                mag.reshape(n_ch, -1))
    for ch, db, rb in where:           # every injected target is detected at its (Doppler, range) cell
        assert words[ch, db, rb] & 1


@pytest.mark.parametrize("mode", ["Greatest Of", "Smallest Of"])
@pytest.mark.parametrize("rr,gr,rd,gd,edge", [(8, 2, 8, 2, "zero"), (8, 2, 8, 2, "wrap"), (4, 1, 5, 3, "zero"), (16, 3, 4, 0, "wrap")])
def test_rd2d_go_so(gpu, mode, rr, gr, rd, gd, edge):
    """2-D GO / SO (build-defined: docs/FIXED_POINT_SPEC.md section 6): greatest / smallest of the means of the lagging
    and the leading half of the training region.  Compile-time windows take the strip walker, the others the tiled kernel."""
    nr, nd, n_ch = 1024, 256, 2
    params = rd_params(nr, nd, ref=rd, guard=gd, edge=edge)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode=mode, refWindowSize=rr, guardWindowSize=gr if gr else 1, divSum=4,
                                 thresholdScaler=4.0)
    if gr == 0:
        pytest.skip("guardWindowSize register must be > 0 (RspChainVanillaTester.scala:51)")
    x, where = targets(n_ch, nd, nr, seed=7 + rr)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    cfg = O.OrcRdCfg(log2nr=10, log2nd=8, mag_mode=O.MAG_JPL, scaler=4.0, ref_r=rr, ref_d=rd, guard_r=gr, guard_d=gd,
                     edge=1 if edge == "wrap" else 0, cfar_mode=O.CFAR_GO if mode == "Greatest Of" else O.CFAR_SO)
    thr, peak, margin, mag = O.rd_f32(x, cfg, want_mag=True)
    compare_f32(words.reshape(n_ch, -1), thr.reshape(n_ch, -1), peak.reshape(n_ch, -1), margin.reshape(n_ch, -1),
                mag.reshape(n_ch, -1))
    ca = O.rd_f32(x, O.OrcRdCfg(log2nr=10, log2nd=8, mag_mode=O.MAG_JPL, scaler=4.0, ref_r=rr, ref_d=rd, guard_r=gr,
                                guard_d=gd, edge=1 if edge == "wrap" else 0))[0]
    assert np.all(thr >= ca - 1e-12) if mode == "Greatest Of" else np.all(thr <= ca + 1e-12)   # GO >= CA >= SO


@pytest.mark.parametrize("wr,wd", [("hann", None), (None, "hamming"), ("blackman", "hann")])
def test_rd2d_windows(gpu, wr, wd):
    """Pre-FFT windows over fast time (range) and slow time (Doppler): SURVEY 8f-n4, no reference item."""
    nr, nd, n_ch = 1024, 256, 2
    params = rd_params(nr, nd, window=wr, windowDoppler=wd)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2, divSum=4,
                                 thresholdScaler=4.0)
    x, _ = targets(n_ch, nd, nr, seed=31)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    code = {None: 0, "hann": 1, "hamming": 2, "blackman": 3}
    cfg = O.OrcRdCfg(log2nr=10, log2nd=8, mag_mode=O.MAG_JPL, scaler=4.0, ref_r=8, ref_d=8, guard_r=2, guard_d=2, edge=0,
                     window_r=code[wr], window_d=code[wd])
    thr, peak, margin, mag = O.rd_f32(x, cfg, want_mag=True)
    compare_f32(words.reshape(n_ch, -1), thr.reshape(n_ch, -1), peak.reshape(n_ch, -1), margin.reshape(n_ch, -1),
                mag.reshape(n_ch, -1))


def test_rd2d_cfg5_shape_one_channel_against_oracle(gpu):
    """BASELINE.json configs[4] shape: one 8192 x 1024 channel map (range_fft<13> + doppler_mag<10> + strip
    walker at their cfg-5 sizes) against the float64 oracle, every cell."""
    nr, nd = 8192, 1024
    params = rd_params(nr, nd)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2, divSum=4,
                                 thresholdScaler=4.0)
    x, where = targets(1, nd, nr, seed=4567, k=6)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    cfg = O.OrcRdCfg(log2nr=13, log2nd=10, mag_mode=O.MAG_JPL, scaler=4.0, ref_r=8, ref_d=8, guard_r=2, guard_d=2, edge=0)
    thr, peak, margin, mag = O.rd_f32(x, cfg, want_mag=True)
    compare_f32(words.reshape(1, -1), thr.reshape(1, -1), peak.reshape(1, -1), margin.reshape(1, -1), mag.reshape(1, -1))
    for ch, db, rb in where:
        assert words[ch, db, rb] & 1


def test_rd2d_cfg5_full_share_properties(gpu):
    """The full per-GPU share of configs[4]: 8 Rx x 8192 x 1024 = 67 108 864 cells in ONE call, checked through
    size-independent properties: (1) channels are independent (the 8 channels are copies of 2 distinct maps:
    equal inputs give bit-equal words, wherever they sit in the batch, and equal the 2-channel call),
    (2) the chain is homogeneous of degree 1 (JPL magnitude, linear CFAR): scaling the input by 2^-3 scales
    every threshold by exactly 2^-3 and leaves every peak flag unchanged, (3) every injected target is found."""
    nr, nd, n_ch = 8192, 1024, 8
    params = rd_params(nr, nd)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2, divSum=4,
                                 thresholdScaler=4.0)
    x2, where = targets(2, nd, nr, seed=4568)
    x = np.ascontiguousarray(x2[[0, 1, 1, 0, 0, 1, 0, 1]])
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        w = dut.stream(x)
        w2 = dut.stream(x2)
        ws = dut.stream(x * np.float32(0.125))
    assert w.shape == (n_ch, nd, nr)
    for i, src in enumerate([0, 1, 1, 0, 0, 1, 0, 1]):
        assert np.array_equal(w[i], w2[src]), f"channel {i} differs from its copy"
    ta, pa = R.unpack_output_f32(w)
    tb, pb = R.unpack_output_f32(ws)
    assert np.array_equal(pa, pb)
    assert np.array_equal(ta * np.float32(0.125), tb)
    for ch, db, rb in where:
        assert w2[ch, db, rb] & 1


@pytest.mark.parametrize("rr,gr,rd,gd,edge", [(4, 1, 5, 3, "zero"), (16, 3, 4, 1, "wrap"), (8, 2, 8, 1, "zero")])
def test_rd2d_run_time_windows(gpu, rr, gr, rd, gd, edge):
    """Windows other than the compile-time (8, 2, 8, 2) of cfg 3 / 5 take the tiled LDS kernel."""
    nr, nd, n_ch = 512, 256, 2
    params = rd_params(nr, nd, ref=rd, guard=gd, edge=edge)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=rr, guardWindowSize=gr, divSum=4,
                                 thresholdScaler=4.0)
    x, where = targets(n_ch, nd, nr, seed=99 + rr)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    cfg = O.OrcRdCfg(log2nr=R.log2Up(nr), log2nd=R.log2Up(nd), mag_mode=O.MAG_JPL, scaler=4.0, ref_r=rr, ref_d=rd,
                     guard_r=gr, guard_d=gd, edge=1 if edge == "wrap" else 0)
    thr, peak, margin, mag = O.rd_f32(x, cfg, want_mag=True)
    compare_f32(words.reshape(n_ch, -1), thr.reshape(n_ch, -1), peak.reshape(n_ch, -1), margin.reshape(n_ch, -1),
                mag.reshape(n_ch, -1))


@pytest.mark.parametrize("edge", ["zero", "wrap"])
def test_rd2d_walker_agrees_with_tiled_kernel(gpu, edge):
    """The register-ring strip walker (compile-time windows) and the tiled LDS kernel are two
    summation orders of the same statistic (the walker's column sums are running sums over 64 rows):
    each is held to the oracle (thresholds in tolerance, every decidable flag equal) and their
    thresholds agree with each other to 1e-5."""
    nr, nd, n_ch = 1024, 256, 2
    params = rd_params(nr, nd, edge=edge)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2, divSum=4,
                                 thresholdScaler=4.0)
    x, _ = targets(n_ch, nd, nr, seed=4242)
    cfg = O.OrcRdCfg(log2nr=10, log2nd=8, mag_mode=O.MAG_JPL, scaler=4.0, ref_r=8, ref_d=8, guard_r=2, guard_d=2,
                     edge=1 if edge == "wrap" else 0)
    thr, peak, margin, mag = O.rd_f32(x, cfg, want_mag=True)
    outs = []
    for tiled in (0, 1):
        with R.FftMagCfarChainVanilla(params) as dut:
            dut.configure(rt)
            dut.set_option(dut.FORCE_TILED_CFAR2D, tiled)
            outs.append(dut.stream(x))
        compare_f32(outs[-1].reshape(n_ch, -1), thr.reshape(n_ch, -1), peak.reshape(n_ch, -1), margin.reshape(n_ch, -1),
                    mag.reshape(n_ch, -1))
    ta, _ = R.unpack_output_f32(outs[0])
    tb, _ = R.unpack_output_f32(outs[1])
    np.testing.assert_allclose(ta, tb, rtol=1e-5)   # half the tolerance each is held to against the oracle


@pytest.mark.parametrize("tiled", [0, 1])
def test_rd2d_detection_list(gpu, tiled):
    """The 2-D chain's detection list three ways: appended by the CFAR kernel itself (fused call), compacted
    from the dense words (stand-alone call) and through the host-buffer call -- all equal to the peaks of the
    dense words; a list capacity smaller than the number of peaks truncates without garbage."""
    nr, nd, n_ch = 512, 256, 3
    params = rd_params(nr, nd)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2, divSum=4,
                                 thresholdScaler=2.0)   # a few hundred peaks
    x, _ = targets(n_ch, nd, nr, seed=77)
    cap = 1 << 16
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        dut.set_option(dut.FORCE_TILED_CFAR2D, tiled)
        d_in = R.DeviceBuffer(x.nbytes); d_in.upload(x)
        d_out, d_out2 = R.DeviceBuffer(x.size * 4), R.DeviceBuffer(x.size * 4)
        d_list, d_cnt = R.DeviceBuffer(cap * 16), R.DeviceBuffer(8)
        d_list2, d_cnt2 = R.DeviceBuffer(cap * 16), R.DeviceBuffer(8)
        dut.process_device(d_in.ptr, n_ch, d_out.ptr)
        dut.detections_device(d_out.ptr, n_ch, d_list.ptr, cap, d_cnt.ptr)
        dut.process_detect_device(d_in.ptr, n_ch, d_out2.ptr, d_list2.ptr, cap, d_cnt2.ptr)
        dut.synchronize()
        dense = d_out.download(np.uint32, x.size).reshape(n_ch, nd, nr)
        assert np.array_equal(dense, d_out2.download(np.uint32, x.size).reshape(n_ch, nd, nr))
        ch, d, r = np.nonzero(dense & 1)
        want = sorted(zip(ch.tolist(), r.tolist(), d.tolist(), dense[ch, d, r].tolist()))
        assert len(want) > 100
        for lst_buf, cnt_buf in ((d_list, d_cnt), (d_list2, d_cnt2)):
            found, stored = (int(v) for v in cnt_buf.download(np.uint32, 2))
            assert found == stored == len(want)
            lst = lst_buf.download(np.uint32, found * 4).reshape(found, 4)
            assert sorted(map(tuple, lst.tolist())) == want      # {frame = channel, bin = range, doppler, word}
        dut.process_detect_device(d_in.ptr, n_ch, d_out2.ptr, d_list2.ptr, 50, d_cnt2.ptr)   # cap < peaks
        dut.synchronize()
        found, stored = (int(v) for v in d_cnt2.download(np.uint32, 2))
        assert found == len(want) and stored == 50
        assert set(map(tuple, d_list2.download(np.uint32, 200).reshape(50, 4).tolist())) <= set(want)
        # count-only call: cap = 0 and NO list buffer (the 1-D chain's count-only form): {found, 0}, nothing stored
        d_cnt2.upload(np.array([0xdead, 0xbeef], np.uint32))
        dut.process_detect_device(d_in.ptr, n_ch, d_out2.ptr, 0, 0, d_cnt2.ptr)
        dut.synchronize()
        assert [int(v) for v in d_cnt2.download(np.uint32, 2)] == [len(want), 0]
        det, found_h = dut.detections(x)
        assert found_h == len(want)
        assert [(int(a), int(b), int(c), int(w)) for a, b, c, w in zip(det["frame"], det["bin"], det["doppler"], det["word"])] == \
            sorted(want, key=lambda t: (t[0], t[2], t[1]))


def test_rd2d_rejects_what_it_does_not_implement(gpu):
    params = rd_params(1024, 512)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(R.RunTimeRspChainParams(fftSize=1024, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2, divSum=4,
                                              peakGrouping=1))
        with pytest.raises(NotImplementedError):
            dut.check()
    with pytest.raises(NotImplementedError):     # GOS has no 2-D definition
        p = rd_params(1024, 512); p.cfarParams.CFARAlgorithm = R.GOSCFARType
        with R.FftMagCfarChainVanilla(p) as dut:
            dut.configure(R.RunTimeRspChainParams(fftSize=1024, refWindowSize=8, guardWindowSize=2, indexLagg=3, indexLead=3))
            dut.check()


# ------------------------------------------------------------------ FIXED16 data path (bit-exact vs orc_rd_fixed)

def rd_params_fx(nr, nd, ref=8, guard=2, edge="zero", window=None, windowDoppler=None, trim="Convergent", bp=12):
    fp = R.FixedPoint(16, bp)
    return R.FftMagCfarVanillaParameters(
        fftParams=R.FFTParams.fixed(numPoints=nr, binPoint=bp, trimType=trim), magParams=R.MAGParams.fixed(binPoint=bp),
        cfarParams=R.CFARParams(protoIn=fp, protoThreshold=fp, protoScaler=fp, fftSize=nr, leadLaggWindowSize=16,
                                guardWindowSize=4, edgeMode=edge),
        dtype=R.FIXED16, dopplerPoints=nd, refDoppler=ref, guardDoppler=guard, window=window, windowDoppler=windowDoppler)


def targets_fx(n_ch, nd, nr, seed, k=3, noise=300, amp=6000):
    """Point targets + white noise as int16 I/Q beats (data[31:16] = re, [15:0] = im)."""
    rng = np.random.default_rng(seed)
    x = noise * (rng.standard_normal((n_ch, nd, nr)) + 1j * rng.standard_normal((n_ch, nd, nr)))
    tr, td = np.arange(nr), np.arange(nd)
    where = []
    for ch in range(n_ch):
        for j in range(k):
            rb, db = int(rng.integers(0, nr)), int(rng.integers(0, nd))
            x[ch] += amp / (j + 1) * np.exp(2j * np.pi * db * td / nd)[:, None] * np.exp(2j * np.pi * rb * tr / nr)[None, :]
            where.append((ch, db, rb))
    re = np.clip(np.rint(x.real), -32768, 32767).astype(np.int64)
    im = np.clip(np.rint(x.imag), -32768, 32767).astype(np.int64)
    return O.pack_iq(re, im).reshape(n_ch, nd, nr), where


def fx_oracle(params, rt, beats, nd, ref_d, guard_d, window_d=None, want_mag=False):
    from helpers import oracle_cfg, WINDOWS
    cfg = oracle_cfg(params, rt)
    return O.rd_fixed(beats, cfg, R.log2Up(nd), ref_d, guard_d, WINDOWS[window_d], n_threads=4, want_mag=want_mag)


@pytest.mark.parametrize("nr,nd,edge,mode,magmode", [
    (256, 256, "zero", "Cell Averaging", 2), (1024, 512, "wrap", "Greatest Of", 2), (512, 256, "zero", "Smallest Of", 0),
    (2048, 256, "wrap", "Cell Averaging", 1), (4096, 512, "zero", "Cell Averaging", 2), (8192, 256, "zero", "Greatest Of", 2),
    (1024, 1024, "zero", "Cell Averaging", 2)])
def test_rd2d_fixed_bit_exact(gpu, nr, nd, edge, mode, magmode):
    """The 2-D chain on the 16-bit FixedPoint data path: every output word equals the oracle's (orc_rd_fixed)."""
    n_ch = 2
    params = rd_params_fx(nr, nd, edge=edge)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode=mode, refWindowSize=8, guardWindowSize=2, divSum=8,
                                 thresholdScaler=3.0, magMode=magmode, logOrLinearMode=0 if magmode == 1 else 1)
    beats, where = targets_fx(n_ch, nd, nr, seed=99 + nr + nd)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(beats)
    ref = fx_oracle(params, rt, beats, nd, 8, 2)
    assert words.shape == ref.shape == (n_ch, nd, nr)
    bad = np.flatnonzero(words.ravel() != ref.ravel())
    assert bad.size == 0, f"{bad.size} words differ, first at {np.unravel_index(bad[0], words.shape)}"
    if magmode == 2 and mode != "Greatest Of":
        for ch, db, rb in where[:1]:       # the strongest injected target of channel 0 is a peak at its cell
            assert words[ch, db, rb] & 1


@pytest.mark.parametrize("rr,gr,rd,gd,mode", [(4, 1, 5, 3, "Cell Averaging"), (16, 3, 4, 0, "Greatest Of"), (2, 1, 7, 1, "Smallest Of")])
def test_rd2d_fixed_windows_and_sizes(gpu, rr, gr, rd, gd, mode):
    """Other window geometries, Q1.15 range + Doppler windows, half-up trim, and the fused detection list."""
    nr, nd, n_ch = 512, 256, 3
    params = rd_params_fx(nr, nd, ref=rd, guard=gd, edge="wrap", window="hann", windowDoppler="hamming", trim="RoundHalfUp")
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode=mode, refWindowSize=rr, guardWindowSize=gr, divSum=7, thresholdScaler=2.5)
    beats, _ = targets_fx(n_ch, nd, nr, seed=5 + rr)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(beats)
        det, found = dut.detections(beats)
    ref = fx_oracle(params, rt, beats, nd, rd, gd, window_d="hamming")
    assert np.array_equal(words, ref)
    ch, d, r = np.nonzero(ref & 1)
    assert found == ch.size == det.size
    got = sorted(zip(det["frame"].tolist(), det["doppler"].tolist(), det["bin"].tolist(), det["word"].tolist()))
    want = sorted(zip(ch.tolist(), d.tolist(), r.tolist(), ref[ch, d, r].tolist()))
    assert got == want


def test_rd2d_fixed_rejects_stage_options(gpu):
    p = rd_params_fx(256, 256)
    p.fftParams.expandLogic[0] = 1
    with pytest.raises(NotImplementedError):
        R.FftMagCfarChainVanilla(p)


@pytest.mark.parametrize("scaler,chunk_mb", [(0.6, 0), (4.0, 1), (0.6, 1)])
def test_rd2d_fused_list_overflow_and_chunks(gpu, scaler, chunk_mb):
    """Fused list under stress: a threshold below the noise (tens of thousands of peaks: every wave overflows its
    LDS staging and re-reads its words) and / or the batch run in chunks of channels that share the scratch maps
    (only the first chunk's range pass may zero the list's cursor)."""
    nr, nd, n_ch = 512, 256, 4
    params = rd_params(nr, nd)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2, divSum=4,
                                 thresholdScaler=scaler)
    x, _ = targets(n_ch, nd, nr, seed=1001)
    cap = 1 << 20
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        if chunk_mb:
            dut.set_option(dut.RD_CHUNK_BYTES, chunk_mb << 20)    # 1 MiB < one channel's 1.5 MiB of intermediates: 1 channel per chunk
        d_in = R.DeviceBuffer(x.nbytes); d_in.upload(x)
        d_out = R.DeviceBuffer(x.size * 4)
        d_list, d_cnt = R.DeviceBuffer(cap * 16), R.DeviceBuffer(8)
        for _ in range(2):                                        # twice: the cursor restarts from zero every call
            dut.process_detect_device(d_in.ptr, n_ch, d_out.ptr, d_list.ptr, cap, d_cnt.ptr)
        dut.synchronize()
        dense = d_out.download(np.uint32, x.size).reshape(n_ch, nd, nr)
        found, stored = (int(v) for v in d_cnt.download(np.uint32, 2))
        lst = d_list.download(np.uint32, stored * 4).reshape(stored, 4)
    ch, d, r = np.nonzero(dense & 1)
    assert found == stored == ch.size
    if scaler < 1:
        assert ch.size > 50000
    key = (lst[:, 0].astype(np.int64) * nd + lst[:, 2]) * nr + lst[:, 1]
    order = np.argsort(key)
    assert np.array_equal(key[order], (ch.astype(np.int64) * nd + d) * nr + r)
    assert np.array_equal(lst[order, 3], dense[ch, d, r])


@pytest.mark.parametrize("mode", ["Cell Averaging", "Greatest Of", "Smallest Of"])
def test_rd2d_fixed_walker_equals_tile_kernel(gpu, mode):
    """FIXED16, the compile-time windows: the strip walker (integer instantiation) and the run-time-window tile kernel
    give the oracle's words, both."""
    nr, nd, n_ch = 1024, 256, 2
    params = rd_params_fx(nr, nd, edge="wrap")
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode=mode, refWindowSize=8, guardWindowSize=2, divSum=8, thresholdScaler=2.0)
    beats, _ = targets_fx(n_ch, nd, nr, seed=321)
    ref = fx_oracle(params, rt, beats, nd, 8, 2)
    for tiled in (0, 1):
        with R.FftMagCfarChainVanilla(params) as dut:
            dut.configure(rt)
            dut.set_option(dut.FORCE_TILED_CFAR2D, tiled)
            assert np.array_equal(dut.stream(beats), ref), tiled


def test_rd2d_host_entry_chunked_by_channel(gpu):
    """The host-buffer entry cuts a batch into pipeline chunks (H2D || kernels || D2H); the 2-D chain's unit is a
    channel.  One channel per chunk, more chunks than staging slots, pageable and pinned buffers: the words of the
    one-launch device call, and the detection call (chunked kernels + one dense compaction) its peaks."""
    nr, nd, n_ch = 512, 256, 5
    params = rd_params(nr, nd)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2, divSum=4,
                                 thresholdScaler=2.0)
    x, _ = targets(n_ch, nd, nr, seed=78)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        d_in = R.DeviceBuffer(x.nbytes); d_in.upload(x)
        d_out = R.DeviceBuffer(x.size * 4)
        dut.process_device(d_in.ptr, n_ch, d_out.ptr)
        dut.synchronize()
        ref = d_out.download(np.uint32, x.size).reshape(n_ch, nd, nr)
        dut.set_option(dut.HOST_CHUNK_BYTES, 1)                      # one channel (1 MiB) per chunk: 5 chunks, 3 slots
        assert np.array_equal(dut.stream(x), ref)
        hin, hout = R.HostBuffer(x.shape, np.complex64), R.HostBuffer(x.size, np.uint32)
        hin.array[...] = x
        assert np.array_equal(dut.stream(hin.array, out=hout.array), ref)
        det, found = dut.detections(x)
        ch, d, r = np.nonzero(ref & 1)
        assert found == ch.size > 50
        assert np.array_equal(det["frame"], ch) and np.array_equal(det["doppler"], d) and np.array_equal(det["bin"], r)
        assert np.array_equal(det["word"], ref[ch, d, r])
