"""Known-answer tests that pin the CPU oracle.  The reference ships no golden vectors and its
specs assert nothing numeric (PARITY UNPINNED, see oracle/rsp_oracle.h), so the anchors are the
facts recoverable from its files: wire formats, 1/N scaling, the JPL formula, the tester's
stimulus and where its peaks must land, the reference's own 2-LSB HW-vs-float tolerance."""
import numpy as np
import pytest

import rsp_chains_amd as R
from oracle import oracle as O
from helpers import make_params, oracle_cfg, tone_beats


# ------------------------------------------------------------------ wire formats
def test_pack_iq_matches_formAXI4StreamComplexData():
    # RspChainTesterUtils.scala:105-109: real in the upper 16 bits, two's complement
    assert O.lib().orc_pack_iq(1, 2) == 0x00010002
    assert O.lib().orc_pack_iq(-1, -2) == 0xFFFFFFFE
    assert O.lib().orc_pack_iq(-32768, 32767) == 0x80007FFF
    z = np.array([3 - 4j, -5 + 6j])
    assert np.array_equal(R.stimulus.formAXI4StreamComplexData(z), O.pack_iq([3, -5], [-4, 6]))


def test_output_word_layout():
    # FftMagCfarChainTester.scala:153,163-167
    w = O.lib().orc_pack_out(1234, 77, 1, 10)
    assert w >> 11 == 1234 and (w >> 1) & 1023 == 77 and w & 1 == 1
    thr, bins, peak = R.unpack_output(np.array([w], np.uint32), 1024)
    assert (thr[0], bins[0], peak[0]) == (1234, 77, 1)
    # a negative (log-domain) threshold survives the tester's arithmetic shift on a signed Int
    w = O.lib().orc_pack_out((-5) & 0xFFFFFFFF, 3, 0, 10)
    thr, bins, peak = R.unpack_output(np.array([w], np.uint32), 1024)
    assert (thr[0], bins[0], peak[0]) == (-5, 3, 0)


# ------------------------------------------------------------------ fixed-point FFT
@pytest.mark.parametrize("n", [16, 256, 1024, 4096])
def test_fft_fixed_impulse_and_dc(n):
    re = np.zeros(n, np.int16); im = np.zeros(n, np.int16)
    re[0] = 16384
    fr, fi = O.fft_fixed(re, im)            # impulse -> flat spectrum of value x/N
    assert np.all(fr == 16384 // n) and np.all(fi == 0)
    re[:] = 1000; im[:] = -2000
    fr, fi = O.fft_fixed(re, im)            # DC -> single bin, net gain 1/N (Tester:77)
    assert fr[0] == 1000 and fi[0] == -2000
    assert np.all(fr[1:] == 0) and np.all(fi[1:] == 0)


@pytest.mark.parametrize("trim", [O.TRIM_FLOOR, O.TRIM_HALF_UP, O.TRIM_CONVERGENT])
def test_fft_fixed_within_reference_tolerance_of_float(trim):
    """checkFFTError's tolerance = 2 (RspChainTesterUtils.scala:221) is the reference's own
    accepted distance between its hardware and a float FFT."""
    rng = np.random.default_rng(5)
    n = 1024
    re = rng.integers(-8000, 8000, n).astype(np.int16)
    im = rng.integers(-8000, 8000, n).astype(np.int16)
    fr, fi = O.fft_fixed(re, im, trim)
    ref = np.fft.fft(re.astype(float) + 1j * im) / n
    err = np.concatenate([fr - ref.real, fi - ref.imag])
    if trim == O.TRIM_CONVERGENT:       # the default: unbiased, inside the reference's own tolerance
        assert np.abs(err).max() <= 2 and abs(err.mean()) < 0.05
    else:                               # floor / half-up carry a -1/2 / +1/2 LSB mean offset
        assert abs(abs(err.mean()) - 0.5) < 0.1 and np.abs(err - err.mean()).max() <= 3


def test_fft_fixed_tone_lands_on_its_bin():
    n = 1024
    z = R.stimulus.calcExpectedNcoOut(n, 32)   # RspChainTesterUtils.scala:174-181
    fr, fi = O.fft_fixed(z.real.astype(np.int16), z.imag.astype(np.int16))
    mag = np.hypot(fr.astype(float), fi)
    assert mag.argmax() == 32 and mag[32] > 16000 and np.delete(mag, 32).max() < 8


def test_twiddle_rom():
    wr, wi = O.twiddles_q14(10)
    assert wr[0] == 16384 and wi[0] == 0
    assert wr[256] == 0 and wi[256] == -16384          # W^(N/4) = -i
    assert wr[128] == 11585 and wi[128] == -11585       # round(2^14 / sqrt 2)
    k = np.arange(512)
    assert np.abs(wr - 16384 * np.cos(2 * np.pi * k / 1024)).max() <= 0.5
    assert np.abs(wi + 16384 * np.sin(2 * np.pi * k / 1024)).max() <= 0.5


# ------------------------------------------------------------------ magnitude
def test_jpl_matches_reference_float_model():
    rng = np.random.default_rng(7)
    re = rng.integers(-32768, 32768, 4000); im = rng.integers(-32768, 32768, 4000)
    cfg = O.default_cfg()
    got = O.mag_fixed(re, im, cfg)
    ref = R.stimulus.jplMag(re + 1j * im)            # RspChainTesterUtils.scala:120-127
    ok = ref <= 32767
    assert np.abs(got[ok] - ref[ok]).max() <= 1      # floor of each term vs floor of the sum
    assert np.all(got[~ok] == 32767)                 # saturation at the 16-bit signed maximum
    # exact cases
    assert O.mag_fixed([3000], [0], cfg)[0] == 3000 and O.mag_fixed([0], [-3000], cfg)[0] == 3000
    assert O.mag_fixed([800], [800], cfg)[0] == max(800 + 100, 700 + 400)


def test_sqr_and_log2_modes():
    cfg = O.default_cfg(mag_mode=O.MAG_SQR)
    assert O.mag_fixed([4096], [0], cfg)[0] == 4096          # 1.0^2 = 1.0 in Q.12
    assert O.mag_fixed([2048], [2048], cfg)[0] == 2048       # 0.25 + 0.25
    cfg = O.default_cfg(mag_mode=O.MAG_LOG2)
    x = np.array([1, 2, 3, 4096, 8192, 12288, 32767])
    got = O.mag_fixed(x, np.zeros_like(x), cfg)
    ref = np.log2(x / 4096.0) * 512                          # Q7.9 (FftMagCfarChain.scala:94-95)
    assert np.abs(got - ref).max() <= 1.5                    # 9-bit mantissa table + rounding
    assert got[3] == 0 and got[4] == 512 and np.all(np.diff(got) > 0)


# ------------------------------------------------------------------ CFAR
def _cfar(mag, **kw):
    cfg = O.default_cfg(log2n=int(np.log2(len(mag))), **kw)
    words, thr = O.cfar_fixed(mag, cfg)
    return words, thr


def test_cfar_flat_input_threshold_and_edges():
    n, R_, G = 256, 16, 4
    mag = np.full(n, 1000, np.int32)
    for mode in (O.CFAR_CA, O.CFAR_GO, O.CFAR_SO):
        words, thr = _cfar(mag, cfar_mode=mode, ref_window=R_, guard_window=G, div_sum=4)
        inner = slice(R_ + G, n - R_ - G)
        assert np.all(thr[inner] == 3500)            # 3.5 x mean (Tester:101, RT:42)
        assert np.all((words[inner] & 1) == 0)       # 1000 < 3500
    # zero-edge policy: the lagging window of the first cell is empty -> SO sees 0, GO sees the lead
    _, thr = _cfar(mag, cfar_mode=O.CFAR_SO, ref_window=R_, guard_window=G, div_sum=4)
    assert thr[0] == 0 and thr[n - 1] == 0
    _, thr = _cfar(mag, cfar_mode=O.CFAR_GO, ref_window=R_, guard_window=G, div_sum=4)
    assert thr[0] == 3500 and thr[n - 1] == 3500
    # cyclic policy: flat everywhere
    _, thr = _cfar(mag, cfar_mode=O.CFAR_SO, ref_window=R_, guard_window=G, div_sum=4, edge=O.EDGE_WRAP)
    assert np.all(thr == 3500)


def test_cfar_mode_ordering_and_peak():
    rng = np.random.default_rng(11)
    n = 512
    mag = rng.integers(50, 150, n).astype(np.int32)
    mag[200] = 5000
    thr = {}
    for mode in (O.CFAR_CA, O.CFAR_GO, O.CFAR_SO):
        words, thr[mode] = _cfar(mag, cfar_mode=mode, ref_window=32, guard_window=4, div_sum=5)
        inner = np.arange(36, n - 36)   # zero-edge policy: SO sees an empty window at the frame ends
        assert list(inner[np.nonzero(words[inner] & 1)[0]]) == [200]
        assert np.array_equal((words >> 1) & (n - 1), np.arange(n))
    assert np.all(thr[O.CFAR_SO] <= thr[O.CFAR_CA]) and np.all(thr[O.CFAR_CA] <= thr[O.CFAR_GO])
    # the guard band hides the target from its neighbours' windows
    assert thr[O.CFAR_GO][197] < 1000 and thr[O.CFAR_GO][200 + 5] > 400


def test_cfar_peak_grouping_and_log_mode():
    n = 256
    mag = np.full(n, 100, np.int32)
    mag[100:103] = [3000, 4000, 3500]
    words, _ = _cfar(mag, ref_window=16, guard_window=4, div_sum=4, peak_grouping=0)
    assert list(np.nonzero(words & 1)[0]) == [100, 101, 102]
    words, _ = _cfar(mag, ref_window=16, guard_window=4, div_sum=4, peak_grouping=1)
    assert list(np.nonzero(words & 1)[0]) == [101]            # only the local maximum survives
    # log domain: threshold = statistic + scaler
    words, thr = _cfar(mag, ref_window=16, guard_window=4, div_sum=4, linear=0, scaler=700,
                       cfar_mode=O.CFAR_CA)
    assert thr[40] == 100 + 700


def test_cfar_gos_order_statistic():
    n = 256
    rng = np.random.default_rng(13)
    mag = rng.permutation(n).astype(np.int32) + 1
    R_, G, k = 16, 2, 11
    words, thr = _cfar(mag, algorithm=1, cfar_mode=O.CFAR_GO, ref_window=R_, guard_window=G,
                       index_lagg=k, index_lead=k, scaler=4096, div_sum=0)
    for c in (40, 100, 200):
        lag = np.sort(mag[c - G - R_:c - G])[k]
        lead = np.sort(mag[c + G + 1:c + G + 1 + R_])[k]
        assert thr[c] == max(lag, lead)


# ------------------------------------------------------------------ the reference tester's procedure
def test_tester_stimulus_gives_four_peaks():
    """FftMagCfarChainVanillaSpec's configuration and stimulus (Tester:53,198-241): tones at
    f = 1/8, 1/4, 1/2 and the noise's DC must be the only detections."""
    params = make_params(1024)
    rt = R.RunTimeRspChainParams()
    for seed in (1, 2, 3):
        out = O.chain_fixed(tone_beats(1, 1024, seed), oracle_cfg(params, rt))
        thr, bins, peaks = R.unpack_output(out, 1024)
        assert list(np.nonzero(peaks)[0]) == [0, 128, 256, 512]
        assert np.array_equal(bins, np.arange(1024))


def test_full_chain_nco_peak_bin_32():
    """RspChainVanillaSpec: 'peak is expected on frequency bin startingPoint*numOfPoints/(4*tableSize)'
    = 16*1024/(4*128) = 32 (RspChainVanillaTester.scala:85,89), formats BP 0 / 3 / 6 (:205-239)."""
    n = 1024
    params = make_params(n, bp=0, leadLagg=32,
                         proto=(R.FixedPoint(16, 0), R.FixedPoint(16, 3), R.FixedPoint(16, 6)))
    rt = R.RunTimeRspChainParams()
    beats = R.stimulus.formAXI4StreamComplexData(R.stimulus.calcExpectedNcoOut(n, 32))
    out = O.chain_fixed(beats, oracle_cfg(params, rt))
    assert list(np.nonzero(out & 1)[0]) == [32]


# ------------------------------------------------------------------ float64 path vs independent numpy
def _cfar_numpy(mag, R_, G, mode, scaler, div, edge):
    n = len(mag)
    thr = np.zeros(n)
    for k in range(n):
        def cell(j):
            if edge == O.EDGE_WRAP:
                return mag[j % n]
            return mag[j] if 0 <= j < n else 0.0
        lag = sum(cell(k - G - R_ + d) for d in range(R_)) * div
        lead = sum(cell(k + G + 1 + d) for d in range(R_)) * div
        s = {O.CFAR_CA: 0.5 * (lag + lead), O.CFAR_GO: max(lag, lead), O.CFAR_SO: min(lag, lead)}[mode]
        thr[k] = s * scaler
    return thr


@pytest.mark.parametrize("edge", [O.EDGE_ZERO, O.EDGE_WRAP])
@pytest.mark.parametrize("mode", [O.CFAR_CA, O.CFAR_GO, O.CFAR_SO])
def test_float_chain_against_numpy(edge, mode):
    n = 256
    x = R.stimulus.chirp_frames(2, n, seed=21)
    cfg = O.default_fcfg(log2n=8, cfar_mode=mode, ref_window=8, guard_window=2, div_sum=3, edge=edge)
    thr, peak, margin, mag = O.chain_f32(x, cfg, want_mag=True)
    X = np.fft.fft(x.astype(np.complex128), axis=1) / n
    u = np.maximum(np.abs(X.real), np.abs(X.imag)); v = np.minimum(np.abs(X.real), np.abs(X.imag))
    mref = np.maximum(u + v / 8, 7 * u / 8 + v / 2)
    assert np.allclose(mag, mref, rtol=1e-12, atol=1e-15)
    for f in range(2):
        tref = _cfar_numpy(mref[f], 8, 2, mode, 3.5, 1 / 8, edge)
        assert np.allclose(thr[f], tref, rtol=1e-12, atol=1e-15)
        assert np.array_equal(peak[f], (mref[f] > tref).astype(np.uint8))


def test_range_doppler_oracle_against_numpy():
    nr, nd = 32, 16
    rng = np.random.default_rng(31)
    x = (rng.standard_normal((2, nd, nr)) + 1j * rng.standard_normal((2, nd, nr))).astype(np.complex64)
    cfg = O.OrcRdCfg(log2nr=5, log2nd=4, mag_mode=O.MAG_JPL, scaler=3.0, ref_r=3, ref_d=2, guard_r=1,
                     guard_d=1, edge=O.EDGE_ZERO)
    thr, peak, margin, mag = O.rd_f32(x, cfg, want_mag=True)
    X = np.fft.fft2(x.astype(np.complex128), axes=(1, 2)) / (nr * nd)
    u = np.maximum(np.abs(X.real), np.abs(X.imag)); v = np.minimum(np.abs(X.real), np.abs(X.imag))
    mref = np.maximum(u + v / 8, 7 * u / 8 + v / 2)
    assert np.allclose(mag, mref, rtol=1e-10)
    hr, hd = 4, 3
    cnt = (2 * hr + 1) * (2 * hd + 1) - 3 * 3
    for ch in range(2):
        for d in (0, 5, 15):
            for r in (0, 3, 17, 31):
                tot = 0.0
                for dd in range(-hd, hd + 1):
                    for rr in range(-hr, hr + 1):
                        if abs(dd) <= 1 and abs(rr) <= 1:
                            continue
                        r2 = r + rr
                        if 0 <= r2 < nr:                        # range: zeros outside; Doppler: cyclic
                            tot += mref[ch, (d + dd) % nd, r2]
                assert np.isclose(thr[ch, d, r], 3.0 * tot / cnt, rtol=1e-9)


def test_range_doppler_fixed_oracle_from_its_parts():
    """orc_rd_fixed restated with the 1-D pieces the KATs above pin (orc_fft_fixed, orc_mag_fixed, the threshold rule of
    orc_cfar_fixed) and a brute-force 2-D training sum; GO >= CA-like ordering of the half statistics."""
    nr, nd, n_ch = 32, 16, 2
    rng = np.random.default_rng(77)
    re = rng.integers(-6000, 6001, size=(n_ch, nd, nr)); im = rng.integers(-6000, 6001, size=(n_ch, nd, nr))
    beats = O.pack_iq(re, im).reshape(n_ch, nd, nr)
    cfg = O.default_cfg(log2n=5, ref_window=3, guard_window=1, div_sum=5, scaler=int(2.5 * 4096), cfar_mode=O.CFAR_CA,
                        edge=O.EDGE_ZERO)
    words, mag = O.rd_fixed(beats, cfg, 4, 2, 1, want_mag=True)
    # magnitudes: rows then columns through the 1-D fixed FFT
    rr_ = np.zeros((n_ch, nd, nr), np.int16); ri_ = np.zeros_like(rr_)
    for ch in range(n_ch):
        for d in range(nd):
            rr_[ch, d], ri_[ch, d] = O.fft_fixed(re[ch, d].astype(np.int16), im[ch, d].astype(np.int16), cfg.trim)
    mref = np.zeros((n_ch, nd, nr), np.int64)
    for ch in range(n_ch):
        for r in range(nr):
            cr, ci = O.fft_fixed(rr_[ch, :, r].copy(), ri_[ch, :, r].copy(), cfg.trim)
            mref[ch, :, r] = O.mag_fixed(cr, ci, cfg)
    assert np.array_equal(mag, mref)
    hr, hd = 4, 3
    for ch in range(n_ch):
        for d in (0, 7, 15):
            for r in (0, 2, 16, 31):
                tot = 0
                for dd in range(-hd, hd + 1):
                    for q in range(-hr, hr + 1):
                        if abs(dd) <= 1 and abs(q) <= 1:
                            continue
                        if 0 <= r + q < nr:
                            tot += int(mref[ch, (d + dd) % nd, r + q])
                thr = ((tot >> 5) * cfg.scaler) >> cfg.bp_scaler           # bp_in = bp_thr: shift by bp_scaler
                thr = min(thr, (1 << (cfg.w_thr - 1)) - 1)
                w = int(words[ch, d, r])
                assert w >> 6 == thr and (w >> 1) & 31 == r and (w & 1) == int(mref[ch, d, r] > thr)
    cfg.cfar_mode = O.CFAR_GO
    go = O.rd_fixed(beats, cfg, 4, 2, 1) >> 6
    cfg.cfar_mode = O.CFAR_SO
    so = O.rd_fixed(beats, cfg, 4, 2, 1) >> 6
    assert np.all(go >= so) and np.any(go > so)
