"""GPU parity: the HIP path called through the C ABI vs the CPU oracle.
FIXED16 must be bit-exact; F32 within the tolerance written in helpers.compare_f32."""
import numpy as np
import pytest

import rsp_chains_amd as R
from oracle import oracle as O
from helpers import (compare_f32, make_params, oracle_cfg, oracle_fcfg, random_beats, tone_beats)

pytestmark = pytest.mark.gpu


def run_fixed(params, rt, beats):
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        return dut.stream(beats)


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, 8192, 16384])
def test_fixed_bit_exact_all_sizes(gpu, n):
    params = make_params(n)
    rt = R.RunTimeRspChainParams(fftSize=n)
    beats = np.concatenate([tone_beats(3, n, 10 + n), random_beats(6, n, n)])  # ragged vs frames/WG
    got = run_fixed(params, rt, beats)
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref)


def test_fixed_reference_tester_procedure(gpu):
    """FftMagCfarChainVanillaSpec (Tester:195-249) with real assertions: tones at 1/8, 1/4,
    1/2 and the noise's DC show up as exactly the four peaks."""
    n = 1024
    params = make_params(n)
    rt = R.RunTimeRspChainParams()
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        out = dut.stream(tone_beats(1, n, 1234))[0]
    thr, bins, peaks = R.unpack_output(out, n)
    assert np.array_equal(bins, np.arange(n))
    assert list(np.nonzero(peaks)[0]) == [0, 128, 256, 512]
    assert np.all(thr >= 0)


@pytest.mark.parametrize("mode", ["Cell Averaging", "Greatest Of", "Smallest Of"])
@pytest.mark.parametrize("edge", ["zero", "wrap"])
@pytest.mark.parametrize("ref,guard", [(16, 4), (64, 2), (2, 1)])
def test_fixed_cfar_modes(gpu, mode, edge, ref, guard):
    n = 1024
    params = make_params(n, edge=edge)
    rt = R.RunTimeRspChainParams(CFARMode=mode, refWindowSize=ref, guardWindowSize=guard,
                                 divSum=R.log2Up(ref) if ref > 1 else 0, peakGrouping=1 if ref == 16 else 0)
    beats = np.concatenate([tone_beats(2, n, 77), random_beats(3, n, 78)])
    got = run_fixed(params, rt, beats)
    ref_out = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref_out)


@pytest.mark.parametrize("trim", ["RoundDown", "RoundHalfUp", "Convergent"])
@pytest.mark.parametrize("mag", [0, 1, 2])
def test_fixed_trim_and_mag_modes(gpu, trim, mag):
    n = 512
    params = make_params(n, trim=trim)
    rt = R.RunTimeRspChainParams(fftSize=n, magMode=mag, logOrLinearMode=0 if mag == 1 else 1,
                                 thresholdScaler=1.25 if mag == 1 else 3.5)
    beats = np.concatenate([tone_beats(2, n, 5), random_beats(2, n, 6, amp=32767)])
    got = run_fixed(params, rt, beats)
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref)


def test_fixed_extreme_inputs(gpu):
    """Full-scale and sign-corner samples: wrap-around and saturation follow the spec."""
    n = 256
    params = make_params(n)
    rt = R.RunTimeRspChainParams(fftSize=n, refWindowSize=16, divSum=4)
    rng = np.random.default_rng(3)
    corner = rng.choice(np.array([-32768, -32767, -1, 0, 1, 32767]), size=(4, n, 2))
    beats = O.pack_iq(corner[..., 0], corner[..., 1])
    beats[0, :] = O.pack_iq(np.full(n, -32768), np.full(n, -32768))
    got = run_fixed(params, rt, beats)
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref)


def test_fixed_bp0_mixed_protos(gpu):
    """RspChainVanillaSpec's formats: data BP 0, threshold BP 3, scaler BP 6
    (RspChainVanillaTester.scala:205-239) on an NCO-like tone (bin 32)."""
    n = 1024
    params = make_params(n, bp=0, leadLagg=32,
                         proto=(R.FixedPoint(16, 0), R.FixedPoint(16, 3), R.FixedPoint(16, 6)))
    rt = R.RunTimeRspChainParams()
    tone = R.stimulus.calcExpectedNcoOut(n, 32)
    beats = R.stimulus.formAXI4StreamComplexData(tone)[None, :]
    got = run_fixed(params, rt, beats)
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref)
    _, _, peaks = R.unpack_output(got[0], n)
    assert list(np.nonzero(peaks)[0]) == [32]


def test_runtime_fft_size(gpu):
    """runTime = true: the stages register selects a smaller FFT than numPoints (Tester:82)."""
    params = make_params(4096)
    rt = R.RunTimeRspChainParams(fftSize=512)
    beats = random_beats(5, 512, 9)
    got = run_fixed(params, rt, beats)
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref)


def test_empty_and_detections(gpu):
    n = 1024
    params = make_params(n)
    rt = R.RunTimeRspChainParams()
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        assert dut.stream(np.zeros(0, np.uint32)).shape == (0, n)
        beats = tone_beats(7, n, 400)
        dense = dut.stream(beats)
        det, found = dut.detections(beats)
        with pytest.raises(ValueError):
            dut.stream(np.zeros(n + 1, np.uint32))  # not a whole frame
    fr, bn = np.nonzero(dense & 1)
    assert found == fr.size == det.size
    assert np.array_equal(det["frame"], fr) and np.array_equal(det["bin"], bn)
    assert np.array_equal(det["word"], dense[fr, bn])


def test_fused_detection_list_matches_dense(gpu):
    """rsp_chain_process_detect_device: the list emitted by the chain kernel equals the peaks of the
    dense words, and the standalone compaction of those words gives the same set."""
    n, frames = 1024, 37
    params = make_params(n)
    rt = R.RunTimeRspChainParams()
    beats = tone_beats(frames, n, 900)
    cap = 4096
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        d_in = R.DeviceBuffer(beats.nbytes); d_in.upload(beats)
        d_out = R.DeviceBuffer(beats.size * 4)
        d_list, d_cnt = R.DeviceBuffer(cap * 16), R.DeviceBuffer(8)
        d_list2, d_cnt2 = R.DeviceBuffer(cap * 16), R.DeviceBuffer(8)
        dut.process_detect_device(d_in.ptr, frames, d_out.ptr, d_list.ptr, cap, d_cnt.ptr)
        dut.detections_device(d_out.ptr, frames, d_list2.ptr, cap, d_cnt2.ptr)
        dut.synchronize()
        dense = d_out.download(np.uint32, beats.size).reshape(frames, n)
        found, stored = (int(v) for v in d_cnt.download(np.uint32, 2))
        found2, stored2 = (int(v) for v in d_cnt2.download(np.uint32, 2))
        assert found == stored and found2 == stored2
        lst = d_list.download(np.uint32, found * 4).reshape(found, 4)
        lst2 = d_list2.download(np.uint32, found2 * 4).reshape(found2, 4)
        # list-only mode (no dense words)
        dut.process_detect_device(d_in.ptr, frames, 0, d_list2.ptr, cap, d_cnt2.ptr)
        dut.synchronize()
        found3 = int(d_cnt2.download(np.uint32, 1)[0])
        lst3 = d_list2.download(np.uint32, found3 * 4).reshape(found3, 4)
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(frames, n)
    assert np.array_equal(dense, ref)
    fr, bn = np.nonzero(dense & 1)
    want = sorted(zip(fr.tolist(), bn.tolist(), dense[fr, bn].tolist()))
    for got in (lst, lst2, lst3):
        assert sorted((int(a), int(b), int(w)) for a, b, _, w in got) == want


def test_detection_list_is_complete_beyond_the_per_frame_staging(gpu):
    """Frames with far more than RSP_FRAME_DET_CAP = 64 peaks (Smallest Of, scaler 0.75 on noise: hundreds
    per frame).  The reference emits every peak (Tester:145-167), so: the host call and the fused
    device call WITH dense words return every one of them; the list-only device call keeps at most 64
    per frame and says so through d_count = {found, stored}."""
    n, frames = 1024, 300                                   # 300 frames: two compaction workgroups
    params = make_params(n)
    rt = R.RunTimeRspChainParams(CFARMode="Smallest Of", thresholdScaler=0.75)
    beats = random_beats(frames, n, 4711, amp=3000)
    beats[5] = 0                                            # frames without a single peak in between
    beats[17:20] = 0
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(frames, n)
    fr, bn = np.nonzero(ref & 1)
    per_frame = np.bincount(fr, minlength=frames)
    assert per_frame.max() > 200 and per_frame.min() == 0
    want = sorted(zip(fr.tolist(), bn.tolist(), ref[fr, bn].tolist()))
    cap = 1 << 18
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        det, found = dut.detections(beats, cap=cap)
        assert found == fr.size == det.size
        assert np.array_equal(det["frame"], fr) and np.array_equal(det["bin"], bn) and np.array_equal(det["word"], ref[fr, bn])
        small, found_small = dut.detections(beats, cap=1000)   # cap smaller than the list: truncated, no garbage
        assert found_small == fr.size and small.size == 1000
        assert set(zip(small["frame"].tolist(), small["bin"].tolist())) <= set(zip(fr.tolist(), bn.tolist()))
        d_in = R.DeviceBuffer(beats.nbytes); d_in.upload(beats)
        d_out = R.DeviceBuffer(beats.size * 4)
        d_list, d_cnt = R.DeviceBuffer(cap * 16), R.DeviceBuffer(8)
        dut.process_detect_device(d_in.ptr, frames, d_out.ptr, d_list.ptr, cap, d_cnt.ptr)
        dut.synchronize()
        cnt = d_cnt.download(np.uint32, 2)
        assert cnt[0] == fr.size and cnt[1] == fr.size
        lst = d_list.download(np.uint32, int(cnt[1]) * 4).reshape(-1, 4)
        assert sorted((int(a), int(b), int(w)) for a, b, _, w in lst) == want
        assert np.array_equal(lst[:, 0], fr)          # frames in ascending order, the overflowing ones (helper workgroups) too
        dut.process_detect_device(d_in.ptr, frames, 0, d_list.ptr, cap, d_cnt.ptr)   # list only
        dut.synchronize()
        cnt = d_cnt.download(np.uint32, 2)
        assert cnt[0] == fr.size and cnt[1] == np.minimum(per_frame, 64).sum()
        lst = d_list.download(np.uint32, int(cnt[1]) * 4).reshape(-1, 4)
        got = set((int(a), int(b), int(w)) for a, b, _, w in lst)
        assert len(got) == cnt[1] and got <= set(want)
        assert np.array_equal(np.bincount(lst[:, 0], minlength=frames), np.minimum(per_frame, 64))


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, 8192, 16384])
def test_f32_all_sizes(gpu, n):
    params = make_params(n, dtype=R.F32, leadLagg=64)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging")
    x = R.stimulus.chirp_frames(5, n, seed=1234 + n)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(params, rt), want_mag=True)
    compare_f32(words, thr, peak, margin, mag)
    # the three injected targets are detected in every frame
    assert np.all((words & 1).sum(axis=1) >= 3)


@pytest.mark.parametrize("mode", ["Cell Averaging", "Greatest Of", "Smallest Of"])
@pytest.mark.parametrize("edge", ["zero", "wrap"])
@pytest.mark.parametrize("mag", [0, 1, 2])
def test_f32_modes(gpu, mode, edge, mag):
    n = 2048
    params = make_params(n, dtype=R.F32, edge=edge, leadLagg=128)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode=mode, magMode=mag, refWindowSize=128, divSum=7,
                                 guardWindowSize=3, peakGrouping=1, logOrLinearMode=0 if mag == 1 else 1,
                                 thresholdScaler=2.0 if mag == 1 else 3.5)
    x = R.stimulus.chirp_frames(4, n, seed=99 + mag)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    thr, peak, margin, magr = O.chain_f32(x, oracle_fcfg(params, rt), want_mag=True)
    compare_f32(words, thr, peak, margin, magr, atol=2.0 ** -9 if mag == 1 else 0.0)


def test_f32_linearity_full_batch(gpu):
    """Size-independent property at BASELINE.json's full cfg-2 size (4096 chirps x 4096 points):
    the chain is homogeneous of degree 1 (JPL magnitude, linear CFAR), so scaling the input by 2^-3
    scales every threshold by exactly 2^-3 and leaves every peak flag unchanged."""
    n, frames = 4096, 4096
    params = make_params(n, dtype=R.F32)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging")
    x = R.stimulus.chirp_frames(64, n, seed=1234)
    x = np.tile(x, (frames // 64, 1))
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        a = dut.stream(x)
        b = dut.stream(x * np.float32(0.125))
    ta, pa = R.unpack_output_f32(a)
    tb, pb = R.unpack_output_f32(b)
    assert np.array_equal(pa, pb)
    assert np.array_equal(ta * np.float32(0.125), tb)
    # identical frames give identical rows (no cross-frame state)
    assert np.array_equal(a[:64], a[64:128])


def test_cpp_host_tester_replays_dumps(gpu, tmp_path):
    """The C++ host mirror (host/RspChain.hpp) run as the reference tester: reads the tester's
    input dumps, writes outputData.txt; must equal the Python host path and the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "rsp-chains_amd", "host")
    subprocess.run(["make", "-C", host], check=True, stdout=subprocess.DEVNULL)
    n = 1024
    z = R.stimulus.getComplexTones(n, 0.125, 0.25, 0.5, shiftRangeFactor=12, seed=777)
    R.dumps.write_input_dumps(str(tmp_path), z)
    out = subprocess.run([os.path.join(host, "tester"), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "peaks 4" in out.stdout
    got = R.dumps.read_output_words(str(tmp_path))
    params, rt = make_params(n), R.RunTimeRspChainParams()
    ref = O.chain_fixed(R.stimulus.formAXI4StreamComplexData(z), oracle_cfg(params, rt))
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("n,ref,guard,idx", [(256, 4, 1, (1, 2)), (1024, 16, 2, (11, 11)), (1024, 32, 4, (24, 8)),
                                            (4096, 64, 4, (48, 48)), (8192, 32, 4, (24, 24)), (8192, 64, 4, (40, 13)),
                                            # the two-run stage with a compile-time index (k = R/2, 3R/4 of 32), odd guards
                                            (4096, 32, 3, (16, 16)), (2048, 32, 1, (24, 24)), (512, 32, 3, (16, 16)),
                                            # 256- / 512-point frames: 18 - 20 window starts per thread on the split path
                                            (512, 32, 4, (20, 9)), (256, 32, 2, (24, 24)), (256, 32, 4, (7, 30)),
                                            # the 64-cell window on the split path at 18 / 19 / 21 starts per thread
                                            (1024, 64, 4, (48, 48)), (512, 64, 2, (10, 50)), (256, 64, 3, (32, 32))])
@pytest.mark.parametrize("mode", ["Cell Averaging", "Greatest Of", "Smallest Of"])
def test_fixed_gos_bit_exact(gpu, n, ref, guard, idx, mode):
    """GOSCFARType (FftMagCfarChainTester.scala:105,123-127): ordered-statistic CFAR, bit-exact."""
    params = make_params(n, alg=R.GOSCFARType, edge="wrap" if ref == 16 else "zero")
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode=mode, refWindowSize=ref, guardWindowSize=guard, divSum=None,
                                 indexLagg=idx[0], indexLead=idx[1], thresholdScaler=1.5, peakGrouping=1 if ref == 4 else 0)
    beats = np.concatenate([tone_beats(2, n, 60 + n), random_beats(3, n, n + 1)])
    got = run_fixed(params, rt, beats)
    ref_out = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref_out)


def test_gosca_runtime_algorithm_select(gpu):
    """GOSCACFARType: register 0x14 picks CA or GOS at run time (Tester:110-118)."""
    n = 1024
    params = make_params(n, alg=R.GOSCACFARType)
    beats = np.concatenate([tone_beats(2, n, 5), random_beats(2, n, 6)])
    for alg in ("CA", "GOS"):
        rt = R.RunTimeRspChainParams(CFARAlgorithm=alg, indexLagg=20, indexLead=20)
        got = run_fixed(params, rt, beats)
        assert np.array_equal(got, O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)), alg


@pytest.mark.parametrize("n,ref,idx", [(8192, 32, 24), (2048, 16, 12), (16384, 32, 24), (16384, 64, 40)])
def test_f32_gos(gpu, n, ref, idx):
    """BASELINE.json configs[3]: OS-CFAR, 32-cell window, 8192-point spectrum (k = 3R/4, G = 4)."""
    params = make_params(n, dtype=R.F32, alg=R.GOSCFARType)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Greatest Of", refWindowSize=ref, guardWindowSize=4, divSum=None,
                                 indexLagg=idx, indexLead=idx, thresholdScaler=2.5)
    x = R.stimulus.chirp_frames(3, n, seed=3456)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(params, rt), want_mag=True)
    compare_f32(words, thr, peak, margin, mag)
    assert np.all((words & 1).sum(axis=1) >= 3)


@pytest.mark.parametrize("n,ref,sub,edge", [(1024, 32, 8, "zero"), (512, 16, 4, "wrap"), (4096, 64, 16, "zero"), (256, 8, 2, "zero")])
def test_cash_mode(gpu, n, ref, sub, edge):
    """cfarMode 3 = CASH on a CACFARType build with includeCASH = true: the subWindowSize register
    at offset 0x2C exists only then (FftMagCfarChainTester.scala:129-132).  FIXED16 bit-exact, F32 in tolerance."""
    params = make_params(n, includeCASH=True, edge=edge)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="CASH", refWindowSize=ref, guardWindowSize=2, subWindowSize=sub,
                                 divSum=R.log2Up(sub), thresholdScaler=3.0, peakGrouping=1 if sub == 4 else 0)
    beats = np.concatenate([tone_beats(2, n, 31), random_beats(2, n, 32)])
    got = run_fixed(params, rt, beats)
    assert np.array_equal(got, O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape))
    paramsf = make_params(n, dtype=R.F32, includeCASH=True, edge=edge)
    x = R.stimulus.chirp_frames(3, n, seed=44)
    with R.FftMagCfarChainVanilla(paramsf) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(paramsf, rt), want_mag=True)
    # sub-window sums of <= 16 cells are differences of prefixes relative to 16-cell blocks (not 256): the common tolerance
    compare_f32(words, thr, peak, margin, mag)


def test_cash_register_only_exists_with_includeCASH(gpu):
    with R.FftMagCfarChainVanilla(make_params(1024)) as dut:
        with pytest.raises(IndexError):
            dut.memWriteWord(0x30002000 + 11 * 4, 8)
        dut.configure(R.RunTimeRspChainParams())
        dut.memWriteWord(0x30002000 + 6 * 4, 3)          # cfarMode = CASH on a build without it
        with pytest.raises(ValueError):
            dut.check()


def test_chunked_launch_path(gpu):
    """Launches whose input would exceed 4 GiB are split (kernels use 32-bit byte offsets).  Force the
    split at 64 frames and check dense words and the fused detection list are unchanged."""
    n, frames = 1024, 200
    params = make_params(n)
    rt = R.RunTimeRspChainParams()
    beats = np.concatenate([tone_beats(8, n, 71), random_beats(frames - 8, n, 72)])
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(frames, n)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        dut.set_option(dut.MAX_FRAMES_PER_LAUNCH, 50)     # rounded up to whole 64-frame pieces
        assert np.array_equal(dut.stream(beats), ref)
        det, found = dut.detections(beats)
    fr, bn = np.nonzero(ref & 1)
    assert found == fr.size and np.array_equal(det["frame"], fr) and np.array_equal(det["bin"], bn)


def test_two_chains_on_two_streams(gpu):
    """Distinct handles are independent (one per host thread): interleaved use gives the same results."""
    n = 512
    pa, pb = make_params(n), make_params(n, dtype=R.F32)
    rta = R.RunTimeRspChainParams(fftSize=n)
    rtb = R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging")
    beats = random_beats(9, n, 5)
    x = R.stimulus.chirp_frames(9, n, seed=6)
    with R.FftMagCfarChainVanilla(pa) as a, R.FftMagCfarChainVanilla(pb) as b:
        a.configure(rta); b.configure(rtb)
        outs = [(a.stream(beats), b.stream(x)) for _ in range(3)]
    for oa, ob in outs:
        assert np.array_equal(oa, outs[0][0]) and np.array_equal(ob, outs[0][1])
    assert np.array_equal(outs[0][0], O.chain_fixed(beats, oracle_cfg(pa, rta)).reshape(9, n))


@pytest.mark.parametrize("n", [16, 32, 64, 128])
def test_small_runtime_fft_sizes(gpu, n):
    """runTime = true: the stages register may select any 2^k <= numPoints (Tester:82).  Frames below
    256 points run on the one-thread-per-frame kernel; FIXED16 bit-exact, F32 in tolerance, all modes."""
    ref_w, guard = (2, 1) if n == 16 else (4, 2)
    for mode, alg, kw in (("Greatest Of", R.CACFARType, {}), ("Cell Averaging", R.CACFARType, dict(peakGrouping=1)),
                          ("Smallest Of", R.GOSCFARType, dict(indexLagg=1, indexLead=ref_w - 1, divSum=None)),
                          ("CASH", R.CACFARType, dict(subWindowSize=ref_w // 2))):
        params = make_params(1024, alg=alg, includeCASH=(mode == "CASH"), edge="wrap" if mode == "Cell Averaging" else "zero")
        rt = R.RunTimeRspChainParams(fftSize=n, CFARMode=mode, refWindowSize=ref_w, guardWindowSize=guard,
                                     **{"divSum": R.log2Up(ref_w), **kw})
        beats = random_beats(70, n, 100 + n)       # 70 frames: more than one 64-frame workgroup
        got = run_fixed(params, rt, beats)
        assert np.array_equal(got, O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)), (n, mode)
    paramsf = make_params(1024, dtype=R.F32)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging", refWindowSize=ref_w, guardWindowSize=guard,
                                 divSum=R.log2Up(ref_w))
    x = R.stimulus.chirp_frames(5, n, seed=n)
    with R.FftMagCfarChainVanilla(paramsf) as dut:
        dut.configure(rt)
        words = dut.stream(x)
        det, found = dut.detections(x)
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(paramsf, rt), want_mag=True)
    compare_f32(words, thr, peak, margin, mag)
    assert found == int((words & 1).sum())


@pytest.mark.parametrize("frames", [3000, 20000])
def test_fused_list_order_and_large_frame_counts(gpu, frames):
    """Up to 16384 frames the frame compaction takes its offsets from a prefix over the frame counts: the list comes out
    with the frames in ascending order (a frame's peaks contiguous, in staging order).  Above that, blocks of 256 frames
    are reserved with one atomic each (any block order); both are complete."""
    n = 256
    params = make_params(n, leadLagg=16, guard=2)
    rt = R.RunTimeRspChainParams(fftSize=n, refWindowSize=16, guardWindowSize=2, thresholdScaler=2.0)
    beats = random_beats(frames, n, 31337, amp=8000)
    cap = 1 << 22
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        d_in = R.DeviceBuffer(beats.nbytes); d_in.upload(beats)
        d_out = R.DeviceBuffer(beats.size * 4)
        d_list, d_cnt = R.DeviceBuffer(cap * 16), R.DeviceBuffer(8)
        dut.process_detect_device(d_in.ptr, frames, d_out.ptr, d_list.ptr, cap, d_cnt.ptr)
        dut.synchronize()
        dense = d_out.download(np.uint32, beats.size).reshape(frames, n)
        found, stored = (int(v) for v in d_cnt.download(np.uint32, 2))
        lst = d_list.download(np.uint32, stored * 4).reshape(stored, 4)
    fr, bn = np.nonzero(dense & 1)
    assert found == stored == fr.size and fr.size > frames
    key = lst[:, 0].astype(np.int64) * n + lst[:, 1]
    if frames <= 16384:
        assert np.array_equal(lst[:, 0], fr)                                         # frames already in order
    else:
        assert np.array_equal(np.sort(key), fr.astype(np.int64) * n + bn)
    assert np.array_equal(np.sort(key), fr.astype(np.int64) * n + bn)
    assert np.array_equal(lst[np.argsort(key), 3], dense[fr, bn])


def test_16384_point_frames(gpu):
    """numPoints = 16384 (FFTParams.fixed accepts any power of two: FftMagCfarChain.scala:78-90): one 1024-thread workgroup
    per frame, 137 KiB of LDS.  CA / GO / SO on the quad tail in both data types (sizes above), a smaller run-time
    fftSize on the same object, the wide beat; what does not fit the 160 KiB says so instead of failing in the launch."""
    n = 16384
    params = make_params(n, guard=8)
    for mode, size in (("Greatest Of", n), ("Smallest Of", n), ("Cell Averaging", 2048)):
        rt = R.RunTimeRspChainParams(fftSize=size, CFARMode=mode, refWindowSize=32, guardWindowSize=8, divSum=5,
                                     thresholdScaler=2.0, peakGrouping=1)
        beats = np.concatenate([tone_beats(2, size, 77), random_beats(3, size, 78)])
        got = run_fixed(params, rt, beats)
        assert np.array_equal(got, O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)), mode
    wide = make_params(n, sendCut=True, guard=8)
    rt = R.RunTimeRspChainParams(fftSize=n, refWindowSize=16, guardWindowSize=4, divSum=4)
    beats = random_beats(3, n, 79)
    got = run_fixed(wide, rt, beats)
    assert np.array_equal(got, O.chain_fixed(beats, oracle_cfg(wide, rt)).reshape(got.shape))
    # per-cell tail (window sizes that are not multiples of 4) and the ordered statistic on FIXED16: the twiddle ROM
    # overlays the frame's LDS region, so both fit (145 / 140 KiB)
    odd = R.RunTimeRspChainParams(fftSize=n, refWindowSize=8, guardWindowSize=3, divSum=3)
    got = run_fixed(params, odd, beats)
    assert np.array_equal(got, O.chain_fixed(beats, oracle_cfg(params, odd)).reshape(got.shape))
    gos = make_params(n, alg=R.GOSCFARType, guard=8)
    rtg = R.RunTimeRspChainParams(fftSize=n, CFARMode="Greatest Of", refWindowSize=32, guardWindowSize=4, divSum=None,
                                  indexLagg=24, indexLead=24, thresholdScaler=1.5)
    got = run_fixed(gos, rtg, beats)
    assert np.array_equal(got, O.chain_fixed(beats, oracle_cfg(gos, rtg)).reshape(got.shape))
    # what does not fit: two order-statistic arrays (indexLagg != indexLead) beside the magnitudes, 207 KiB
    rt2 = R.RunTimeRspChainParams(fftSize=n, CFARMode="Greatest Of", refWindowSize=32, guardWindowSize=4, divSum=None,
                                  indexLagg=24, indexLead=8, thresholdScaler=1.5)
    with R.FftMagCfarChainVanilla(gos) as dut:
        dut.configure(rt2)
        with pytest.raises(NotImplementedError, match="160 KiB"):
            dut.stream(beats)
        dut.configure(rtg)                              # the object is still usable
        assert dut.stream(beats).shape == (3, n)
    # the 2-D chain keeps its 8192-point range FFT
    with pytest.raises(NotImplementedError, match="range FFT holds up to 8192"):
        R.FftMagCfarChainVanilla(R.FftMagCfarVanillaParameters(
            fftParams=R.FFTParams.fixed(numPoints=n), magParams=R.MAGParams.fixed(),
            cfarParams=R.CFARParams(fftSize=n, leadLaggWindowSize=16), dtype=R.F32, dopplerPoints=256, refDoppler=8, guardDoppler=2))


@pytest.mark.parametrize("n,ref,guard,idx", [(512, 32, 20, (24, 24)), (512, 32, 9, (16, 16)), (256, 32, 12, (24, 24)), (256, 32, 5, (3, 17)),
                                             (1024, 64, 33, (48, 48)), (1024, 64, 31, (20, 60)), (512, 64, 17, (32, 32)), (256, 64, 9, (48, 5))])
def test_gos_short_frames_wide_guards(gpu, n, ref, guard, idx):
    """Ordered statistic on 256- / 512-point frames with guard windows that change the number of window starts per thread
    (18 / 19 at 512 points, 19 / 20 at 256; the 64-cell window off its split path's run): every instantiation, FIXED16
    bit-exact and fp32 in tolerance."""
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Smallest Of", refWindowSize=ref, guardWindowSize=guard, divSum=None,
                                 indexLagg=idx[0], indexLead=idx[1], thresholdScaler=1.5)
    params = make_params(n, alg=R.GOSCFARType, guard=48)
    beats = np.concatenate([tone_beats(2, n, 90 + n), random_beats(9, n, n + 3)])
    got = run_fixed(params, rt, beats)
    assert np.array_equal(got, O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape))
    pf = make_params(n, dtype=R.F32, alg=R.GOSCFARType, guard=48)
    x = R.stimulus.chirp_frames(5, n, seed=77 + guard)
    with R.FftMagCfarChainVanilla(pf) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(pf, rt), want_mag=True)
    compare_f32(words, thr, peak, margin, mag)


@pytest.mark.parametrize("n,ref,guard", [(256, 8, 2), (1024, 16, 6), (1024, 32, 2), (4096, 32, 10), (4096, 64, 2), (8192, 16, 2), (512, 4, 2)])
@pytest.mark.parametrize("mode,edge,grouping", [("Cell Averaging", "zero", 0), ("Greatest Of", "wrap", 1), ("Smallest Of", "zero", 1)])
def test_quad_tail_even_guards(gpu, n, ref, guard, mode, edge, grouping):
    """guardWindowSize = 2 mod 4 on the quad tail (8-byte window-edge reads, per-half prefix blocks): FIXED16 bit-exact
    against the oracle AND identical to the per-cell tail's words; fp32 in the common tolerance (16-cell-block prefixes
    for the windows of at most 16 cells)."""
    params = make_params(n, guard=16, edge=edge)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode=mode, refWindowSize=ref, guardWindowSize=guard,
                                 divSum=int(np.log2(ref)), thresholdScaler=2.0, peakGrouping=grouping)
    beats = np.concatenate([tone_beats(2, n, 12 + n + guard), random_beats(7, n, n + ref)])
    oracle = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(len(beats), n)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        got = dut.stream(beats)
        dut.set_option(dut.FORCE_GENERIC_TAIL, 1)
        per_cell = dut.stream(beats)
    assert np.array_equal(got, oracle) and np.array_equal(per_cell, oracle)
    pf = make_params(n, dtype=R.F32, guard=16, edge=edge)
    x = R.stimulus.chirp_frames(5, n, seed=guard + n)
    with R.FftMagCfarChainVanilla(pf) as dut:
        dut.configure(rt)
        words = dut.stream(x)
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(pf, rt), want_mag=True)
    compare_f32(words, thr, peak, margin, mag)
