"""Elaboration options of FFTParams.fixed / CFARParams that every reference configuration leaves at its default
(/root/reference/src/main/scala/FftMagCfarChain.scala:82,86-87,107) and the pre-FFT window:
sendCut = true, useBitReverse = false, keepMSBorLSB(s) = false, expandLogic(s) = 1.  All BUILD-DEFINED
(docs/FIXED_POINT_SPEC.md sections 2.1, 3, 6); GPU == oracle bit for bit (FIXED16), in tolerance (F32)."""
import numpy as np
import pytest

import rsp_chains_amd as R
from oracle import oracle as O
from helpers import compare_f32, make_params, oracle_cfg, oracle_fcfg, random_beats, tone_beats

pytestmark = pytest.mark.gpu

# (numPoints, run-time fftSize, algorithm, guard): small kernel / quad tail / per-cell tail / GOS kernel
SHAPES = [(64, 64, R.CACFARType, 2), (1024, 1024, R.CACFARType, 4), (1024, 512, R.CACFARType, 3),
          (4096, 4096, R.CACFARType, 4), (1024, 1024, R.GOSCFARType, 4), (8192, 8192, R.CACFARType, 4)]


def rt_for(n, alg, guard, **kw):
    ref = 16 if n >= 256 else 4
    gos = alg == R.GOSCFARType
    return R.RunTimeRspChainParams(fftSize=n, refWindowSize=ref, guardWindowSize=guard, divSum=None if gos else 4,
                                   indexLagg=5 if gos else None, indexLead=9 if gos else None, thresholdScaler=2.5, **kw)


def run(params, rt, beats):
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        return dut.stream(beats)


@pytest.mark.parametrize("npts,n,alg,guard", SHAPES)
def test_fixed_send_cut(gpu, npts, n, alg, guard):
    """sendCut = true: 64-bit beat {word, cut}; the word is the sendCut = false word, the cut the magnitude."""
    params = make_params(npts, alg=alg, guard=8, sendCut=True)
    plain = make_params(npts, alg=alg, guard=8)
    rt = rt_for(n, alg, guard)
    beats = np.concatenate([tone_beats(2, n, 3 + n) if n >= 256 else random_beats(2, n, 1), random_beats(67, n, n)])
    got = run(params, rt, beats)
    assert got.shape == (69, n, 2)
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref)
    assert np.array_equal(got[..., 0], run(plain, rt, beats))
    # detection list from the wide beats (stand-alone compaction and host call)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        det, found = dut.detections(beats)
    fr, bn = np.nonzero(got[..., 0] & 1)
    assert found == fr.size and np.array_equal(det["frame"], fr) and np.array_equal(det["bin"], bn)
    assert np.array_equal(det["word"], got[fr, bn, 0])


@pytest.mark.parametrize("n", [128, 2048])
def test_f32_send_cut(gpu, n):
    params = make_params(n, dtype=R.F32, sendCut=True, guard=8)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging", refWindowSize=16, guardWindowSize=4, divSum=4)
    x = R.stimulus.chirp_frames(5, n, seed=n)
    got = run(params, rt, x)
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(params, rt), want_mag=True)
    compare_f32(got[..., 0], thr, peak, margin, mag)
    cut = got[..., 1].view(np.float32).astype(np.float64)
    # one bin's magnitude: relative 2e-5, plus the fp32 FFT's absolute error floor, 1e-7 of the frame's peak
    assert np.all(np.abs(cut - mag) <= 2e-5 * np.abs(mag) + 1e-7 * np.abs(mag).max(axis=-1, keepdims=True))


@pytest.mark.parametrize("npts,n,alg,guard", SHAPES)
def test_fixed_use_bit_reverse_false(gpu, npts, n, alg, guard):
    """useBitReverse = false: the FFT block streams bin bitrev(p) at position p and the magnitude / CFAR blocks
    work on that order (the CFAR's bin field is its own position counter)."""
    params = make_params(npts, alg=alg, guard=8, useBitReverse=False)
    rt = rt_for(n, alg, guard, peakGrouping=1)
    beats = np.concatenate([tone_beats(2, n, 4 + n) if n >= 256 else random_beats(2, n, 2), random_beats(5, n, n + 1)])
    got = run(params, rt, beats)
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref)
    assert not np.array_equal(got, run(make_params(npts, alg=alg, guard=8), rt, beats))


@pytest.mark.parametrize("n", [64, 512, 4096])
def test_f32_use_bit_reverse_false(gpu, n):
    params = make_params(n, dtype=R.F32, useBitReverse=False)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Greatest Of", refWindowSize=8, guardWindowSize=4, divSum=3)
    x = R.stimulus.chirp_frames(4, n, seed=9 + n)
    got = run(params, rt, x)
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(params, rt), want_mag=True)
    compare_f32(got, thr, peak, margin, mag)


@pytest.mark.parametrize("npts,n,alg,guard", SHAPES)
@pytest.mark.parametrize("seed", [0, 1])
def test_fixed_stage_options(gpu, npts, n, alg, guard, seed):
    """keepMSBorLSB(s) = false (the stage drops its MSB: no halving, wrap) and expandLogic(s) = 1 (the stage keeps
    its extra bit; the 16 MSBs of the grown word go on to the magnitude block), random per stage, all trim types."""
    rng = np.random.default_rng(100 * seed + n)
    stages = R.log2Up(npts)
    expand = [int(v) for v in rng.integers(0, 2, stages)] if seed else [0] * stages
    keep = [bool(v) for v in rng.integers(0, 2, stages)]
    params = make_params(npts, alg=alg, guard=8, expandLogic=expand, keepMSBorLSB=keep,
                         trim=["Convergent", "RoundHalfUp", "RoundDown"][(n + seed) % 3])
    rt = rt_for(n, alg, guard)
    # small amplitudes: with stages that do not halve, full-scale inputs only exercise the wrap-around
    beats = np.concatenate([random_beats(4, n, n + seed, amp=40), random_beats(3, n, n + 7, amp=32767)])
    got = run(params, rt, beats)
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref), (expand, keep)


def test_stage_options_are_fixed_point_only(gpu):
    with pytest.raises(ValueError):
        R.FftMagCfarChainVanilla(make_params(1024, dtype=R.F32, keepMSBorLSB=[False] * 10))


@pytest.mark.parametrize("window", ["hann", "hamming", "blackman"])
@pytest.mark.parametrize("npts,n,alg,guard", SHAPES[:5])
def test_fixed_window(gpu, window, npts, n, alg, guard):
    params = make_params(npts, alg=alg, guard=8, window=window)
    rt = rt_for(n, alg, guard)
    beats = np.concatenate([tone_beats(2, n, 5 + n) if n >= 256 else random_beats(2, n, 3), random_beats(5, n, n + 2)])
    got = run(params, rt, beats)
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(got.shape)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("window", ["hann", "blackman"])
@pytest.mark.parametrize("n", [32, 1024, 8192])
def test_f32_window(gpu, window, n):
    params = make_params(n, dtype=R.F32, window=window)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging", refWindowSize=16 if n > 32 else 4,
                                 guardWindowSize=4 if n > 32 else 1, divSum=4 if n > 32 else 2)
    x = R.stimulus.chirp_frames(4, n, seed=5 + n)
    got = run(params, rt, x)
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(params, rt), want_mag=True)
    compare_f32(got, thr, peak, margin, mag)
    assert not np.array_equal(got, run(make_params(n, dtype=R.F32), rt, x))



def test_host_entry_pipeline_chunks_and_pinned_buffers(gpu):
    """rsp_chain_process moves the batch as chunks on three streams (H2D || kernels || D2H); pageable memory goes
    through the pinned staging ring, rsp_host_alloc'd memory over the link in place.  Every route gives the words of
    the one-launch device call -- with chunks smaller than the batch, a ragged last chunk, more chunks than staging
    slots -- and the detection call (chunked kernels + ONE compaction) the dense peaks."""
    n, frames = 1024, 1000
    params = make_params(n)
    rt = rt_for(n, R.CACFARType, 4)
    beats = np.concatenate([tone_beats(8, n, 5), random_beats(frames - 8, n, 9)])
    ref = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(frames, n)
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(rt)
        assert np.array_equal(dut.stream(beats), ref)                       # one chunk (4 MB < 16 MiB)
        dut.set_option(dut.HOST_CHUNK_BYTES, 300 * 1024)                    # 64-frame chunks: 16 of them, ragged last
        assert np.array_equal(dut.stream(beats), ref)
        hin, hout = R.HostBuffer((frames, n), np.uint32), R.HostBuffer(frames * n, np.uint32)
        hin.array[...] = beats
        hout.array[...] = 0
        got = dut.stream(hin.array, out=hout.array)                         # pinned both ways
        assert np.array_equal(got, ref) and np.array_equal(hout.array.reshape(frames, n), ref)
        out2 = np.zeros(frames * n, np.uint32)
        assert np.array_equal(dut.stream(hin.array, out=out2), ref)         # pinned in, pageable out
        det, found = dut.detections(beats)
        fr, bn = np.nonzero(ref & 1)
        assert found == fr.size and np.array_equal(det["frame"], fr) and np.array_equal(det["bin"], bn)
        assert np.array_equal(det["word"], ref[fr, bn])
        with pytest.raises(ValueError):
            dut.stream(beats, out=np.zeros(5, np.uint32))
