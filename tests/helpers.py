"""Shared builders for the parity tests: the same chain configuration expressed
(a) as the reference-shaped parameter objects driving the GPU path and (b) as the
oracle's register struct."""
import numpy as np

import rsp_chains_amd as R
from oracle import oracle as O

MODE_NAMES = {0: "Cell Averaging", 1: "Greatest Of", 2: "Smallest Of", 3: "CASH"}


WINDOWS = {None: 0, "hann": 1, "hamming": 2, "blackman": 3}


def make_params(n, dtype=R.FIXED16, bp=12, alg=R.CACFARType, edge="zero", trim="Convergent",
                leadLagg=64, guard=4, proto=None, includeCASH=False, sendCut=False, useBitReverse=True,
                expandLogic=(), keepMSBorLSB=(), window=None):
    pin, pthr, psc = proto or (R.FixedPoint(16, bp),) * 3
    return R.FftMagCfarVanillaParameters(
        fftParams=R.FFTParams.fixed(numPoints=n, binPoint=bp, trimType=trim, useBitReverse=useBitReverse,
                                    expandLogic=list(expandLogic), keepMSBorLSB=list(keepMSBorLSB)),
        magParams=R.MAGParams.fixed(binPoint=bp),
        cfarParams=R.CFARParams(protoIn=pin, protoThreshold=pthr, protoScaler=psc, leadLaggWindowSize=leadLagg,
                                guardWindowSize=guard, fftSize=n, CFARAlgorithm=alg, edgeMode=edge,
                                includeCASH=includeCASH, sendCut=sendCut),
        dtype=dtype, window=window)


def stage_masks(params, rt):
    """(keep_lsb_mask, expand_mask) of the ACTIVE stages: a run-time size below numPoints uses the last
    log2(fftSize) stages of the elaborated pipeline (docs/FIXED_POINT_SPEC.md section 3)."""
    f = params.fftParams
    m, m_max = R.log2Up(rt.fftSize), R.log2Up(f.numPoints)
    keep = expand = 0
    for s in range(m):
        es = s + m_max - m
        if f.expandLogic[es]:
            expand |= 1 << s
        elif not f.keepMSBorLSB[es]:
            keep |= 1 << s
    return keep, expand


def oracle_cfg(params, rt):
    """orc_cfg equivalent of params + the register writes chain.configure(rt) performs."""
    c, f, m = params.cfarParams, params.fftParams, params.magParams
    gos = c.CFARAlgorithm == R.GOSCFARType or (c.CFARAlgorithm == R.GOSCACFARType and rt.CFARAlgorithm == "GOS")
    return O.default_cfg(
        log2n=R.log2Up(rt.fftSize), trim={"RoundDown": 0, "RoundHalfUp": 1, "Convergent": 2}[f.trimType],
        mag_mode=rt.magMode, bp_data=m.binPoint, bp_log=m.binPointLog, log2_lut_width=m.log2LookUpWidth,
        bp_in=c.protoIn.binaryPoint, bp_thr=c.protoThreshold.binaryPoint, w_thr=c.protoThreshold.width,
        bp_scaler=c.protoScaler.binaryPoint,
        scaler=int(rt.thresholdScaler * 2.0 ** c.protoThreshold.binaryPoint), linear=rt.logOrLinearMode,
        div_sum=0 if c.CFARAlgorithm == R.GOSCFARType else rt.divSum, peak_grouping=rt.peakGrouping,
        algorithm=1 if gos else 0, cfar_mode={v: k for k, v in MODE_NAMES.items()}[rt.CFARMode],
        ref_window=rt.refWindowSize, guard_window=rt.guardWindowSize,
        index_lagg=rt.indexLagg or 0, index_lead=rt.indexLead or 0, sub_window=rt.subWindowSize or 0,
        edge={"zero": 0, "wrap": 1}[c.edgeMode], keep_lsb_mask=stage_masks(params, rt)[0],
        expand_mask=stage_masks(params, rt)[1], no_bit_reverse=0 if f.useBitReverse else 1,
        send_cut=int(c.sendCut), window=WINDOWS[params.window])


def oracle_fcfg(params, rt):
    c = params.cfarParams
    gos = c.CFARAlgorithm == R.GOSCFARType or (c.CFARAlgorithm == R.GOSCACFARType and rt.CFARAlgorithm == "GOS")
    raw = int(rt.thresholdScaler * 2.0 ** c.protoThreshold.binaryPoint)
    return O.default_fcfg(
        log2n=R.log2Up(rt.fftSize), mag_mode=rt.magMode, scaler=raw / 2.0 ** c.protoScaler.binaryPoint,
        linear=rt.logOrLinearMode, div_sum=0 if c.CFARAlgorithm == R.GOSCFARType else rt.divSum,
        peak_grouping=rt.peakGrouping, algorithm=1 if gos else 0,
        cfar_mode={v: k for k, v in MODE_NAMES.items()}[rt.CFARMode], ref_window=rt.refWindowSize,
        guard_window=rt.guardWindowSize, index_lagg=rt.indexLagg or 0, index_lead=rt.indexLead or 0,
        edge={"zero": 0, "wrap": 1}[c.edgeMode], sub_window=rt.subWindowSize or 0,
        no_bit_reverse=0 if params.fftParams.useBitReverse else 1, window=WINDOWS[params.window])


def random_beats(n_frames, n, seed, amp=12000):
    rng = np.random.default_rng(seed)
    re = rng.integers(-amp, amp + 1, size=(n_frames, n))
    im = rng.integers(-amp, amp + 1, size=(n_frames, n))
    return O.pack_iq(re, im)


def tone_beats(n_frames, n, seed, bp=12):
    """The tester's stimulus (FftMagCfarChainTester.scala:53), one seed per frame."""
    frames = [R.stimulus.formAXI4StreamComplexData(
        R.stimulus.getComplexTones(n, 0.125, 0.25, 0.5, shiftRangeFactor=bp, seed=seed + f)) for f in range(n_frames)]
    return np.stack(frames)


def compare_f32(words, thr_ref, peak_ref, margin_ref, mag_ref=None, rtol=2e-5, min_decided=0.98, atol=0.0):
    """fp32 device result vs float64 oracle.  Tolerance: |thr - ref| <= tol = rtol * max(|ref|, frame
    peak magnitude * 2^-10): the reference's own HW-vs-float acceptance is 2 LSB of a 16-bit word
    (RspChainTesterUtils.scala:221,231) = 6e-5 of full scale; we ask for 3x tighter.
    Peak flags: a flag is an AND of comparisons between two quantities (cut vs threshold, cut vs a
    neighbour) that each carry at most `tol` of fp32 error, so it is compared on EVERY cell whose
    decision margin in the oracle (orc_cfar_f64: distance of the deciding comparison from a tie)
    exceeds 2 tol.  A cell is excluded only by its own margin; `min_decided` (>= 0.98 everywhere) is a
    guard that the comparison stays meaningful, not a quota that hides disagreements.
    atol: log2-magnitude mode only -- fp32 rounding on near-null bins is amplified by the log, so
    that mode is held to 1 LSB (2^-9) of the reference's Q7.9 log format (FftMagCfarChain.scala:94-95)
    instead of a relative bound."""
    assert min_decided >= 0.98
    thr, peak = R.unpack_output_f32(words)
    thr = thr.astype(np.float64).reshape(thr_ref.shape)
    peak = peak.reshape(peak_ref.shape)
    floor = (np.abs(thr_ref).max(axis=-1, keepdims=True) if mag_ref is None
             else np.abs(mag_ref).max(axis=-1, keepdims=True)) * 2.0 ** -10
    tol = rtol * np.maximum(np.abs(thr_ref), floor) + atol
    err = np.abs(thr - thr_ref)
    assert np.all(err <= tol), f"threshold error {np.max(err / tol):.2f} x tolerance"
    decided = margin_ref > 2 * tol
    assert decided.mean() >= min_decided, f"only {decided.mean():.4f} of the cells have a decidable flag"
    bad = decided & (peak != peak_ref)
    assert not bad.any(), f"{int(bad.sum())} peak flags differ on cells with margin > 2 tol (first: {np.argwhere(bad)[0]})"
    return float(np.max(err / tol))
