"""Host-side mirror of the reference interface: parameter validation (= elaboration-time
requires), RunTimeRspChainParams' requires, stimulus helpers.  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

import rsp_chains_amd as R
from rsp_chains_amd import _native as N
from helpers import make_params


def validate(params):
    cp = params.to_c()
    rc = N.lib().rsp_chain_validate_params(C.byref(cp))
    return rc, N.lib().rsp_last_error().decode()


def test_reference_parameter_sets_validate():
    # the six concrete parameter sets of the reference differ only in binary points
    for bp, proto, ll in ((12, None, 64), (0, None, 64),
                          (0, (R.FixedPoint(16, 0), R.FixedPoint(16, 3), R.FixedPoint(16, 6)), 32)):
        assert validate(make_params(1024, bp=bp, proto=proto, leadLagg=ll))[0] == 0


@pytest.mark.parametrize("mutate,code", [
    (lambda p: setattr(p.fftParams, "numPoints", 1000), N.RSP_ERR_INVALID),
    (lambda p: (setattr(p.fftParams, "numPoints", 8), setattr(p.cfarParams, "fftSize", 8)), N.RSP_ERR_UNSUPPORTED),
    (lambda p: (setattr(p.fftParams, "numPoints", 32768), setattr(p.cfarParams, "fftSize", 32768)), N.RSP_ERR_UNSUPPORTED),
    (lambda p: (setattr(p.fftParams, "numPoints", 16384), setattr(p.cfarParams, "fftSize", 16384), setattr(p, "dopplerPoints", 256),
                setattr(p, "refDoppler", 4), setattr(p, "dtype", R.F32)), N.RSP_ERR_UNSUPPORTED),
    (lambda p: setattr(p.fftParams, "dataWidth", 18), N.RSP_ERR_UNSUPPORTED),
    (lambda p: setattr(p.fftParams, "expandLogic", [2] * 10), N.RSP_ERR_INVALID),
    (lambda p: (setattr(p.fftParams, "expandLogic", [1] * 10), setattr(p, "dtype", R.F32)), N.RSP_ERR_INVALID),
    (lambda p: (setattr(p.fftParams, "useBitReverse", False), setattr(p, "dopplerPoints", 256), setattr(p, "refDoppler", 4),
                setattr(p, "dtype", R.F32)), N.RSP_ERR_UNSUPPORTED),
    (lambda p: setattr(p, "window", "kaiser"), None),
    (lambda p: setattr(p.cfarParams, "fftSize", 512), N.RSP_ERR_INVALID),
    (lambda p: setattr(p.cfarParams, "leadLaggWindowSize", 48), N.RSP_ERR_INVALID),
    (lambda p: setattr(p.cfarParams, "leadLaggWindowSize", 512), N.RSP_ERR_UNSUPPORTED),
    (lambda p: setattr(p.cfarParams, "protoThreshold", R.FixedPoint(24, 12)), N.RSP_ERR_INVALID),
    (lambda p: setattr(p, "beatBytes", 8), N.RSP_ERR_UNSUPPORTED),
    (lambda p: setattr(p, "magAddress", R.AddressSet(0x30000100, 0xFF)), N.RSP_ERR_INVALID),
])
def test_invalid_parameters_are_rejected_with_a_reason(mutate, code):
    p = make_params(1024)
    mutate(p)
    if code is None:   # rejected by the host mirror before it reaches the C ABI
        with pytest.raises(ValueError, match="requirement failed"):
            validate(p)
        return
    rc, msg = validate(p)
    assert rc == code and len(msg) > 10


def test_every_option_the_reference_types_accept_validates():
    """FFTParams.fixed / CFARParams fields that round 1 rejected: sendCut, useBitReverse = false, per-stage
    expandLogic / keepMSBorLSB (FftMagCfarChain.scala:82,86-87,107), + the window extension."""
    for kw in (dict(sendCut=True), dict(useBitReverse=False), dict(expandLogic=[1, 0] * 5), dict(keepMSBorLSB=[False] * 10),
               dict(window="hann"), dict(sendCut=True, useBitReverse=False, expandLogic=[0, 1] * 5, window="blackman")):
        assert validate(make_params(1024, **kw))[0] == 0, kw


def test_runtime_params_requires():
    # RspChainVanillaTester.scala:50-61
    R.RunTimeRspChainParams()
    for kw in (dict(refWindowSize=24), dict(fftSize=1000), dict(guardWindowSize=0),
               dict(refWindowSize=4, guardWindowSize=4), dict(subWindowSize=32), dict(indexLead=32),
               dict(indexLagg=40)):
        with pytest.raises(ValueError, match="requirement failed"):
            R.RunTimeRspChainParams(**kw)
    d = R.RunTimeRspChainParams()
    assert (d.CFARMode, d.refWindowSize, d.guardWindowSize, d.thresholdScaler, d.divSum, d.magMode,
            d.logOrLinearMode, d.peakGrouping) == ("Greatest Of", 32, 4, 3.5, 5, 2, 1, 0)


def test_log2up_and_fft_params_fixed():
    assert [R.log2Up(x) for x in (1, 2, 3, 1024, 1025)] == [1, 1, 2, 10, 11]
    f = R.FFTParams.fixed(numPoints=4096)
    assert len(f.expandLogic) == 12 and all(f.keepMSBorLSB) and f.protoIQ == R.FixedPoint(16, 12)
    with pytest.raises(ValueError):
        R.FFTParams.fixed(numPoints=1024, expandLogic=[0] * 9)


def test_stimulus_restatements():
    # getComplexTones: tones at 1/8, 1/4, 1/2 with amplitudes 0.4/0.2/0.1, truncation toward zero
    z = R.stimulus.getComplexTones(1024, 0.125, 0.25, 0.5, shiftRangeFactor=12, seed=3)
    assert np.all(z.real == np.trunc(z.real)) and np.all(np.abs(z.real) < 2 ** 15)
    X = np.abs(np.fft.fft(z)) / 1024 / 4096
    assert abs(X[128] - 0.4) < 0.02 and abs(X[256] - 0.2) < 0.02 and abs(X[512] - 0.1) < 0.02
    assert X[0] > 0.8            # mean of sqrt(U1 + U2) ~ 0.97: the DC peak the tester's plot shows
    assert np.array_equal(z, R.stimulus.getComplexTones(1024, 0.125, 0.25, 0.5, shiftRangeFactor=12, seed=3))
    # the quirk SURVEY §4 warns about: integer-division frequencies are all zero -> a DC-only signal
    z0 = R.stimulus.getComplexTones(64, 1 // 8, 1 // 4, 1 // 2, shiftRangeFactor=12, seed=1)
    assert np.all(z0.imag == 0)
    nco = R.stimulus.calcExpectedNcoOut(1024, 32)
    assert nco[0] == np.trunc(np.cos(2 * np.pi * 32 / 1024) * 2 ** 14) + 1j * np.trunc(np.sin(2 * np.pi * 32 / 1024) * 2 ** 14)
    with pytest.raises(ValueError):
        R.stimulus.calcExpectedNcoOut(16, 16)
    assert R.stimulus.formAXI4StreamRealData([1, -1])[1] == 0xFFFF0000


def test_shard_range_partitions_exactly():
    from rsp_chains_amd.dist import shard_range
    for n in (0, 1, 7, 64, 4096):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_tester_dump_formats_round_trip(tmp_path):
    """inputDataReal/Imag.txt, outputData.txt, thresholdData.txt in the tester's %04x format
    (FftMagCfarChainTester.scala:56-68,155-175)."""
    z = R.stimulus.getComplexTones(64, 0.125, 0.25, 0.5, shiftRangeFactor=12, seed=9) - 3000
    R.dumps.write_input_dumps(str(tmp_path), z)
    lines = open(tmp_path / "inputDataReal.txt").read().split()
    assert len(lines) == 64 and all(len(s) in (4, 8) for s in lines)
    assert any(len(s) == 8 and s.startswith("ffff") for s in lines)      # negative Int -> 8 digits
    assert np.array_equal(R.dumps.read_input_dumps(str(tmp_path)), z)
    words = np.array([(77 << 7) | (5 << 1) | 1, ((-3) << 7) & 0xFFFFFFFF], np.uint32)
    R.dumps.write_output_dumps(str(tmp_path), words, 64)
    assert open(tmp_path / "thresholdData.txt").read().split() == ["004d", "fffffffd"]
    assert np.array_equal(R.dumps.read_output_words(str(tmp_path)), words)
