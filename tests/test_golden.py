"""Committed fixtures (tests/golden/, made by make_golden.py from the oracle with fixed seeds;
BUILD-GENERATED -- the reference holds none).  CPU: the oracle still reproduces them.  GPU: the
HIP path reproduces them through the C ABI without the oracle in the loop."""
import os

import numpy as np
import pytest

import rsp_chains_amd as R
from helpers import compare_f32, make_params

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = {"CA": "Cell Averaging", "GO": "Greatest Of", "SO": "Smallest Of"}


def test_oracle_reproduces_golden():
    from oracle import oracle as O
    from helpers import oracle_cfg, oracle_fcfg
    g = np.load(os.path.join(G, "fixed_n1024_tester.npz"))
    assert np.array_equal(O.chain_fixed(g["beats"], oracle_cfg(make_params(1024), R.RunTimeRspChainParams())).reshape(2, 1024), g["words"])
    g = np.load(os.path.join(G, "fixed_n256_modes.npz"))
    for key in g.files:
        if key == "beats":
            continue
        m, edge = key.split("_")
        rt = R.RunTimeRspChainParams(fftSize=256, CFARMode=MODES[m], refWindowSize=16, divSum=4, peakGrouping=1)
        assert np.array_equal(O.chain_fixed(g["beats"], oracle_cfg(make_params(256, edge=edge), rt)).reshape(3, 256), g[key])
    g = np.load(os.path.join(G, "f32_n4096_cfg2.npz"))
    rt = R.RunTimeRspChainParams(fftSize=4096, CFARMode="Cell Averaging")
    thr, peak, margin = O.chain_f32(g["x"], oracle_fcfg(make_params(4096, dtype=R.F32), rt))
    assert np.allclose(thr, g["thr"], rtol=1e-12) and np.array_equal(peak, g["peak"])


@pytest.mark.gpu
def test_gpu_reproduces_golden_fixed(gpu):
    g = np.load(os.path.join(G, "fixed_n1024_tester.npz"))
    with R.FftMagCfarChainVanilla(make_params(1024)) as dut:
        dut.configure(R.RunTimeRspChainParams())
        assert np.array_equal(dut.stream(g["beats"]), g["words"])
    g = np.load(os.path.join(G, "fixed_n256_modes.npz"))
    for key in g.files:
        if key == "beats":
            continue
        m, edge = key.split("_")
        with R.FftMagCfarChainVanilla(make_params(256, edge=edge)) as dut:
            dut.configure(R.RunTimeRspChainParams(fftSize=256, CFARMode=MODES[m], refWindowSize=16, divSum=4,
                                                  peakGrouping=1))
            assert np.array_equal(dut.stream(g["beats"]), g[key]), key


@pytest.mark.gpu
def test_gpu_reproduces_golden_f32(gpu):
    g = np.load(os.path.join(G, "f32_n4096_cfg2.npz"))
    with R.FftMagCfarChainVanilla(make_params(4096, dtype=R.F32)) as dut:
        dut.configure(R.RunTimeRspChainParams(fftSize=4096, CFARMode="Cell Averaging"))
        words = dut.stream(g["x"])
    # compare_f32's floor uses the frame's peak magnitude; the fixture stores it per frame
    mag_ref = np.broadcast_to(g["mag_max"][:, None], g["thr"].shape)
    compare_f32(words, g["thr"], g["peak"], g["margin"], mag_ref)
