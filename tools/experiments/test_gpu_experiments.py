"""Parity of the experimental 1-D chain kernels (side libraries: tools/build_experiment.sh; run through
tools/ab_experiment.sh with RSP_CHAIN_LIB pointing at the side library).  Not part of tests/: the product
library contains none of these kernels."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import rsp_chains_amd as R  # noqa: E402
from oracle import oracle as O  # noqa: E402
from helpers import compare_f32, make_params, oracle_fcfg  # noqa: E402


@pytest.mark.parametrize("mode,edge,grouping,ref", [("Cell Averaging", "zero", 0, 32), ("Greatest Of", "wrap", 1, 64),
                                                   ("Smallest Of", "zero", 0, 16)])
def test_f32_experimental_kernel_equals_plain(mode, edge, grouping, ref):
    """RSP_OPT_EXPERIMENT: the software-pipelined persistent kernels of this directory run the plain quad kernel's
    arithmetic on frames that arrive by LDS-DMA, so its words are BIT-identical -- over more frames than there are
    resident workgroups (every workgroup walks several frame groups), with dense words, the fused detection list,
    and the list-only call (no word stores: the uncounted wait)."""
    n, frames = 4096, 1500
    params = make_params(n, dtype=R.F32, edge=edge, leadLagg=64)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode=mode, refWindowSize=ref, divSum=R.log2Up(ref), peakGrouping=grouping)
    x = R.stimulus.chirp_frames(frames, n, seed=77)
    out = {}
    for pipe in (1, 0):
        with R.FftMagCfarChainVanilla(params) as dut:
            dut.configure(rt)
            dut.set_option(dut.EXPERIMENT, pipe)
            words = dut.stream(x)
            det, found = dut.detections(x)
            d_in = R.DeviceBuffer(x.nbytes); d_in.upload(x)
            cap = 1 << 16
            d_list, d_cnt = R.DeviceBuffer(16 * cap), R.DeviceBuffer(8)
            dut.process_detect_device(d_in.ptr, frames, 0, d_list.ptr, cap, d_cnt.ptr)   # list only
            dut.synchronize()
            cnt = d_cnt.download(np.uint32, 2)
            lst = d_list.download(np.uint32, int(cnt[1]) * 4).reshape(-1, 4)
        out[pipe] = (words, det, found, cnt, lst[np.lexsort((lst[:, 1], lst[:, 0]))])
    # the LDS-DMA kernel runs the plain kernel's arithmetic (bit-identical words); the register-prefetch kernel squares
    # its pass-0 twiddles (2-8 ulp): every frame against the float64 oracle instead, and its lists against its own words
    same = all(np.array_equal(a, b) for a, b in zip(out[1], out[0]))
    fr, bn = np.nonzero(out[1][0] & 1)
    assert out[1][2] == fr.size and np.array_equal(out[1][1]["frame"], fr) and np.array_equal(out[1][1]["bin"], bn)
    assert int(out[1][3][0]) == fr.size and out[1][4].shape[0] == int(out[1][3][1])
    thr, peak, margin, mag = O.chain_f32(x, oracle_fcfg(params, rt), want_mag=True, n_threads=8)
    worst = compare_f32(out[1][0], thr, peak, margin, mag, rtol=2e-5 if same else 6e-5)   # squared twiddles: ~2x the product's error
    print(f"experiment words {'bit-identical to' if same else 'differ from'} the product kernel's; worst threshold error {worst:.3f} x tolerance")
