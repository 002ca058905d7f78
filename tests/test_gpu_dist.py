"""The detection gather on the `nccl` backend (= RCCL on ROCm), device tensors end to end: a FRESH child process
(started before anything in it touches the GPU) initialises a 1-rank RCCL group, runs the chain into a packed
{header, list} tensor through the C ABI and all-gathers it with the code bench.py --gpus N uses.  One rank is what a
one-GPU box offers: it exercises communicator creation, the collective on device memory, its ordering against the
chain's stream and the sized payload -- not the cross-GPU transport (no N > 1 hardware run exists yet: DESIGN.md 4)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    import torch, torch.distributed as dist
    import rsp_chains_amd as R
    from rsp_chains_amd.dist import PackedGatherer, gather_packed, gathered_complete, merge_gathered, unpack_gathered
    from helpers import make_params
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)

    def check(dut, x, n_units, cells_per_unit, cap):
        d_in = torch.from_numpy(x).to(dev)
        words = torch.empty(n_units * cells_per_unit, dtype=torch.int32, device=dev)
        packed = torch.zeros(cap + 1, 4, dtype=torch.int32, device=dev)     # row 0 = (found, stored), rows 1.. = list
        dut.set_stream(stream.cuda_stream)
        dut.process_detect_device(d_in.data_ptr(), n_units, words.data_ptr(), packed[1:].data_ptr(), cap, packed[0].data_ptr())
        pg = PackedGatherer(cap, min_rows=4)
        view = pg.gather(packed)                    # same stream: ordered behind the chain's kernels
        if not pg.settle(view):                     # the first guess was too small: the gatherer has grown
            assert not gathered_complete(view)
            view = pg.gather(packed)
        assert pg.settle(view) and view.shape[1] < cap + 1, (view.shape, cap)
        lists, stored, found = unpack_gathered(view)
        merged = merge_gathered(lists, stored, [0])
        w = words.cpu().numpy().view(np.uint32)
        peaks = np.flatnonzero(w & 1)
        assert int(found[0]) == int(stored[0]) == peaks.size == merged.shape[0], (int(found[0]), int(stored[0]), peaks.size)
        return merged, w, peaks

    # 1-D chain, FIXED16, the tester's stimulus
    n, frames = 1024, 96
    params = make_params(n)
    beats = np.stack([R.stimulus.formAXI4StreamComplexData(R.stimulus.getComplexTones(n, 0.125, 0.25, 0.5, shiftRangeFactor=12, seed=s)) for s in range(frames)])
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(R.RunTimeRspChainParams())
        merged, w, peaks = check(dut, beats, frames, n, 4096)
        got = sorted((int(a) * n + int(b), int(wd) & 0xffffffff) for a, b, _, wd in merged)
        assert got == [(int(p), int(w[p])) for p in peaks]
        # the lists of K steps in one collective (bench.py --gpus N, cfg 2): the chain writes step j's list into slot j
        K, cap = 3, 4096
        batch = torch.zeros(K, cap + 1, 4, dtype=torch.int32, device=dev)
        d_in = torch.from_numpy(beats).to(dev)
        words = torch.empty(frames * n, dtype=torch.int32, device=dev)
        for j in range(K):   # step j: the first frames - 8 j frames of the batch
            dut.process_detect_device(d_in.data_ptr(), frames - 8 * j, words.data_ptr(), batch[j][1:].data_ptr(), cap, batch[j][0].data_ptr())
        pg = PackedGatherer(cap, min_rows=4)
        vb = pg.gather_batch(batch)
        if not pg.settle(vb):
            vb = pg.gather_batch(batch)
        assert pg.settle(vb) and vb.shape[:2] == (1, K) and vb.shape[2] < cap + 1
        for j in range(K):
            lj, sj, fj = unpack_gathered(vb[:, j])
            mj = merge_gathered(lj, sj, [0])
            want = [(int(p), int(w[p])) for p in peaks if p < (frames - 8 * j) * n]
            assert sorted((int(a) * n + int(b), int(wd) & 0xffffffff) for a, b, _, wd in mj) == want, j
    # 2-D chain (the cfg-5 code path of bench.py --gpus N), fp32, list appended by the CFAR kernel
    nr, nd, n_ch = 1024, 256, 2
    p2 = R.FftMagCfarVanillaParameters(fftParams=R.FFTParams.fixed(numPoints=nr), magParams=R.MAGParams.fixed(),
                                       cfarParams=R.CFARParams(fftSize=nr, leadLaggWindowSize=16), dtype=R.F32,
                                       dopplerPoints=nd, refDoppler=8, guardDoppler=2)
    rng = np.random.default_rng(5)
    x = (0.05 * (rng.standard_normal((n_ch, nd, nr)) + 1j * rng.standard_normal((n_ch, nd, nr)))).astype(np.complex64)
    x[0, :, :] += (0.4 * np.exp(2j * np.pi * (37 * np.arange(nr)[None, :] / nr + 11 * np.arange(nd)[:, None] / nd))).astype(np.complex64)
    with R.FftMagCfarChainVanilla(p2) as dut:
        dut.configure(R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2, divSum=4, thresholdScaler=4.0))
        merged, w, peaks = check(dut, x, n_ch, nd * nr, 8192)
        got = sorted((int(c) * nd * nr + int(d) * nr + int(b)) for c, b, d, _ in merged)
        assert got == peaks.tolist() and (0 * nd + 11) * nr + 37 in got
    dist.barrier(); dist.destroy_process_group()
    print("nccl 1-rank gather ok")
""")


def test_one_rank_rccl_gather_of_device_lists(gpu, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "nccl 1-rank gather ok" in out.stdout
