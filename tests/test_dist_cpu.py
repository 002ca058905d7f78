"""N > 1 path on CPU: two gloo ranks shard the frames, each produces its detection list (here
from the oracle standing in for the GPU kernel), and the lists are all-gathered by the same
code the GPU bench uses over RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import rsp_chains_amd as R
    from rsp_chains_amd.dist import (shard_range, gather_detections, gather_packed, unpack_gathered, merge_gathered,
                                     gathered_complete, PackedGatherer)
    from oracle import oracle as O
    from helpers import make_params, oracle_cfg, tone_beats
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, frames, cap = 256, 11, 64
    params = make_params(n); rt = R.RunTimeRspChainParams(fftSize=n, refWindowSize=16, divSum=4)
    beats = tone_beats(frames, n, 4242)
    lo, hi = shard_range(frames, rank, world)
    words = O.chain_fixed(beats[lo:hi], oracle_cfg(params, rt)).reshape(hi - lo, n)
    fr, bn = np.nonzero(words & 1)
    lst = np.zeros((cap, 4), np.int32)
    k = len(fr)
    lst[:k, 0], lst[:k, 1], lst[:k, 3] = fr, bn, words[fr, bn].astype(np.int64).astype(np.int32)
    # (found, stored) as the C ABI writes them through d_count; rank 1's list is truncated by one entry
    st = k if rank == 0 else max(k - 1, 0)
    lists, counts, founds = gather_detections(torch.from_numpy(lst), torch.tensor([k, st], dtype=torch.int32), cap)
    firsts = [shard_range(frames, r, world)[0] for r in range(world)]
    mt = merge_gathered(lists, counts, firsts)
    assert mt.shape[0] == int(counts.sum()) and int(founds.sum()) >= mt.shape[0]
    try:
        gather_detections(torch.from_numpy(lst), torch.tensor([k], dtype=torch.int32), cap)
        raise SystemExit("a one-word count tensor must be refused")
    except ValueError:
        pass
    lists, counts, founds = gather_detections(torch.from_numpy(lst), torch.tensor([k, k], dtype=torch.int32), cap)
    merged = merge_gathered(lists, counts, firsts)
    full = O.chain_fixed(beats, oracle_cfg(params, rt)).reshape(frames, n)
    gfr, gbn = np.nonzero(full & 1)
    want = sorted(zip(gfr.tolist(), gbn.tolist()))
    got = sorted(zip(merged[:, 0].tolist(), merged[:, 1].tolist()))
    assert got == want, (rank, got[:5], want[:5])
    assert int(counts.sum()) == len(want)
    # the one-collective form bench.py uses: row 0 = header (found, stored), rows 1.. = list.  Rank 1
    # pretends its list was truncated (found > stored): only the stored rows may be merged.
    packed = torch.zeros((cap + 1, 4), dtype=torch.int32)
    stored = k if rank == 0 else max(k - 1, 0)
    packed[0, 0], packed[0, 1] = k, stored
    packed[1:] = torch.from_numpy(lst)
    packed[1 + stored:] = -7            # garbage beyond the stored rows must never be read
    l2, s2, f2 = unpack_gathered(gather_packed(packed))
    m2 = merge_gathered(l2, s2, firsts)
    assert int(f2.sum()) == len(want) and m2.shape[0] == int(s2.sum())
    assert (m2[:, 1] >= 0).all() and set(zip(m2[:, 0].tolist(), m2[:, 1].tolist())) <= set(want)
    # payload sized to the lists: too few rows is detected, the gatherer grows and the repeat is complete
    packed[0, 1] = k
    packed[1:] = torch.from_numpy(lst)
    small = gather_packed(packed, rows=2)
    assert small.shape == (world, 2, 4) and not gathered_complete(small)
    pg = PackedGatherer(cap, min_rows=2)
    v = pg.gather(packed)
    if not pg.settle(v):
        v = pg.gather(packed)
    assert pg.settle(v) and v.shape[1] <= cap + 1 and v.shape[1] >= int(s2.max()) + 1
    l3, s3, f3 = unpack_gathered(v)
    m3 = merge_gathered(l3, s3, firsts)
    assert sorted(zip(m3[:, 0].tolist(), m3[:, 1].tolist())) == want
    # the lists of K steps in one collective: slice j of the batch is the list rotated by j entries
    K = 3
    batch = torch.full((K, cap + 1, 4), -9, dtype=torch.int32)
    for j in range(K):
        batch[j, 0, 0], batch[j, 0, 1] = k, k
        batch[j, 1:1 + k] = torch.from_numpy(np.roll(lst[:k], j, axis=0))
    vb = pg.gather_batch(batch)
    assert vb.shape == (world, K, pg.rows, 4) and pg.settle(vb)
    for j in range(K):
        lj, sj, fj = unpack_gathered(vb[:, j])
        mj = merge_gathered(lj, sj, firsts)
        assert sorted(zip(mj[:, 0].tolist(), mj[:, 1].tolist())) == want
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok", len(want))
""")


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("ok") == 2  # both ranks (their prints may interleave)
