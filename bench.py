#!/usr/bin/env python3
"""Headline benchmark: range-Doppler cells/s through FFT -> logMag -> CFAR on MI355X.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 1 channel, 4096-point range FFT + JPL magnitude +
CA-CFAR (R = 32, G = 4), 4096-chirp batch, fp32 -- 16 777 216 cells per step per GPU.
A step = one pass of the fused chain kernel over one batch already resident in HBM,
plus the compaction of its peak cells into a detection list.  Multi-GPU: chirps shard
embarrassingly (weak scaling: every rank owns a full batch); the only collective is
the RCCL all-gather of the detection lists: the lists of 4 consecutive steps share
one block, one all-gather per block on a side stream under the following steps.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (chain1d):
algorithmic bytes (12 B/cell: 8 in + 4 out) / its mean launch duration,
measured with HIP events on the kernel's stream, against HBM peak 8.0 TB/s.
`cpu_baseline` = the CPU oracle (oracle/rsp_oracle.c, float64 port; the reference's
Chisel simulation cannot run here) timed on this box's host cores, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
N_SETS = 4             # rotate 4 x (134 MB in + 67 MB out) = 805 MB > 2 x the 256 MiB Infinity Cache


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--fft", type=int, default=4096)
    ap.add_argument("--chirps", type=int, default=4096)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="cfg2 = the headline (BASELINE.json configs[1]); cfg3 / cfg4 are extra lines, same JSON shape")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-json", default=None, help="PMC-derived HBM bytes/launch (profiles/*.json)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import rsp_chains_amd as R

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RSP_BENCH_FORCE_DIST=1 runs the collective path with a 1-rank RCCL group (rehearsal on a 1-GPU box)
    use_dist = world > 1 or bool(os.environ.get("RSP_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" is RCCL on ROCm

    if args.workload != "cfg2":
        return other_workload(args, torch, dist, R, rank, local_rank, world, dev, use_dist)
    n, frames = args.fft, args.chirps
    cells = n * frames
    params = R.FftMagCfarVanillaParameters(
        fftParams=R.FFTParams.fixed(numPoints=n), magParams=R.MAGParams.fixed(),
        cfarParams=R.CFARParams(fftSize=n), dtype=R.F32, device=local_rank)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging", refWindowSize=32,
                                 guardWindowSize=4, divSum=5, thresholdScaler=3.5)
    dut = R.FftMagCfarChainVanilla(params)
    dut.configure(rt)
    # an explicit (non-default) stream: handle 0 would mean "the chain's own stream" to the C ABI, and
    # the events below must be recorded on the stream the kernels really run on
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    dut.set_stream(main_stream.cuda_stream)
    assert main_stream.cuda_stream != 0

    # ---- synthetic chirp frames, generated on the device (3 point targets of
    # amplitude 0.4/0.2/0.1 at random range bins + complex white noise, sigma 0.05) ----
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    t = torch.arange(n, device=dev, dtype=torch.float32)
    ins, outs = [], []
    cap = 1 << 15  # list capacity per rank and step (expected ~22 k peaks): 512 KiB
    for s in range(N_SETS):
        x = 0.05 * torch.randn(frames, n, 2, device=dev, generator=g)
        bins = torch.randint(0, n, (frames, 3), device=dev, generator=g).to(torch.float32)
        for j, a in enumerate((0.4, 0.2, 0.1)):
            ph = (2 * np.pi / n) * ((bins[:, j:j + 1] * t[None, :]) % n)
            x[..., 0] += a * torch.cos(ph)
            x[..., 1] += a * torch.sin(ph)
        ins.append(x.contiguous())
        outs.append(torch.empty(frames, n, dtype=torch.int32, device=dev))
    # detection lists: packed [cap + 1, 4] per step (row 0 = count, rows 1.. = list).  G consecutive steps
    # share one contiguous block so that ONE all-gather moves G lists (the collective's host-side cost,
    # tens of microseconds, is of the order of a whole step); two blocks alternate, the gather of one
    # runs on a side stream under the steps that fill the other.
    G = N_SETS
    rows = cap + 1
    blocks = [torch.zeros(G * rows, 4, dtype=torch.int32, device=dev) for _ in range(2)]
    lists = [[blocks[b][k * rows:(k + 1) * rows] for k in range(G)] for b in range(2)]
    if use_dist:
        from rsp_chains_amd.dist import gather_packed
        comm_stream = torch.cuda.Stream(device=dev)
        g_blocks = [torch.empty(world * G * rows, 4, dtype=torch.int32, device=dev) for _ in range(2)]
        ready = [torch.cuda.Event() for _ in range(2)]     # block b written
        gathered = [torch.cuda.Event() for _ in range(2)]  # block b gathered (reusable)

    def gather_block(b):
        ready[b].record(main_stream)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ready[b])
            gather_packed(blocks[b], out=g_blocks[b])
            gathered[b].record(comm_stream)

    last = {"b": 0, "k": 0, "open": False}

    def step(i):
        s = i % N_SETS
        b, k = divmod(i % (2 * G), G)
        if use_dist and k == 0 and i >= 2 * G:
            main_stream.wait_event(gathered[b])  # do not overwrite a block still being gathered
        lst = lists[b][k]
        dut.process_detect_device(ins[s].data_ptr(), frames, outs[s].data_ptr(), lst[1:].data_ptr(), cap,
                                  lst[0, :1].data_ptr())
        last.update(b=b, k=k, open=k != G - 1)
        if use_dist and k == G - 1:
            gather_block(b)

    def flush():  # a run that stops inside a block still gathers it
        if use_dist and last["open"]:
            gather_block(last["b"])
            last["open"] = False

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    flush()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    flush()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- dominant kernel alone: one HIP event pair per launch, recorded by the library on the
    # stream the kernel runs on, over a repeat of the timed region ----
    torch.cuda.synchronize()
    dut.profile_enable(True)
    for i in range(args.steps):
        step(args.warmup + args.steps + i)
    flush()
    fence()
    tot_ms, launches = dut.profile_read()
    dut.profile_enable(False)
    kernel_ms = tot_ms / max(launches, 1)

    n_det = int(lists[last["b"]][last["k"]][0, 0].item())

    if rank == 0:
        bytes_per_launch = 12.0 * cells
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tj = args.traffic_json or os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get("chain1d_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "range-Doppler cells/sec (FFT+CFAR)",
            "value": cells * world * args.steps / elapsed,
            "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg2: 1-ch {n}-pt range FFT + JPL logMag + CA-CFAR (R=32,G=4), "
                                   f"{frames}-chirp batch per GPU, fp32, dense words + detection list",
                       "cells_per_step_per_gpu": cells, "buffer_sets": N_SETS,
                       "detections_last_step": n_det,
                       "sharding": "chirps/channels per rank; one RCCL all-gather per 4 steps moves their 4 detection lists (side stream)"},
            "roofline": {"bound": "hbm", "kernel": "chain1d_kernel<12,f32>",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "frac_of_measured_copy_6290": achieved / 6290.0},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ins[0], n, frames)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def other_workload(args, torch, dist, R, rank, local_rank, world, dev, use_dist):
    """cfg3: 8 Rx x 4096 x 512 2-D range-Doppler + 2-D CA-CFAR (28 B/cell algorithmic);
    cfg4: OS-CFAR (R = 32, k = 24, G = 4) on 8192-point spectra, 2048 chirps (12 B/cell)."""
    if args.workload in ("cfg3", "cfg5"):
        nr, nd, units, bpc = (4096, 512, 8, 28.0) if args.workload == "cfg3" else (8192, 1024, 8, 28.0)
        params = R.FftMagCfarVanillaParameters(
            fftParams=R.FFTParams.fixed(numPoints=nr), magParams=R.MAGParams.fixed(),
            cfarParams=R.CFARParams(fftSize=nr, leadLaggWindowSize=16), dtype=R.F32, device=local_rank,
            dopplerPoints=nd, refDoppler=8, guardDoppler=2)
        rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2,
                                     divSum=4, thresholdScaler=4.0)
        shape = (units, nd, nr)
        name = (f"{args.workload}: 8 Rx per GPU, {nr}x{nd} range-Doppler 2-D FFT + JPL logMag + 2-D CA-CFAR "
                "(ref 8x8, guard 2x2), fp32, dense words + compact detection list"
                + ("; Rx channels sharded over ranks, RCCL all-gather of the lists" if args.workload == "cfg5" else ""))
        kernel = f"range_fft<{nr.bit_length() - 1}> + doppler_mag<{nd.bit_length() - 1}> + cfar2d"
    else:
        nr, units, bpc = 8192, 2048, 12.0
        params = R.FftMagCfarVanillaParameters(
            fftParams=R.FFTParams.fixed(numPoints=nr), magParams=R.MAGParams.fixed(),
            cfarParams=R.CFARParams(fftSize=nr, CFARAlgorithm=R.GOSCFARType), dtype=R.F32, device=local_rank)
        rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Greatest Of", refWindowSize=32, guardWindowSize=4, divSum=None,
                                     indexLagg=24, indexLead=24, thresholdScaler=2.5)
        shape, name = (units, nr), "cfg4: OS-CFAR (32-cell window, k = 24, G = 4) on 8192-pt spectra, 2048-chirp batch, fp32"
        kernel = "chain1d_gos_kernel<13,f32>"
    dut = R.FftMagCfarChainVanilla(params)
    dut.configure(rt)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    dut.set_stream(stream.cuda_stream)
    g = torch.Generator(device=dev)
    g.manual_seed(2345 + rank)
    ins = [(0.05 * torch.randn(*shape, 2, device=dev, generator=g)).contiguous() for _ in range(N_SETS)]
    for x in ins:   # a few strong cells so that the detector has something to find
        x.view(-1, 2)[:: 65537, 0] += 20.0
    cells = int(np.prod(shape))
    outs = [torch.empty(cells, dtype=torch.int32, device=dev) for _ in range(N_SETS)]
    cap = 1 << 15
    lists = [torch.zeros(cap + 1, 4, dtype=torch.int32, device=dev) for _ in range(N_SETS)]  # row 0 = count
    counts = [p[0, :1] for p in lists]
    gather = args.workload == "cfg5" and use_dist
    if gather:
        from rsp_chains_amd.dist import gather_packed
        comm = torch.cuda.Stream(device=dev)
        g_lists = [torch.empty(world * (cap + 1), 4, dtype=torch.int32, device=dev) for _ in range(N_SETS)]
        ready = [torch.cuda.Event() for _ in range(N_SETS)]
        gathered = [torch.cuda.Event() for _ in range(N_SETS)]

    def step(i):
        s = i % N_SETS
        if gather and i >= N_SETS:
            stream.wait_event(gathered[s])
        dut.process_device(ins[s].data_ptr(), units, outs[s].data_ptr())
        if args.workload == "cfg5":
            dut.detections_device(outs[s].data_ptr(), units, lists[s][1:].data_ptr(), cap, counts[s].data_ptr())
        if gather:
            ready[s].record(stream)
            with torch.cuda.stream(comm):
                comm.wait_event(ready[s])
                gather_packed(lists[s], out=g_lists[s])
                gathered[s].record(comm)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    torch.cuda.synchronize()
    dut.timer_start()
    for i in range(args.steps):
        step(i)
    # HIP events around the same steps (GPU time only); a short event-bracketed loop starts with an idle
    # queue, so it can read higher than the wall-clock mean of the saturated loop above: take the lower
    kernel_ms = min(dut.timer_stop() / args.steps, elapsed / args.steps * 1e3)
    if rank == 0:
        achieved = bpc * cells / (kernel_ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "range-Doppler cells/sec (FFT+CFAR)", "value": cells * world * args.steps / elapsed,
            "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "cells_per_step_per_gpu": cells, "buffer_sets": N_SETS},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": bpc * cells}}), flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(x_dev, n, frames):
    """The oracle as the timed CPU baseline (kind "port": the reference's Chisel/verilator
    simulation cannot be built here).  Sample: the first `sample` chirps of the same batch."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    try:  # the box's CPU share (cgroup quota), not the host's core count
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    sample = min(frames, 2048)
    x = x_dev[:sample].cpu().numpy().view(np.complex64).reshape(sample, n)
    cfg = O.default_fcfg(log2n=n.bit_length() - 1, cfar_mode=O.CFAR_CA, ref_window=32, guard_window=4,
                         div_sum=5, scaler=3.5)
    O.chain_f32(x[:64], cfg, n_threads=cores)  # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    reps = 0
    while True:
        O.chain_f32(x, cfg, n_threads=cores)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or reps >= 20:
            break
    return {"value": sample * n * reps / dt, "unit": "cells/s", "cores": cores, "kind": "port",
            "sample": f"{sample} of {frames} chirps x {n} points, {reps} passes, float64 oracle, OpenMP x{cores}"}


if __name__ == "__main__":
    main()
