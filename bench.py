#!/usr/bin/env python3
"""Headline benchmark: range-Doppler cells/s through FFT -> logMag -> CFAR on MI355X.

  python bench.py --gpus 1 --steps 60 --warmup 8
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N = 1 (the headline, BASELINE.json configs[1] = "cfg2"): 1 channel, 4096-point range FFT + JPL
magnitude + CA-CFAR (R = 32, G = 4), 4096-chirp batch, fp32 -- 16 777 216 cells per step.  A step =
one pass of the fused chain kernel over one batch already resident in HBM plus the compaction of
its peak cells into a detection list.  The same line carries, under "extra", driver-visible numbers
for the other single-GPU configurations (cfg3: 8 Rx x 4096 x 512 2-D chain; cfg4: OS-CFAR on
8192-point spectra; cfg5_share: one GPU's 8-Rx share of the 64-Rx 8192 x 1024 configuration) and
two CPU baselines timed on this box's host cores: the float64 oracle on all cores ("cpu_baseline")
and the bit-accurate fixed-point model of the reference's own configuration on ONE thread
("cpu_baseline_fixed", SURVEY 8d-i: the closest analogue of the Chisel simulation, which cannot run here).

N > 1 (BASELINE.json configs[4] = "cfg5"): 64 Rx channels of 8192 x 1024 range-Doppler maps sharded
over the N ranks (contiguous channels per rank, rsp_chains_amd.dist.shard_range), 2-D FFT + JPL
magnitude + 2-D CA-CFAR per channel, per-rank compaction, and ONE RCCL all-gather of the packed
detection lists per step on a side stream.  Total work is fixed (64 channels): "scaling": "strong".
No data-path collective; the gather's cost is reported separately.

Timing: W warm-up steps, then 5 blocks of K steps, each block bracketed by barrier +
synchronize on both sides and reduced with MAX over ranks; the line reports the MEDIAN block
(`blocks_ms` lists all five).  `roofline` is for the dominant kernel: algorithmic bytes
/ its mean launch duration measured with one HIP event pair per launch on the stream the kernel
runs on (rsp_chain_profile_*) over a repeat of the same blocks -- the event records perturb the
stream by ~1 us per step, so they are kept out of the blocks `value` comes from.
Prints ONE JSON line (rank 0).
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
N_BLOCKS = 5
METRIC = "range-Doppler cells/sec (FFT+CFAR)"


def median(v):
    s = sorted(v)
    return s[len(s) // 2]


class Fence:
    def __init__(self, torch, dist, use_dist, dev):
        self.torch, self.dist, self.use_dist, self.dev = torch, dist, use_dist, dev

    def __call__(self):
        self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.use_dist:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def timed_blocks(step, steps, warmup, fence, first=0):
    """warm-up, then N_BLOCKS blocks of `steps` steps; returns per-block seconds (max over ranks)."""
    i = first
    for _ in range(warmup):
        step(i)
        i += 1
    out = []
    for _ in range(N_BLOCKS):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(i)
            i += 1
        fence()
        out.append(fence.max_over_ranks(time.perf_counter() - t0))
    return out, i


def kernel_ms(dut, step, steps, first, fence):
    """mean duration of the chain kernel(s) per step: HIP event pair per launch, same step sequence."""
    fence()
    dut.profile_enable(True)
    for i in range(steps):
        step(first + i)
    fence()
    tot_ms, launches = dut.profile_read()
    dut.profile_enable(False)
    return tot_ms / max(launches, 1), launches


def roofline(kernel, kms, algorithmic_bytes, traffic=None, traffic_source=None):
    achieved = algorithmic_bytes / (kms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": kms,
         "algorithmic_bytes_per_launch": algorithmic_bytes, "frac_of_measured_copy_6290": achieved / 6290.0}
    if traffic_source:
        r["traffic_source"] = traffic_source
    return r


def static_traffic(key):
    """HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs, gfx950 FETCH_SIZE x2
    correction: profiles/README.md).  Static: it describes the kernel as profiled for this round, not this run."""
    for name in ("traffic_r02.json", "traffic_r01.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                v = json.load(open(path)).get(key)
                if v:
                    return v, f"profiles/{name} (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel)"
            except Exception:
                pass
    return None, None


# ---------------------------------------------------------------------------------------- workloads

def synth_frames(torch, dev, g, frames, n):
    """3 point targets of amplitude 0.4/0.2/0.1 at random range bins + complex white noise, sigma 0.05"""
    t = torch.arange(n, device=dev, dtype=torch.float32)
    x = 0.05 * torch.randn(frames, n, 2, device=dev, generator=g)
    bins = torch.randint(0, n, (frames, 3), device=dev, generator=g).to(torch.float32)
    for j, a in enumerate((0.4, 0.2, 0.1)):
        ph = (2 * np.pi / n) * ((bins[:, j:j + 1] * t[None, :]) % n)
        x[..., 0] += a * torch.cos(ph)
        x[..., 1] += a * torch.sin(ph)
    return x.contiguous()


def synth_maps(torch, dev, g, n_ch, nd, nr):
    """per channel 3 point targets = complex exponentials in fast time (range bin) and slow time
    (Doppler bin), amplitudes 0.4/0.2/0.1, + complex white noise, sigma 0.05 per component"""
    x = 0.05 * torch.randn(n_ch, nd, nr, 2, device=dev, generator=g)
    xc = torch.view_as_complex(x)
    tr = torch.arange(nr, device=dev, dtype=torch.float32)
    td = torch.arange(nd, device=dev, dtype=torch.float32)
    rb = torch.randint(0, nr, (n_ch, 3), device=dev, generator=g).to(torch.float32)
    db = torch.randint(0, nd, (n_ch, 3), device=dev, generator=g).to(torch.float32)
    for ch in range(n_ch):
        for j, a in enumerate((0.4, 0.2, 0.1)):
            pr = torch.polar(torch.ones_like(tr), (2 * np.pi / nr) * ((rb[ch, j] * tr) % nr))
            pd = torch.polar(torch.full_like(td, a), (2 * np.pi / nd) * ((db[ch, j] * td) % nd))
            xc[ch] += pd[:, None] * pr[None, :]
    return x.contiguous()


def make_cfg2(R, torch, dev, local_rank, rank, n=4096, frames=4096):
    params = R.FftMagCfarVanillaParameters(
        fftParams=R.FFTParams.fixed(numPoints=n), magParams=R.MAGParams.fixed(),
        cfarParams=R.CFARParams(fftSize=n), dtype=R.F32, device=local_rank)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging", refWindowSize=32,
                                 guardWindowSize=4, divSum=5, thresholdScaler=3.5)
    dut = R.FftMagCfarChainVanilla(params)
    dut.configure(rt)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    ins = [synth_frames(torch, dev, g, frames, n) for _ in range(N_SETS)]
    outs = [torch.empty(frames, n, dtype=torch.int32, device=dev) for _ in range(N_SETS)]
    cap = 1 << 15  # list capacity per step (expected ~13 k peaks): 512 KiB
    lists = [torch.zeros(cap + 1, 4, dtype=torch.int32, device=dev) for _ in range(N_SETS)]  # row 0 = {found, stored}

    def step(i):
        s = i % N_SETS
        dut.process_detect_device(ins[s].data_ptr(), frames, outs[s].data_ptr(), lists[s][1:].data_ptr(), cap,
                                  lists[s][0].data_ptr())

    name = (f"cfg2: 1-ch {n}-pt range FFT + JPL logMag + CA-CFAR (R=32,G=4), {frames}-chirp batch, fp32, "
            "dense words + detection list")
    return dict(dut=dut, step=step, cells=n * frames, bytes=12.0 * n * frames, kernel=f"chain1d_quad_kernel<{n.bit_length() - 1},f32>",
                name=name, ins=ins, lists=lists, n=n, frames=frames)


def make_rd(R, torch, dev, local_rank, rank, nr, nd, n_ch, tag, sets=N_SETS, with_list=True):
    params = R.FftMagCfarVanillaParameters(
        fftParams=R.FFTParams.fixed(numPoints=nr), magParams=R.MAGParams.fixed(),
        cfarParams=R.CFARParams(fftSize=nr, leadLaggWindowSize=16), dtype=R.F32, device=local_rank,
        dopplerPoints=nd, refDoppler=8, guardDoppler=2)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Cell Averaging", refWindowSize=8, guardWindowSize=2,
                                 divSum=4, thresholdScaler=4.0)
    dut = R.FftMagCfarChainVanilla(params)
    dut.configure(rt)
    g = torch.Generator(device=dev)
    g.manual_seed(2345 + rank)
    shape = (n_ch, nd, nr)
    ins = [synth_maps(torch, dev, g, n_ch, nd, nr) for _ in range(sets)]
    cells = int(np.prod(shape))
    outs = [torch.empty(cells, dtype=torch.int32, device=dev) for _ in range(sets)]
    cap = 1 << 16
    lists = [torch.zeros(cap + 1, 4, dtype=torch.int32, device=dev) for _ in range(sets)]

    def step(i):
        s = i % sets
        if with_list:   # the CFAR kernel appends its peaks itself: no second pass over the dense words
            dut.process_detect_device(ins[s].data_ptr(), n_ch, outs[s].data_ptr(), lists[s][1:].data_ptr(), cap,
                                      lists[s][0].data_ptr())
        else:
            dut.process_device(ins[s].data_ptr(), n_ch, outs[s].data_ptr())

    name = (f"{tag}: {n_ch} Rx x {nr} x {nd} range-Doppler 2-D FFT + JPL logMag + 2-D CA-CFAR (ref 8x8, guard 2x2), fp32, "
            "dense words" + (" + detection list appended by the CFAR kernel" if with_list else ""))
    kernel = f"range_fft<{nr.bit_length() - 1}> + doppler_mag<{nd.bit_length() - 1}> + cfar2d_walk"
    return dict(dut=dut, step=step, cells=cells, bytes=28.0 * cells, kernel=kernel, name=name, lists=lists, sets=sets, cap=cap)


def make_cfg4(R, torch, dev, local_rank, rank, nr=8192, frames=2048):
    params = R.FftMagCfarVanillaParameters(
        fftParams=R.FFTParams.fixed(numPoints=nr), magParams=R.MAGParams.fixed(),
        cfarParams=R.CFARParams(fftSize=nr, CFARAlgorithm=R.GOSCFARType), dtype=R.F32, device=local_rank)
    rt = R.RunTimeRspChainParams(fftSize=nr, CFARMode="Greatest Of", refWindowSize=32, guardWindowSize=4, divSum=None,
                                 indexLagg=24, indexLead=24, thresholdScaler=2.5)
    dut = R.FftMagCfarChainVanilla(params)
    dut.configure(rt)
    g = torch.Generator(device=dev)
    g.manual_seed(3456 + rank)
    ins = [synth_frames(torch, dev, g, frames, nr) for _ in range(N_SETS)]
    outs = [torch.empty(frames, nr, dtype=torch.int32, device=dev) for _ in range(N_SETS)]

    def step(i):
        s = i % N_SETS
        dut.process_device(ins[s].data_ptr(), frames, outs[s].data_ptr())

    return dict(dut=dut, step=step, cells=nr * frames, bytes=12.0 * nr * frames, kernel="chain1d_gos_kernel<13,f32>",
                name=f"cfg4: OS-CFAR (32-cell window, k = 24, G = 4) on {nr}-pt spectra, {frames}-chirp batch, fp32, dense words")


def run_extra(w, torch, steps, warmup):
    """one extra configuration on this GPU: median block + kernel time, freed afterwards"""
    stream = torch.cuda.current_stream()
    w["dut"].set_stream(stream.cuda_stream)
    fence = Fence(torch, None, False, None)
    blocks, nxt = timed_blocks(w["step"], steps, warmup, fence)
    kms, _ = kernel_ms(w["dut"], w["step"], steps, nxt, fence)
    sec = median(blocks)
    det = None
    if w.get("lists") is not None:
        found, stored = (int(v) for v in w["lists"][0][0, :2].tolist())
        det = {"found": found, "stored": stored}
    return {"workload": w["name"], "cells_per_step": w["cells"], "steps": steps, "ms_per_step": sec / steps * 1e3,
            "detections_per_step": det,
            "value": w["cells"] * steps / sec, "unit": "cells/s", "blocks_ms": [b / steps * 1e3 for b in blocks],
            "roofline": roofline(w["kernel"], kms, w["bytes"])}


# ---------------------------------------------------------------------------------------- main

def main():
    # stdout carries exactly ONE line, the JSON: everything else that may write to file descriptor 1 (RCCL prints
    # its version banner there when a communicator is created) is sent to stderr for the whole run
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--workload", default=None, choices=[None, "cfg2", "cfg5"],
                    help="override: cfg2 weak-scaled on N ranks / cfg5 on one rank (rehearsals); default = cfg2 at N = 1, cfg5 at N > 1")
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
    # an explicit (non-default) stream: handle 0 would mean "the chain's own stream" to the C ABI, and
    # the events must be recorded on the stream the kernels really run on
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    assert main_stream.cuda_stream != 0
    fence = Fence(torch, dist, use_dist, dev)
    workload = args.workload or ("cfg2" if world == 1 and not use_dist else "cfg5")
    if workload == "cfg5":
        line = run_cfg5(args, torch, dist, R, rank, local_rank, world, dev, use_dist, main_stream, fence)
    else:
        line = run_cfg2(args, torch, dist, R, rank, local_rank, world, dev, use_dist, main_stream, fence)
    if rank == 0:
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if use_dist:
        dist.destroy_process_group()


def run_cfg2(args, torch, dist, R, rank, local_rank, world, dev, use_dist, main_stream, fence):
    w = make_cfg2(R, torch, dev, local_rank, rank)
    dut, step = w["dut"], w["step"]
    dut.set_stream(main_stream.cuda_stream)
    blocks, nxt = timed_blocks(step, args.steps, args.warmup, fence)
    kms, launches = kernel_ms(dut, step, args.steps, nxt, fence)
    sec = median(blocks)
    found, stored = (int(v) for v in w["lists"][(nxt - 1) % N_SETS][0, :2].tolist())
    line = None
    if rank == 0:
        traffic, src = static_traffic("chain1d_hbm_bytes_per_launch")
        line = {
            "metric": METRIC, "value": w["cells"] * world * args.steps / sec, "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "blocks": N_BLOCKS, "blocks_ms": [b / args.steps * 1e3 for b in blocks], "timing": "median of 5 blocks of `steps` steps",
            "config": {"workload": w["name"], "cells_per_step_per_gpu": w["cells"], "buffer_sets": N_SETS,
                       "detections_last_step": {"found": found, "stored": stored}},
            "roofline": roofline(w["kernel"], kms, w["bytes"], traffic, src),
        }
    if world == 1 and not use_dist and not args.no_extra:
        x0 = w["ins"][0]
        host_sample = x0[:2048].cpu().numpy().view(np.complex64).reshape(2048, w["n"])
        del w, dut, step
        torch.cuda.empty_cache()
        extra = {}
        steps_x = max(8, args.steps // 3)
        for tag, make in (("cfg3", lambda: make_rd(R, torch, dev, local_rank, rank, 4096, 512, 8, "cfg3")),
                          ("cfg4", lambda: make_cfg4(R, torch, dev, local_rank, rank)),
                          ("cfg5_share", lambda: make_rd(R, torch, dev, local_rank, rank, 8192, 1024, 8,
                                                        "cfg5 share (8 of 64 Rx)", sets=2))):
            wx = make()
            extra[tag] = run_extra(wx, torch, steps_x if tag != "cfg5_share" else max(4, steps_x // 2), 3)
            del wx
            torch.cuda.empty_cache()
        line["extra"] = extra
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host_sample, 4096, 4096)
            line["cpu_baseline_fixed"] = cpu_baseline_fixed()
    return line


def run_cfg5(args, torch, dist, R, rank, local_rank, world, dev, use_dist, main_stream, fence):
    """BASELINE.json configs[4]: 64 Rx x 8192 x 1024, channels sharded over the ranks, RCCL gather of the lists."""
    from rsp_chains_amd.dist import gather_packed, shard_range
    total_ch, nr, nd = 64, 8192, 1024
    lo, hi = shard_range(total_ch, rank, world)
    n_ch = hi - lo
    w = make_rd(R, torch, dev, local_rank, rank, nr, nd, n_ch, f"cfg5: {total_ch} Rx sharded {n_ch}/GPU", sets=2)
    dut, sets, cap = w["dut"], w["sets"], w["cap"]
    dut.set_stream(main_stream.cuda_stream)
    lists = w["lists"]
    gather = use_dist
    if gather:
        comm = torch.cuda.Stream(device=dev)
        g_lists = [torch.empty(world * (cap + 1), 4, dtype=torch.int32, device=dev) for _ in range(sets)]
        ready = [torch.cuda.Event() for _ in range(sets)]
        gathered = [torch.cuda.Event() for _ in range(sets)]
    state = {"gather": gather}

    def step(i):
        s = i % sets
        if state["gather"] and i >= sets:
            main_stream.wait_event(gathered[s])   # do not overwrite a list still being gathered
        w["step"](i)
        if state["gather"]:
            ready[s].record(main_stream)
            with torch.cuda.stream(comm):
                comm.wait_event(ready[s])
                gather_packed(lists[s], out=g_lists[s])
                gathered[s].record(comm)

    blocks, nxt = timed_blocks(step, args.steps, args.warmup, fence)
    sec = median(blocks)
    # the same steps without the collective: what the gather costs end to end
    state["gather"] = False
    blocks_ng, nxt = timed_blocks(step, args.steps, 1, fence, first=nxt)
    sec_ng = median(blocks_ng)
    kms, _ = kernel_ms(dut, w["step"], args.steps, nxt, fence)
    gather_ms = None
    if gather:  # the collective alone, back to back on its stream
        fence()
        t0 = time.perf_counter()
        for k in range(20):
            gather_packed(lists[k % sets], out=g_lists[k % sets])
        fence()
        gather_ms = fence.max_over_ranks(time.perf_counter() - t0) / 20 * 1e3
    if rank != 0:
        return None
    cells_total = total_ch * nd * nr
    return {
        "metric": METRIC, "value": cells_total * args.steps / sec, "unit": "cells/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "blocks": N_BLOCKS, "blocks_ms": [b / args.steps * 1e3 for b in blocks], "timing": "median of 5 blocks of `steps` steps",
        "config": {"workload": w["name"] + "; RCCL all-gather of the packed lists, one per step, side stream",
                   "cells_per_step_total": cells_total, "channels_per_gpu": n_ch, "buffer_sets": sets,
                   "sharding": "contiguous channels per rank (dist.shard_range); no data-path collective"},
        "gather": {"ms_per_collective_alone": gather_ms, "ms_per_step_without_gather": sec_ng / args.steps * 1e3,
                   "bytes_per_rank": (cap + 1) * 16},
        "scaling_note": ("N > 1 lines run BASELINE.json configs[4] (64 Rx, fixed total work); the N = 1 line's headline is configs[1] "
                         "(a different, lighter workload per cell), so scaling efficiency is value(N) / (N/8 x 8 x the N = 1 line's "
                         "extra.cfg5_share.value), i.e. against the same 8-Rx-per-GPU share measured on one GPU"),
        "per_gpu_value": cells_total * args.steps / sec / world,
        "roofline": roofline(w["kernel"], kms, 28.0 * n_ch * nd * nr),
    }


# ---------------------------------------------------------------------------------------- CPU baselines

def host_cores():
    cores = os.cpu_count() or 1
    try:  # the box's CPU share (cgroup quota), not the host's core count
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except Exception:
        pass
    return cores


def cpu_baseline(x, n, frames):
    """The oracle as the timed CPU baseline (kind "port": the reference's Chisel/verilator
    simulation cannot be built here).  Sample: the first chirps of the same batch."""
    from oracle import oracle as O
    cores = host_cores()
    sample = x.shape[0]
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
            "sample": f"{sample} of {frames} chirps x {n} points, {reps} passes, float64 oracle (oracle/rsp_oracle.c), OpenMP x{cores}"}


def cpu_baseline_fixed():
    """the bit-accurate fixed-point model (oracle/rsp_oracle.c orc_chain_fixed) on ONE thread, the
    reference's own configuration (BASELINE.json configs[0]: 1024-point FixedPoint SDF-FFT + JPL magnitude +
    16-cell CA-CFAR, G = 4) -- the closest analogue of the Chisel treadle/verilator simulation."""
    import rsp_chains_amd as R
    from oracle import oracle as O
    n, frames = 1024, 2048
    beats = np.stack([R.stimulus.formAXI4StreamComplexData(
        R.stimulus.getComplexTones(n, 0.125, 0.25, 0.5, shiftRangeFactor=12, seed=1234 + f)) for f in range(64)])
    beats = np.tile(beats, (frames // 64, 1))
    cfg = O.default_cfg(log2n=10, cfar_mode=O.CFAR_CA, ref_window=16, guard_window=4, div_sum=4)
    O.chain_fixed(beats[:16], cfg)
    t0 = time.perf_counter()
    reps = 0
    while True:
        O.chain_fixed(beats, cfg)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 8.0 or reps >= 50:
            break
    return {"value": frames * n * reps / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": f"{frames} frames x {n} points x {reps} passes, FIXED16 bit-accurate model, cfg1 parameters (R=16, G=4, CA), single thread"}


if __name__ == "__main__":
    main()
