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

N > 1 (BASELINE.json configs[4] = "cfg5" at N = 8): 8 Rx channels of 8192 x 1024 range-Doppler maps PER GPU
(8 N channels in all, contiguous channels per rank, rsp_chains_amd.dist.shard_range; N = 8 is the 64-Rx
configuration), 2-D FFT + JPL magnitude + 2-D CA-CFAR per channel, the list appended by the CFAR kernel, and ONE
RCCL all-gather per step of the packed detection lists -- sized to the lists, not to their capacity -- on a side
stream.  Per-GPU work is fixed: "scaling": "weak"; the N = 1 line carries the same 8-Rx share as
`scaling_baseline` (machine-readable: value(N) / (N x scaling_baseline.value) is the efficiency of THIS workload;
the N = 1 headline is a different, lighter configuration).  No data-path collective; the gather's cost is
reported separately.

Timing: W warm-up steps, then a PRE-HEAT of back-to-back steps until the step time is stationary (>= 150 ms; the
chip's clock moves under load and a 1-ms block measures a transient), then 5 blocks.  A block = R passes over the K
steps, R chosen so that a block is >= 50 ms of GPU time whatever K is; every block is bracketed by barrier +
synchronize on both sides and reduced with MAX over ranks.  `value` = the MEDIAN block (sustained), `best_block`
the fastest; `blocks_ms` lists all five and `sclk_mhz` the shader clock the driver reported before / after each
(sysfs pp_dpm_sclk), so that a slow block is attributable.  `roofline` is for the dominant kernel: algorithmic bytes
 / its mean launch duration measured with one HIP event pair per launch on the stream the kernel runs on
(rsp_chain_profile_*) over a repeat of one block, outside the blocks `value` comes from.  For the 1-D chain the pair
is bound to the kernel's dispatch (hipExtLaunchKernelGGL: the kernel's own begin / end timestamps, the quantity
rocprofv3's kernel trace reports in profiles/); the 2-D chain's pair brackets its three kernels and their two
boundaries.
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
BLOCK_MIN_S = 0.050    # a timed block is at least this much GPU time (a pass over --steps is repeated inside it)
PREHEAT_MIN_S, PREHEAT_MAX_S = 0.150, 1.0
METRIC = "range-Doppler cells/sec (FFT+CFAR)"


class SclkReader:
    """Current shader clock (MHz) from the amdgpu driver's sysfs table of the device this rank runs on.  None when the
    table is not readable on this box (the line then says so; the clock is a label, never an input)."""

    def __init__(self, torch, dev):
        import glob
        self.path = None
        want = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
        except Exception:
            pass
        cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        for c in cands:
            try:
                real = os.path.realpath(os.path.dirname(c))
            except OSError:
                continue
            if want and want in real:
                self.path = c
                break
        if self.path is None and len(cands) == 1:
            self.path = cands[0]

    def __call__(self):
        if not self.path:
            return None
        try:
            for line in open(self.path):
                if line.rstrip().endswith("*"):
                    return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
        except Exception:
            return None
        return None


CFG2_CAP = 1 << 15
RD_CAP = 1 << 16
GATHER_EVERY = 8   # N > 1, cfg 2: detection lists per RCCL all-gather


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

    def sync_only(self):
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.use_dist:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def timed_blocks(step, steps, warmup, fence, first=0, sclk=None, flush=None):
    """warm-up, pre-heat to a stationary step time, then N_BLOCKS blocks of `reps` passes over `steps` steps.
    Returns a dict: per-block seconds PER STEP (max over ranks), reps, the clock readings, the next step index."""
    i = first
    for _ in range(warmup):
        step(i)
        i += 1
    # pre-heat: windows of >= 25 ms until two consecutive ones agree within 3 % (and >= 150 ms in all).  Every decision
    # is taken on MAX-over-ranks times, so that all ranks run the same number of steps (a step may hold a collective).
    fence()
    total, prev, per_step = 0.0, None, None
    n_win = max(steps, 8)
    while True:
        t0 = time.perf_counter()
        for _ in range(n_win):
            step(i)
            i += 1
        fence.sync_only()
        dt = fence.max_over_ranks(time.perf_counter() - t0)
        total += dt
        per_step = dt / n_win
        if dt < 0.025 and total < PREHEAT_MAX_S:
            n_win = int(n_win * max(2.0, 0.03 / max(dt, 1e-6))) + 1
            continue
        stationary = prev is not None and abs(per_step - prev) <= 0.03 * prev
        if (total >= PREHEAT_MIN_S and stationary) or total >= PREHEAT_MAX_S:
            break
        prev = per_step
    preheat_s = total
    reps = max(1, int(np.ceil(BLOCK_MIN_S / max(per_step * steps, 1e-9))))
    out, clocks = [], []
    for _ in range(N_BLOCKS):
        fence()
        c0 = sclk() if sclk else None
        t0 = time.perf_counter()
        for _ in range(reps * steps):
            step(i)
            i += 1
        if flush:
            flush(i)   # inside the timed region: lists not yet gathered (a partial batch) travel before the clock stops
        fence()
        dt = time.perf_counter() - t0
        clocks.append([c0, sclk() if sclk else None])
        out.append(fence.max_over_ranks(dt) / (reps * steps))
    return {"per_step_s": out, "reps": reps, "sclk_mhz": clocks, "next": i, "preheat_s": preheat_s}


def kernel_ms(dut, step, steps, first, fence):
    """mean duration of the chain kernel(s) per step: HIP event pair per launch, same step sequence."""
    steps = min(steps, 1000)  # one event pair per launch: bounded
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
    for name in ("traffic_r03.json", "traffic_r02.json", "traffic_r01.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                v = json.load(open(path))
                for k in key.split("/"):
                    v = v.get(k) if isinstance(v, dict) else None
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


def make_cfg2(R, torch, dev, local_rank, rank, n=4096, frames=4096, lists=None):
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
    cap = CFG2_CAP  # list capacity per step (expected ~13 k peaks): 512 KiB
    if lists is None:
        lists = [torch.zeros(cap + 1, 4, dtype=torch.int32, device=dev) for _ in range(N_SETS)]  # row 0 = {found, stored}

    def step(i):
        s, lst = i % N_SETS, lists[i % len(lists)]
        dut.process_detect_device(ins[s].data_ptr(), frames, outs[s].data_ptr(), lst[1:].data_ptr(), cap, lst[0].data_ptr())

    name = (f"cfg2: 1-ch {n}-pt range FFT + JPL logMag + CA-CFAR (R=32,G=4), {frames}-chirp batch, fp32, "
            "dense words + detection list")
    return dict(dut=dut, step=step, cells=n * frames, bytes=12.0 * n * frames, kernel=f"chain1d_quad_kernel<{n.bit_length() - 1},f32>",
                name=name, ins=ins, lists=lists, n=n, frames=frames, cap=cap)


def make_rd(R, torch, dev, local_rank, rank, nr, nd, n_ch, tag, sets=N_SETS, with_list=True, lists=None):
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
    cap = RD_CAP
    if lists is None:
        lists = [torch.zeros(cap + 1, 4, dtype=torch.int32, device=dev) for _ in range(sets)]

    def step(i):
        s, lst = i % sets, lists[i % len(lists)]
        if with_list:   # the CFAR kernel appends its peaks itself: no second pass over the dense words
            dut.process_detect_device(ins[s].data_ptr(), n_ch, outs[s].data_ptr(), lst[1:].data_ptr(), cap, lst[0].data_ptr())
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


def make_fixed(R, torch, dev, local_rank, n, frames, ref):
    """FIXED16 1-D chain = the reference's own arithmetic (FixedPoint(16.W, 12.BP), FftMagCfarChain.scala:79-81):
    the tester's stimulus (3 tones + noise, RspChainTesterUtils.scala:56-67), CA-CFAR, G = 4; 8 B per cell"""
    params = R.FftMagCfarVanillaParameters(
        fftParams=R.FFTParams.fixed(numPoints=n), magParams=R.MAGParams.fixed(),
        cfarParams=R.CFARParams(fftSize=n), dtype=R.FIXED16, device=local_rank)
    rt = R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging", refWindowSize=ref, guardWindowSize=4,
                                 divSum=ref.bit_length() - 1, thresholdScaler=3.5)
    dut = R.FftMagCfarChainVanilla(params)
    dut.configure(rt)
    uniq = 64
    beats = np.stack([R.stimulus.formAXI4StreamComplexData(
        R.stimulus.getComplexTones(n, 0.125, 0.25, 0.5, shiftRangeFactor=12, seed=1234 + f)) for f in range(uniq)])
    host = np.tile(beats, (frames // uniq, 1)).astype(np.uint32)
    ins = [torch.from_numpy(host.view(np.int32)).to(dev) for _ in range(N_SETS)]
    outs = [torch.empty(frames, n, dtype=torch.int32, device=dev) for _ in range(N_SETS)]

    def step(i):
        s = i % N_SETS
        dut.process_device(ins[s].data_ptr(), frames, outs[s].data_ptr())

    return dict(dut=dut, step=step, cells=n * frames, bytes=8.0 * n * frames, kernel=f"chain1d_quad_kernel<{n.bit_length() - 1},fixed16>",
                name=f"FIXED16 (the reference's data type): {n}-pt FFT + JPL + CA-CFAR (R={ref},G=4), {frames}-frame batch, dense words")


def summarise(w, steps, tb, kms, traffic=None, traffic_src=None):
    per = tb["per_step_s"]
    sec = median(per)
    det = None
    if w.get("lists") is not None:
        found, stored = (int(v) for v in w["lists"][0][0, :2].tolist())
        det = {"found": found, "stored": stored}
    return {"workload": w["name"], "cells_per_step": w["cells"], "steps": steps, "ms_per_step": sec * 1e3,
            "detections_per_step": det, "value": w["cells"] / sec, "best_block": w["cells"] / min(per), "unit": "cells/s",
            "blocks_ms": [b * 1e3 for b in per], "block_reps": tb["reps"], "sclk_mhz": tb["sclk_mhz"],
            "roofline": roofline(w["kernel"], kms, w["bytes"], traffic, traffic_src)}


def run_extra(w, torch, steps, warmup, sclk, traffic_key=None):
    """one extra configuration on this GPU: sustained median block + kernel time, freed afterwards"""
    stream = torch.cuda.current_stream()
    w["dut"].set_stream(stream.cuda_stream)
    fence = Fence(torch, None, False, None)
    tb = timed_blocks(w["step"], steps, warmup, fence, sclk=sclk)
    kms, _ = kernel_ms(w["dut"], w["step"], steps * tb["reps"], tb["next"], fence)
    traffic, src = static_traffic(traffic_key) if traffic_key else (None, None)
    return summarise(w, steps, tb, kms, traffic, src)


def host_entry(R, local_rank, n=4096, frames=4096):
    """The host-buffer entry rsp_chain_process -- what the reference-side binding calls (FftMagCfarChainTester.scala:137,
    145-151 -> JNI) -- on the cfg-2 batch: PCIe-inclusive, never `value`.  Pageable NumPy arrays (staged through the
    pinned ring) and buffers from rsp_host_alloc (DMA in place); both the chunked H2D || kernel || D2H pipeline."""
    params = R.FftMagCfarVanillaParameters(fftParams=R.FFTParams.fixed(numPoints=n), magParams=R.MAGParams.fixed(),
                                           cfarParams=R.CFARParams(fftSize=n), dtype=R.F32, device=local_rank)
    out = {}
    with R.FftMagCfarChainVanilla(params) as dut:
        dut.configure(R.RunTimeRspChainParams(fftSize=n, CFARMode="Cell Averaging", refWindowSize=32, guardWindowSize=4,
                                              divSum=5, thresholdScaler=3.5))
        x = np.tile(R.stimulus.chirp_frames(64, n, seed=1), (frames // 64, 1))
        hin, hout = R.HostBuffer((frames, n), np.complex64, local_rank), R.HostBuffer(frames * n, np.uint32, local_rank)
        hin.array[...] = x
        pageable_out = np.zeros(frames * n, np.uint32)
        for tag, a, o in (("pageable", x, pageable_out), ("pinned", hin.array, hout.array)):
            dut.stream(a, out=o)
            t = []
            for _ in range(5):
                t0 = time.perf_counter()
                dut.stream(a, out=o)
                t.append(time.perf_counter() - t0)
            out[tag] = {"ms_per_batch": median(t) * 1e3, "cells_per_s": n * frames / median(t)}
        out["results_identical"] = bool(np.array_equal(pageable_out, hout.array))
        hin.free()
        hout.free()
    out["entry"] = "rsp_chain_process, cfg-2 batch (134 MB in + 67 MB out over PCIe), chunked H2D || kernel || D2H on three streams"
    return out


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
                    help="default cfg2 (configs[1]) per GPU at every N, + configs[4] as extra.cfg5 at N > 1; cfg5 = configs[4] as the line itself")
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
    sclk = SclkReader(torch, dev)
    # the SAME per-GPU workload at every N (value(N) / value(1) is then a scaling efficiency); at N > 1 the line also
    # carries BASELINE.json configs[4] (cfg 5) as extra.cfg5
    workload = args.workload or "cfg2"
    if workload == "cfg5":
        line = run_cfg5(args, torch, dist, R, rank, local_rank, world, dev, use_dist, main_stream, fence, sclk)
    else:
        line = run_cfg2(args, torch, dist, R, rank, local_rank, world, dev, use_dist, main_stream, fence, sclk)
    if rank == 0:
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
    if use_dist:
        dist.destroy_process_group()


class GatherLeg:
    """The only exchange of the N > 1 path: the RCCL all-gather of the packed detection lists (row 0 = {found, stored}),
    issued on a side stream under the following steps' kernels and sized to the lists (dist.PackedGatherer).
    every = K: the lists of K consecutive steps travel in ONE collective (dist.gather_packed_batch; double-buffered
    batches of K list slots, the chain writes step i's list straight into slot i % K); K = 1: one collective per step."""

    def __init__(self, torch, dev, world, main_stream, cap, enabled, every=1):
        from rsp_chains_amd.dist import PackedGatherer
        self.torch, self.main, self.cap, self.on, self.k = torch, main_stream, cap, enabled, every
        self.batches = [torch.zeros(every, cap + 1, 4, dtype=torch.int32, device=dev) for _ in range(2)]
        self.lists = [self.batches[b][j] for b in range(2) for j in range(every)]   # list of step i = lists[i % (2 K)]
        self.pg = PackedGatherer(cap) if enabled else None
        self.last = 0
        if enabled:
            self.comm = torch.cuda.Stream(device=dev)
            self.stage = torch.empty(every * (cap + 1), 4, dtype=torch.int32, device=dev)
            self.g_out = [torch.empty(world * every * (cap + 1), 4, dtype=torch.int32, device=dev) for _ in range(2)]
            self.ready = [torch.cuda.Event() for _ in range(2)]
            self.gathered = [torch.cuda.Event() for _ in range(2)]
            self.views = [None, None]

    def _gather(self, b):
        self.ready[b].record(self.main)
        with self.torch.cuda.stream(self.comm):
            self.comm.wait_event(self.ready[b])
            self.views[b] = self.pg.gather_batch(self.batches[b], out=self.g_out[b], stage=self.stage)
            self.gathered[b].record(self.comm)

    def wrap(self, inner):
        def step(i):
            b = (i // self.k) % 2
            if self.on and i % self.k == 0 and self.views[b] is not None:
                self.main.wait_event(self.gathered[b])   # do not overwrite a batch still being gathered
            inner(i)
            if self.on and (i + 1) % self.k == 0:
                self._gather(b)
                self.last = i + 1
        return step

    def flush(self, i):
        """lists of steps [last, i) are still local: gather their (partial) batch"""
        if self.on and i > self.last and i % self.k != 0:
            self._gather((i // self.k) % 2)
            self.last = i

    def size_payload(self, step, fence):
        """off the clock: a few steps, read their headers, grow the travelling row count if a list did not fit"""
        if not self.on:
            return
        for i in range(2 * self.k):
            step(i)
        fence()
        for v in self.views:
            self.pg.settle(v)

    def complete(self, fence):
        """every list of the last gathered batches fitted the rows that travelled?"""
        if not self.on:
            return True
        fence()
        return all(self.pg.settle(v) for v in self.views if v is not None)

    def alone_ms(self, fence):
        if not self.on:
            return None
        fence()
        t0 = time.perf_counter()
        for j in range(20):
            self.pg.gather_batch(self.batches[j % 2], out=self.g_out[j % 2], stage=self.stage)
        fence()
        return fence.max_over_ranks(time.perf_counter() - t0) / 20 * 1e3

    def report(self, gather_ms, sec_without, complete):
        return {"lists_per_collective": self.k, "ms_per_collective_alone": gather_ms,
                "ms_per_step_without_gather": sec_without * 1e3,
                "rows_per_list": self.pg.rows if self.on else None,
                "bytes_per_rank_per_collective": self.pg.rows * 16 * self.k if self.on else None,
                "capacity_rows": self.cap + 1, "every_gathered_list_complete": complete}


def run_cfg2(args, torch, dist, R, rank, local_rank, world, dev, use_dist, main_stream, fence, sclk):
    # N > 1: the same batch on every GPU (chirps shard across ranks; weak scaling).  A 49-us step cannot hide a collective
    # of its own (12 us per step on a 1-rank group: launch latency + the CU share of the RCCL kernel), so the lists of
    # GATHER_EVERY steps travel in one all-gather; the per-step form is measured beside it.
    leg = GatherLeg(torch, dev, world, main_stream, CFG2_CAP, use_dist, every=GATHER_EVERY if use_dist else 1)
    w = make_cfg2(R, torch, dev, local_rank, rank, lists=leg.lists if use_dist else None)
    dut = w["dut"]
    dut.set_stream(main_stream.cuda_stream)
    step = leg.wrap(w["step"])
    leg.size_payload(step, fence)
    tb = timed_blocks(step, args.steps, args.warmup, fence, sclk=sclk, flush=leg.flush)
    nxt = tb["next"]
    per = tb["per_step_s"]
    sec = median(per)
    gather_rep = None
    if use_dist:
        complete = leg.complete(fence)
        leg.on = False   # the same steps without the collective: what the gather costs end to end
        tb_ng = timed_blocks(step, args.steps, 1, fence, first=nxt)
        nxt = tb_ng["next"]
        leg.on = True
        gather_rep = leg.report(leg.alone_ms(fence), median(tb_ng["per_step_s"]), complete)
        leg.on = False
        # one collective per step, for comparison
        leg1 = GatherLeg(torch, dev, world, main_stream, CFG2_CAP, True, every=1)
        w1 = make_cfg2(R, torch, dev, local_rank, rank, lists=leg1.lists)
        w1["dut"].set_stream(main_stream.cuda_stream)
        step1 = leg1.wrap(w1["step"])
        leg1.size_payload(step1, fence)
        tb1 = timed_blocks(step1, args.steps, 2, fence, first=2, flush=leg1.flush)
        gather_rep["ms_per_step_with_one_collective_per_step"] = median(tb1["per_step_s"]) * 1e3
        del leg1, w1, step1
        torch.cuda.empty_cache()
    kms, launches = kernel_ms(dut, w["step"], args.steps * tb["reps"], nxt, fence)
    found, stored = (int(v) for v in w["lists"][(nxt - 1) % len(w["lists"])][0, :2].tolist())
    line = None
    if rank == 0:
        traffic, src = static_traffic("chain1d_hbm_bytes_per_launch")
        line = {
            "metric": METRIC, "value": w["cells"] * world / sec, "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "best_block": w["cells"] * world / min(per),
            "blocks": N_BLOCKS, "blocks_ms": [b * 1e3 for b in per], "block_reps": tb["reps"], "sclk_mhz": tb["sclk_mhz"],
            "preheat_ms": tb["preheat_s"] * 1e3,
            "timing": (f"pre-heated to a stationary step time, then {N_BLOCKS} blocks of block_reps x steps steps (>= 50 ms each); "
                       "value = median block (sustained), best_block = fastest; sclk_mhz = driver-reported shader clock before / after each block"),
            "config": {"workload": w["name"] + ("" if not use_dist else f"; the same batch on each of {world} GPUs (chirps shard "
                                                f"across ranks, weak scaling) + ONE RCCL all-gather of the packed detection lists per {GATHER_EVERY} steps, side stream"),
                       "cells_per_step_per_gpu": w["cells"], "buffer_sets": N_SETS,
                       "detections_last_step": {"found": found, "stored": stored}},
            "roofline": roofline(w["kernel"], kms, w["bytes"], traffic, src),
        }
        line["roofline"]["kernel_ms_launches"] = launches
        if use_dist:
            line["gather"] = gather_rep
            line["per_gpu_value"] = w["cells"] / sec
    if use_dist and not args.no_extra:
        # BASELINE.json configs[4] (64 Rx sharded 8 per GPU at N = 8) in the same run, as its own record
        del w, dut, step, leg
        torch.cuda.empty_cache()
        sub = run_cfg5(args, torch, dist, R, rank, local_rank, world, dev, use_dist, main_stream, fence, sclk)
        if rank == 0:
            line["extra"] = {"cfg5": sub}
        return line
    if world == 1 and not use_dist and not args.no_extra:
        x0 = w["ins"][0]
        host_sample = x0[:2048].cpu().numpy().view(np.complex64).reshape(2048, w["n"])
        del w, dut, step
        torch.cuda.empty_cache()
        extra = {}
        steps_x = max(8, args.steps // 3)
        jobs = (("cfg3", lambda: make_rd(R, torch, dev, local_rank, rank, 4096, 512, 8, "cfg3"), "cfg3_2d_chain/hbm_bytes_all_three_kernels"),
                ("cfg4", lambda: make_cfg4(R, torch, dev, local_rank, rank), "cfg4_gos/hbm_bytes_per_launch"),
                ("cfg5_share", lambda: make_rd(R, torch, dev, local_rank, rank, 8192, 1024, 8, "cfg5 share (8 of 64 Rx)", sets=2),
                 "cfg5_share_2d_chain/hbm_bytes_all_three_kernels"),
                ("fixed16_cfg1", lambda: make_fixed(R, torch, dev, local_rank, 1024, 16384, 16), "fixed16_cfg1/hbm_bytes_per_launch"),
                ("fixed16_4096", lambda: make_fixed(R, torch, dev, local_rank, 4096, 4096, 32), "fixed16_4096/hbm_bytes_per_launch"))
        for tag, make, tkey in jobs:
            wx = make()
            extra[tag] = run_extra(wx, torch, steps_x if tag != "cfg5_share" else max(4, steps_x // 2), 3, sclk, tkey)
            del wx
            torch.cuda.empty_cache()
        extra["host_entry"] = host_entry(R, local_rank)
        line["extra"] = extra
        # the same per-GPU work the N > 1 lines run (8 Rx of 8192 x 1024 per GPU): the 1-GPU point of THAT scaling curve
        line["scaling_baseline"] = {"workload": extra["cfg5_share"]["workload"], "value": extra["cfg5_share"]["value"],
                                    "unit": "cells/s", "n_gpus": 1,
                                    "note": "the 1-GPU point of extra.cfg5 of the N > 1 lines (8 Rx of 8192 x 1024 per GPU, weak scaling); `value` is configs[1] per GPU at every N"}
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host_sample, 4096, 4096)
            line["cpu_baseline_fixed"] = cpu_baseline_fixed()
    return line


def run_cfg5(args, torch, dist, R, rank, local_rank, world, dev, use_dist, main_stream, fence, sclk):
    """BASELINE.json configs[4] at N = 8: 8 Rx x 8192 x 1024 per GPU (64 Rx on eight), RCCL gather of the lists."""
    from rsp_chains_amd.dist import shard_range
    per_gpu, nr, nd = 8, 8192, 1024
    total_ch = per_gpu * world
    lo, hi = shard_range(total_ch, rank, world)
    n_ch = hi - lo
    leg = GatherLeg(torch, dev, world, main_stream, RD_CAP, use_dist, every=1)   # a 0.5-ms step hides its own collective
    w = make_rd(R, torch, dev, local_rank, rank, nr, nd, n_ch, f"cfg5: {total_ch} Rx, {n_ch} per GPU", sets=2,
                lists=leg.lists if use_dist else None)
    dut, sets, cap = w["dut"], w["sets"], w["cap"]
    dut.set_stream(main_stream.cuda_stream)
    step = leg.wrap(w["step"])
    leg.size_payload(step, fence)
    tb = timed_blocks(step, args.steps, args.warmup, fence, sclk=sclk, flush=leg.flush)
    per = tb["per_step_s"]
    sec = median(per)
    complete = leg.complete(fence)
    gather = leg.on
    leg.on = False   # the same steps without the collective: what the gather costs end to end
    tb_ng = timed_blocks(step, args.steps, 1, fence, first=tb["next"])
    sec_ng = median(tb_ng["per_step_s"])
    kms, _ = kernel_ms(dut, w["step"], args.steps * tb["reps"], tb_ng["next"], fence)
    leg.on = gather
    gather_ms = leg.alone_ms(fence)
    if rank != 0:
        return None
    cells_total = total_ch * nd * nr
    traffic, src = static_traffic("cfg5_share_2d_chain/hbm_bytes_all_three_kernels")
    return {
        "metric": METRIC, "value": cells_total / sec, "unit": "cells/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "best_block": cells_total / min(per),
        "blocks": N_BLOCKS, "blocks_ms": [b * 1e3 for b in per], "block_reps": tb["reps"], "sclk_mhz": tb["sclk_mhz"],
        "timing": (f"pre-heated, then {N_BLOCKS} blocks of block_reps x steps steps (>= 50 ms each); value = median block, "
                   "MAX over ranks per block"),
        "config": {"workload": w["name"] + "; RCCL all-gather of the packed lists, one per step, side stream",
                   "cells_per_step_total": cells_total, "channels_per_gpu": n_ch, "buffer_sets": sets,
                   "sharding": "contiguous channels per rank (dist.shard_range); no data-path collective"},
        "gather": leg.report(gather_ms, sec_ng, complete),
        "scaling_note": ("weak scaling: 8 Rx of 8192 x 1024 per GPU at every N (N = 8 is BASELINE.json configs[4]); the 1-GPU point of this "
                         "curve is the N = 1 line's `scaling_baseline` (= extra.cfg5_share)"),
        "per_gpu_value": cells_total / sec / world,
        "roofline": roofline(w["kernel"], kms, 28.0 * n_ch * nd * nr, traffic, src),
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
