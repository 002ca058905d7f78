#!/usr/bin/env python3
"""gpurun_out/<round>/collect/ -> profiles/ (tracked): bench lines, kernel stats CSV of the bench command, PMC
summary + HBM traffic json (headline kernel, per-cell tail for comparison, 2-D chain kernels)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", rnd, "collect"); dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
for a, b in (("bench.json", f"bench_{rnd}.json"), ("bench_under_rocprof.json", f"bench_{rnd}_under_rocprof.json"),
             ("bench_cfg5_1rank_rccl.json", f"bench_{rnd}_cfg5_1rank_rccl.json"), ("pcie.json", f"pcie_inclusive_{rnd}.json")):
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, b))
for f in glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if r and ("rsp::" in r[0] or "rccl" in r[0].lower())]
    csv.writer(open(os.path.join(dst, f"rocprof_{rnd}_kernel_stats.csv"), "w")).writerows(keep)


def pmc(prefix, match):
    out = {}
    for f in glob.glob(os.path.join(src, prefix + "_*", "*", "*counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if match(r["Kernel_Name"]):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    return out


def traffic(p):
    f, w = p.get("FETCH_SIZE", {}).get("mean_per_launch"), p.get("WRITE_SIZE", {}).get("mean_per_launch")
    # MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half of a
    # wide coalesced stream's reads, WRITE_SIZE is exact
    return (2 * f * 1024 + w * 1024) if f and w else None


quad = pmc("pmc", lambda k: "chain1d_quad" in k)
gen = pmc("pmcgen", lambda k: "chain1d_kernel" in k)
doc = {"chain1d_hbm_bytes_per_launch": traffic(quad), "algorithmic_bytes_per_launch": 12 * 4096 * 4096,
       "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950), write bytes = WRITE_SIZE x 1024",
       "workload": "chain1d_quad_kernel<12,f32>, 4096 chirps x 4096 points, tools/prof_chain.py",
       "pmc": quad, "per_cell_tail_kernel_for_comparison": {"hbm_bytes_per_launch": traffic(gen), "pmc": gen}}
for tag, cells in (("rd3", 8 * 4096 * 512), ("rd5", 8 * 8192 * 1024)):
    ks = {}
    for name, m in (("range_fft", "range_fft"), ("doppler_mag", "doppler"), ("cfar2d_walk", "cfar2d")):
        p = pmc("pmc" + tag, lambda k, m=m: m in k)
        ks[name] = {"hbm_bytes_per_launch": traffic(p), "pmc": p}
    tot = sum(v["hbm_bytes_per_launch"] or 0 for v in ks.values())
    doc["cfg3_2d_chain" if tag == "rd3" else "cfg5_share_2d_chain"] = {
        "cells": cells, "algorithmic_bytes": 28 * cells, "hbm_bytes_all_three_kernels": tot,
        "hbm_bytes_per_cell": tot / cells if tot else None, "kernels": ks}
json.dump(doc, open(os.path.join(dst, f"traffic_{rnd}.json"), "w"), indent=1)
b = json.load(open(os.path.join(dst, f"bench_{rnd}.json")))
print("headline", b["value"], b["ms_per_step"], b["roofline"]["kernel_ms"], b["roofline"]["frac"], "traffic", doc["chain1d_hbm_bytes_per_launch"])
for k, v in b.get("extra", {}).items():
    print(k, v["ms_per_step"], v["roofline"]["kernel_ms"], v["roofline"]["frac"])
for k in ("cfg3_2d_chain", "cfg5_share_2d_chain"):
    print(k, "B/cell moved", doc[k]["hbm_bytes_per_cell"])
