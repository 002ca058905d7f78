#!/usr/bin/env python3
"""gpurun_out/<round>/collect/ -> profiles/ (tracked): bench lines, rocprofv3 kernel-stats CSVs (the bench command and one
per shape), PMC summary + HBM traffic json (headline kernel, 2-D chain at both shapes, cfg 4, FIXED16), ablation table,
instruction-cost table.  Usage: tools/summarise_profiles.py [round]"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", rnd, "collect"); dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
for a, b in (("bench.json", f"bench_{rnd}.json"), ("bench_under_rocprof.json", f"bench_{rnd}_under_rocprof.json"),
             ("bench_1rank_rccl.json", f"bench_{rnd}_1rank_rccl.json"), ("pcie.json", f"pcie_inclusive_{rnd}.json"),
             ("ablate.txt", f"ablate_{rnd}.txt"), ("ablate_fx4096.txt", f"ablate_{rnd}_fx4096.txt"),
             ("ablate_gos.txt", f"ablate_{rnd}_gos.txt"), ("ablate_ca8192.txt", f"ablate_{rnd}_ca8192.txt"), ("valubench.txt", f"valubench_{rnd}.txt")):
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, b))


def kernel_stats(sub, out_name):
    for f in glob.glob(os.path.join(src, sub, "*", "*kernel_stats.csv")):
        rows = list(csv.reader(open(f)))
        keep = [rows[0]] + [r for r in rows[1:] if r and ("rsp::" in r[0] or "rccl" in r[0].lower())]
        csv.writer(open(os.path.join(dst, out_name), "w")).writerows(keep)
        return {r[0]: dict(zip(rows[0], r)) for r in keep[1:]}
    return {}


kernel_stats("stats", f"rocprof_{rnd}_kernel_stats.csv")
per_shape = {}
for tag in ("cfg2", "cfg3", "cfg5", "cfg4", "fx1024", "fx4096"):
    per_shape[tag] = kernel_stats("stats_" + tag, f"rocprof_{rnd}_{tag}_kernel_stats.csv")


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
doc = {"chain1d_hbm_bytes_per_launch": traffic(quad), "algorithmic_bytes_per_launch": 12 * 4096 * 4096,
       "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950), write bytes = WRITE_SIZE x 1024",
       "workload": "chain1d_quad_kernel<12,f32>, 4096 chirps x 4096 points, tools/prof_chain.py",
       "pmc": quad}
for key, prefix, match, algo, what in (
        ("cfg4_gos", "pmcgos", lambda k: "chain1d_gos" in k, 12 * 8192 * 2048, "chain1d_gos_kernel<13,f32>, 2048 x 8192, R = 32, k = 24"),
        ("fixed16_cfg1", "pmcfx1", lambda k: "chain1d_quad" in k, 8 * 1024 * 16384, "chain1d_quad_kernel<10,fixed16>, 16384 x 1024, R = 16"),
        ("fixed16_4096", "pmcfx4", lambda k: "chain1d_quad" in k, 8 * 4096 * 4096, "chain1d_quad_kernel<12,fixed16>, 4096 x 4096, R = 32")):
    p = pmc(prefix, match)
    doc[key] = {"hbm_bytes_per_launch": traffic(p), "algorithmic_bytes_per_launch": algo, "workload": what, "pmc": p}
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
    if "roofline" in v:
        print(k, v["ms_per_step"], v["roofline"]["kernel_ms"], v["roofline"]["frac"])
for k in ("cfg3_2d_chain", "cfg5_share_2d_chain"):
    print(k, "B/cell moved", doc[k]["hbm_bytes_per_cell"])
for tag, rows in per_shape.items():
    for name, r in rows.items():
        print(tag, name[:70], r.get("AverageNs") or r.get("Average"))
