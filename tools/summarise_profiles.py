#!/usr/bin/env python3
"""gpurun_out/r01/ -> profiles/ (tracked): kernel stats CSV, PMC summary, traffic json, bench json."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r01"); dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench_r01.json"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "bench_r01_under_rocprof.json"))
for f in glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if r and ("rsp::" in r[0] or "rccl" in r[0].lower())]
    csv.writer(open(os.path.join(dst, "rocprof_r01_kernel_stats.csv"), "w")).writerows(keep)
pmc = {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "chain1d" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
fetch_kb = pmc.get("FETCH_SIZE", {}).get("mean_per_launch"); write_kb = pmc.get("WRITE_SIZE", {}).get("mean_per_launch")
traffic = None
if fetch_kb and write_kb:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB-like units of 1024 B; on gfx950
    # FETCH_SIZE reports exactly half of a coalesced streaming read (calibrated here: 2 x FETCH =
    # 134 MB = the input this kernel must read once); WRITE_SIZE is exact.
    traffic = 2 * fetch_kb * 1024 + write_kb * 1024
json.dump({"chain1d_hbm_bytes_per_launch": traffic, "fetch_size_raw_kb": fetch_kb, "write_size_raw_kb": write_kb,
           "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950), write bytes = WRITE_SIZE x 1024",
           "algorithmic_bytes_per_launch": 12 * 4096 * 4096, "pmc": pmc,
           "workload": "chain1d_kernel<12,f32>, 4096 chirps x 4096 points, tools/prof_chain.py"},
          open(os.path.join(dst, "traffic_r01.json"), "w"), indent=1)
for w in ("cfg3", "cfg4", "cfg5"):
    f = os.path.join(src, f"bench_{w}.json")
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, f"bench_r01_{w}.json"))
for w in ("cfg3", "cfg4"):
    for f in glob.glob(os.path.join(src, f"stats_{w}", "*", "*kernel_stats.csv")):
        rows = list(csv.reader(open(f)))
        keep = [rows[0]] + [r for r in rows[1:] if r and "rsp::" in r[0]]
        csv.writer(open(os.path.join(dst, f"rocprof_r01_{w}_kernel_stats.csv"), "w")).writerows(keep)
if os.path.exists(os.path.join(src, "pcie.json")):
    shutil.copy(os.path.join(src, "pcie.json"), os.path.join(dst, "pcie_inclusive_r01.json"))
print(open(os.path.join(dst, "bench_r01.json")).read()[:600]); print("traffic", traffic)
