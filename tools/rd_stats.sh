cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/rd_stats
python3 $R/tools/prof_rd.py "$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rd_stats -- python3 $R/tools/prof_rd.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/rd_stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60].ljust(60), r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min_us", float(r["MinNs"]) / 1e3)
PY
