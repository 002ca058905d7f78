cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/prof_rd.py 4096 512 8 10
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rd_stats -- python3 $R/tools/prof_rd.py 4096 512 8 10 > /dev/null 2>&1
cat $R/gpurun_out/rd_stats/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
