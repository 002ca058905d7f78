cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1 2 4 6 7; do
  RSP_DBG2D=$m rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl2d_$m -- python3 $R/tools/prof_rd.py 4096 512 8 6 > /dev/null 2>&1
  echo -n "dbg=$m: "; grep cfar2d $R/gpurun_out/abl2d_$m/*/*kernel_stats.csv | cut -d, -f4
done
