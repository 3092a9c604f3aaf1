# usage (on the GPU box): bash tools/x_share_trace.sh  -- launch timeline of one rank's eighth of the headline frame
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_share -- python3 $R/tools/share_trace.py $R > $R/gpurun_out/tl_share.log 2>&1 || exit 1
python3 $R/tools/timeline.py $R/gpurun_out/tl_share > $R/gpurun_out/tl_share.txt
tail -34 $R/gpurun_out/tl_share.txt
