cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_single -- python3 $R/bench.py --no-cpu --steps 3 --warmup 1 --in-flight 1 > $R/gpurun_out/tl_single.log 2>&1 || exit 1
python3 $R/tools/timeline.py $R/gpurun_out/tl_single > $R/gpurun_out/tl_single.txt
