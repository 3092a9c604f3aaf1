#!/bin/bash
# Runs on the GPU box: the bench lines kept under profiles/r03 (after tools/keep_profile.py, so that they cite the kept traffic files).
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { name=$1; shift; timeout -k 10 280 python3 bench.py --no-cpu "$@" > gpurun_out/line_$name.json 2> gpurun_out/line_$name.err || { tail -5 gpurun_out/line_$name.err; exit 1; }; echo $name done; }
run balls4_4k_depth8_exact --orbit
run balls4_4k_depth8_exact_serial --serial
run rings4_4k_depth8_exact --scene rings --size 4 --orbit
run rings4_4k_depth8_exact_serial --scene rings --size 4 --serial
run mount8l8_4k_depth8_exact --scene mount --size 8 --lights 8 --orbit
run mount8l8_4k_depth8_exact_serial --scene mount --size 8 --lights 8 --serial
run lathe96_1080p_depth8_exact --scene lathe --size 96 --width 1920 --height 1080 --orbit
run lathe96_1080p_depth8_exact_serial --scene lathe --size 96 --width 1920 --height 1080 --serial
run mount8l8_8k_depth8_exact --scene mount --size 8 --lights 8 --width 7680 --height 4320
run tetra6_8k_depth8_exact --scene tetra --size 6 --width 7680 --height 4320
run rings6_4k_depth8_exact --scene rings --size 6
run tetra6_4k_depth8_exact --scene tetra --size 6 --orbit
run tetra6_4k_depth8_exact_serial --scene tetra --size 6 --serial
