#!/bin/bash
# Runs on the GPU box: bench line (frames in flight, with the moving-camera figure), the serial-schedule bench
# line, then the serial-schedule profile (kernel trace + PMC passes) of one workload; summaries under
# gpurun_out/prof_<tag>/.  usage: tools/r3_profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
timeout -k 10 280 python3 bench.py --no-cpu --orbit "$@" > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
timeout -k 10 280 python3 bench.py --no-cpu --serial "$@" > gpurun_out/bench_${TAG}_serial.json 2>> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
bash tools/profile.sh $TAG --serial "$@" || exit 1
OUT=gpurun_out/prof_$TAG
python3 tools/pmc_sum.py $OUT $OUT/traffic.json > $OUT/pmc_sum.csv
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -8 $OUT/kernel_stats.csv | cut -c1-160
