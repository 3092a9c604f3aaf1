#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats and separate PMC passes of the
# bench command.  Output under gpurun_out/prof_<tag>/; copy the summaries to profiles/.
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r00}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "kernel-trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '+')
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc pass $pass failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
done
find "$OUT" -name "*.csv" | head -50
