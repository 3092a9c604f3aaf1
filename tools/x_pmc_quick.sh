#!/bin/bash
# (scratch) the two SQ counter passes of a serial-schedule bench run, summed per kernel.  usage: tools/x_pmc_quick.sh <tag> [bench args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pq_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu --serial $*"
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '+')
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
done
python3 $ROOT/tools/pmc_sum.py $OUT > $OUT/pmc_sum.csv
grep "NoCounters" $OUT/pmc_sum.csv | grep "k_shadow\|k_deep\|k_intersect<false"
