#!/bin/bash
# usage: tools/x_bench.sh <lib suffix or ""> <tag> [bench args]  -- bench line of a variant library (frames in flight as bench.py chooses)
SUF=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
[ -n "$SUF" ] && export RTAMD_LIB=$ROOT/rust-raytracer_amd/librtamd_$SUF.so
timeout -k 10 250 python3 bench.py --no-cpu --steps 20 --warmup 2 "$@" > gpurun_out/xb_$TAG.json 2> gpurun_out/xb_$TAG.err || { tail -3 gpurun_out/xb_$TAG.err; exit 1; }
python3 - "$TAG" <<'PY'
import json,sys
j=json.loads([l for l in open("gpurun_out/xb_%s.json"%sys.argv[1]) if l.startswith("{")][-1])
f=j["frame_ms"]
print(sys.argv[1], "Mrays/s", j["value"], "step", f["step_wall"], "single", f["single_frame"], "host", f["single_frame_host"], "in flight", f["frames_in_flight"], "serial", f["kernels_serial"]["ms_frame"])
PY
