#!/bin/bash
# usage: tools/x_time.sh <lib suffix or ""> <tag> [bench args]  -- serial bench of a variant library
SUF=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
[ -n "$SUF" ] && export RTAMD_LIB=$ROOT/rust-raytracer_amd/librtamd_$SUF.so
timeout -k 10 200 python3 bench.py --no-cpu --serial --steps 6 --warmup 1 "$@" > gpurun_out/x_$TAG.json 2> gpurun_out/x_$TAG.err || { tail -3 gpurun_out/x_$TAG.err; exit 1; }
python3 - "$TAG" <<'PY'
import json,sys
j=json.loads([l for l in open("gpurun_out/x_%s.json"%sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], j["frame_ms"]["kernels_serial"])
PY
