#!/bin/bash
# Runs on the GPU box: short bench of every library build under variants/ (and the in-tree one).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
for lib in rust-raytracer_amd/librtamd.so variants/*.so; do
  [ -f "$lib" ] || continue
  RTAMD_LIB=$ROOT/$lib timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu "$@" > gpurun_out/bv.log 2>&1 || { echo "$lib FAILED"; tail -3 gpurun_out/bv.log; exit 1; }
  python3 - "$lib" <<'PY'
import json, sys
j = json.loads([l for l in open("gpurun_out/bv.log") if l.startswith("{")][-1])
print(sys.argv[1], "Mrays/s", j["value"], "device_frame_ms", j["frame_ms"]["device_frame"], "lane_use", j["roofline"].get("walk_lane_use"))
PY
done
