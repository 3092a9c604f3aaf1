#!/bin/bash
# Runs on the GPU box (through gpurun): traversal + render parity tests, then a short bench.
# usage: tools/gpu_check.sh <tag> [bench args...]
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
timeout -k 10 400 python -u -m pytest tests/test_gpu_traverse.py tests/test_gpu_render.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_$TAG.log 2>&1
rc=$?
tail -4 gpurun_out/t_$TAG.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu "$@" > gpurun_out/b_$TAG.log 2>&1 || { tail -5 gpurun_out/b_$TAG.log; exit 1; }
python3 - "$TAG" <<'PY'
import json, sys
line = [l for l in open("gpurun_out/b_%s.log" % sys.argv[1]) if l.startswith("{")][-1]
j = json.loads(line)
w = j["roofline"]["work"]
rays = j["config"]["rays_per_frame"]["total"]
print("Mrays/s", j["value"], "frame_ms", j["frame_ms"], "frac", j["roofline"]["frac"])
print({k: round(w[k] / rays, 2) for k in ("bound_tests", "bound_passes", "sphere_tests", "lane_slots", "node_visits", "poly_tests") if k in w})
PY
