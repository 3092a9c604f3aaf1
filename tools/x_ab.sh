#!/bin/bash
# A/B on the GPU box: bench lines for a list of environment settings.  usage: tools/x_ab.sh "<bench args>" "ENV1=.. ENV2=.." "ENV..." ...
ARGS=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
k=0
for e in "$@"; do
  k=$((k+1))
  env $e timeout -k 10 150 python bench.py --no-cpu --steps 20 $ARGS > gpurun_out/ab_$k.json 2> gpurun_out/ab_$k.err || { tail -3 gpurun_out/ab_$k.err; exit 1; }
  python3 - "$e" gpurun_out/ab_$k.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
f = j["frame_ms"]
r = j["roofline"]
print("%-34s value %8.1f step %.3f single %.3f serial: l0 %.3f deep %.3f shadow %.3f frame %.3f fif %d lane_use %.3f frac %.4f" % (sys.argv[1], j["value"], f["step_wall"], f["single_frame"],
      f["kernels_serial"]["ms_intersect0"], f["kernels_serial"]["ms_intersect_deep"], f["kernels_serial"]["ms_shadow"], f["kernels_serial"]["ms_frame"], f["frames_in_flight"], r.get("walk_lane_use") or 0, r["frac"]))
PY
done
