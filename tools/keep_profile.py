#!/usr/bin/env python3
"""Copies the summaries tools/r3_profile.sh left under gpurun_out/prof_<tag>/ (and the two bench lines beside it)
into profiles/<round>/ under the workload's name, stamping the traffic file with the commit the library was built
from (HEAD; refuses if the kernel sources differ from it).
usage: tools/keep_profile.py <tag> <workload name, e.g. balls4k_depth8_exact> [round directory, default r03]"""
import json, os, shutil, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r03"
dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "rust-raytracer_amd/csrc", "include", "bench.py"],
                       capture_output=True, text=True).stdout.strip()
if dirty:
    sys.exit("sources differ from HEAD:\n" + dirty)
commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
src, dst = os.path.join(root, "gpurun_out", "prof_" + tag), os.path.join(root, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, f"kernel_stats_{name}_serial.csv"))
shutil.copy(os.path.join(src, "pmc_sum.csv"), os.path.join(dst, f"pmc_{name}_serial.csv"))
with open(os.path.join(src, "traffic.json")) as f:
    t = json.load(f)
t["commit"] = commit
with open(os.path.join(dst, f"traffic_{name}_serial.json"), "w") as f:
    json.dump(t, f, indent=1)
for suffix, out in (("", f"bench_{name}.json"), ("_serial", f"bench_{name}_serial.json")):
    p = os.path.join(root, "gpurun_out", f"bench_{tag}{suffix}.json")
    line = [l for l in open(p) if l.startswith("{")][-1]
    with open(os.path.join(dst, out), "w") as f:
        f.write(line)
print("kept", name, "at", commit)
