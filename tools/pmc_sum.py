#!/usr/bin/env python3
"""Sums rocprofv3 --pmc output (counter_collection.csv files under a directory) per kernel and
counter.  usage: tools/pmc_sum.py <dir> [traffic.json] > summary.csv

With a second argument it also writes the memory-side bytes of one timed frame: FETCH_SIZE and
WRITE_SIZE (KiB) of the kernels a timed frame launches, FETCH_SIZE doubled (on gfx950 it tallies
128-byte requests at 64 bytes), divided by the number of frames in the run."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    sums = defaultdict(float)
    disp = defaultdict(set)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                short = name.split("(")[0].replace("void ", "")
                key = (short, row["Counter_Name"])
                sums[key] += float(row["Counter_Value"])
                disp[key].add((path, row["Dispatch_Id"]))
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "counter", "sum_over_dispatches", "dispatches"])
    for key in sorted(sums):
        w.writerow([key[0], key[1], sums[key], len(disp[key])])
    if len(sys.argv) > 2:
        # timed frames launch the NoCounters kernels; k_resolve runs in every frame, instrumented or not
        frames_timed = max((len(d) for k, d in disp.items() if "k_intersect<true" in k[0] and "NoCounters" in k[0]
                            and k[1] == "FETCH_SIZE"), default=0)
        frames_all = max((len(d) for k, d in disp.items() if "k_resolve<true>" in k[0] and k[1] == "FETCH_SIZE"), default=0)
        per = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
        for (kernel, counter), v in sums.items():
            if counter not in per:
                continue
            if "NoCounters" in kernel and frames_timed:
                per[counter] += v * 1024.0 / frames_timed
            elif "k_resolve" in kernel and frames_all:
                per[counter] += v * 1024.0 / frames_all
        out = {"frames": frames_timed, "fetch_bytes_per_frame": int(2 * per["FETCH_SIZE"]),
               "write_bytes_per_frame": int(per["WRITE_SIZE"]),
               "hbm_bytes_per_frame": int(2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]),
               "method": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, per-kernel sums of the timed "
                         "frames' kernels, FETCH_SIZE x 2 (gfx950), / frames"}
        # per launch, for the kernels of the timed frames (meaningful for a run on the serial schedule,
        # where a frame launches each of them a fixed number of times)
        names = {"k_shadow": "k_shadow<", "k_intersect0": "k_intersect<true", "k_intersect_deep": "k_intersect<false", "k_deep": "k_deep<",
                 "k_light": "k_light<", "k_resolve0": "k_resolve<true", "k_resolve_deep": "k_resolve<false"}
        for label, needle in names.items():
            f_sum = w_sum = 0.0
            n = 0
            for (kernel, counter), v in sums.items():
                if needle not in kernel or "WorkCounters" in kernel:
                    continue
                if counter == "FETCH_SIZE":
                    f_sum += v * 1024.0
                    n = max(n, len(disp[(kernel, counter)]))
                elif counter == "WRITE_SIZE":
                    w_sum += v * 1024.0
            if n:
                out[label] = {"launches": n, "fetch_bytes_per_launch": int(2 * f_sum / n), "write_bytes_per_launch": int(w_sum / n),
                              "hbm_bytes_per_launch": int((2 * f_sum + w_sum) / n)}
        with open(sys.argv[2], "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
