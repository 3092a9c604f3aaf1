#!/usr/bin/env python3
"""Prints the launches of the timed frames in a rocprofv3 kernel_trace.csv (instrumented
frames and scene set-up left out): kernel, hardware queue, start, end, duration, in
microseconds from the first launch.  Frames in flight overlap; the queue tells them apart.
usage: tools/timeline.py <dir>"""
import csv
import glob
import os
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "rt_exact::" in r["Kernel_Name"] or "rt_fast::" in r["Kernel_Name"]]
    rows = [r for r in rows if "WorkCounters" not in r["Kernel_Name"] and "k_appraise" not in r["Kernel_Name"]
            and "k_prepare" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the instrumented frames at the end launch resolve kernels too: stop at the last timed trace kernel
    last = max(i for i, r in enumerate(rows) if "NoCounters" in r["Kernel_Name"])
    while last + 1 < len(rows) and "k_resolve" in rows[last + 1]["Kernel_Name"] and \
            int(rows[last + 1]["Start_Timestamp"]) - int(rows[last]["End_Timestamp"]) < 1_000_000:
        last += 1
    rows = rows[:last + 1]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt_exact::", "").replace("rt_fast::", "")
        print("%-36s q%-3s start %9.1f us  end %9.1f us  dur %8.1f us" % (n, r["Queue_Id"], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))


if __name__ == "__main__":
    main()
