#!/usr/bin/env python3
"""Prints the launches of the last complete frame in a rocprofv3 kernel_trace.csv.
usage: tools/timeline.py <dir>"""
import csv
import glob
import os
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "rt_exact::" in r["Kernel_Name"] or "rt_fast::" in r["Kernel_Name"]]
    rows = [r for r in rows if "WorkCounters" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_intersect<true" in r["Kernel_Name"]]
    fr = rows[idx[-2]:idx[-1]] if len(idx) > 1 else rows[idx[-1]:]
    t0 = int(fr[0]["Start_Timestamp"])
    for r in fr:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rt_exact::", "").replace("rt_fast::", "")
        print("%-40s q%-3s start %9.1f us  end %9.1f us  dur %8.1f us" % (n, r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))


if __name__ == "__main__":
    main()
