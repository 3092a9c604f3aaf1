#!/usr/bin/env python3
"""Sums rocprofv3 --pmc output (counter_collection.csv files under a directory) per kernel and
counter.  usage: tools/pmc_sum.py <dir> > summary.csv"""
import csv
import glob
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


if __name__ == "__main__":
    main()
