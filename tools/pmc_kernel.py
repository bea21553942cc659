#!/usr/bin/env python3
"""Sum rocprofv3 PMC counters per kernel over the *counter_collection.csv files of one or more pass directories.
usage: pmc_kernel.py <kernel-name substring> <dir> [<dir> ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    key, dirs = sys.argv[1], sys.argv[2:]
    acc, n = defaultdict(float), defaultdict(int)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if key in row["Kernel_Name"]:
                        acc[row["Counter_Name"]] += float(row["Counter_Value"])
                        n[row["Counter_Name"]] += 1
    for k in sorted(acc):
        print("%-34s %16.4g per launch (%d launches)" % (k, acc[k] / max(n[k], 1), n[k]))


if __name__ == "__main__":
    main()
