"""Summarise rocprofv3 --pmc CSVs (one directory per pass) per kernel: mean counter value per dispatch.
Usage: pmc_summary.py <pmc_dir> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if filt and filt not in k:
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, vals in sorted(cs.items()):
            print(f"   {c:<28} n={len(vals):<4} mean={sum(vals)/len(vals):.6g}")


if __name__ == "__main__":
    main()
