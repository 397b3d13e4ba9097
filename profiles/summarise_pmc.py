"""Sums rocprofv3 --pmc counter CSVs per kernel: python profiles/summarise_pmc.py gpurun_out/pmc_r01 [kernel-substring]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "k_render<1, false>"
    out = {}
    for path in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
        sums, calls = defaultdict(float), defaultdict(set)
        for row in csv.DictReader(open(path)):
            if want not in row["Kernel_Name"]:
                continue
            sums[row["Counter_Name"]] += float(row["Counter_Value"])
            calls[row["Counter_Name"]].add(row["Dispatch_Id"])
        for k, v in sums.items():
            out[k] = {"per_launch": v / max(len(calls[k]), 1), "launches": len(calls[k])}
    print(json.dumps({"kernel": want, "counters": out}, indent=1))


if __name__ == "__main__":
    main()
