"""Prints per-dispatch counter values of k_render launches: python pmc_table.py gpurun_out/pmc_<tag>"""
import collections
import csv
import glob
import sys

for grp in sorted(glob.glob(sys.argv[1] + "/*/")):
    for path in glob.glob(grp + "**/*counter_collection.csv", recursive=True):
        d = collections.OrderedDict()
        for row in csv.DictReader(open(path)):
            if "k_render" not in row["Kernel_Name"]:
                continue
            d.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
        ids = sorted(d)
        names = sorted(d[ids[0]])
        print(grp, names)
        last = None
        for i in ids:
            line = " ".join(f"{d[i][n]:.4g}" for n in names)
            if line != last:
                print(i, line)
            last = line
