#!/usr/bin/env python3
"""Prints valu_issue.csv as one row per instruction kind: per resident-waves-per-SIMD setting, the cycles
a wave spends per counted instruction and the instructions all waves of a SIMD issue per cycle."""
import csv
import sys
from collections import OrderedDict

rows = list(csv.reader(open(sys.argv[1] if len(sys.argv) > 1 else "valu_issue.csv")))
t = OrderedDict()
for r in rows[1:]:
    t.setdefault(r[0], []).append(r)
ws = [r[1] for r in next(iter(t.values()))]
print("%-56s" % "op", "  ".join("W=%s cyc  i/c " % w for w in ws))
for k, v in t.items():
    print("%-56s" % k[:56], "  ".join("%6.2f %.3f" % (float(r[3]), float(r[4])) for r in v), "  MHz %s..%s" % (v[0][2], v[-1][2]))
