"""Per-variant instruction counts of profiles/experiments/phase_costs.py from the rocprofv3 counter CSVs of
profiles/run_phase_costs.sh: per wave and sample (a wave renders 64 pixels x 16 taps), second frame of every variant."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
names = ["room_2_lights", "room_1_light", "room_0_lights", "teapot", "teapot_0_lights"]
vals = defaultdict(dict)  # variant -> counter -> per wave-sample
for path in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(path)) if "k_render" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    for r in rows:
        k = ids.index(int(r["Dispatch_Id"]))
        if k % 2 == 1 and k // 2 < len(names):
            vals[names[k // 2]][r["Counter_Name"]] = vals[names[k // 2]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
print(f"{'variant':16s}" + "".join(f"{c.replace('SQ_', ''):>16s}" for c in cols))
for n in names:
    if n in vals:
        print(f"{n:16s}" + "".join(f"{vals[n].get(c, 0) / 129600 / 16:16.1f}" for c in cols))
for log in sorted(glob.glob(os.path.join(root, "sq1.log"))):
    print(open(log).read().strip())
