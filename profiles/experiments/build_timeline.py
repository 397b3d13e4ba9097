"""python profiles/experiments/build_timeline.py <dir with *_kernel_trace.csv>: the kernels of the first tree build, start offset and duration (ns)."""
import csv, glob, sys
t = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
b = [i for i, n in enumerate(names) if "k_sort_keys" in n][0]
e = [i for i, n in enumerate(names) if "k_octant_variant0" in n][0]
t0 = int(rows[b]["Start_Timestamp"])
for r in rows[b:e + 1]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:28]
    print(f'{n:30s} start {int(r["Start_Timestamp"]) - t0:9d}  dur {int(r["End_Timestamp"]) - int(r["Start_Timestamp"]):8d}  grid {r.get("Grid_Size_X", "?")} wg {r.get("Workgroup_Size_X", "?")}')
