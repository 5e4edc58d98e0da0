"""Group a rocprofv3 kernel trace by (kernel, grid size): python benchmarks/trace_groups.py <kernel_trace.csv> [top]"""
import csv, re, sys
from collections import defaultdict

rows = csv.DictReader(open(sys.argv[1]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
acc = defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel\w*)(<[^>]*>)?", n)
    name = (m.group(0) if m else n)[:48]
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]))
    a = acc[(name, grid)]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in acc.values())
for (name, grid), (c, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%8.1f ms %5.1f%% calls=%6d avg=%8.1fus wgs=%6d  %s" % (us / 1e3, 100 * us / tot, c, us / c, grid, name))
