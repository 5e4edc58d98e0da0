"""GPU idle time between decode steps, from a rocprofv3 --kernel-trace CSV: gaps of more than
20 us between the end of one kernel and the start of the next (python step_gaps.py DIR)."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(f))]
rows.sort()
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = (s1 - e0) / 1e3
    if g > 20:
        gaps.append((g, n0, n1))
g = np.array([x[0] for x in gaps])
mid = g[(g > 20) & (g < 2000)]
print("gaps > 20us:", len(g), "median", round(float(np.median(mid)), 1), "us  p90", round(float(np.percentile(mid, 90)), 1))
busy = sum(e - s for s, e, _ in rows) / 1e6
print("kernel busy ms", round(busy, 1), "span ms", round((rows[-1][1] - rows[0][0]) / 1e6, 1))
from collections import Counter
print(Counter((a, b) for _, a, b in gaps).most_common(4))
