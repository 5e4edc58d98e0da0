"""Summarise a rocprofv3 --kernel-trace --stats CSV: python benchmarks/prof_summary.py <dir> [top] [json_out]"""
import csv, glob, json, re, sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
f = glob.glob(d + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6))
out = []
for r in rows[:top]:
    n = r["Name"]
    m = re.search(r"(\w+_kernel\w*)(<[^>]*>)?", n)
    name = (m.group(0) if m else n)[:70]
    out.append({"kernel": name, "calls": int(r["Calls"]), "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 2),
                "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "pct": float(r["Percentage"])})
    print("%9.1f ms %5.1f%% calls=%6s avg=%8.1fus  %s" % (out[-1]["total_ms"], out[-1]["pct"], r["Calls"], out[-1]["avg_us"], name))
if len(sys.argv) > 3:
    json.dump({"total_kernel_ms": round(tot / 1e6, 1), "top_kernels": out}, open(sys.argv[3], "w"), indent=1)
