"""Median / p10 / p90 duration per kernel from a rocprofv3 --kernel-trace CSV (averages hide the
split between eager prefill dispatches and graph-replayed decode dispatches of the same kernel):
python benchmarks/prof_medians.py <dir> [top]"""
import collections, csv, glob, re, statistics, sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel\w*)(<[^>]*>)?", n)
    by[(m.group(0) if m else n)[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-70s %7s %9s %9s %9s %10s" % ("kernel", "calls", "p10_us", "median_us", "p90_us", "total_ms"))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:top]:
    v.sort()
    print("%-70s %7d %9.1f %9.1f %9.1f %10.1f" % (k, len(v), v[len(v) // 10], statistics.median(v), v[len(v) * 9 // 10],
                                                  sum(v) / 1e3))
