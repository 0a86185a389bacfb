"""Print the top rows of a rocprofv3 kernel_stats.csv (and per-size stats of the fused step kernel from the trace)."""
import csv
import sys

stats = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(stats)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over %d kernels" % (tot / 1e6, len(rows)))
for r in rows[:top]:
    print("%6.2f%% %9.1f ms  calls %6s  avg %9.1f us  %s" % (float(r["Percentage"]), int(r["TotalDurationNs"]) / 1e6,
                                                           r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:100]))
for r in rows:
    if "vqa::" in r["Name"]:
        print("VQA %6.3f%% %9.2f ms  calls %6s  avg %9.2f us  %s" % (float(r["Percentage"]), int(r["TotalDurationNs"]) / 1e6,
                                                                 r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:90]))
