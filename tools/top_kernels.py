"""Top kernels of a rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv): name, calls, average µs, share.
    python tools/top_kernels.py <kernel_stats.csv> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in rows[:n]:
    name = r["Name"].replace("void ", "").replace("kpl::(anonymous namespace)::", "")
    print("%-64s %6s calls  %10.1f us avg  %5s %%" % (name[:64], r["Calls"], float(r["AverageNs"]) / 1e3, r.get("Percentage", "")[:5]))
