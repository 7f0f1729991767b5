#!/bin/bash
# usage (on the GPU box, via gpurun): tools/trace_sorted.sh <tag> [args of tools/time_sorted.py]: kernels of the sorted stage by total time
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/tools/time_sorted.py "$@" > $R/gpurun_out/${tag}_trace.log 2>&1 < /dev/null
echo "trace rc=$?"
f=$(find $R/gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    name = r["Name"].replace("void kpl::(anonymous namespace)::", "")[:70]
    print("%-70s calls %5s avg %10.1f us  total %8.2f ms  %5s%%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
find $R/gpurun_out/${tag}_trace -name "*kernel_trace.csv" -delete
