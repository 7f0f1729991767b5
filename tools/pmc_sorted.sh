#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc_sorted.sh <tag> "<counters...>" [args of tools/time_sorted.py]: per-kernel means of the counters
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$tag
timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/time_sorted.py "$@" > $R/gpurun_out/pmc_$tag.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$tag.log; exit 1; }
python3 - $R/gpurun_out/pmc_$tag <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "sorted" in k or "feature_search" in k:
        print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
