#!/bin/bash
# usage (on the GPU box, via gpurun): tools/timeline_bench.sh <tag> [bench args]: bench.py --lean under rocprofv3 --kernel-trace;
# from the start / end stamps of every dispatch of the timed steps: how long the feature kernels run alone, overlap each other,
# and how much of a step no feature kernel is running at all (-> gpurun_out/<tag>_timeline.json)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py --lean --steps 40 --warmup 10 --repeats 2 --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_trace.log 2>&1 < /dev/null
echo "trace rc=$?"
f=$(find $R/gpurun_out/${tag}_trace -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 - "$f" $R/gpurun_out/${tag}_timeline.json <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    short = name.replace("void kpl::(anonymous namespace)::", "").replace("kpl::(anonymous namespace)::", "").split("(")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
ev.sort()
feat = [e for e in ev if e[2].startswith("feature_kernel<false")]
# the timed region: the last 80 feature launches (2 repeats x 40 steps); take the middle 60 of them
feat = feat[-80:][10:70]
t0, t1 = feat[0][0], feat[-1][1]
win = [e for e in ev if e[1] > t0 and e[0] < t1]
def covered(intervals, depth):
    pts = []
    for a, b, _ in intervals:
        pts.append((max(a, t0), 1)); pts.append((min(b, t1), -1))
    pts.sort()
    cur, last, tot = 0, t0, 0
    for t, d in pts:
        if cur >= depth: tot += t - last
        cur += d; last = t
    return tot
span = t1 - t0
f_only = [e for e in win if e[2].startswith("feature_kernel")]
forest = [e for e in win if e[2].startswith("forest")]
other = [e for e in win if not e[2].startswith("feature_kernel") and not e[2].startswith("forest")]
out = {"steps": len(feat), "ms_per_step": span / len(feat) / 1e6,
       "feature_running_frac": covered(f_only, 1) / span, "two_feature_kernels_frac": covered(f_only, 2) / span,
       "forest_running_frac": covered(forest, 1) / span, "any_kernel_frac": covered(win, 1) / span,
       "feature_or_forest_frac": covered(f_only + forest, 1) / span,
       "other_kernels_running_frac": covered(other, 1) / span,
       "feature_avg_ms": sum(b - a for a, b, _ in feat) / len(feat) / 1e6}
print(json.dumps(out))
# the dispatches of two steps in the middle of the timed region, in start order: offset, duration (us), kernel
mid = feat[len(feat) // 2][0]
seq = [e for e in ev if e[0] >= mid and e[0] < mid + int(2.2 * span / len(feat))]
with open(sys.argv[2].replace(".json", "_sequence.txt"), "w") as f:
    for a, b2, name in seq:
        f.write("%9.1f %8.1f  %s\n" % ((a - mid) / 1e3, (b2 - a) / 1e3, name[:60]))
open(sys.argv[2], "w").write(json.dumps(out) + "\n")
PY
find $R/gpurun_out/${tag}_trace -name "*kernel_trace.csv" -delete
