#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof.sh <tag> [bench args]
# 1. bench.py under rocprofv3 --kernel-trace --stats (csv)   -> gpurun_out/<tag>_trace/
# 2. own passes for --pmc FETCH_SIZE and --pmc WRITE_SIZE     -> gpurun_out/pmc_<tag>_fetch|write/
# every step has its own timeout; nothing here reads stdin
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_trace
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py --lean --steps 100 --warmup 10 "$@" > $R/gpurun_out/${tag}_trace.log 2>&1 < /dev/null
echo "trace rc=$?"
grep '^{"metric' $R/gpurun_out/${tag}_trace.log | tail -1 > $R/gpurun_out/${tag}_bench.json
f=$(find $R/gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-160 "$f"
for c in FETCH_SIZE WRITE_SIZE; do
  l=$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  bash $R/tools/pmc.sh ${tag}_$l $c "$@" < /dev/null | cut -c1-300
done
