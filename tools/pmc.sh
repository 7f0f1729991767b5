#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc.sh <tag> "<counters...>" [bench args]
# runs bench.py under rocprofv3 --pmc (own pass, no tracing) and prints per-kernel means
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$tag
timeout 150 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --lean "$@" > $R/gpurun_out/pmc_$tag.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$tag.log; exit 1; }
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$tag
