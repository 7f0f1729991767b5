#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc_cfg.sh <tag> "<counters...>" <kernel name substring> cfgN [cfgM ...]
# tools/run_configs.py under rocprofv3 --pmc (own pass, no tracing); prints per-dispatch means of the matching kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; ctrs=$2; kern=$3; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$tag
timeout 400 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/run_configs.py "$@" > $R/gpurun_out/pmc_$tag.log 2>&1 < /dev/null || { tail -3 $R/gpurun_out/pmc_$tag.log; exit 1; }
python3 $R/tools/pmc_last.py $R/gpurun_out/pmc_$tag "$kern"
find $R/gpurun_out/pmc_$tag -name "*.csv" -size +8M -delete
