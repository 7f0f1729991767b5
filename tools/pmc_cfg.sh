#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc_cfg.sh <tag> "<counters...>" cfgN [cfgM ...]
# tools/run_configs.py under rocprofv3 --pmc (own pass, no tracing)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$tag
timeout 280 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/run_configs.py "$@" > $R/gpurun_out/pmc_$tag.log 2>&1 < /dev/null || { tail -3 $R/gpurun_out/pmc_$tag.log; exit 1; }
python3 $R/tools/pmc_last.py $R/gpurun_out/pmc_$tag "score_kernel<false>"
