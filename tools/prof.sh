#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof.sh <tag> [bench args]
# 1. bench.py under rocprofv3 --kernel-trace --stats (csv)                      -> gpurun_out/<tag>_trace/
# 2. own passes for --pmc FETCH_SIZE, --pmc WRITE_SIZE, SQ counter sets, TA busy -> gpurun_out/pmc_<tag>_{fetch,write,sq,ta,valu,lds}/
# 2b. tools/valu_ceiling (VALU issue ceilings of this box)                       -> gpurun_out/<tag>_valu_ceiling.json
# 3. the hash of the kernel sources the counters were taken on                  -> gpurun_out/<tag>_sha256.txt
# Back in the build container: python tools/save_profile.py <tag>  (copies the summaries to profiles/ and
# refreshes profiles/counters.json, which bench.py quotes only while that hash matches its own kernels).
# Every step has its own timeout; nothing here reads stdin.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_sha256())" > $R/gpurun_out/${tag}_sha256.txt
rm -rf $R/gpurun_out/${tag}_trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py --lean --steps 40 --warmup 10 --repeats 2 "$@" > $R/gpurun_out/${tag}_trace.log 2>&1 < /dev/null
echo "trace rc=$?"
grep '^{"metric' $R/gpurun_out/${tag}_trace.log | tail -1 > $R/gpurun_out/${tag}_bench.json
f=$(find $R/gpurun_out/${tag}_trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-160 "$f" | head -14
# keep what travels back small: the per-dispatch trace is not needed, the stats are
find $R/gpurun_out/${tag}_trace -name "*kernel_trace.csv" -delete
bash $R/tools/pmc.sh ${tag}_fetch FETCH_SIZE "$@" < /dev/null | cut -c1-200 | grep -A2 "feature_kernel<false, 2>\|forest_kernel<false>"
bash $R/tools/pmc.sh ${tag}_write WRITE_SIZE "$@" < /dev/null | cut -c1-200 | grep -A2 "feature_kernel<false, 2>\|forest_kernel<false>"
bash $R/tools/pmc.sh ${tag}_sq "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "$@" < /dev/null | cut -c1-200 | grep -A9 "feature_kernel<false, 2>\|forest_kernel<false>"
bash $R/tools/pmc.sh ${tag}_ta "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "$@" < /dev/null | cut -c1-200 | grep -A4 "feature_kernel<false, 2>\|forest_kernel<false>"
# instruction classes for the VALU issue model (tools/valu_model.py) and the LDS side of the forest kernel
bash $R/tools/pmc.sh ${tag}_valu "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 GRBM_GUI_ACTIVE" "$@" < /dev/null | cut -c1-200 | grep -A9 "feature_kernel<false, 2>\|forest_kernel<false>"
bash $R/tools/pmc.sh ${tag}_lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "$@" < /dev/null | cut -c1-200 | grep -A8 "feature_kernel<false, 2>\|forest_kernel<false>"
# lane utilisation of the VALU: enabled lanes summed over the VALU instructions against 64 per instruction
bash $R/tools/pmc.sh ${tag}_lanes "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" "$@" < /dev/null | cut -c1-200 | grep -A6 "feature_kernel<false, 2>\|forest_kernel<false>" || echo "lane counters not available on this box"
# the VALU issue ceilings of this box (tools/valu_ceiling.hip)
[ -x $R/tools/valu_ceiling ] && $R/tools/valu_ceiling > $R/gpurun_out/${tag}_valu_ceiling.json && echo "valu ceiling measured"
