cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_round6.py tests/test_gpu_standin_sort.py -q -x 2>&1 | grep -E "assert|Error|record|passed|failed" | head -20 > gpurun_out/s32_tests.log
