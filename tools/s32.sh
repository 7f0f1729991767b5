cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_sorted.py tests/test_gpu_default_point.py tests/test_gpu_round6.py -q -x 2>&1 | tail -2 > gpurun_out/s32_tests.log
for r in 12 14 16; do
  python3 tools/time_sorted.py rmul=$r 2>/dev/null >> gpurun_out/s32_sorted.jsonl
  KPL_LIB_PATH=$PWD/build/variants/libkpl_old.so python3 tools/time_sorted.py rmul=$r 2>/dev/null >> gpurun_out/s32_sorted.jsonl
done
