cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_sorted.py tests/test_gpu_default_point.py tests/test_gpu_round6.py tests/test_gpu_standin_sort.py -q -x 2>&1 | tail -12 > gpurun_out/s32_tests.log
for r in 3 4 5 6 7 8 10; do
  python3 tools/time_sorted.py rmul=$r 2>/dev/null >> gpurun_out/s32_sorted.jsonl
done
