cd $GRAFT_REPO_ROOT
python3 tools/first_call.py > gpurun_out/r06_base_first_call.jsonl 2>&1
for r in 6 8 10 12; do
  python3 tools/time_sorted.py rmul=$r >> gpurun_out/r06_base_sorted.jsonl 2>&1
  python3 tools/time_sorted.py canonical rmul=$r >> gpurun_out/r06_base_sorted.jsonl 2>&1
done
python3 tools/run_configs.py cfg0 > gpurun_out/r06_base_cfg0.jsonl 2>&1
