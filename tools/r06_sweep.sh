# tools/r06_sweep.sh (on the GPU box): sorted / canonical feature stage by radius -> gpurun_out/r06f_sweep.jsonl
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r06f_sweep.jsonl
for r in 3 4 5 6 7 8 9 10 11 12 14 15 16; do
  python3 tools/time_sorted.py rmul=$r 2>/dev/null >> gpurun_out/r06f_sweep.jsonl
  python3 tools/time_sorted.py canonical rmul=$r 2>/dev/null >> gpurun_out/r06f_sweep.jsonl
done
