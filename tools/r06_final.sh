# tools/r06_final.sh (on the GPU box): the measurements the round-6 documents quote, in one go -> gpurun_out/r06f_*
cd $GRAFT_REPO_ROOT
python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r06f_tests.log
bash tools/r06_sweep.sh
python3 tools/run_configs.py cfg0 cfg1 cfg2 cfg3 cfg4 cfg5 cfg0b > gpurun_out/r06f_run_configs.jsonl 2> gpurun_out/r06f_run_configs.err
python3 tools/first_call.py > gpurun_out/r06f_first_call.jsonl 2>&1
bash tools/prof.sh r06z > gpurun_out/r06f_prof.log 2>&1
for k in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric' > gpurun_out/r06f_bench_default_$k.json; done
python3 bench.py --gpus 1 --steps 200 --warmup 20 2>/dev/null | grep '^{"metric' > gpurun_out/r06f_bench_200.json
