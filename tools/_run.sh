python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 100 --warmup 10 > gpurun_out/r2_bench_a.json 2> gpurun_out/r2_bench_a.err; tail -c 4000 gpurun_out/r2_bench_a.json; tail -3 gpurun_out/r2_bench_a.err
bash tools/prof.sh r02a
python tools/run_configs.py cfg1 cfg2 cfg4 cfg5 > gpurun_out/r2_configs_a.jsonl 2> gpurun_out/r2_configs_a.err; cat gpurun_out/r2_configs_a.jsonl | cut -c1-900; tail -3 gpurun_out/r2_configs_a.err
