python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/prof.sh r02c 2>&1 | grep -A14 '"Name","Calls"' | cut -c1-150
python bench.py > gpurun_out/r2_bench_c.json 2> gpurun_out/r2_bench_c.err; tail -2 gpurun_out/r2_bench_c.err; cut -c1-300 gpurun_out/r2_bench_c.json
python tools/run_configs.py cfg1 cfg2 cfg3 cfg4 cfg5 > gpurun_out/r2_configs_c.jsonl 2> gpurun_out/r2_configs_c.err; tail -2 gpurun_out/r2_configs_c.err; cut -c1-400 gpurun_out/r2_configs_c.jsonl
python tools/run_cfg3.py --rounds 10 2>/dev/null | tail -1
