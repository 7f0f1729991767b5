true
python bench.py --steps 100 --warmup 10 > gpurun_out/r2_bench_b.json 2> gpurun_out/r2_bench_b.err; tail -3 gpurun_out/r2_bench_b.err; python - <<'PY'
import json
j=json.load(open('gpurun_out/r2_bench_b.json'))
print(j['value'], j['ms_per_step'], j['phases_ms']); print(j['single_view']); print(j.get('host_buffer_path')); print(j.get('single_view_cfg1')); print(j['roofline']['valu_busy'], j['roofline']['traffic'], j['roofline']['hbm_counter_frac'], j['roofline']['counters'])
PY
