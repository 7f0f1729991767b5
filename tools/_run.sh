python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 60 --warmup 10 --lean --no-cpu-baseline --groups 1 --repeats 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['phases_ms'], j['parity'])
"
timeout 100 python tools/fuzz_parity.py 40 31 2>&1 | tail -1
