for x in 0 1 2 3; do echo X=$x; KPL_X=$x python bench.py --steps 60 --warmup 10 --lean --no-cpu-baseline --no-parity --groups 1 --repeats 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['phases_ms'])
"; done
