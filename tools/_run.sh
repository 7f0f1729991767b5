python bench.py --steps 60 --warmup 10 --lean --no-cpu-baseline --groups 1 --repeats 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['phases_ms'], j['parity'])
"
for g in 2 3 4; do python bench.py --steps 60 --warmup 10 --lean --no-cpu-baseline --groups $g --repeats 3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('groups', j['config']['batches_in_flight'], j['value'], j['ms_per_step'])
"; done
