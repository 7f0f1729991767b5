for sp in 43 42 25 23 85 83; do echo SPLIT=$sp; KPL_SPLIT=$sp python tools/run_configs.py cfg5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['gpu_Mpts'], j['gpu_ms'], j['phases_ms']['forest_ms'], j['parity'])
"; done
