python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for g in 1 2; do python bench.py --steps 100 --warmup 10 --groups $g --lean --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['phases_ms'], j['parity'])
"; done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/bench.py --lean --steps 50 --warmup 5 --no-cpu-baseline --groups 1 > /dev/null 2>&1; f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -8
