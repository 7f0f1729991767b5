python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python tools/run_cfg3.py --rounds 10 2>&1 | tail -2
