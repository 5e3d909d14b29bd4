#!/bin/bash
# GPU-box helper: parity suite + one bench line reduced to the per-kernel times.
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['frame']['kernel_ms'])"
