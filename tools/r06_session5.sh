#!/bin/bash
export TMPDIR=/tmp; root=$(pwd); o=gpurun_out/s5b; mkdir -p $o
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "deep_strips or direct_launches or invert" 2>&1 | tail -5
for sm in 6 8 12 16; do
  echo "== FDH_DEEP_STRIP_MIN=$sm"
  FDH_DEEP_STRIP_MIN=$sm timeout 600 python tools/deep_sweep.py 1920 1080 -- 0 8 16 24 2>&1 | grep -v "^#" | tee -a $o/deep_1080.txt
done
for sm in 8 12 16; do
  echo "== FDH_DEEP_STRIP_MIN=$sm (720p)"
  FDH_DEEP_STRIP_MIN=$sm timeout 600 python tools/deep_sweep.py 1280 720 -- 0 12 24 2>&1 | grep -v "^#" | tee -a $o/deep_720.txt
done
