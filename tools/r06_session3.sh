#!/bin/bash
export TMPDIR=/tmp; root=$(pwd); o=gpurun_out/s3; mkdir -p $o
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "quarter_strips" 2>&1 | tail -15
for lib in figdraw_amd/libfigdraw_hip.so build/libfigdraw_hip_nopf.so build/libfigdraw_hip_prio.so figdraw_amd/libfigdraw_hip.so build/libfigdraw_hip_nopf.so; do
  echo "== $lib"
  FIGDRAW_HIP_LIB=$root/$lib python tools/narrow_sweep.py 1920 1080 -- 0 24 32 48 2>&1 | tee -a $o/narrow_1080.txt
done
for lib in figdraw_amd/libfigdraw_hip.so build/libfigdraw_hip_nopf.so; do
  echo "== $lib"
  FIGDRAW_HIP_LIB=$root/$lib python tools/narrow_sweep.py 3840 2160 1 -- 0 32 2>&1 | tee -a $o/narrow_4k.txt
  FIGDRAW_HIP_LIB=$root/$lib python tools/narrow_sweep.py 1280 720 -- 0 24 32 48 2>&1 | tee -a $o/narrow_720.txt
done
