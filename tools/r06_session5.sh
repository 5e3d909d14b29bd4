#!/bin/bash
export TMPDIR=/tmp; root=$(pwd); o=gpurun_out/s5; mkdir -p $o
for lib in figdraw_amd/libfigdraw_hip.so build/libfigdraw_hip_noprio.so figdraw_amd/libfigdraw_hip.so build/libfigdraw_hip_noprio.so; do
for sm in 8 16; do
  echo "== $lib FDH_DEEP_STRIP_MIN=$sm"
  FIGDRAW_HIP_LIB=$root/$lib FDH_DEEP_STRIP_MIN=$sm timeout 600 python tools/deep_sweep.py 1920 1080 -- 0 12 24 2>&1 | grep -v "^#" | tee -a $o/deep_1080.txt
done
done
