#!/bin/bash
export TMPDIR=/tmp; root=$(pwd); o=gpurun_out/s6; mkdir -p $o
for cfg in "0 12" "24 16" "1 1" "24 200"; do set -- $cfg
  echo "=== FDH_DEEP_MIN=$1 FDH_DEEP_STRIP_MIN=$2"
  FDH_DEEP_MIN=$1 FDH_DEEP_STRIP_MIN=$2 FIGDRAW_HIP_LIB=build/libfigdraw_hip_timing.so python tools/wave_timeline.py 1920 1080 2>&1 | tee -a $o/timeline_1080.txt
done
