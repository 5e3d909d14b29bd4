#!/bin/bash
export TMPDIR=/tmp; root=$(pwd); o=gpurun_out/s4; mkdir -p $o
for lib in build/libfigdraw_hip_nopf.so build/libfigdraw_hip_sched_max-ilp.so build/libfigdraw_hip_sched_max-memory-clause.so build/libfigdraw_hip_ilp_w5.so build/libfigdraw_hip_ilp_w4.so build/libfigdraw_hip_w4.so build/libfigdraw_hip_nopf.so; do
  echo "== $lib"
  FIGDRAW_HIP_LIB=$root/$lib python tools/narrow_sweep.py 1920 1080 -- 0 2>&1 | tee -a $o/sched_1080.txt
  FIGDRAW_HIP_LIB=$root/$lib python tools/narrow_sweep.py 3840 2160 -- 0 2>&1 | tee -a $o/sched_4k.txt
done
