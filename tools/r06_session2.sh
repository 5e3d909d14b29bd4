#!/bin/bash
export TMPDIR=/tmp; root=$(pwd); o=gpurun_out/s2; mkdir -p $o

FIGDRAW_HIP_LIB=build/libfigdraw_hip_timing.so python tools/wave_timeline.py 1920 1080 > $o/timeline_1080.txt 2>&1; cat $o/timeline_1080.txt
FIGDRAW_HIP_LIB=build/libfigdraw_hip_timing.so python tools/wave_timeline.py 3840 2160 1 > $o/timeline_4k.txt 2>&1; cat $o/timeline_4k.txt
