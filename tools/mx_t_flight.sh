#!/bin/bash
# in-flight value against the blocks-per-wave of the two-pass matrix blur (FDH_MX_T): fewer, longer waves leave LDS for another frame's pass
run() { echo "== $*"; env "$@" bash tools/flight_sweep.sh 4 4; }
run FDH_BLUR_FUSED=0
run FDH_BLUR_FUSED=0 FDH_MX_T=3
run FDH_BLUR_FUSED=0 FDH_MX_T=6
run FDH_BLUR_FUSED=0 FDH_MX_T=8
run FDH_BLUR_FUSED=0 FDH_MX_T=12
