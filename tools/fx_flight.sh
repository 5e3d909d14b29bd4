#!/bin/bash
# in-flight value with the fused blur forced and its LDS request padded (FDH_FX_LDS, KB per wave)
run() { echo "== $*"; env "$@" bash tools/flight_sweep.sh 4 4; }
run A=0
run FDH_BLUR_FUSED=1
run FDH_BLUR_FUSED=1 FDH_FX_LDS=30
run FDH_BLUR_FUSED=1 FDH_FX_LDS=38
run FDH_BLUR_FUSED=1 FDH_FX_LDS=50
