#!/bin/bash
# GPU-box helper: the GPU suite once per non-default switch (combination).  Every route must pass: the switches choose kernels and
# hand-overs, never pixels.  (End of round 4 this found a stripe of a large blur node running other blur kernels than the whole frame
# on the two-pass route.)  usage: bash tools/suite_off_defaults.sh
for e in "FDH_BLUR_FUSED=0" "FDH_FORCE_KERNEL_PATHS=1" "FDH_FORCE_KERNEL_PATHS=2" "FDH_FORCE_KERNEL_PATHS=3" "FDH_FORCE_BLUR_PATH=1" "FDH_FORCE_BLUR_PATH=2" \
         "FDH_FORCE_BLUR_PATH=3" "FDH_SYNC_SUBMIT=1 FDH_WALK_THREADS=0" "FDH_BIN_SUBGRIDS=0 FDH_VRAM_STAGING=0" "FDH_VRAM_STORE=2 FDH_WALK_AFFINITY=0" "FDH_INK_BOUNDS=0 FDH_FOLD_CLEAR=0" \
         "FDH_BLUR_FUSED=0 FDH_FORCE_KERNEL_PATHS=3 FDH_WALK_THREADS=3" "FDH_FORCE_KERNEL_PATHS=8" "FDH_FORCE_KERNEL_PATHS=19" \
         "FDH_DEEP_MIN=0 FDH_DIRECT=0" "FDH_DEEP_MIN=1 FDH_DEEP_STRIP_MIN=1" "FDH_DEEP_MIN=8 FDH_DEEP_STRIP_MIN=4 FDH_DIRECT=0"; do
  printf "%-70s " "$e"
  env $e timeout 1200 python3 -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
done
