#!/bin/bash
# the final set: profiles (r06_e), then the forced-build rows of the switch suite again (their deep-strip assertion was the test's)
bash tools/r06_session10.sh
for e in "FDH_FORCE_KERNEL_PATHS=1" "FDH_FORCE_KERNEL_PATHS=2" "FDH_FORCE_KERNEL_PATHS=3" "FDH_BLUR_FUSED=0 FDH_FORCE_KERNEL_PATHS=3 FDH_WALK_THREADS=3" "FDH_FORCE_KERNEL_PATHS=8" "FDH_FORCE_KERNEL_PATHS=19"; do
  printf "%-70s " "$e"
  env $e timeout 1200 python3 -m pytest tests -q -m gpu 2>&1 < /dev/null | grep -E "passed|failed" | tail -1
done | tee gpurun_out/r06_suite_forced_rows.txt
