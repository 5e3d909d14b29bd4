#!/bin/bash
# deep strips with seven shading waves per strip (variant library deep7) against the product's three: identity, then the thresholds
export TMPDIR=/tmp
FIGDRAW_HIP_LIB=$PWD/build/libfigdraw_hip_deep7.so timeout 600 python3 -m pytest tests/test_hip_parity.py -q -m gpu -k "deep_strips" 2>&1 < /dev/null | tail -2
for rep in 1 2; do
for lib in figdraw_amd/libfigdraw_hip.so build/libfigdraw_hip_deep7.so; do
  for sm in 16 8; do
    echo "== $(basename $lib .so) FDH_DEEP_STRIP_MIN=$sm"
    for size in "1920 1080" "1280 720"; do
      FIGDRAW_HIP_LIB=$PWD/$lib FDH_DEEP_STRIP_MIN=$sm timeout 300 python3 tools/deep_sweep.py $size -- 0 12 24 2>&1 < /dev/null | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
    done
  done
done
done
