#!/bin/bash
# the GPU suite on the 16 x 24 small-blur tile; then configs 10 / 11 (the <3> build) against round 5's library on the same box
export TMPDIR=/tmp; o=gpurun_out/s15; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 < /dev/null | tail -4 | tee $o/suite.txt
for i in 1 2; do
for lib in build/libfigdraw_hip_r05g.so figdraw_amd/libfigdraw_hip.so; do
  echo "== $lib" | tee -a $o/cfg_ab.txt
  for c in 10 11; do FIGDRAW_HIP_LIB=$PWD/$lib timeout 300 python3 tools/perf_configs.py $c 2>$o/err_${c}_$(basename $lib .so).txt < /dev/null | python3 -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d.items(): print('  ', k, 'frame', v['frame_us'], 'bin', v['kernel_us']['bin'], 'composite', v['kernel_us']['composite_all'])
" 2>/dev/null | tee -a $o/cfg_ab.txt || true; done
done
done
tail -4 $o/err_10_libfigdraw_hip_r05g.txt
