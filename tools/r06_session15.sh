#!/bin/bash
# configs 4 / 10 / 11 (the <2> and <3> builds) against round 5's library on the same box (the C player linked against each library)
export TMPDIR=/tmp; o=gpurun_out/s15; mkdir -p $o; rm -f $o/cfg_ab.txt
for i in 1; do
for lib in figdraw_amd/libfigdraw_hip.so; do
  echo "== $lib" | tee -a $o/cfg_ab.txt
  for c in 10 11 6 7; do FIGDRAW_HIP_LIB=$PWD/$lib timeout 300 python3 tools/perf_configs.py $c 2>$o/err_${c}_$(basename $lib .so).txt < /dev/null | python3 -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d.items(): print('  ', k, 'frame', v['frame_us'], 'bin', v['kernel_us']['bin'], 'composite', v['kernel_us']['composite_all'])
" 2>/dev/null | tee -a $o/cfg_ab.txt || true; done
done
done
tail -3 $o/err_10_libfigdraw_hip_r05g.txt
