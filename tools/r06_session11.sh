#!/bin/bash
# same-box A/B: round 5's final library (build/libfigdraw_hip_r05g.so, built from commit bc298da) against the tree's
export TMPDIR=/tmp; o=gpurun_out/s11; mkdir -p $o
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "every_kernel_build or direct_launches or deep_strips" 2>&1 | tail -3
python3 tools/ab_kernels.py 3 r05g tree 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee $o/ab_kernels.txt
for i in 1 2; do
for lib in build/libfigdraw_hip_r05g.so figdraw_amd/libfigdraw_hip.so; do
  echo "== $lib" | tee -a $o/cfg_ab.txt
  for c in 2 4 9 10 11; do FIGDRAW_HIP_LIB=$PWD/$lib python3 tools/perf_configs.py $c 2>$o/err_$c.txt | python3 -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d.items(): print('  ', k, 'frame', v['frame_us'], 'bin', v['kernel_us']['bin'], 'composite', v['kernel_us']['composite_all'])
" 2>/dev/null | tee -a $o/cfg_ab.txt || tail -3 $o/err_$c.txt; done
done
done
tail -3 $o/err_10.txt
