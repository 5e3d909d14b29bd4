#!/bin/bash
run() { lib=$1; shift; for c in "$@"; do FIGDRAW_HIP_LIB=$lib python3 tools/perf_configs.py $c 2>/dev/null | python3 -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d.items():
    if isinstance(v, dict) and 'frame_us' in v: print('  ', k, 'frame', v['frame_us'], 'bin', v['kernel_us']['bin'], 'composite', v['kernel_us']['composite_all'], 'parity', v.get('parity_max_lsb'), v.get('parity_pixels_differing'))
"; done; }
echo "== product"; run $PWD/figdraw_amd/libfigdraw_hip.so 4 9 10 11
echo "== aw4 (atlas build at 4 waves/SIMD, spill-free)"; run $PWD/build/libfigdraw_hip_aw4.so 4 11
echo "== sw3 (all-paths build at 3 waves/SIMD)"; run $PWD/build/libfigdraw_hip_sw3.so 9 10 11
