#!/bin/bash
# GPU-box helper: tools/perf_configs.py for some configs on the product library and on variant libraries (build/libfigdraw_hip_<name>.so), same box.
# usage: bash tools/cfg_ab.sh "<configs>" name...     e.g.  bash tools/cfg_ab.sh "4 9 10 11" r05b nnanall
run() { lib=$1; shift; for c in "$@"; do FIGDRAW_HIP_LIB=$lib python3 tools/perf_configs.py $c 2>/dev/null | python3 -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d.items():
    if isinstance(v, dict) and 'frame_us' in v: print('  ', k, 'frame', v['frame_us'], 'bin', v['kernel_us']['bin'], 'composite', v['kernel_us']['composite_all'], 'parity', v.get('parity_max_lsb'), v.get('parity_pixels_differing'))
"; done; }
cfgs=${1:-"4 9 10 11"}; shift
echo "== product"; run $PWD/figdraw_amd/libfigdraw_hip.so $cfgs
for n in "$@"; do echo "== $n"; run $PWD/build/libfigdraw_hip_$n.so $cfgs; done
