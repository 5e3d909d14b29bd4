#!/bin/bash
# host-side A/B: configs 6 / 7 / 8 through fdh_render_frame (dynamic_us_per_frame), round 5's library against the tree's on one box
export TMPDIR=/tmp
for i in 1 2; do
for lib in build/libfigdraw_hip_r05g.so figdraw_amd/libfigdraw_hip.so; do
  echo "== $lib"
  for c in 6 7 8; do FIGDRAW_HIP_LIB=$PWD/$lib timeout 300 python3 tools/perf_configs.py $c 2>/dev/null < /dev/null | python3 -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d.items(): print('  ', k, 'frame', v['frame_us'], 'dynamic', v.get('dynamic_us_per_frame'), v.get('dynamic_us_per_frame_runs'))
" 2>/dev/null || true; done
done
done
