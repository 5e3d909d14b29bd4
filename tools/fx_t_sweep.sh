#!/bin/bash
# GPU-box helper: the fused full-frame blur (k_blur_fx) at 4K for several blocks-per-segment T.  T is the launcher's choice in the
# product; a fixed T is an experiment build: for t in 4 5 6; do make -C figdraw_amd/csrc variant NAME=fxt$t DEFS=-DFDH_FX_T=$t; done
# usage: bash tools/fx_t_sweep.sh [t ...]   (0 = the product library)
for t in ${@:-0 4 5 6}; do
  lib=$PWD/figdraw_amd/libfigdraw_hip.so; [ "$t" != 0 ] && lib=$PWD/build/libfigdraw_hip_fxt$t.so
  [ -f $lib ] || { echo "T = $t: $lib not built"; continue; }
  FIGDRAW_HIP_LIB=$lib T=$t python3 - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
w, h = 3840, 2160
c = HipContext(device=0)
c.set_blur_route(1)
c.render_frame(make_render_tree_100(w, h, frame=0, full_frame_blur=True), w, h)
c.replay(5); c.profile(40)
print("T =", os.environ["T"], "(0: the launcher's choice)  k_blur_fx", round(1e3 * c.frame_stats().ms_blur_fused, 2), "us")
PY
done
