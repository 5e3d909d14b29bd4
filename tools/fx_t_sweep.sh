#!/bin/bash
# GPU-box helper: the fused full-frame blur (k_blur_fx) at 4K for several blocks-per-segment T (FDH_FX_T): launch time from bench.py's own events
for t in ${@:-0 4 5 6 7 8 9 10 12}; do
  FDH_FX_T=$t python3 - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
w, h = 3840, 2160
c = HipContext(device=0)
c.set_blur_route(1)
c.render_frame(make_render_tree_100(w, h, frame=0, full_frame_blur=True), w, h)
c.replay(5); c.profile(40)
print("T =", os.environ["FDH_FX_T"], "(0: the launcher's choice)  k_blur_fx", round(1e3 * c.frame_stats().ms_blur_fused, 2), "us")
PY
done
