#!/usr/bin/env python3
"""GPU-box helper for traces: N dynamic frames (fdh_render_frame) of the bench scene on one context, one at a time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figdraw_amd.context import HipContext, _F4
from figdraw_amd.scenes import make_render_tree_100
w, h = int(os.environ.get("W", 3840)), int(os.environ.get("H", 2160))
c = HipContext(device=0)
c.set_walk_threads(int(os.environ.get("TH", -1)))
sc = make_render_tree_100(w, h, frame=0, full_frame_blur=True)
cs = sc.to_c()
c.L.fdh_set_ui_scale(c.h, 1.0)
for i in range(int(os.environ.get("N", 40))):
    c.L.fdh_render_frame(c.h, cs.byref(), float(w), float(h), 1, _F4(1, 1, 1, 1))
c.sync()
