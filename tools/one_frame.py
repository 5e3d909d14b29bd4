#!/usr/bin/env python3
"""GPU-box helper for counter runs: renders the bench frame a few times on one context and exits (FIGDRAW_HIP_LIB picks the build)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figdraw_amd.context import HipContext  # noqa: E402
from figdraw_amd.scenes import make_render_tree_100  # noqa: E402

w, h = int(os.environ.get("W", 3840)), int(os.environ.get("H", 2160))
c = HipContext(device=0)
c.render_frame(make_render_tree_100(w, h, frame=0, full_frame_blur=os.environ.get("BLUR", "1") != "0"), w, h)
c.replay(int(os.environ.get("N", 6)))
c.sync()
