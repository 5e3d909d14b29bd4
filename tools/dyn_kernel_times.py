#!/usr/bin/env python3
"""GPU-box helper (run under rocprofv3 --kernel-trace --stats): the bench frame rendered DYNAMICALLY (fdh_render_frame, eight
animation frames in rotation) on one context, N frames -- kernel durations of fresh frames, as opposed to replayed ones."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figdraw_amd import call_stream as CS  # noqa: E402
from figdraw_amd.context import HipContext  # noqa: E402
from figdraw_amd.scenes import make_render_tree_100  # noqa: E402

w, h = 3840, 2160
cs = [make_render_tree_100(w, h, frame=f, full_frame_blur=True).to_c() for f in range(8)]
c = HipContext(device=0)
c.set_blur_route(int(os.environ.get("ROUTE", "0")))
P = CS.Player()
P.play_scenes([c], cs, 50, w, h)
t = P.play_scenes([c], cs, int(os.environ.get("N", 300)), w, h)
print("dynamic, one context: %.1f us per frame" % (t / int(os.environ.get("N", 300)) * 1e6))
