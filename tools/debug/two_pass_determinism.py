#!/usr/bin/env python3
"""GPU box: is the two-pass matrix-pipe blur route deterministic?  The same frame rendered N times on one context (and on fresh ones),
whole and as a stripe; prints how many pixels ever differ from the first render and where."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
sc = make_render_tree_100(w, h, frame=2, full_frame_blur=True)
for route in (0, 1):
    for stripe in (None, (272, 408)):
        c = HipContext(device=0); c.set_blur_route(route)
        if stripe: c.set_stripe(*stripe)
        ref = None; ever = None
        for k in range(30):
            c.render_frame(sc, w, h)
            got = c.read_pixels().copy()
            if stripe: got = got[stripe[0]:stripe[1]]
            if ref is None: ref = got; ever = np.zeros(got.shape[:2], bool); continue
            ever |= (got != ref).any(axis=2)
        ys, xs = np.nonzero(ever)
        print(f"route {route} stripe {stripe}: {int(ever.sum())} px ever differ over 30 renders" + (f": {list(zip(ys.tolist(), xs.tolist()))[:8]}" if ever.any() else ""))
