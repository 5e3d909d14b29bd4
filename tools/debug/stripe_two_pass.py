#!/usr/bin/env python3
"""GPU box: rows of a stripe against the same rows of the whole frame, two-pass blur routes (fdh_set_blur_route(0)), by scene part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
from figdraw_amd import scene as S

w, h = 1920, 1080
for route in (0, 1):
    for full_blur in (True, False):
        sc = make_render_tree_100(w, h, frame=2, full_frame_blur=full_blur)
        for cull in (0, 1):
            c = HipContext(device=0); c.set_blur_route(route); c.set_cull(cull)
            c.render_frame(sc, w, h); whole = c.read_pixels().copy()
            for y0, y1 in ((0, 136), (272, 408), (944, 1080)):
                c.set_stripe(y0, y1)
                c.render_frame(sc, w, h)
                got = c.read_pixels()[y0:y1]
                d = np.abs(got.astype(int) - whole[y0:y1].astype(int))
                bad = d.any(axis=2)
                ys, xs = np.nonzero(bad)
                print(f"route {route} full_frame_blur {full_blur} cull {cull} stripe {y0}-{y1}: {int(bad.sum())} px differ, max {int(d.max())}" +
                      (f", rows {ys.min() + y0}..{ys.max() + y0} cols {xs.min()}..{xs.max()}" if bad.any() else ""))
            c.set_stripe(0, 0)
