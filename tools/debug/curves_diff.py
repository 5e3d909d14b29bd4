#!/usr/bin/env python3
"""GPU-box helper: the curves scene by primitive kind (0 quadratic bezier, 1 cubic -> adaptive quadratic spans, 2 line, 3 arc with
joins): where the HIP frame differs from the oracle's."""
import os, sys
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_curves_scene
from oracle import oracle as O
w, h, n, seed = 1280, 720, 300, 7
for kind in (0, 1, 2, 3):
    sc = make_curves_scene(w, h, n=n, seed=seed, only_kind=kind)
    ctx = HipContext(device=0); ctx.render_frame(sc, w, h); got = ctx.read_pixels().astype(int); ctx.close()
    o = O.Oracle(threads=8); o.render_frame(sc, w, h); want = o.read_pixels().astype(int)
    d = np.abs(got - want).max(axis=2)
    ys, xs = np.nonzero(d > 1)
    print("kind", kind, "n>0", int((d > 0).sum()), "n>1", len(ys), "max", d.max(), [(int(x), int(y), int(d[y, x])) for y, x in list(zip(ys, xs))[:8]])
