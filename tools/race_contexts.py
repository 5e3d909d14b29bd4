#!/usr/bin/env python3
"""GPU-box helper: NC contexts (env NC, default 3) replay different frames with a full-frame blur concurrently, many times; every
context must end up with the frame it renders alone.  usage: python3 tools/race_contexts.py [iterations]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
w, h = [int(v) for v in os.environ.get('SIZE', '1280x720').split('x')]
COPIES = int(os.environ.get('COPIES', '40'))
NC = int(os.environ.get('NC', '3'))
ONLY1 = os.environ.get('ONLY1')
RADII = [float(v) for v in os.environ.get('RADII', '18').split(',')]  # per context, cycled: different radii = different kernel instantiations
scenes = [make_render_tree_100(w, h, frame=f, copies=COPIES, full_frame_blur=(not ONLY1 or f == 1), full_frame_blur_radius=RADII[f % len(RADII)]) for f in range(NC)]
hip = HipContext(device=0)
alone = []
for sc in scenes:
    hip.render_frame(sc, w, h); alone.append(hip.read_pixels())
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    ctxs = [HipContext(device=0) for _ in scenes]
    for c, sc in zip(ctxs, scenes): c.render_frame(sc, w, h)
    for _ in range(6):
        for c in ctxs: c.replay_async(3)
    for i, (c, want) in enumerate(zip(ctxs, alone)):
        c.sync(); got = c.read_pixels()
        if not np.array_equal(got, want):
            bad += 1
            d = np.abs(got.astype(int) - want.astype(int)).max(axis=2); ys, xs = np.nonzero(d)
            if os.environ.get('SAMPLES') and bad <= 3:
                for k in range(0, len(ys), max(1, len(ys) // 6)): print("   px", xs[k], ys[k], "got", got[ys[k], xs[k]], "want", want[ys[k], xs[k]])
            if bad < 100: print("iter", it, "ctx", i, "differing", len(ys), "max", d.max(), "bbox x", xs.min(), xs.max(), "y", ys.min(), ys.max(),
                  "\n", (got.astype(int) - want.astype(int))[ys.min():ys.max() + 1:4, xs.min():xs.max() + 1:4, 0] if len(ys) < 3000 and bad <= 2 else "",
                  "x%32", np.unique(xs % 32)[:8], "y%32", np.unique(ys % 32)[:8], "blocks", sorted(set(zip((xs // 32).tolist(), (ys // 32).tolist())))[:6])
        c.close()
print("bad", bad, "of", NC, "contexts")
