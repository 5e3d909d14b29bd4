#!/usr/bin/env python3
"""GPU-box probe: how well the frame's parts overlap between contexts.  Three variants of the bench frame -- as it is, without the
full-frame blur node, without any blur node -- each through fdh_render_frame from C with 1 and 4 contexts in flight."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figdraw_amd import call_stream as CS  # noqa: E402
from figdraw_amd.context import HipContext  # noqa: E402
from figdraw_amd.scene import FigKind  # noqa: E402
from figdraw_amd.scenes import make_render_tree_100  # noqa: E402

w, h = 3840, 2160
N = 400
ROUTE = int(os.environ.get("ROUTE", "-1"))  # fdh_set_blur_route: -1 per frame, 0 two passes, 1 fused
P = CS.Player()


def variants():
    full = [make_render_tree_100(w, h, frame=f, full_frame_blur=True) for f in range(8)]
    small = [make_render_tree_100(w, h, frame=f, full_frame_blur=False) for f in range(8)]
    none = [make_render_tree_100(w, h, frame=f, full_frame_blur=False) for f in range(8)]
    for sc in none:
        for nd in sc.layers[0].nodes:
            if nd.kind == FigKind.nkBackdropBlur:
                nd.kind = FigKind.nkFrame
    return (("bench frame", full), ("no full-frame blur", small), ("no blur at all", none))


for name, scenes in variants():
    cs = [s.to_c() for s in scenes]
    for F, T in ((1, 1), (4, 1), (4, 2), (4, 4), (8, 4), (8, 8)):
        ctxs = [HipContext(device=0) for _ in range(F)]
        for c in ctxs:
            c.set_blur_route(ROUTE)
            c.render_frame(scenes[0], w, h)
            c.sync()
        best = []
        for rep in range(5):
            P.play_scenes(ctxs, cs, 40, w, h, threads=T)
            best.append(P.play_scenes(ctxs, cs, N, w, h, threads=T) / N * 1e6)
        best.sort()
        st = ctxs[0].frame_stats()
        print(f"{name:20s} route={ROUTE} F={F} host threads={T}: {best[2]:6.1f} us/frame (min {best[0]:.1f})  phases {st.n_phases} blurs {st.n_blurs}", flush=True)
        for c in ctxs:
            c.close()
