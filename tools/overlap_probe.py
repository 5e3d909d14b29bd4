#!/usr/bin/env python3
"""GPU-box probe: do the compositor (VALU-bound) and the full-frame blur (matrix pipe + memory) overlap when they come from different
contexts?  Four contexts render blur-free bench frames, two render frames that are a background and a full-frame blur node; each
group alone, then both together, one host thread per context.  Together ~ max(alone): they overlap; ~ sum: they take turns."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figdraw_amd import call_stream as CS  # noqa: E402
from figdraw_amd.context import HipContext  # noqa: E402
from figdraw_amd.scene import Fig, FigKind, RenderList, Renders  # noqa: E402
from figdraw_amd.scenes import make_render_tree_100, rect, rgba  # noqa: E402

w, h = 3840, 2160
N = 60  # frames per context
P = CS.Player()


def comp_scene(f):
    sc = make_render_tree_100(w, h, frame=f, full_frame_blur=False)
    for nd in sc.layers[0].nodes:
        if nd.kind == FigKind.nkBackdropBlur:
            nd.kind = FigKind.nkFrame
    return sc


def blur_scene(f):
    lst = RenderList()
    lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, w, h), fill=rgba(200, 210 + f, 230, 255)))
    lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(300 + 40 * f, 200, 900, 700), fill=rgba(30, 90, 160, 255), corners=[20] * 4))
    lst.addRoot(Fig(kind=FigKind.nkBackdropBlur, screenBox=rect(0, 0, w, h), blur=18.0))
    out = Renders()
    out.setLayer(0, lst)
    return out


def run(scenes, label, route=-1):
    n = len(scenes)
    ctxs = [HipContext(device=0) for _ in range(n)]
    for c in ctxs:
        c.set_blur_route(route)
    cs = [s.to_c() for s in scenes]
    for c, s in zip(ctxs, scenes):
        c.render_frame(s, w, h)
        c.sync()
    ts = []
    for rep in range(5):
        P.play_scenes(ctxs, cs, 10 * n, w, h, threads=n)
        ts.append(P.play_scenes(ctxs, cs, N * n, w, h, threads=n))
    ts.sort()
    st = [c.frame_stats() for c in ctxs]
    print(f"{label:44s} {n} contexts x {N} frames: {ts[2] * 1e3:7.2f} ms  = {ts[2] / N * 1e6:6.1f} us per round of frames", flush=True)
    for c in ctxs:
        c.close()
    return ts[2]


comp = [comp_scene(f) for f in range(4)]
blur = [blur_scene(f) for f in range(2)]
for route in (0, 1):
    a = run(comp, "compositor frames alone", route)
    b = run(blur, f"blur frames alone ({'fused kernel' if route else 'two passes'})", route)
    ab = run(comp + blur, "both together", route)
    print(f"   together / max(alone) = {ab / max(a, b):.2f}   together / sum(alone) = {ab / (a + b):.2f}", flush=True)
