#!/usr/bin/env python3
"""GPU-box soak of the DYNAMIC path with frames in flight: NC contexts take bursts of fdh_render_frame calls from the C player (one
calling thread, or one thread per context), eight different frames in rotation, full-frame blur on either route; after every burst
each context must hold exactly the frame a lone synchronous context renders for the scene it drew last.
env: SIZE=1920x1080 NC=4 THREADS=1 ROUTE=-1|0|1 COPIES=100    usage: dynamic_soak.py [bursts]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from figdraw_amd import call_stream as CS  # noqa: E402
from figdraw_amd.context import HipContext  # noqa: E402
from figdraw_amd.scenes import make_render_tree_100  # noqa: E402

w, h = [int(v) for v in os.environ.get("SIZE", "1920x1080").split("x")]
NC, T, ROUTE = int(os.environ.get("NC", "4")), int(os.environ.get("THREADS", "1")), int(os.environ.get("ROUTE", "-1"))
COPIES = int(os.environ.get("COPIES", "100"))
bursts = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NS = 8
MIX = os.environ.get("MIX")  # MIX=1: the eight frames are different KINDS of frame -- bench tree, rotated tree, curves, curves under rotation,
# rotated glyph rows, the reference's clip benchmark -- so that contexts in flight run different compositor builds (<4>, <8>, <3>, <0>)
# and both builds of the bin kernel beside each other
if MIX:
    from figdraw_amd.scenes import load_glyph_fixture, make_clip_mask_benchmark, make_curves_scene, make_glyph_scene, make_rotated_tree
    imgs = load_glyph_fixture(os.path.join(ROOT, "tests", "golden", "glyphs_ubuntu20.npz"))
    scenes = [make_render_tree_100(w, h, frame=0, copies=COPIES, full_frame_blur=True), make_rotated_tree(w, h, 1, copies=COPIES),
              make_curves_scene(w, h, n=300, seed=5), make_curves_scene(w, h, n=200, seed=9, rotation=11.0),
              make_glyph_scene(w, h, imgs, cols=40, rows=30, rotation=3.0), make_clip_mask_benchmark("sub_clip", w, h, rows=60, cols=6),
              make_render_tree_100(w, h, frame=5, copies=COPIES, full_frame_blur=False), make_rotated_tree(w, h, 4, copies=COPIES // 2)]
else:
    scenes = [make_render_tree_100(w, h, frame=f, copies=COPIES, full_frame_blur=True) for f in range(NS)]
cs = [s.to_c() for s in scenes]
ref = HipContext(device=0, sync_submit=True)
ref.set_blur_route(0)
if MIX:
    import ref_scenes as RS  # noqa: E402
    used = RS.used_images(scenes[4], imgs)
    for k in sorted(used): ref.put_image(k, used[k])
want = []
for sc in scenes:
    ref.render_frame(sc, w, h)
    want.append(ref.read_pixels())
ref.close()
P = CS.Player()
ctxs = [HipContext(device=0) for _ in range(NC)]
for c in ctxs:
    c.set_blur_route(ROUTE)
    if MIX:
        for k in sorted(used): c.put_image(k, used[k])
rnd = random.Random(7)
bad = runs = 0
for b in range(bursts):
    frames = rnd.randrange(NC, 6 * NC + 3)
    P.play_scenes(ctxs, cs, frames, w, h, threads=T)
    for i, c in enumerate(ctxs):
        k_last = max(k for k in range(frames) if k % NC == i)
        runs += 1
        got = c.read_pixels()
        if not np.array_equal(got, want[k_last % NS]):
            bad += 1
            d = (got != want[k_last % NS]).any(axis=2)
            ys, xs = np.nonzero(d)
            if bad <= 5:
                print(f"burst {b} ctx {i} frames {frames}: {len(ys)} px differ, bbox x {xs.min()}..{xs.max()} y {ys.min()}..{ys.max()}", flush=True)
for c in ctxs:
    c.close()
print(f"SIZE={w}x{h} NC={NC} THREADS={T} ROUTE={ROUTE}{' MIX' if MIX else ''}: bad {bad} of {runs} context-bursts")
