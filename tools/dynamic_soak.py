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
import numpy as np  # noqa: E402

from figdraw_amd import call_stream as CS  # noqa: E402
from figdraw_amd.context import HipContext  # noqa: E402
from figdraw_amd.scenes import make_render_tree_100  # noqa: E402

w, h = [int(v) for v in os.environ.get("SIZE", "1920x1080").split("x")]
NC, T, ROUTE = int(os.environ.get("NC", "4")), int(os.environ.get("THREADS", "1")), int(os.environ.get("ROUTE", "-1"))
COPIES = int(os.environ.get("COPIES", "100"))
bursts = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NS = 8
scenes = [make_render_tree_100(w, h, frame=f, copies=COPIES, full_frame_blur=True) for f in range(NS)]
cs = [s.to_c() for s in scenes]
ref = HipContext(device=0, sync_submit=True)
ref.set_blur_route(0)
want = []
for sc in scenes:
    ref.render_frame(sc, w, h)
    want.append(ref.read_pixels())
ref.close()
P = CS.Player()
ctxs = [HipContext(device=0) for _ in range(NC)]
for c in ctxs:
    c.set_blur_route(ROUTE)
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
print(f"SIZE={w}x{h} NC={NC} THREADS={T} ROUTE={ROUTE}: bad {bad} of {runs} context-bursts")
