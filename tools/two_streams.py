#!/usr/bin/env python3
"""Throughput of K independent contexts (own stream, own surfaces) replaying the bench frame concurrently on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
w, h, n = 3840, 2160, 200
for k in (1, 2, 3):
    ctxs = [HipContext(device=0) for _ in range(k)]
    for i, c in enumerate(ctxs):
        c.render_frame(make_render_tree_100(w, h, frame=i, full_frame_blur=True), w, h)
        c.replay(5)
    for c in ctxs: c.sync()
    t = time.perf_counter()
    for r in range(n // 10):  # interleave the enqueues so every stream always has work queued
        for c in ctxs: c.replay_async(10)
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t
    print(f"{k} context(s): {dt / (n * k) * 1e6:.1f} us per frame, {w * h * n * k / dt / 1e6:.0f} Mpix/s")
    for c in ctxs: c.close()
