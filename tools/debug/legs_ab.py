#!/usr/bin/env python3
"""GPU-box helper: the bench's `value` leg (fdh_render_frame per frame) against its per-call leg on the same four contexts,
alternating, with the host-side split of each (why do they differ by 5 % when the GPU work is the same?)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from figdraw_amd import call_stream as CS
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
w, h, F, NS, N = 3840, 2160, 4, 8, 400
scenes = [make_render_tree_100(w, h, k, full_frame_blur=True) for k in range(NS)]
cs = [s.to_c() for s in scenes]
ctxs = [HipContext(device=0) for _ in range(F)]
for i, c in enumerate(ctxs): c.render_frame(scenes[i], w, h); c.sync()
rec = HipContext(record_only=True); streams = []
for sc in scenes:
    rec.record_begin(); rec.render_frame(sc, w, h); streams.append(CS.pack(rec.record_calls()))
P = CS.Player()
def leg(name, fn, pool=None):
    for c in ctxs: c.set_walk_threads(-1 if pool is None else pool)
    fn(40)
    for c in ctxs: c.sync(); c.host_times()
    t0 = time.perf_counter(); fn(N)
    for c in ctxs: c.sync()
    dt = (time.perf_counter() - t0) / N
    ht = {}
    for c in ctxs:
        for k, v in c.host_times().items(): ht[k] = ht.get(k, 0) + v
    print(f"{name:28s} {dt*1e6:6.1f} us/frame = {w*h/dt/1e9:6.1f} Gpix/s | host per frame: " + " ".join(f"{k}={v/N/1e3:.1f}" for k, v in ht.items() if v), flush=True)
for r in range(3):
    leg("render_frame, pool", lambda n: P.play_scenes(ctxs, cs, n, w, h))
    leg("render_frame, caller alone", lambda n: P.play_scenes(ctxs, cs, n, w, h), pool=0)
    leg("per call", lambda n: P.play_frames(ctxs, streams, n, w, h))
