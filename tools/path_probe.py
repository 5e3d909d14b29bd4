#!/usr/bin/env python3
"""GPU-box probe of the DYNAMIC paths (tree -> pixels every frame): fdh_render_frame and the per-call stream, from C
(tools/call_player.c), with 1..4 contexts in flight, submit thread on and off.  Prints one line per configuration."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from figdraw_amd import call_stream as CS  # noqa: E402
from figdraw_amd.context import HipContext  # noqa: E402
from figdraw_amd.scenes import make_render_tree_100  # noqa: E402

print("cpus:", os.cpu_count(), "affinity:", len(os.sched_getaffinity(0)), flush=True)
w, h = int(os.environ.get("W", 3840)), int(os.environ.get("H", 2160))
N = int(os.environ.get("N", 400))
scenes = [make_render_tree_100(w, h, frame=f, full_frame_blur=True) for f in range(8)]
cs = [s.to_c() for s in scenes]
rec = HipContext(record_only=True)
streams = []
for s in scenes:
    rec.record_begin()
    rec.render_frame(s, w, h)
    streams.append(CS.pack(rec.record_calls()))
P = CS.Player()
for sync_submit in (False, True):
    for F in (1, 2, 4):
        ctxs = [HipContext(device=0, sync_submit=sync_submit) for _ in range(F)]
        for c in ctxs:
            c.render_frame(scenes[0], w, h)
            c.sync()
        for kind in ("scenes", "calls"):
            best = []
            for rep in range(5):
                if kind == "scenes":
                    P.play_scenes(ctxs, cs, 40, w, h)
                    t = P.play_scenes(ctxs, cs, N, w, h)
                else:
                    P.play_frames(ctxs, streams, 40, w, h)
                    t = P.play_frames(ctxs, streams, N, w, h)
                best.append(t / N * 1e6)
            st = ctxs[0].frame_stats()
            print(f"submit={'sync ' if sync_submit else 'async'} F={F} {kind:6s}: us/frame median {np.median(best):7.1f} min {min(best):7.1f}  -> {w * h / np.median(best) / 1e3:7.1f} Gpix/s"
                  f"   host record {1e3 * st.ms_host_record:5.1f} prep {1e3 * st.ms_host_upload:5.1f} launch {1e3 * st.ms_host_launch:5.1f} us", flush=True)
        for c in ctxs:
            c.close()
