#!/usr/bin/env python3
"""Per-frame cost of the DYNAMIC path (fdh_render_frame: C++ tree walk + record upload + kernels every frame) next to
the replay path bench.py times (records resident).  The scene is marshalled to the C structs once."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from figdraw_amd import context as C
from figdraw_amd.scenes import make_render_tree_100

w, h = 3840, 2160
ctx = C.HipContext(device=0)
scenes = [make_render_tree_100(w, h, frame=f, full_frame_blur=True).to_c() for f in range(4)]
col = C._F4(1.0, 1.0, 1.0, 1.0)
ctx.W, ctx.H = w, h
for cs in scenes:
    ctx._ck(ctx.L.fdh_render_frame(ctx.h, cs.byref(), float(w), float(h), 1, col))
ctx.sync()
n = 200
t = time.perf_counter()
for i in range(n):
    ctx._ck(ctx.L.fdh_render_frame(ctx.h, scenes[i & 3].byref(), float(w), float(h), 1, col))
t_host = time.perf_counter() - t
ctx.sync()
t_all = time.perf_counter() - t
st = ctx.frame_stats()
print(f"host per frame: record {st.ms_host_record * 1e3:.1f} us, build + copies {st.ms_host_upload * 1e3:.1f} us, launches {st.ms_host_launch * 1e3:.1f} us")
ctx.replay(10); ctx.sync()
t = time.perf_counter(); ctx.replay(n); ctx.sync(); t_rep = time.perf_counter() - t
print(f"dynamic: {t_all / n * 1e6:.1f} us/frame ({w * h * n / t_all / 1e6:.0f} Mpix/s), host-side submit {t_host / n * 1e6:.1f} us/frame; "
      f"replay: {t_rep / n * 1e6:.1f} us/frame")
