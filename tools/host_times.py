#!/usr/bin/env python3
"""GPU box: where the calling thread's time goes per frame (fdh_debug_host_times), bench scene, by pool threads; one context,
frames back to back (the submit thread overlaps).  usage: python3 tools/host_times.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from figdraw_amd.context import HipContext, _F4
from figdraw_amd.scenes import make_render_tree_100, make_clip_mask_benchmark

def run(name, sc, w, h, threads, retained=False, n=200):
    ctx = HipContext(device=0)
    ctx.set_walk_threads(threads)
    cs = sc.to_c()
    L = ctx.L
    L.fdh_set_ui_scale(ctx.h, 1.0)
    if retained:
        ctx.scene_retain(sc, w, h)
    acc = {}
    t0 = None
    for i in range(n + 30):
        if i == 30:
            ctx.sync(); t0 = time.perf_counter()
        if retained: L.fdh_scene_render(ctx.h)
        else: L.fdh_render_frame(ctx.h, cs.byref(), float(w), float(h), 1, _F4(1, 1, 1, 1))
        if i >= 30:
            for k, v in ctx.host_times().items(): acc[k] = acc.get(k, 0) + v
    ctx.sync()
    dt = (time.perf_counter() - t0) / n
    st = ctx.frame_stats()
    print(f"{name} threads={threads} retained={retained}: {dt*1e6:.1f} us/frame | " + " ".join(f"{k}={v/n/1000:.1f}" for k, v in acc.items()) + f" | stats record={st.ms_host_record*1e3:.1f} upload={st.ms_host_upload*1e3:.1f} launch={st.ms_host_launch*1e3:.1f}", flush=True)
    ctx.close()

w, h = 3840, 2160
sc = make_render_tree_100(w, h, 0, full_frame_blur=True)
for th in (0, 1, 3, 7):
    run("bench", sc, w, h, th)
run("bench", sc, w, h, 0, retained=True)
sc7 = make_clip_mask_benchmark("sub_clip")
for th in (0, 3, 7):
    run("config7", sc7, 1200, 800, th)
