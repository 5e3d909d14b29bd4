#!/usr/bin/env python3
"""Host side alone: fdh_render_frame on FDH_CREATE_RECORD_ONLY contexts (tree walk -> draw records, no device) for the bench
scene and the reference's own benchmark trees, by pool threads and culling.  us per frame (best of 25 batches) + digest.
usage: python3 tools/walk_probe.py > gpurun_out/walk_probe.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from figdraw_amd.context import HipContext, _F4
from figdraw_amd.scenes import make_clip_mask_benchmark, make_non_clip_benchmark, make_render_tree_100


def t(name, sc, w, h, cull, threads, n=300):
    ctx = HipContext(record_only=True)
    ctx.set_cull(cull); ctx.set_walk_threads(threads)
    cs = sc.to_c()
    L = ctx.L
    L.fdh_set_ui_scale(ctx.h, 1.0)
    for _ in range(30): L.fdh_render_frame(ctx.h, cs.byref(), float(w), float(h), 1, _F4(1, 1, 1, 1))
    best = 1e9
    for rep in range(25):
        t0 = time.perf_counter()
        for _ in range(n): L.fdh_render_frame(ctx.h, cs.byref(), float(w), float(h), 1, _F4(1, 1, 1, 1))
        best = min(best, (time.perf_counter() - t0) / n)
    print(f"{name:8s} cull={cull} pool_threads={threads}: {best * 1e6:7.1f} us/frame  digest {ctx.record_digest():016x} forked_groups {ctx.walk_stats()[1]}", flush=True)
    ctx.close()


scs = [("bench", make_render_tree_100(3840, 2160, 0, full_frame_blur=True), 3840, 2160), ("config6", make_non_clip_benchmark(), 1200, 800),
       ("config7", make_clip_mask_benchmark("sub_clip"), 1200, 800), ("config8", make_clip_mask_benchmark("rect_mask"), 1200, 800)]
print("host cores:", os.cpu_count())
for name, sc, w, h in scs:
    for cull in (0, 1):
        for th in (0, 1, 2, 3, 5, 7):
            t(name, sc, w, h, cull, th)
