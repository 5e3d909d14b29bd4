#!/usr/bin/env python3
"""Per-strip draw classification of the bench frame, from the instrumented build (make -C figdraw_amd/csrc stats).
Run on the GPU box:  FIGDRAW_HIP_LIB=build/libfigdraw_hip_stats.so python3 tools/strip_stats.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from figdraw_amd import context as ctx_mod  # noqa: E402
from figdraw_amd.scenes import make_render_tree_100  # noqa: E402

w, h = 3840, 2160
ctx = ctx_mod.HipContext(device=0)
L = ctx_mod.load()
buf = (C.c_ulonglong * 64)()
L.fdh_debug_counters(buf, 1)
ctx.render_frame(make_render_tree_100(w, h, frame=0, full_frame_blur=True), w, h)
ctx.sync()
L.fdh_debug_counters(buf, 1)
c = list(buf)
names = {0: "strip-draws evaluated", 1: "slow path (one pixel slot)", 2: "fast: elliptical", 3: "fast: vertex-colour gradient",
         4: "fast: 3-stop fill", 5: "cls1 (alpha==1 on the strip)", 6: "cls2 (no-op strip)", 7: "cls1 and strip inside quad"}
for i in range(8):
    print(f"{names[i]:34s} {c[i]:10d}")
for m in range(32):
    if c[8 + m]:
        print(f"fast mode {m:2d}: {c[8 + m]:10d}   of which cls1 {c[40 + (m & 15)] if m < 16 else 0:10d}")
for i, n in ((32, "core: stroke no-op"), (33, "core: solid uniform blend"), (34, "core: other (gradient / push / blur)")):
    print(f"{n:34s} {c[i]:10d}")
print("strips:", (w // 32) * (h // 8))
