#!/usr/bin/env python3
"""GPU-box helper: where the dynamic path's frame time goes -- time inside the fdh_render_frame calls vs the final wait."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from figdraw_amd import context as C_
from figdraw_amd.scenes import make_render_tree_100
w, h = 3840, 2160
ctx = C_.HipContext(device=0)
cs = [make_render_tree_100(w, h, frame=f, full_frame_blur=True).to_c() for f in range(4)]
col = C_._F4(1.0, 1.0, 1.0, 1.0)
n = 200
for same in (False, True):
    for i in range(8):
        ctx._ck(ctx.L.fdh_render_frame(ctx.h, cs[0 if same else i & 3].byref(), float(w), float(h), 1, col))
    ctx.sync()
    t0 = time.perf_counter()
    per = []
    for i in range(n):
        a = time.perf_counter()
        ctx._ck(ctx.L.fdh_render_frame(ctx.h, cs[0 if same else i & 3].byref(), float(w), float(h), 1, col))
        per.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    st = ctx.frame_stats()
    per.sort()
    print(f"{'same frame' if same else 'four frames in rotation'}: {1e6 * (t2 - t0) / n:.1f} us/frame; inside the calls {1e6 * (t1 - t0) / n:.1f} us/frame "
          f"(median call {1e6 * per[n // 2]:.1f}, p90 {1e6 * per[int(n * .9)]:.1f}, max {1e6 * per[-1]:.1f}); final wait {1e6 * (t2 - t1):.0f} us; "
          f"host_record {1e3 * st.ms_host_record:.1f} host_upload_prep {1e3 * st.ms_host_upload:.1f} host_launch {1e3 * st.ms_host_launch:.1f} uploaded {ctx.last_upload_bytes()}")
