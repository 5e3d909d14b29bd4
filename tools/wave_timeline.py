#!/usr/bin/env python3
"""When the compositor's waves run: per-wave start / duration / draws shaded of ONE full-frame launch, from the timing build
(make -C figdraw_amd/csrc variant NAME=timing SINGLE=1 DEFS="-DFDH_STATS=1 -DFDH_TIMING=1"; clock64 per wave, no atomics).
    FIGDRAW_HIP_LIB=build/libfigdraw_hip_timing.so python3 tools/wave_timeline.py [width height [blur]]
Prints the launch's span, the waves in flight over time (deciles of the span), the duration of a wave against the draws it shaded, and
what a perfectly balanced launch of the same waves would take -- the numbers behind "the 1080p launch waits on its tail" (round 6)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from figdraw_amd import context as ctx_mod  # noqa: E402
from figdraw_amd.scenes import make_render_tree_100  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
blur = len(sys.argv) > 3 and sys.argv[3] != "0"
ctx = ctx_mod.HipContext(device=0)
L = ctx_mod.load()
ctx.render_frame(make_render_tree_100(w, h, frame=0, full_frame_blur=blur), w, h)
ctx.replay(5)
ctx.sync()
ctx.profile(5)
st = ctx.frame_stats()
print(f"{w}x{h}: timing build composite_main {st.ms_composite_main * 1000:.1f} us, frame {st.ms_total * 1000:.1f} us, {st.n_draws} draws")
buf = (C.c_ulonglong * 128)()
L.fdh_debug_counters(buf, 1)
wt = np.zeros((65536, 16), dtype=np.uint64)
L.fdh_debug_wave_times(wt.ctypes.data_as(C.c_void_p))  # reset
ctx.replay(1)
ctx.sync()
L.fdh_debug_wave_times(wt.ctypes.data_as(C.c_void_p))
ok = wt[:, 6] == 1
t = wt[ok].astype(np.float64)
start, dur, draws = t[:, 1], t[:, 0], t[:, 5]
# (the later phases' launches overwrite the low rows; the full-frame launch is the bulk: keep waves that start within its span)
t0 = np.percentile(start, 1)
keep = (start >= t0 - 1e5) & (start < t0 + 5e6)
start, dur, draws = start[keep] - start[keep].min(), dur[keep], draws[keep]
end = start + dur
span = end.max()
MHZ = span / (st.ms_composite_main * 1000.0)  # ticks per us, calibrated: the launch's span in ticks against its measured time (clock64 = shader clock, ~2.1 - 2.4 GHz)
print(f"clock: {MHZ:.0f} ticks per us (span of the launch in ticks / its event-timed duration)")
print(f"waves {len(start)}, span {span / MHZ:.1f} us, sum of wave durations {dur.sum() / MHZ / 1e3:.1f} ms = {dur.sum() / span:.0f} waves in flight on average (6144 slots at 6 per SIMD)")
print(f"wave duration us: p10 {np.percentile(dur, 10) / MHZ:.1f}  p50 {np.percentile(dur, 50) / MHZ:.1f}  p90 {np.percentile(dur, 90) / MHZ:.1f}  p99 {np.percentile(dur, 99) / MHZ:.1f}  max {dur.max() / MHZ:.1f}")
print(f"draws shaded per wave: p50 {np.percentile(draws, 50):.0f}  p90 {np.percentile(draws, 90):.0f}  p99 {np.percentile(draws, 99):.0f}  max {draws.max():.0f}  total {draws.sum():.0f}")
for lo, hi in ((0, 0), (1, 4), (5, 9), (10, 19), (20, 29), (30, 39), (40, 59), (60, 999)):
    m = (draws >= lo) & (draws <= hi)
    if m.any():
        print(f"  waves with {lo:3d}..{hi:3d} draws: {int(m.sum()):6d}   duration mean {dur[m].mean() / MHZ:6.2f} us  ({dur[m].mean() / max(draws[m].mean(), 1) / MHZ * 1e3:6.0f} ns per draw)   start mean {start[m].mean() / MHZ:6.2f} us")
print("waves in flight at deciles of the span:", [int(((start <= f * span) & (end > f * span)).sum()) for f in np.linspace(0.05, 0.95, 10)])
last = np.argsort(end)[-5:]
print("the five waves that end last: (start us, duration us, draws)", [(round(start[i] / MHZ, 1), round(dur[i] / MHZ, 1), int(draws[i])) for i in last])
