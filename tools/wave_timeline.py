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
L.fdh_debug_wave_times.argtypes = [C.c_void_p]
L.fdh_debug_wave_times(wt.ctypes.data)  # (a read clears the rows)
ctx.replay(1)
ctx.sync()
L.fdh_debug_wave_times(wt.ctypes.data)
ok = wt[:, 6] != 0
t = wt[ok].astype(np.float64)
print("rows flagged:", int(ok.sum()))
WALL = 100.0  # wall_clock64: 100 MHz, one counter for the whole device (clock64 is per XCD)
start, end, cyc, draws, role = t[:, 1] / WALL, t[:, 2] / WALL, t[:, 0], t[:, 5], t[:, 6]
# (the later phases' launches overwrite the low rows; the full-frame launch is the bulk: keep the waves that start within 200 us of the median start)
keep = np.abs(start - np.median(start)) < 200.0
start, end, cyc, draws, role = start[keep], end[keep], cyc[keep], draws[keep], role[keep]
t0 = start.min()
start -= t0
end -= t0
dur = end - start
span = end.max()
print(f"waves {len(start)}, span {span:.1f} us (event-timed launch: {st.ms_composite_main * 1000:.1f}), sum of wave durations {dur.sum() / 1e3:.1f} ms = {dur.sum() / span:.0f} waves in flight on average (6144 slots at 6 per SIMD); deep bins: {st.deep_bins:.0f}")
print(f"shader clock: {np.median(cyc[dur > 1] / dur[dur > 1]):.0f} cycles per us")
for rl, name in ((1, "one-wave strips"), (2, "deep strips' blenders"), (3, "deep strips' shaders")):
    k = role == rl
    if not k.any():
        continue
    if rl == 2:  # (columns 11 / 15 of a blender's row: cycles between asking for a slot and having it, polls that found it unpublished)
        tw, npoll = t[:, 11][keep][k], t[:, 15][keep][k]
        print(f"   blenders: {100.0 * tw.sum() / cyc[k].sum():.0f} % of their cycles between asking for a slot and having it; {npoll.sum() / max(draws[k].sum(), 1):.2f} polls per list entry found the slot unpublished")
    d_, n_, s_, e_ = dur[k], draws[k], start[k], end[k]
    print(f"-- {name}: {int(k.sum())} waves; duration us p50 {np.percentile(d_, 50):.1f} p90 {np.percentile(d_, 90):.1f} p99 {np.percentile(d_, 99):.1f} max {d_.max():.1f}; list entries walked p50 {np.percentile(n_, 50):.0f} p90 {np.percentile(n_, 90):.0f} max {n_.max():.0f}; start p50 {np.percentile(s_, 50):.1f} max {s_.max():.1f}; end max {e_.max():.1f}")
    for lo, hi in ((0, 0), (1, 4), (5, 9), (10, 19), (20, 29), (30, 39), (40, 59), (60, 999)):
        m = (n_ >= lo) & (n_ <= hi)
        if m.any():
            print(f"     {lo:3d}..{hi:3d} entries: {int(m.sum()):6d} waves   duration mean {d_[m].mean():6.2f} us  ({d_[m].sum() / max(n_[m].sum(), 1) * 1e3:6.0f} ns per entry)   start mean {s_[m].mean():6.2f}   end max {e_[m].max():6.2f}")
print("waves in flight at 5 %, 15 %, .. 95 % of the span:", [int(((start <= f * span) & (end > f * span)).sum()) for f in np.linspace(0.05, 0.95, 10)])
print("waves started by then:                            ", [int((start <= f * span).sum()) for f in np.linspace(0.05, 0.95, 10)])
last = np.argsort(end)[-6:]
print("the six waves that end last: (role, start us, duration us, entries)", [(int(role[i]), round(float(start[i]), 1), round(float(dur[i]), 1), int(draws[i])) for i in last])
