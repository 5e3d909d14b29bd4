#!/usr/bin/env python3
"""GPU-box helper: where k_blur_small's time goes, wall clocks of every wave of one launch from the timing build
(make -C figdraw_amd/csrc variant NAME=timing SINGLE=1 DEFS="-DFDH_STATS=1 -DFDH_TIMING=1"; FIGDRAW_HIP_LIB=build/libfigdraw_hip_timing.so)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from figdraw_amd import context as ctx_mod
from figdraw_amd.scenes import make_render_tree_100
w, h = 3840, 2160
ctx = ctx_mod.HipContext(device=0); L = ctx_mod.load()
ctx.render_frame(make_render_tree_100(w, h, frame=0, full_frame_blur=True), w, h); ctx.replay(5); ctx.sync()
wt = np.zeros((65536, 16), dtype=np.uint64)
L.fdh_debug_wave_times.argtypes = [C.c_void_p]
L.fdh_debug_wave_times(wt.ctypes.data)
ctx.replay(1); ctx.sync()
L.fdh_debug_wave_times(wt.ctypes.data)
rows = np.nonzero(wt[:, 6] == 8)[0]
t = wt[rows].astype(np.float64) / 100.0
t0 = t[:, 1].min()
wave = rows % 16
print(f"{len(rows)} waves of {len(rows) // 16} tiles; us after the launch's first wave, p10 / p50 / p90 / max")
def q(name, v): print(f"  {name:44s} {np.percentile(v,10):6.2f} {np.percentile(v,50):6.2f} {np.percentile(v,90):6.2f} {v.max():6.2f}")
q("wave enters", t[:, 1] - t0)
q("window asked for (addresses done, loads out)", t[:, 2] - t0)
q("window in LDS (behind the barrier)", t[:, 3] - t0)
for lo, hi in ((0, 3), (4, 7), (8, 11), (12, 15)):
    m = (wave >= lo) & (wave <= hi)
    q(f"waves {lo}-{hi}: own horizontal tasks done", t[m, 4] - t0)
q("behind the second barrier", t[:, 5] - t0)
for lo, hi in ((0, 3), (4, 15)):
    m = (wave >= lo) & (wave <= hi)
    q(f"waves {lo}-{hi}: leaves", t[m, 7] - t0)
