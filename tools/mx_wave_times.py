#!/usr/bin/env python3
"""GPU-box helper: per-wave phase times of the matrix-pipe blur passes from an instrumented build
(make -C figdraw_amd/csrc variant NAME=timing DEFS="-DFDH_STATS=1 -DFDH_TIMING=1" with the k_blur_mx probes applied)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from figdraw_amd import context as ctx_mod
from figdraw_amd.scenes import make_render_tree_100
w, h = 3840, 2160
ctx = ctx_mod.HipContext(device=0); L = ctx_mod.load()
ctx.render_frame(make_render_tree_100(w, h, frame=0, full_frame_blur=True), w, h); ctx.sync()
wt = np.zeros((65536, 16), dtype=np.uint64)
L.fdh_debug_wave_times(wt.ctypes.data_as(C.c_void_p))  # (reading clears the rows)
ctx.replay(1); ctx.sync()
L.fdh_debug_wave_times(wt.ctypes.data_as(C.c_void_p))  # exactly one frame
for mark, name in ((2, "horizontal"), (3, "vertical")):
    sel = np.nonzero(wt[:, 6] == mark)[0]
    r = wt[sel].astype(np.float64)
    spans = []
    for x in range(8):  # s_memtime is per XCD: compare start times only inside one
        q = wt[sel[(sel % 8) == x]].astype(np.float64)
        if len(q): spans.append(((q[:, 9] + q[:, 0]).max() - q[:, 9].min(), q[:, 9].max() - q[:, 9].min(), len(q)))
    print(name, "per XCD: (kernel span, spread of wave start times, waves) in us:", [(round(a / 2100, 1), round(b / 2100, 1), n) for a, b, n in spans])
    if not len(r): continue
    tick = 1.0 / 2.1  # s_memtime ticks are shader cycles here; ~2.1 GHz under this load -> ns
    print(f"{name}: {len(r)} waves; per wave (ns): total {r[:,0].mean()*tick:.0f} (p10 {np.percentile(r[:,0],10)*tick:.0f}, p90 {np.percentile(r[:,0],90)*tick:.0f}, max {r[:,0].max()*tick:.0f}) "
          f"prologue {r[:,1].mean()*tick:.0f}; per block: wait {(r[:,2]/r[:,8]).mean()*tick:.0f} stores {(r[:,3]/r[:,8]).mean()*tick:.0f} "
          f"lds+mfma {(r[:,4]/r[:,8]).mean()*tick:.0f} epilogue {(r[:,5]/r[:,8]).mean()*tick:.0f} issue {(r[:,7]/r[:,8]).mean()*tick:.0f}; "
          f"launch span {(r[:,9].max() + r[r[:,9].argmax(),0] - r[:,9].min())*tick/1000:.1f} us, start spread {(r[:,9].max()-r[:,9].min())*tick/1000:.1f} us")

# where the slow waves are: mean / max lifetime by position along the filter direction (sa) -- the first and the last
# segment of a line group fetch k-steps that cross the frame edge (one texel per lane, clamped)
for mark, name, n_lines, n_along in ((2, "horizontal", (h + 31) // 32, (w + 127) // 128), (3, "vertical", (w + 31) // 32, (h + 127) // 128)):
    sel = np.nonzero(wt[:, 6] == mark)[0]
    if not len(sel): continue
    total = n_lines * n_along
    per = (total + 7) // 8
    item = (sel & 7) * per + (sel >> 3)
    if mark == 2:
        sa = item % n_along
    else:
        band = item // (16 * n_along); rem = item - band * 16 * n_along
        bw = np.minimum(16, n_lines - band * 16)
        sa = rem // bw
    life = wt[sel, 0].astype(np.float64) / 2100.0
    print(name, "wave lifetime (us) by segment along the filter: ", " ".join(f"{int(k)}:{life[sa == k].mean():.1f}/{life[sa == k].max():.1f}" for k in np.unique(sa)))
