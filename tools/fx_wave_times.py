#!/usr/bin/env python3
"""GPU-box helper: per-wave phase times of the fused full-frame blur (k_blur_fx) from an instrumented build
(make -C figdraw_amd/csrc variant NAME=timing SINGLE=1 DEFS="-DFDH_STATS=1 -DFDH_TIMING=1"; FIGDRAW_HIP_LIB=build/libfigdraw_hip_timing.so)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from figdraw_amd import context as ctx_mod
from figdraw_amd.scenes import make_render_tree_100
w, h = 3840, 2160
ctx = ctx_mod.HipContext(device=0); L = ctx_mod.load()
tree = make_render_tree_100(w, h, frame=0, full_frame_blur=True)
lst = tree.layers[0]  # (without the small blur node and its panel: the launches behind k_blur_fx write their own rows over its)
lst.nodes = lst.nodes[:-2]; lst.rootIds = list(range(len(lst.nodes)))
ctx.render_frame(tree, w, h); ctx.sync()
wt = np.zeros((65536, 16), dtype=np.uint64)
L.fdh_debug_wave_times(wt.ctypes.data_as(C.c_void_p))  # (reading clears the rows)
ctx.replay(1); ctx.sync()
L.fdh_debug_wave_times(wt.ctypes.data_as(C.c_void_p))  # exactly one frame
sel = np.nonzero(wt[:, 6] == 7)[0]
r = wt[sel].astype(np.float64)
tick = 1.0 / 2.1  # shader cycles at ~2.1 GHz -> ns
nb = r[:, 8]
print(f"k_blur_fx: {len(r)} waves, {nb.mean():.1f} H-blocks each; per wave (ns): total {r[:,0].mean()*tick:.0f} (p10 {np.percentile(r[:,0],10)*tick:.0f}, p90 {np.percentile(r[:,0],90)*tick:.0f}, "
      f"max {r[:,0].max()*tick:.0f}) prologue {r[:,1].mean()*tick:.0f}; per H-block: wait {(r[:,2]/nb).mean()*tick:.0f} h-product {(r[:,3]/nb).mean()*tick:.0f} "
      f"[dma issue {(r[:,12]/nb).mean()*tick:.0f} + rounding {(r[:,11]/nb).mean()*tick:.0f} + v-product =] {(r[:,4]/nb).mean()*tick:.0f} epilogue {(r[:,5]/nb).mean()*tick:.0f} dma-late+stores {(r[:,7]/nb).mean()*tick:.0f}")
# which waves live longest: by strip (column position) and by segment (row position) -- the T the launcher chose is what n_hblocks says
T = int(round(nb.max())) - 2
n_sg, n_seg = (w // 32 + 3) // 4, (h // 32 + T - 1) // T
total = n_sg * n_seg
per = (total + 7) // 8
wgs = sel // 4
item = (wgs % 8) * per + wgs // 8
seg, sg, wv = item // n_sg, item % n_sg, sel % 4
strip = 4 * sg + wv
life = r[:, 0] * tick / 1000
print(f"T = {T}: {n_sg} strip groups x {n_seg} segments")
print("mean / max life (us) by strip:", " ".join(f"{int(k)}:{life[strip == k].mean():.1f}/{life[strip == k].max():.1f}" for k in np.unique(strip) if k < 6 or k > w // 32 - 7 or k % 16 == 0))
for k in (1, n_seg - 1):
    m_ = seg == k
    print(f"segment {k}: per H-block (ns) wait {(r[m_,2]/nb[m_]).mean()*tick:.0f} h-product {(r[m_,3]/nb[m_]).mean()*tick:.0f} rounding + v-product {(r[m_,4]/nb[m_]).mean()*tick:.0f} "
          f"epilogue {(r[m_,5]/nb[m_]).mean()*tick:.0f} stores {(r[m_,7]/nb[m_]).mean()*tick:.0f} prologue {r[m_,1].mean()*tick:.0f}; V blocks {r[m_,10].mean():.1f}")
print("mean / max life (us) by segment:", " ".join(f"{int(k)}:{life[seg == k].mean():.1f}/{life[seg == k].max():.1f}" for k in np.unique(seg)))
# edge strip groups (their shared window crosses the frame's left / right edge; their blocks lie on the fused quad's border) against the rest
def brk(name, m_):
    print(f"{name}: {int(m_.sum())} waves, life {life[m_].mean():.1f} us; per H-block (ns) wait {(r[m_,2]/nb[m_]).mean()*tick:.0f} h-product {(r[m_,3]/nb[m_]).mean()*tick:.0f} "
          f"dma issue {(r[m_,12]/nb[m_]).mean()*tick:.0f} rounding + v-product {((r[m_,4]-r[m_,12])/nb[m_]).mean()*tick:.0f} epilogue {(r[m_,5]/nb[m_]).mean()*tick:.0f} "
          f"stores {(r[m_,7]/nb[m_]).mean()*tick:.0f} prologue {r[m_,1].mean()*tick:.0f}")
inner_seg = (seg > 0) & (seg < n_seg - 1)
brk("left edge group, inner segments", (sg == 0) & inner_seg)
for k in range(4): brk(f"   its wave {k}", (sg == 0) & inner_seg & (wv == k))
brk("right edge group, inner segments", (sg == n_sg - 1) & inner_seg)
for k in range(4): brk(f"   its wave {k}", (sg == n_sg - 1) & inner_seg & (wv == k))
brk("interior groups, inner segments", (sg > 0) & (sg < n_sg - 1) & inner_seg)
brk("interior groups, first segment", (sg > 0) & (sg < n_sg - 1) & (seg == 0))
brk("interior groups, last segment", (sg > 0) & (sg < n_sg - 1) & (seg == n_seg - 1))
