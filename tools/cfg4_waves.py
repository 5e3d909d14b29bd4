#!/usr/bin/env python3
"""GPU-box helper: the compositor launch of config 4 / 11 (10 000 glyph quads over a gradient at 4K) wave by wave, from the timing build
(make -C figdraw_amd/csrc variant NAME=timing SINGLE=1 DEFS="-DFDH_STATS=1 -DFDH_TIMING=1"; FIGDRAW_HIP_LIB=build/libfigdraw_hip_timing.so; ROT=2: config 11)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from figdraw_amd import context as ctx_mod
from figdraw_amd.scenes import make_glyph_scene, load_glyph_fixture
import ref_scenes as RS
imgs = load_glyph_fixture(os.path.join(ROOT, "tests", "golden", "glyphs_ubuntu20.npz"))
ctx = ctx_mod.HipContext(atlas_size=1024, device=0); L = ctx_mod.load()
w, h = 3840, 2160
rot = float(os.environ.get("ROT", "0"))
sc = make_glyph_scene(w, h, imgs, rotation=rot) if rot else make_glyph_scene(w, h, imgs)
for k, v in RS.used_images(sc, imgs).items(): ctx.put_image(k, v)
ctx.render_frame(sc, w, h); ctx.replay(5); ctx.sync(); ctx.profile(5); st = ctx.frame_stats()
print(f"timing build: composite {st.ms_composite * 1e3:.1f} us, {st.n_draws} draws")
wt = np.zeros((65536, 16), dtype=np.uint64)
L.fdh_debug_wave_times.argtypes = [C.c_void_p]
L.fdh_debug_wave_times(wt.ctypes.data)
ctx.replay(1); ctx.sync()
L.fdh_debug_wave_times(wt.ctypes.data)
ok = wt[:, 6] == 1
t = wt[ok].astype(np.float64)
start, end, cyc, n = t[:, 1] / 100.0, t[:, 2] / 100.0, t[:, 0], t[:, 5]
keep = np.abs(start - np.median(start)) < 500.0
start, end, cyc, n, t = start[keep], end[keep], cyc[keep], n[keep], t[keep]
t0 = start.min(); start -= t0; end -= t0; dur = end - start; span = end.max()
print(f"{len(start)} strips; launch span {span:.1f} us; sum of wave lives {dur.sum() / 1e3:.2f} ms = {dur.sum() / span:.0f} waves in flight on average (of 256 CUs x 4 SIMDs x waves per SIMD)")
print("waves in flight at 5 %, 15 %, .. 95 % of the span:", [int(((start <= f * span) & (end > f * span)).sum()) for f in np.linspace(0.05, 0.95, 10)])
clk = np.median(cyc[dur > 1] / dur[dur > 1])
print(f"shader clock {clk:.0f} cycles per us; wave life us p10 {np.percentile(dur,10):.1f} p50 {np.percentile(dur,50):.1f} p90 {np.percentile(dur,90):.1f} max {dur.max():.1f}; list entries walked p50 {np.percentile(n,50):.0f} p90 {np.percentile(n,90):.0f} max {n.max():.0f}")
for lo, hi in ((0, 1), (2, 3), (4, 5), (6, 8), (9, 12), (13, 20), (21, 999)):
    m = (n >= lo) & (n <= hi)
    if m.any():
        print(f"  {lo:3d}..{hi:3d} entries: {int(m.sum()):6d} strips, life mean {dur[m].mean():6.2f} us = {dur[m].sum() / max(n[m].sum(), 1) * 1e3:6.0f} ns per entry; in it: record fetch {t[m, 3].mean() / clk:5.2f} us, shading {t[m, 4].mean() / clk:5.2f} us; starts at {start[m].mean():5.1f} us")
