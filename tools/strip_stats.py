#!/usr/bin/env python3
"""Per-strip draw classification of the bench frame, from the instrumented build (make -C figdraw_amd/csrc stats).
Run on the GPU box:  FIGDRAW_HIP_LIB=build/libfigdraw_hip_stats.so python3 tools/strip_stats.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from figdraw_amd import context as ctx_mod  # noqa: E402
from figdraw_amd.scenes import make_clip_mask_benchmark, make_curves_scene, make_render_tree_100, make_rotated_tree  # noqa: E402

w, h = 3840, 2160
ctx = ctx_mod.HipContext(device=0)
L = ctx_mod.load()
buf = (C.c_ulonglong * 128)()
which = sys.argv[1] if len(sys.argv) > 1 else "bench"  # bench | bench1080 (BASELINE config 2: the same tree at 1920 x 1080, no full-frame blur) | rotated | curves | sub_clip | rect_mask (the reference's clip benchmark, 1200 x 800)
if which in ("sub_clip", "rect_mask"): w, h = 1200, 800
if which == "bench1080": w, h = 1920, 1080
scene = {"bench1080": lambda: make_render_tree_100(w, h, frame=0), "bench": lambda: make_render_tree_100(w, h, frame=0, full_frame_blur=True), "rotated": lambda: make_rotated_tree(w, h, 0), "curves": lambda: make_curves_scene(w, h), "sub_clip": lambda: make_clip_mask_benchmark("sub_clip"), "rect_mask": lambda: make_clip_mask_benchmark("rect_mask")}[which]()
ctx.render_frame(scene, w, h)
ctx.replay(5)
ctx.sync()
ctx.profile(5)
st = ctx.frame_stats()
print(f"this build: composite_main {st.ms_composite_main * 1000:.1f} us, frame {st.ms_total * 1000:.1f} us")
L.fdh_debug_counters(buf, 1)  # reset after warm-up
ctx.replay(1)
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
for i, n in ((32, "core: stroke no-op"), (33, "core: solid uniform blend"), (34, "core: other (gradient / push / blur)"),
             (35, "core: plain colour, record not fetched")):
    print(f"{n:34s} {c[i]:10d}")
for i, n in ((60, "rotated quad, 4-wide"), (63, "  of which core strips"), (61, "bezier, 4-wide: evaluated"), (62, "bezier, 4-wide: strip skipped")):
    print(f"{n:34s} {c[i]:10d}")
for code, n in enumerate(("", "fill", "drop shadow", "inner shadow", "AA stroke", "fill, elliptical", "drop shadow, elliptical", "inner shadow, elliptical",
                          "AA stroke, elliptical")):
    if code:
        print(f"packed edge path {code} ({n:24s}) {c[48 + code]:10d}")
for i, n in ((64, "packed edge: strip inside the quad (no per-pixel quad test)"), (65, "packed edge: quad test per pixel"), (71, "packed edge: black source"),
             (66, "generic path: strip covered (core or inside the quad)"), (67, "generic path: quad test per pixel"),
             (68, "3-stop fill: strip below the middle stop"), (69, "3-stop fill: strip above the middle stop"), (70, "3-stop fill: strip straddles the middle stop")):
    print(f"{n:62s} {c[i]:10d}")
if hasattr(L, "fdh_debug_wave_times") and os.environ.get("FDH_TIMING"):
    import numpy as np
    wt = np.zeros((65536, 16), dtype=np.uint64)
    L.fdh_debug_wave_times(wt.ctypes.data_as(C.c_void_p))
    ids = np.nonzero(wt[:, 6] == 1)[0]
    wt = wt[wt[:, 6] == 1].astype(np.float64)
    per_xcd = [wt[(ids & 7) == x, 0].sum() / 1000 for x in range(8)]
    print("sum of wave durations per XCD (kcycles):", [round(v) for v in per_xcd], " balanced kernel = sum/5120 slots =",
          round(wt[:, 0].sum() / 5120 / 1000, 1), "kcycles;  worst XCD / 640 slots =", round(max(per_xcd) / 640, 1))

    n = len(wt)
    for i in range(6):
        c[50 + i] = wt[:, i].sum()
    for i, nm in enumerate(("mode 3 fill", "mode 7 drop shadow", "mode 9 inset shadow", "mode 12 stroke / other")):
        print(f"edge draws {nm:24s} n={int(wt[:, 12 + i].sum()):7d}  mean {wt[:, 8 + i].sum() / max(wt[:, 12 + i].sum(), 1):7.0f} cycles  total {wt[:, 8 + i].sum() / 1e6:7.1f} Mcycles")
    core_t = np.floor(wt[:, 7] / 1024).sum(); core_n = (wt[:, 7] % 1024).sum()
    print(f"shade: core draws {int(core_n)} mean {core_t / max(core_n, 1):.0f} cycles; other draws {int(c[55] - core_n)} mean "
          f"{(c[54] - core_t) / max(c[55] - core_n, 1):.0f} cycles")
    c[56] = n
    print("wave duration p50/p90/max",
          np.percentile(wt[:, 0], 50) / 1000, np.percentile(wt[:, 0], 90) / 1000, wt[:, 0].max() / 1000)
if c[56]:
    n = c[56]
    print(f"timing build: {n} waves, mean per wave in kilo-cycles (s_memtime): total {c[50] / n / 1000:.2f}  "
          f"record-wait {c[53] / n / 1000:.2f}  shade {c[54] / n / 1000:.2f}  draws/wave {c[55] / n:.2f}")
print("strips:", (w // 32) * (h // 8))
