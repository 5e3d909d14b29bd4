#!/usr/bin/env python3
"""GPU-box helper: every BASELINE.json config on the current build -- frame time (one frame at a time, replayed resident
records), per-kernel times (fdh_profile: each launch's own timestamps), rate, and parity against the oracle (max LSB, pixels
differing).  Prints one JSON document (-> profiles/<tag>_configs.json).

Configs 6 - 8 are the reference's OWN benchmark workloads (the only scenes it times itself): examples/windy_non_clip_benchmark.nim
(180 x 10 cells, 1200 x 800) and examples/windy_clip_mask_benchmark.nim (clip + sub-clip / clip + rect-mask, 180 x 6 cells); for
those the dynamic path (fdh_render_frame per frame, what the reference's loop times) is reported beside the replay figure.

usage: python3 tools/perf_configs.py [only]     only = 1..11 : run just that config (for a rocprofv3 --kernel-trace --stats pass per config)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import ref_scenes as RS
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import (load_glyph_fixture, make_clip_mask_benchmark, make_curves_scene, make_glyph_scene, make_non_clip_benchmark, make_render_tree_100,
                                make_rotated_tree)
from oracle import oracle as O

only = int(sys.argv[1]) if len(sys.argv) > 1 else 0
out = {}

# ---- the algorithmic model of a frame (extended to the atlas / MSDF / bezier modes for round 5): flops per COVERED
# fragment counted off the reference's fragment shader, x the fragments the frame's draws cover (quad areas), and the bytes a frame
# must move.  The shader lines each figure is counted from:
#   ClipAA(3) 25, DropShadow(7) 35 + exp, InsetShadow(9) 70 + exp, AnnularAA(12) 28, BackdropBlur(17) 25: SURVEY.md 8(d) (atlas.frag:252-405)
#   elliptical corners + 30 (sdEllipticalRoundedBox, atlas.frag:51-115)
#   Atlas(0) 32: one LINEAR texture2D = 3 lerps x 4 channels x 2 flops + 4 for the texel coordinates, x the tint, 4 (atlas.frag:284-295)
#   Msdf / Mtsdf(13 - 16) 48: the same fetch (28) + median 4 + msdfScreenPxRange ~10 + threshold, clamp, alpha 6 (atlas.frag:296-318)
#   bezier strokes(18 - 20) ~120: sdBezier's closed-form cubic (atlas.frag:121-209: dot products, a sqrt, acos / cos or two cube roots)
#   + 16 per fragment for the fixed-function blend and the RGBA8 round (utils/glutils.nim:150-154)
FLOPS = {0: 32, 3: 25, 7: 36, 9: 71, 12: 28, 13: 48, 14: 48, 15: 48, 16: 48, 17: 25, 18: 120, 19: 120, 20: 120}
VALU_PEAK_TFLOPS, HBM_PEAK_GBS = 157.3, 8000.0


def algorithmic(sc, w, h, images=None, atlas_size=1024):
    """fragments by mode (quad areas clipped to nothing: the shader runs on every fragment of a quad inside the frame or not -- the
    frame's own rectangle is applied), flops, bytes: from the BackendContext calls the scene decomposes into (a record-only context)"""
    rec = HipContext(record_only=True, atlas_size=atlas_size)
    for k in sorted(images or {}):
        rec.put_image(k, images[k])
    rec.record_begin()
    rec.render_frame(sc, w, h)
    calls = rec.record_calls()
    rec.close()
    frags, ellip, texels, n_draws = {}, 0, {}, 0
    scale, stack = 1.0, []
    for c in calls:
        name = c[0]
        if name == "save_transform":
            stack.append(scale)
        elif name == "restore_transform":
            scale = stack.pop() if stack else 1.0
        elif name == "scale":
            scale *= abs(float(c[1]) * float(c[2] if len(c) > 2 and c[2] is not None else c[1]))
        elif name == "draw_rounded_rect_sdf":
            area = max(float(c[1][2]), 0.0) * max(float(c[1][3]), 0.0) * scale
            mode = int(c[5])
            frags[mode] = frags.get(mode, 0.0) + min(area, float(w * h))
            if list(c[3]) != list(c[4]):
                ellip += min(area, float(w * h))
            n_draws += 1
        elif name in ("draw_image", "draw_image_adj"):
            sz = c[4] if len(c) > 4 else [0, 0]
            iw, ih = (float(sz[0]), float(sz[1])) if sz and sz[0] > 0 and sz[1] > 0 else ((images[c[1]].shape[1], images[c[1]].shape[0]) if images and c[1] in images else (0.0, 0.0))
            frags[0] = frags.get(0, 0.0) + iw * ih * scale
            if images and c[1] in images: texels[c[1]] = images[c[1]].shape[0] * images[c[1]].shape[1]
            n_draws += 1
        elif name == "draw_msdf":
            frags[13] = frags.get(13, 0.0) + float(c[4][0]) * float(c[4][1]) * scale
            if images and c[1] in images: texels[c[1]] = images[c[1]].shape[0] * images[c[1]].shape[1]
            n_draws += 1
        elif name == "draw_quadratic_bezier_sdf":
            n_draws += 1  # (its quad's area is not in the call: counted from the library's own fragment total below)
        elif name in ("draw_filled_quad", "draw_rect", "draw_backdrop_blur"):
            n_draws += 1
    return frags, ellip, sum(texels.values()), n_draws


def run(key, what, ctx, sc, w, h, n=100, oracle_kw=None, images=None):
    # the context's FIRST frame, timed where the library starts: the scene already marshalled into FdhFig arrays, one fdh_render_frame
    # + fdh_sync from C (allocations, staging sets, the kernels' first launches, upload).  (Round 3 timed ctx.render_frame from
    # Python here: 23 - 90 ms of which all but a few were ctypes marshalling of the node list.)
    from figdraw_amd import call_stream as CS
    cs0 = sc.to_c()
    first = CS.Player().play_scenes([ctx], [cs0], 1, w, h)
    t = time.perf_counter()
    ctx.render_frame(sc, w, h)
    ctx.sync()
    through_python = time.perf_counter() - t
    got = ctx.read_pixels()
    ctx.replay(10)
    ctx.replay(n)
    ms = ctx.frame_stats().ms_total
    ctx.profile(5)   # (the first profiled launches of a context can carry a one-off: round 5's first full run put 24.6 us on a 5.3-us launch)
    ctx.profile(20)
    st = ctx.frame_stats()
    e = {"workload": what, "width": w, "height": h, "draws": st.n_draws, "phases": st.n_phases, "blur_nodes": st.n_blurs,
         "frame_us": round(1e3 * ms, 1), "mpixels_per_s": round(w * h / ms / 1e3, 0), "gfragments_per_s": round(st.fragments / ms / 1e6, 1),
         "kernel_us": {"bin": round(1e3 * st.ms_bin, 1), "composite_phase0": round(1e3 * st.ms_composite_main, 1),
                       "composite_all": round(1e3 * st.ms_composite, 1), "blur_h_all": round(1e3 * st.ms_blur_h, 1),
                       "blur_v_all": round(1e3 * st.ms_blur_v, 1), "largest_blur_h": round(1e3 * st.ms_blur_big_h, 1),
                       "largest_blur_v": round(1e3 * st.ms_blur_big_v, 1)},
         "first_frame_host_ms": round(1e3 * first, 2), "frame_through_python_binding_ms": round(1e3 * through_python, 1)}
    # ---- roofline of the frame's compositing (round 5): algorithmic flops over the compositor launches' time against the FP32 vector
    # peak, algorithmic bytes over the frame's time against the HBM peak
    frags, ellip, atlas_texels, _ = algorithmic(sc, w, h, images, (oracle_kw or {}).get("atlas_size", 1024))
    counted = sum(frags.values())
    other = max(float(st.fragments) - counted, 0.0)  # bezier strokes, filled quads, shadows' extra quads: what the calls' rectangles do not give
    curve_frames = key == "config10"
    flops = sum(FLOPS.get(m, 25) * a for m, a in frags.items()) + 30 * ellip + (120 if curve_frames else 25) * other + 16 * (counted + other)
    byts = 4 * w * h + 128 * st.n_draws + 4 * atlas_texels + int(st.bytes_algorithmic - 4 * w * h - 128 * st.n_draws if st.n_blurs else 0)
    comp_s = st.ms_composite * 1e-3
    e["roofline"] = {"kernel": "k_composite_tiles (all phases of the frame)", "bound": "valu", "unit": "TFLOP/s", "peak": VALU_PEAK_TFLOPS,
                     "achieved": round(flops / comp_s / 1e12, 2) if comp_s > 0 else None, "frac": round(flops / comp_s / 1e12 / VALU_PEAK_TFLOPS, 4) if comp_s > 0 else None,
                     "algorithmic_flops": int(flops), "fragments_by_mode": {str(m): int(a) for m, a in sorted(frags.items())}, "fragments_other": int(other),
                     "fragments_elliptical": int(ellip),
                     "hbm": {"algorithmic_bytes": int(byts), "atlas_texels_touched": int(atlas_texels), "achieved_GBs": round(byts / (ms * 1e-3) / 1e9, 1),
                             "frac": round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                     "note": "flops per covered fragment counted off atlas.frag (tools/perf_configs.py FLOPS) x quad areas; bytes = 4 W H + 128 B x draws + atlas "
                             "texels of the images drawn (once each) + the blur nodes' bytes (SURVEY.md 8d)"}
    if key in ("config6", "config7", "config8"):  # the reference's loop: renderFrame per frame, 20 warm-up + 120 timed
        cs = sc.to_c()
        from figdraw_amd import call_stream as CS
        P = CS.Player()
        P.play_scenes([ctx], [cs], 20, w, h)
        runs = sorted(round(P.play_scenes([ctx], [cs], 120, w, h) / 120 * 1e6, 1) for _ in range(7))  # 120 frames = a few ms: one late wake-up of a
        e["dynamic_us_per_frame"] = runs[len(runs) // 2]  # pool thread doubles a single run; the median of seven is the figure, all seven are kept
        e["dynamic_us_per_frame_runs"] = runs
        e["dynamic_fps"] = round(1e6 / e["dynamic_us_per_frame"], 0)
    if not only:  # parity leg (the oracle takes seconds per frame; skipped under rocprofv3)
        orc = O.Oracle(threads=min(os.cpu_count() or 1, 16), **(oracle_kw or {}))
        for k in sorted(images or {}):
            orc.put_image(k, images[k])
        orc.render_frame(sc, w, h)
        d = np.abs(got.astype(int) - orc.read_pixels().astype(int))
        e["parity_vs_oracle"] = {"max_lsb": int(d.max()), "pixels_differing": int((d.max(axis=2) > 0).sum())}
    out[key] = e


if only in (0, 1, 2, 3, 5):
    ctx = HipContext(device=0)
    if only in (0, 1):
        run("config1", "tests/trender_rgb_boxes_sdf.nim scene, 800x600 (4 nodes)", ctx, RS.rgb_boxes_sdf(800.0, 600.0), 800, 600)
    if only in (0, 2):
        run("config2", "S100@1080p: renderlist_100 (304 nodes) at 1920x1080", ctx, make_render_tree_100(1920, 1080, 0), 1920, 1080)
    if only in (0, 3):
        run("config3", "S300@4K: renderlist_100 + full-frame blur(18) at 3840x2160 (the bench frame)", ctx,
            make_render_tree_100(3840, 2160, 0, full_frame_blur=True), 3840, 2160)
    if only in (0, 5):
        run("config5_one_frame", "one 7680x4320 frame of config 3's tree (the 8-frame, striped run: bench.py --mode stripes)", ctx,
            make_render_tree_100(7680, 4320, 0, full_frame_blur=True), 7680, 4320, 30)
    ctx.close()
if only in (0, 4):
    imgs = load_glyph_fixture(os.path.join(ROOT, "tests", "golden", "glyphs_ubuntu20.npz"))
    ctx = HipContext(atlas_size=1024, device=0)
    sc = make_glyph_scene(3840, 2160, imgs)
    used = RS.used_images(sc, imgs)
    for k in sorted(used):
        ctx.put_image(k, used[k])
    run("config4", "T10k@4K: 10 000 glyph quads (5 000 coverage + 5 000 MSDF) over a 3-stop gradient, 3840x2160", ctx, sc, 3840, 2160,
        oracle_kw={"atlas_size": 1024}, images=used)
    ctx.close()
if only in (0, 6, 7, 8):
    ctx = HipContext(device=0)
    if only in (0, 6):
        run("config6", "examples/windy_non_clip_benchmark.nim: 180 x 10 rounded cells, 1200x800 (1801 nodes, all roots)", ctx, make_non_clip_benchmark(), 1200, 800)
    if only in (0, 7):
        run("config7", "examples/windy_clip_mask_benchmark.nim, clip + sub-clip: clipping viewport, 180 x 6 cells each a second clip level (4322 nodes), 1200x800",
            ctx, make_clip_mask_benchmark("sub_clip"), 1200, 800)
    if only in (0, 8):
        run("config8", "examples/windy_clip_mask_benchmark.nim, clip + rect-mask: the same table with NfRectMaskContent cells, 1200x800",
            ctx, make_clip_mask_benchmark("rect_mask"), 1200, 800)
    ctx.close()
if only in (0, 11):
    imgs = load_glyph_fixture(os.path.join(ROOT, "tests", "golden", "glyphs_ubuntu20.npz"))
    ctx = HipContext(atlas_size=1024, device=0)
    sc = make_glyph_scene(3840, 2160, imgs, rotation=2.0)
    used = RS.used_images(sc, imgs)
    for k in sorted(used):
        ctx.put_image(k, used[k])
    run("config11", "T10k@4K rotated: config 4's glyph rows turned by 2 degrees about their centres, its MSDF images by 6: every atlas quad a rotated quad", ctx, sc, 3840, 2160,
        oracle_kw={"atlas_size": 1024}, images=used)
    ctx.close()
if only in (0, 9, 10):  # the compositor's one-pixel-slot build (<3>): rotated quads, bezier strokes -- SURVEY.md 8(a) a5 / f1
    ctx = HipContext(device=0)
    if only in (0, 9):
        run("config9", "R300@4K: the renderlist_100 tree with every rectangle rotated (-30 .. 30 degrees), 3840x2160, no blur: every draw a rotated quad",
            ctx, make_rotated_tree(3840, 2160, 0), 3840, 2160)
    if only in (0, 10):
        run("config10", "C1500@4K: 1500 stroked nkDrawable curves / lines / arcs (quadratic-bezier SDF spans, rotated boxes, join quads), 3840x2160",
            ctx, make_curves_scene(3840, 2160), 3840, 2160)
    ctx.close()
print(json.dumps(out, indent=1))
