#!/usr/bin/env python3
"""GPU-box helper: hostile arguments through the per-call entry points -- NaN, infinities, 1e30, negative and zero sizes, huge blur
radii, 1-pixel and very wide frames, unbalanced masks -- must come back as an error code or a (possibly empty) frame, never a crash
or a hang, and the context must render a normal frame correctly afterwards.  usage: timeout 300 python3 tools/abuse.py [rounds]"""
import math
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import ref_scenes as RS  # noqa: E402
from figdraw_amd.context import FigdrawHipError, HipContext  # noqa: E402

import faulthandler
faulthandler.enable()
VERBOSE = os.environ.get("VERBOSE")
rnd = random.Random(int(os.environ.get("SEED", "11")))
BAD = [float("nan"), float("inf"), -float("inf"), 1e30, -1e30, 1e-30, 0.0, -0.0, -5.0, 3.4e38, 65536.0, 1e7]


def val(lo, hi):
    return rnd.choice(BAD) if rnd.random() < 0.3 else rnd.uniform(lo, hi)


def col():
    return (rnd.randrange(256), rnd.randrange(256), rnd.randrange(256), rnd.randrange(256))


ref = HipContext(device=0, sync_submit=True)
good = RS.nested_clips(640.0, 360.0)
ref.render_frame(good, 640, 360)
want = ref.read_pixels()
ctx = HipContext(device=0)
from figdraw_amd.scenes import load_glyph_fixture  # noqa: E402
imgs = load_glyph_fixture(os.path.join(ROOT, 'tests', 'golden', 'glyphs_ubuntu20.npz'))
for k in (1065, 1105, 2065, 2105, 3000):
    ctx.put_image(k, imgs[k])
errors = frames = 0
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for it in range(rounds):
    w, h = rnd.choice([(1, 1), (3, 2), (640, 360), (333, 217), (8192, 16), (16, 4096), (1920, 1080)])
    try:
        ctx.begin_frame(w, h, rnd.random() < 0.8, (val(0, 1), val(0, 1), val(0, 1), val(0, 1)))
        depth = 0
        for k in range(rnd.randrange(1, 40)):
            op = rnd.randrange(13)
            rect = (val(-50, w), val(-50, h), val(-10, w), val(-10, h))
            rx, ry = tuple(val(0, 40) for _ in range(4)), tuple(val(0, 40) for _ in range(4))
            if VERBOSE:
                print(it, (w, h), 'op', op, rect, rx, ry, flush=True)
            if op < 4:
                ctx.draw_rounded_rect_sdf(rect, [col() for _ in range(4)], rx, ry, rnd.choice([3, 7, 8, 9, 11, 12, 17]), val(0, 30), val(-5, 20), (val(-20, 20), val(-20, 20)),
                                          rnd.randrange(5), col(), col(), val(0, 1))
            elif op == 4:
                ctx.draw_backdrop_blur(rect, rx, ry, val(0, 200))
            elif op == 5 and depth < 20:
                ctx.begin_mask(rect, rx, ry); ctx.end_mask(); depth += 1
            elif op == 6 and depth > 0:
                ctx.pop_mask(); depth -= 1
            elif op == 7:
                ctx.save_transform(); ctx.translate(val(-100, 100), val(-100, 100)); ctx.rotate(val(-7, 7)); ctx.scale(val(-3, 3), val(-3, 3))
                ctx.draw_rect(rect, col()); ctx.restore_transform()
            elif op == 8:
                ctx.draw_quadratic_bezier_sdf(rect, {"kind": 0, "axis": 0, "start": col(), "mid": col(), "stop": col(), "mid_pos": 128},
                                              (val(0, w), val(0, h)), (val(0, w), val(0, h)), (val(0, w), val(0, h)), val(0, 30), rnd.randrange(3))
            elif op == 9:
                ctx.draw_filled_quad([val(-10, w) for _ in range(8)], [col() for _ in range(4)])
            elif op == 10:
                ctx.set_text_subpixel(rnd.random() < 0.5, val(0, 1))
                ctx.draw_image(rnd.choice([1065, 1105, 3000, 424242]), (val(-20, w), val(-20, h)), [col() for _ in range(4)], (val(-5, 300), val(-5, 300)), rnd.random() < 0.3)
            elif op == 11:
                ctx.draw_msdf(rnd.choice([2065, 2105, 424242]), (val(-20, w), val(-20, h)), col(), (val(-5, 300), val(-5, 300)), val(0, 16), val(-1, 2), val(-2, 6),
                              rnd.random() < 0.5, rnd.random() < 0.3)
            else:
                ctx.begin_rect_mask(rect, rx, ry); ctx.draw_rect(rect, col()); ctx.pop_rect_mask()
        if rnd.random() < 0.15:
            depth += 1  # leave a mask open on purpose: end_frame must refuse
        for _ in range(depth if rnd.random() < 0.85 else 0):
            ctx.pop_mask()
        if VERBOSE:
            print(it, 'end_frame', flush=True)
        ctx.end_frame()
        ctx.sync()
        frames += 1
    except FigdrawHipError:
        errors += 1
        # A refused call leaves the frame open (as in the reference: the caller finishes it); what was refused at end_frame is an
        # unbalanced mask stack, so: pop until the frame closes.  It must close.
        for _ in range(64):
            try:
                ctx.end_frame(); ctx.sync()
                break
            except FigdrawHipError as e:
                if "beginFrame was not called" in str(e):
                    break
                try:
                    ctx.pop_mask()
                except FigdrawHipError:
                    pass
        else:
            raise SystemExit("abuse: the frame could not be closed after an error")
    if it % 25 == 24:
        ctx.render_frame(good, 640, 360)
        assert np.array_equal(ctx.read_pixels(), want), f"round {it}: the context no longer renders a normal frame correctly"
ctx.render_frame(good, 640, 360)
assert np.array_equal(ctx.read_pixels(), want)
print(f"abuse: {rounds} rounds, {frames} frames rendered, {errors} refused with an error code; the context still renders correctly")
