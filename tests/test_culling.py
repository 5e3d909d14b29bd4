"""CPU suite: culling (fdh_set_cull) changes no pixel.

The reference hands every draw to GL and lets the rasteriser clip (examples/windy_non_clip_benchmark.nim:82-108 submits 180
rows of cells to a window that shows 34; examples/windy_clip_mask_benchmark.nim:147-186 scrolls 180 clipped rows through a
viewport).  The library drops draws whose pixel bounds miss the frame and does not walk the subtree of a node whose clip mask
lies outside.  Proof without a GPU: the call stream the culling front-end produces (FDH_CREATE_RECORD_ONLY context, cull mode 2 =
"also while recording") is replayed through the ORACLE's backend and must give the oracle's own frame of the whole scene, bit
for bit -- same arithmetic, fewer calls.  The GPU suite compares the HIP frames with and without culling."""
import numpy as np
import pytest

import ref_scenes as RS
from figdraw_amd import scene as S
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_clip_mask_benchmark, make_non_clip_benchmark
from oracle import oracle as O


def _culled_calls(scene, w, h, mode=2):
    ctx = HipContext(record_only=True)
    ctx.set_cull(mode)
    ctx.record_begin()
    ctx.render_frame(scene, w, h)
    calls = ctx.record_calls()
    n = ctx.culled_draws()
    ctx.close()
    return calls, n


def _oracle_frames(scene, w, h, calls):
    a = O.Oracle(threads=8)
    a.render_frame(scene, w, h)
    want = a.read_pixels()
    b = O.Oracle(threads=8)
    b.W, b.H = w, h
    b.replay(calls)
    return want, b.read_pixels()


def _off_frame_scene(w, h, seed):
    """random scene + nodes pushed off every edge of the frame: shadows that reach back in, clips whose content overflows into
    the frame, rect masks just outside, a blur node outside, rotated boxes straddling a corner"""
    rng = np.random.default_rng(seed)
    sc = RS.random_scene(seed, float(w), float(h), n=40, clips=True, blur=True)
    lst = sc.layers[0]
    for k in range(24):
        side = k % 4
        bw, bh = float(rng.uniform(20, 120)), float(rng.uniform(20, 90))
        off = float(rng.uniform(0.5, 30))
        x = {0: -bw - off, 1: w + off, 2: float(rng.uniform(0, w)), 3: float(rng.uniform(0, w))}[side]
        y = {0: float(rng.uniform(0, h)), 1: float(rng.uniform(0, h)), 2: -bh - off, 3: h + off}[side]
        flags = [0, S.FigFlags.NfClipContent, S.FigFlags.NfRectMaskContent][k % 3]
        shadows = []
        if k % 2 == 0:  # a drop shadow large enough to reach back into the frame
            shadows = [S.RenderShadow(style=S.ShadowStyle.DropShadow, blur=float(rng.uniform(4, 30)), spread=float(rng.uniform(0, 20)),
                                      x=float(rng.uniform(-20, 20)), y=float(rng.uniform(-20, 20)), fill=S.fill(S.rgba(0, 0, 0, 160)))]
        parent = lst.addRoot(S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(x, y, bw, bh), flags=flags, corners=[int(rng.integers(0, 12))] * 4,
                                   fill=S.rgba(int(rng.integers(0, 255)), 90, 160, 220), shadows=shadows,
                                   rotation=float(rng.uniform(-40, 40)) if k % 5 == 0 else 0.0))
        # children that overflow the (off-frame) parent back into the frame
        lst.addChild(parent, S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(x - 60, y - 60, bw + 120, bh + 120),
                                   fill=S.rgba(20, 200, int(rng.integers(0, 255)), 200), corners=[6] * 4))
    lst.addRoot(S.Fig(kind=S.FigKind.nkBackdropBlur, screenBox=S.rect(w + 5.0, 10.0, 80.0, 60.0), blur=9.0, fill=S.rgba(0, 0, 0, 0)))
    lst.addRoot(S.Fig(kind=S.FigKind.nkBackdropBlur, screenBox=S.rect(w - 40.0, h - 30.0, 80.0, 60.0), blur=6.0, fill=S.rgba(255, 255, 255, 40)))
    return sc


@pytest.mark.parametrize("seed,w,h", [(3, 320, 200), (11, 411, 263), (29, 256, 256)])
def test_culled_call_stream_gives_the_oracles_frame(seed, w, h):
    sc = _off_frame_scene(w, h, seed)
    calls, n = _culled_calls(sc, w, h)
    assert n > 0
    want, got = _oracle_frames(sc, w, h, calls)
    assert np.array_equal(want, got)


@pytest.mark.parametrize("kind", ["non_clip", "sub_clip", "rect_mask"])
def test_reference_benchmark_tables_culled(kind):
    """the reference's own benchmark trees at a reduced window: most rows lie below it"""
    w, h = 300, 200
    sc = make_non_clip_benchmark(w, h, rows=40, cols=4) if kind == "non_clip" else make_clip_mask_benchmark(kind, w, h, rows=40, cols=3)
    calls, n = _culled_calls(sc, w, h)
    full, _ = _culled_calls(sc, w, h, mode=0)
    assert len(calls) < len(full) // 2  # rows below the window are gone from the stream
    want, got = _oracle_frames(sc, w, h, calls)
    assert np.array_equal(want, got)


def test_cull_mode_1_leaves_recorded_streams_alone():
    """mode 1 (the default) does not cull while the call recorder runs: recorded streams stay the reference's, call for call"""
    sc = make_non_clip_benchmark(300, 200, rows=40, cols=4)
    a, _ = _culled_calls(sc, 300, 200, mode=1)
    b, _ = _culled_calls(sc, 300, 200, mode=0)
    assert a == b


def test_culling_keeps_every_visible_record():
    """records of a frame with and without culling: the culled frame's records are a subsequence of the full frame's (digest of
    the draws with non-empty bounds is unchanged for a scene without clips, where nothing but invisible draws can go)"""
    w, h = 300, 200
    sc = make_non_clip_benchmark(w, h, rows=40, cols=4)
    counts = {}
    for mode in (0, 1):
        ctx = HipContext(record_only=True)
        ctx.set_cull(mode)
        ctx.render_frame(sc, w, h)
        counts[mode] = ctx.culled_draws()
        ctx.close()
    assert counts[0] == 0 and counts[1] > 100


def test_draw_image_adj_call_stream_matches_the_oracles():
    """drawImageAdj through both restatements' recorders: same call, same arguments (CPU)"""
    import os

    from conftest import GOLDEN
    from figdraw_amd.scenes import load_glyph_fixture

    img = load_glyph_fixture(os.path.join(GOLDEN, "glyphs_ubuntu20.npz"))[3000]
    out = []
    for be in (O.Oracle(atlas_size=256), HipContext(atlas_size=256, record_only=True)):
        be.put_image(3000, img)
        be.record_begin()
        be.begin_frame(64, 64)
        be.draw_image_adj(3000, (3.5, 4.0), (255, 128, 0, 255), (40.0, 30.0))
        be.end_frame()
        out.append(be.record_calls())
    assert out[0] == out[1] and out[0][1][0] == "draw_image_adj"
