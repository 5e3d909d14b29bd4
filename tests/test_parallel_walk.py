"""CPU suite: the scene front-end on the walk pool (fdh_set_walk_threads) records what the calling thread alone records.

The reference walks its node tree on one thread (figrender.nim:1756-1839, 1960-2002).  The library cuts large sibling groups
into chunks that pool threads decompose into their own arrays; the upload gathers the pieces in painter's order.  Contract:
`fdh_debug_record_digest` -- draw records in the form the calls produced them, bounds, quad extensions, phase table -- is the
same for every thread count, on every scene: what crosses chunk boundaries (bounds of the clips open around a group, extension
indices, phase summaries) is put right by the calling thread.  FDH_CREATE_RECORD_ONLY contexts: no GPU needed; the GPU suite
compares pixels."""
import copy

import numpy as np
import pytest

import ref_scenes as RS
from figdraw_amd import scene as S
from figdraw_amd.context import FigdrawHipError, HipContext
from figdraw_amd.scenes import make_clip_mask_benchmark, make_non_clip_benchmark, make_render_tree_100


def _digests(scene, w, h, threads=(0, 1, 2, 5), cull=1, ui_scale=1.0, setup=None):
    out = {}
    for t in threads:
        ctx = HipContext(record_only=True)
        if setup:
            setup(ctx)
        ctx.set_cull(cull)
        ctx.set_walk_threads(t)
        ctx.render_frame(scene, w, h, ui_scale=ui_scale)
        first = ctx.record_digest()
        ctx.render_frame(scene, w, h, ui_scale=ui_scale)  # a second frame: lanes and pool threads are reused
        assert ctx.record_digest() == first
        out[t] = (first, ctx.walk_stats()[1])
        ctx.close()
    return out


def _assert_same(d, expect_parallel=True):
    ref = d[0][0]
    assert d[0][1] == 0
    for t, (dig, groups) in d.items():
        assert dig == ref, f"{t} pool threads record something else than the calling thread alone"
        if t > 0 and expect_parallel:
            assert groups > 0, f"nothing went to the pool with {t} threads"


@pytest.mark.parametrize("cull", [0, 1])
def test_bench_scene_and_reference_benchmark_tables(cull):
    _assert_same(_digests(make_render_tree_100(3840, 2160, 3, full_frame_blur=True), 3840, 2160, cull=cull))  # two blur roots: three groups
    _assert_same(_digests(make_non_clip_benchmark(), 1200, 800, cull=cull))
    _assert_same(_digests(make_clip_mask_benchmark("sub_clip"), 1200, 800, cull=cull))  # the group sits under an open clip
    _assert_same(_digests(make_clip_mask_benchmark("rect_mask"), 1200, 800, cull=cull))  # ... and its cells open the fast rect mask


def _wide_scene(seed, w, h, n=260):
    """many sibling roots of every kind the random generator knows + one parent with many children under a clip and a rotation"""
    sc = RS.random_scene(seed, float(w), float(h), n=n, clips=True, blur=True)
    lst = sc.layers[0]
    rng = np.random.default_rng(seed)
    parent = lst.addRoot(S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(w * 0.1, h * 0.1, w * 0.7, h * 0.7), fill=S.rgba(240, 240, 250, 200),
                               flags=S.FigFlags.NfClipContent, corners=[12] * 4, rotation=7.0))
    for k in range(150):
        x, y = float(rng.uniform(0, w * 0.8)), float(rng.uniform(0, h * 0.8))
        flags = [0, S.FigFlags.NfRectMaskContent, S.FigFlags.NfClipContent][k % 3]
        cell = lst.addChild(parent, S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(x, y, 60, 40), flags=flags, corners=[k % 9] * 4,
                                          fill=S.rgba(int(rng.integers(0, 255)), 120, 200, 230), rotation=float(k % 7) * 3.0 if k % 4 == 0 else 0.0,
                                          stroke=S.RenderStroke(weight=2.0 if k % 2 else 0.0, fill=S.fill(S.rgba(0, 0, 0, 200)))))
        lst.addChild(cell, S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(x - 8, y + 5, 76, 12), fill=S.rgba(250, 80, 60, 255), corners=[3] * 4))
    return sc


@pytest.mark.parametrize("seed,w,h", [(5, 1280, 720), (6, 801, 603), (9, 1920, 1080)])
def test_random_wide_scenes(seed, w, h):
    sc = _wide_scene(seed, w, h)
    _assert_same(_digests(sc, w, h))
    _assert_same(_digests(sc, w, h, cull=0, ui_scale=1.5, threads=(0, 3)))


def test_every_reference_test_scene():
    """the small scenes of the reference's own tests: too few siblings for the pool -- same digest, nothing forked"""
    scenes = {k: v[:3] for k, v in RS.REFERENCE_PNG_SCENES.items()}
    scenes.update(RS.SWIFTSHADER_SCENES)
    for name in sorted(scenes):
        fn, w, h = scenes[name]
        d = _digests(fn(float(w), float(h)), w, h, threads=(0, 3))
        assert d[0][0] == d[3][0], name


def test_a_group_that_needs_the_calling_thread_falls_back():
    """drawables sample the 4x4 white "rect" atlas image, created on first use (glcontext.nim:966-970) -- on the calling thread:
    a pool thread that finds it missing hands the group back (SerialOnly); the next frame runs on the pool"""
    w, h = 640, 480
    lst = S.RenderList()
    for k in range(120):
        lst.addRoot(S.Fig(kind=S.FigKind.nkDrawable, screenBox=S.rect(5.0 * k, 10.0, 40, 40), drawStroke=S.RenderStroke(weight=3.0, fill=S.fill(S.rgba(10, 20, 30, 255)), join=S.StrokeJoin.sjMiter, cap=S.StrokeCap.scButt),
                          drawOps=[S.drawableBezier([(0, 0), (40, 40), (0, 40), (40, 0), (5, 30)], steps=2)]))
    sc = S.Renders()
    sc.setLayer(0, lst)
    ref = HipContext(record_only=True)
    ref.set_walk_threads(0)
    ref.render_frame(sc, w, h)
    ctx = HipContext(record_only=True)
    ctx.set_walk_threads(3)
    ctx.render_frame(sc, w, h)
    assert ctx.record_digest() == ref.record_digest()
    first_groups = ctx.walk_stats()[1]
    ctx.render_frame(sc, w, h)
    assert ctx.record_digest() == ref.record_digest()
    assert first_groups == 0 and ctx.walk_stats()[1] == 1  # frame 1 fell back (the image was made), frame 2 forked
    ref.close()
    ctx.close()


def test_an_error_in_a_pool_thread_reaches_the_caller():
    """nodes nested deeper than the walk allows inside a forked group: the frame fails with the walker's own error"""
    lst = S.RenderList()
    for k in range(100):
        lst.addRoot(S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(3.0 * k, 3.0, 20, 20), fill=S.rgba(1, 2, 3, 255)))
    deep = lst.addRoot(S.Fig(kind=S.FigKind.nkFrame, screenBox=S.rect(0, 0, 10, 10)))
    for _ in range(2100):
        deep = lst.addChild(deep, S.Fig(kind=S.FigKind.nkFrame, screenBox=S.rect(0, 0, 10, 10)))
    for k in range(100):
        lst.addRoot(S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(3.0 * k, 40.0, 20, 20), fill=S.rgba(1, 2, 3, 255)))
    sc = S.Renders()
    sc.setLayer(0, lst)
    for t in (0, 3):
        ctx = HipContext(record_only=True)
        ctx.set_walk_threads(t)
        with pytest.raises(FigdrawHipError) as e:
            ctx.render_frame(sc, 640, 480)
        assert "2048" in str(e.value)
        ok = S.Renders()
        ok.setLayer(0, S.RenderList())
        ok.layers[0].addRoot(S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(1, 1, 20, 20), fill=S.rgba(1, 2, 3, 255)))
        ctx.render_frame(ok, 64, 48)  # the context is usable afterwards
        ctx.close()


def test_many_groups_in_one_frame_consolidate():
    """more pieces than the upload's run table holds (kMaxUploadRuns): the digest (which walks the pieces) stays the serial one; the
    GPU suite renders such a frame"""
    w, h = 800, 600
    lst = S.RenderList()
    for g in range(14):  # 14 parents x 60 children: 14 forked groups in one frame
        parent = lst.addRoot(S.Fig(kind=S.FigKind.nkFrame, screenBox=S.rect(0, 40.0 * g, w, 40)))
        for k in range(60):
            lst.addChild(parent, S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(12.0 * k, 40.0 * g + 4, 10, 30), corners=[2] * 4,
                                       fill=S.rgba((k * 37) & 255, (g * 53) & 255, 90, 255)))
    sc = S.Renders()
    sc.setLayer(0, lst)
    d = _digests(sc, w, h, threads=(0, 3))
    assert d[0][0] == d[3][0] and d[3][1] == 14
