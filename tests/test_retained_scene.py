"""Retained scenes (fdh_scene_*): the library-side half of the reference's RenderFragments (renderfragments.nim:426-544).

The contract: after ANY sequence of edits, fdh_scene_render hands the kernels exactly what fdh_render_frame of the edited tree
would -- the same draw records, bounds, quad extensions and phase table (fdh_debug_record_digest), hence the same pixels --
while decomposing only the roots the edits touched.  The CPU tests run on FDH_CREATE_RECORD_ONLY contexts (no GPU); the GPU
test compares pixels."""
import copy
import random

import numpy as np
import pytest

import ref_scenes as RS
from figdraw_amd.context import HipContext
from figdraw_amd.scene import (Fig, FigFlags, FigKind, RenderList, Renders, RenderShadow, RenderStroke, ShadowStyle, fill, linear,
                               rect, rgba)
from figdraw_amd.scenes import make_render_tree_100


def _subtrees(lst):
    """a RenderList as one node list per root: node 0 the root (parent -1), later parents relative to the list"""
    root_of = []
    for i, n in enumerate(lst.nodes):
        root_of.append(i if n.parent < 0 else root_of[n.parent])
    out = []
    for r in lst.rootIds:
        idx = [i for i in range(len(lst.nodes)) if root_of[i] == r]
        pos = {g: k for k, g in enumerate(idx)}
        sub = []
        for g in idx:
            f = copy.deepcopy(lst.nodes[g])
            f.parent = -1 if g == r else pos[f.parent]
            sub.append(f)
        out.append(sub)
    return out


def _flatten(subtrees):
    lst = RenderList()
    for sub in subtrees:
        base = len(lst.nodes)
        for k, f in enumerate(sub):
            g = copy.deepcopy(f)
            if k == 0:
                g.parent = -1
                lst.rootIds.append(base)
            else:
                g.parent = f.parent + base
            lst.nodes.append(g)
    sc = Renders()
    sc.setLayer(0, lst)
    return sc


def Renders_of(lst):
    sc = Renders()
    sc.setLayer(0, lst)
    return sc


def _fresh_digest(scene, w, h):
    ctx = HipContext(record_only=True)
    ctx.render_frame(scene, w, h)
    d = ctx.record_digest()
    ctx.close()
    return d


def _random_subtree(rnd, w, h):
    def node(parent):
        f = Fig(kind=FigKind.nkRectangle, screenBox=rect(rnd.uniform(-20, w - 40), rnd.uniform(-20, h - 30), rnd.uniform(8, 220), rnd.uniform(8, 160)),
                fill=rgba(rnd.randrange(256), rnd.randrange(256), rnd.randrange(256), rnd.choice([255, 255, 180, 90])),
                corners=[rnd.randrange(0, 30)] * 4)
        f.parent = parent
        if rnd.random() < 0.4:
            f.stroke = RenderStroke(weight=rnd.uniform(1, 6), fill=fill(rgba(0, 0, 0, rnd.choice([255, 160]))))
        if rnd.random() < 0.3:
            f.shadows = [RenderShadow(style=ShadowStyle.DropShadow, blur=rnd.uniform(2, 12), spread=rnd.uniform(0, 4), x=3, y=4, fill=fill(rgba(0, 0, 0, 120)))]
        if rnd.random() < 0.2:
            f.rotation = rnd.uniform(-40, 40)
        if rnd.random() < 0.15:
            f.fill = linear(rgba(250, 40, 40, 255), rgba(40, 40, 250, 255))
        return f

    sub = [node(-1)]
    if rnd.random() < 0.3:
        sub[0].flags |= FigFlags.NfClipContent
    for _ in range(rnd.randrange(0, 4)):
        sub.append(node(rnd.randrange(0, len(sub))))
    return sub


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_property_updates_reach_only_their_roots(seed):
    """fdh_scene_update_nodes on random node ranges of a random scene (clips, rect masks, rotations, blur nodes)"""
    rnd = random.Random(seed)
    w, h = 640, 480
    sc = RS.random_scene(seed, float(w), float(h), n=60, clips=True, blur=(seed != 2))
    lst = next(iter(sc.layers.values()))
    ctx = HipContext(record_only=True)
    ctx.scene_retain(sc, w, h)
    assert ctx.record_digest() == _fresh_digest(sc, w, h)
    n_roots = len(lst.rootIds)
    for step in range(12):
        first = rnd.randrange(0, len(lst.nodes))
        count = min(rnd.randrange(1, 5), len(lst.nodes) - first)
        for i in range(first, first + count):
            n = lst.nodes[i]
            x, y, bw, bh = n.screenBox
            n.screenBox = rect(x + rnd.uniform(-9, 9), y + rnd.uniform(-9, 9), max(1.0, bw + rnd.uniform(-5, 5)), max(1.0, bh + rnd.uniform(-5, 5)))
            if rnd.random() < 0.5:
                n.fill = fill(rgba(rnd.randrange(256), rnd.randrange(256), rnd.randrange(256), rnd.choice([255, 128])))
            if rnd.random() < 0.3:
                n.corners = [rnd.randrange(0, 25)] * 4
        ctx.scene_update_nodes(0, first, lst.nodes[first:first + count])
        ctx.scene_render()
        assert ctx.record_digest() == _fresh_digest(sc, w, h), (seed, step)
        walked, reused = ctx.scene_stats()
        assert walked + reused == n_roots and walked <= count + 4  # the touched roots + the roots holding blur nodes
    ctx.scene_render()  # nothing edited: only roots with blur nodes are walked again
    walked, reused = ctx.scene_stats()
    assert walked <= 3 and ctx.record_digest() == _fresh_digest(sc, w, h)
    ctx.close()


@pytest.mark.parametrize("seed", [5, 6])
def test_structural_edits_replace_insert_remove(seed):
    """fdh_scene_replace_root / fdh_scene_insert_root / removal against a mirror of the tree kept as one subtree per root"""
    rnd = random.Random(seed)
    w, h = 512, 384
    sc = RS.random_scene(seed, float(w), float(h), n=40, clips=True, blur=True)
    subs = _subtrees(next(iter(sc.layers.values())))
    ctx = HipContext(record_only=True)
    ctx.scene_retain(_flatten(subs), w, h)
    for step in range(24):
        op = rnd.choice(["replace", "replace", "insert", "remove"]) if len(subs) > 3 else "insert"
        if op == "replace":
            slot = rnd.randrange(len(subs))
            subs[slot] = _random_subtree(rnd, w, h)
            ctx.scene_replace_root(0, slot, subs[slot])
        elif op == "insert":
            slot = rnd.randrange(len(subs) + 1)
            subs.insert(slot, _random_subtree(rnd, w, h))
            ctx.scene_insert_root(0, slot, subs[slot])
        else:
            slot = rnd.randrange(len(subs))
            del subs[slot]
            ctx.scene_replace_root(0, slot, [])
        ctx.scene_render()
        assert ctx.record_digest() == _fresh_digest(_flatten(subs), w, h), (seed, step, op)
        walked, reused = ctx.scene_stats()
        assert walked + reused == len(subs) and walked <= 4
    ctx.close()


def test_baseline_workload_one_rect_moves():
    """the bench scene (S300@4K, 304 roots): moving one rectangle re-decomposes 1 root + the 2 blur roots, reuses 301"""
    w, h = 3840, 2160
    sc = make_render_tree_100(w, h, frame=0, full_frame_blur=True)
    lst = next(iter(sc.layers.values()))
    ctx = HipContext(record_only=True)
    ctx.scene_retain(sc, w, h)
    assert ctx.scene_stats() == (len(lst.rootIds), 0)
    n = lst.nodes[17]
    x, y, bw, bh = n.screenBox
    n.screenBox = rect(x + 31.5, y - 12.25, bw, bh)
    ctx.scene_update_nodes(0, 17, [n])
    ctx.scene_render()
    walked, reused = ctx.scene_stats()
    assert walked == 3 and reused == len(lst.rootIds) - 3
    assert ctx.record_digest() == _fresh_digest(sc, w, h)
    ctx.close()


def test_bad_edits_are_rejected():
    from figdraw_amd.context import FigdrawHipError

    ctx = HipContext(record_only=True)
    with pytest.raises(FigdrawHipError):
        ctx.scene_render()  # nothing retained
    fn, w, h = RS.SWIFTSHADER_SCENES["oneframe"]
    sc = fn(float(w), float(h))
    ctx.scene_retain(sc, w, h)
    lst = next(iter(sc.layers.values()))
    with pytest.raises(FigdrawHipError):
        ctx.scene_update_nodes(0, len(lst.nodes) - 1, lst.nodes[:3])  # range past the end
    with pytest.raises(FigdrawHipError):
        ctx.scene_replace_root(0, len(lst.rootIds), [lst.nodes[0]])  # no such slot
    bad = copy.deepcopy(lst.nodes[0])
    bad.parent = 0
    with pytest.raises(FigdrawHipError):
        ctx.scene_replace_root(0, 0, [bad])  # a subtree must start with its root
    ctx.close()


def _text_scene(n_glyphs=6, variants=False, x0=10.3):
    from figdraw_amd.scene import Glyph, TextRect

    r = Renders()
    r.addRoot(0, Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, 256, 128), fill=rgba(240, 240, 240, 255)))
    gl = [Glyph(image_id=7, x=x0 + 9.37 * i, y=2.0, subpixel_shift=-1.0, variant_ids=[100 + k for k in range(10)] if variants else None) for i in range(n_glyphs)]
    r.addRoot(0, Fig(kind=FigKind.nkText, screenBox=rect(20, 30, 200, 30), glyphs=gl, textRects=[TextRect(1.0, 20.0, 50.0, 2.0, kind=1, fill=fill(rgba(200, 0, 0, 255)))]))
    r.addRoot(0, Fig(kind=FigKind.nkRectangle, screenBox=rect(30, 80, 60, 20), fill=rgba(10, 60, 200, 255), corners=[4] * 4))
    return r


def _text_ctx():
    ctx = HipContext(atlas_size=256, record_only=True)
    img = np.zeros((10, 8, 4), np.uint8)
    for k in [7] + [100 + k for k in range(10)]:
        ctx.put_image(k, img)
    return ctx


def _fresh_text_digest(scene, subpixel, variants):
    ctx = _text_ctx()
    ctx.set_text_subpixel(subpixel, 0.0, glyph_variants=variants)
    ctx.render_frame(scene, 256, 128)
    d = ctx.record_digest()
    ctx.close()
    return d


def test_text_settings_invalidate_the_per_root_cache():
    """fdh_set_text_subpixel_* between two fdh_scene_render calls: the text roots' cached records were made under the old
    settings (glyph snapping, shift, variant keys -- figrender.nim:464-476) and must be decomposed again"""
    sc = _text_scene(variants=True)
    ctx = _text_ctx()
    ctx.scene_retain(sc, 256, 128)
    seen = set()
    for subpixel, variants in ((False, False), (True, False), (True, True), (False, False)):
        ctx.set_text_subpixel(subpixel, 0.0, glyph_variants=variants)
        ctx.scene_render()
        assert ctx.record_digest() == _fresh_text_digest(sc, subpixel, variants), (subpixel, variants)
        seen.add(ctx.record_digest())
    assert len(seen) == 3  # the three settings really give three different record streams
    ctx.close()


def test_variant_ids_arriving_late_fall_back_to_the_glyphs_own_image():
    """a retained scene without variant ids, then an update whose side arrays bring some: the OLD glyphs must keep drawing
    their own image under glyph-variant positioning (they were back-filled with key 0 = a missing image once)"""
    base = _text_scene(variants=False)
    ctx = _text_ctx()
    ctx.set_text_subpixel(True, 0.0, glyph_variants=True)
    ctx.scene_retain(base, 256, 128)
    lst_old = next(iter(base.layers.values()))
    newer = _text_scene(variants=True, x0=31.7)
    lst_new = next(iter(newer.layers.values()))
    extra = copy.deepcopy(lst_new.nodes[1])
    extra.screenBox = rect(20, 70, 200, 30)
    ctx.scene_insert_root(0, 3, [extra])
    ctx.scene_render()
    # the same tree built in one go: old glyphs without ids, new glyphs with ids (to_c fills the gaps with image_id)
    whole = _text_scene(variants=False)
    next(iter(whole.layers.values())).addRoot(copy.deepcopy(extra))
    assert ctx.record_digest() == _fresh_text_digest(whole, True, True)
    # every text root was decomposed again (the table's appearance changes how all glyphs are treated) ...
    assert ctx.scene_stats()[0] >= 2
    # ... and a later property edit of the OLD text node still finds its glyphs' fallback keys (7, never 0 = a missing image)
    old = lst_old.nodes[1]
    old.screenBox = rect(22, 30, 200, 30)
    ctx.scene_update_nodes(0, 1, [old])
    ctx.record_begin()
    ctx.scene_render()
    keys = [c[1] for c in ctx.record_calls() if c[0] == "draw_image"]
    assert keys == [7] * 6
    ctx.close()


def test_failed_structural_edit_leaves_the_scene_untouched():
    """fdh_scene_replace_root / insert_root with a bad side range (or over the node budget) must fail BEFORE anything is
    committed: same digest afterwards, same roots walked, and further edits still work"""
    from figdraw_amd.context import FigdrawHipError
    import ctypes as C

    sc = _text_scene()
    ctx = _text_ctx()
    ctx.scene_retain(sc, 256, 128)
    want = ctx.record_digest()
    lst = next(iter(sc.layers.values()))
    good = copy.deepcopy(lst.nodes[1])
    cs, ptr, n, side = ctx._marshal_nodes([good])
    nodes = C.cast(ptr, C.POINTER(type(cs.struct.layers[0].nodes[0])))
    nodes[0].glyph_first = 3
    nodes[0].glyph_count = 1000  # past the side arrays
    for call in (ctx.L.fdh_scene_replace_root, ctx.L.fdh_scene_insert_root):
        assert call(ctx.h, 0, 1, ptr, n, side) == -1
        ctx.scene_render()
        assert ctx.record_digest() == want
        assert ctx.scene_stats()[0] == 0  # nothing was marked dirty, no root was dropped or added
    nodes[0].glyph_count = 2  # and the same call with a valid range goes through
    assert ctx.L.fdh_scene_replace_root(ctx.h, 0, 1, ptr, n, side) == 0
    ctx.scene_render()
    assert ctx.scene_stats() == (1, 2)
    # node budget: 32767 nodes per layer (FigIdx is int16)
    big = [copy.deepcopy(lst.nodes[2]) for _ in range(32767)]
    for i, f in enumerate(big):
        f.parent = -1 if i == 0 else 0
    d0 = ctx.record_digest()
    with pytest.raises(FigdrawHipError):
        ctx.scene_insert_root(0, 0, big)
    ctx.scene_render()
    assert ctx.record_digest() == d0 and sum(ctx.scene_stats()) == 3
    ctx.close()


def test_animated_text_node_does_not_grow_the_side_arrays():
    """every update appends the new node's glyphs / text rects to the retained side arrays; they are compacted once dead
    entries dominate, and the records stay those of a fresh render"""
    ctx = _text_ctx()
    sc = _text_scene(n_glyphs=40)
    ctx.scene_retain(sc, 256, 128)
    lst = next(iter(sc.layers.values()))
    for step in range(300):  # 300 x 40 glyphs appended = 12 000 > the 4096-entry threshold, 40 live
        node = lst.nodes[1]
        node.screenBox = rect(20 + (step % 7), 30, 200, 30)
        ctx.scene_update_nodes(0, 1, [node])
        if step % 60 == 59:
            ctx.scene_render()
            assert ctx.record_digest() == _fresh_text_digest(sc, False, False), step
            assert ctx.scene_stats() == (1, 2)
    ctx.close()


@pytest.mark.gpu
def test_retained_scene_pixels_equal_full_render():
    """GPU: after each edit the retained context's frame equals a full fdh_render_frame of the edited tree, bit for bit, and
    the oracle's frame of the edited tree within the suite's parity bar"""
    rnd = random.Random(11)
    w, h = 1280, 720
    sc = make_render_tree_100(w, h, frame=2, full_frame_blur=True)
    subs = _subtrees(next(iter(sc.layers.values())))
    from oracle import oracle as O

    orc = O.Oracle(threads=8)
    ret, ref = HipContext(device=0), HipContext(device=0)
    ret.scene_retain(_flatten(subs), w, h)
    for step in range(8):
        slot = rnd.randrange(len(subs))
        if step % 3 == 2:
            subs.insert(slot, _random_subtree(rnd, w, h))
            ret.scene_insert_root(0, slot, subs[slot])
        else:
            subs[slot] = _random_subtree(rnd, w, h)
            ret.scene_replace_root(0, slot, subs[slot])
        ret.scene_render()
        ref.render_frame(_flatten(subs), w, h)
        got = ret.read_pixels()
        assert np.array_equal(got, ref.read_pixels()), step
        walked, reused = ret.scene_stats()
        assert walked <= 4 and reused >= len(subs) - 4
        # ... and the frame is the ORACLE's frame of the edited tree (parity, not only self-consistency)
        orc.render_frame(_flatten(subs), w, h)
        d = np.abs(got.astype(int) - orc.read_pixels().astype(int)).max(axis=2)
        assert d.max() <= 1 and (d > 0).sum() <= 0.005 * w * h, (step, int(d.max()), int((d > 0).sum()))
    # a property edit that keeps the record count: only the 256-byte chunks that changed travel to the device
    lst = _flatten(subs).layers[0]
    ret.scene_retain(_flatten(subs), w, h)
    node = lst.nodes[lst.rootIds[40]]
    x, y, bw, bh = node.screenBox
    node.screenBox = rect(x + 17.0, y + 9.0, bw, bh)
    ret.scene_update_nodes(0, lst.rootIds[40], [node])
    ret.scene_render()
    assert 0 < ret.last_upload_bytes() <= 8192 < ret.frame_stats().n_draws * 128  # (a record is 128 bytes: the block is ~100 KB)
    ref.render_frame(Renders_of(lst), w, h)
    got = ret.read_pixels()
    assert np.array_equal(got, ref.read_pixels())
    orc.render_frame(Renders_of(lst), w, h)  # after fdh_scene_update_nodes too: the oracle's frame of the edited tree
    d = np.abs(got.astype(int) - orc.read_pixels().astype(int)).max(axis=2)
    assert d.max() <= 1 and (d > 0).sum() <= 0.005 * w * h, (int(d.max()), int((d > 0).sum()))
    ret.close()
    ref.close()
