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


@pytest.mark.gpu
def test_retained_scene_pixels_equal_full_render():
    """GPU: after each edit the retained context's frame equals a full fdh_render_frame of the edited tree, bit for bit"""
    rnd = random.Random(11)
    w, h = 1280, 720
    sc = make_render_tree_100(w, h, frame=2, full_frame_blur=True)
    subs = _subtrees(next(iter(sc.layers.values())))
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
        assert np.array_equal(ret.read_pixels(), ref.read_pixels()), step
        walked, reused = ret.scene_stats()
        assert walked <= 4 and reused >= len(subs) - 4
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
    assert np.array_equal(ret.read_pixels(), ref.read_pixels())
    ret.close()
    ref.close()
