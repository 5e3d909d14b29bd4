#!/usr/bin/env python3
"""Narrow a fuzz mismatch down to single nodes: render every root subtree of RS.random_scene(seed) alone (over the
background) on the HIP path and on the oracle and report the ones that differ by more than 1 LSB."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import ref_scenes as RS
from figdraw_amd.context import HipContext
from figdraw_amd.scene import RenderList, Renders
from figdraw_amd.scenes import load_glyph_fixture
from oracle import oracle as O

seed, w, h, clips = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == '1'
imgs = load_glyph_fixture(os.path.join(ROOT, 'tests', 'golden', 'glyphs_ubuntu20.npz'))
sc = RS.random_scene(seed, float(w), float(h), n=50, clips=clips, blur=(seed % 2 == 1), images=imgs)
used = RS.used_images(sc, imgs)
ctx = HipContext(atlas_size=1024, device=0); orc = O.Oracle(atlas_size=1024, threads=8)
for k, v in used.items():
    ctx.put_image(k, v); orc.put_image(k, v)
def cmp(r):
    ctx.render_frame(r, w, h); orc.render_frame(r, w, h)
    a, b = ctx.read_pixels().astype(int), orc.read_pixels().astype(int)
    d = np.abs(a - b).max(axis=2)
    return d
d = cmp(sc)
ys, xs = np.nonzero(d > 1)
print("full scene: max", d.max(), "pixels > 1:", len(ys), list(zip(xs[:8], ys[:8])))
lst = sc.layers[0]
def subtree(i):
    out = [i]
    for j, n in enumerate(lst.nodes):
        if n.parent == i:
            out += subtree(j)
    return out
for root in lst.rootIds[1:]:
    ids = sorted(subtree(root))
    r = Renders(); l2 = RenderList()
    l2.addRoot(copy.deepcopy(lst.nodes[lst.rootIds[0]]))
    remap = {}
    for i in ids:
        n = copy.deepcopy(lst.nodes[i]); n.childCount = 0
        remap[i] = l2.addRoot(n) if n.parent == -1 else l2.addChild(remap[n.parent], n)
    r.layers[0] = l2
    dd = cmp(r)
    if dd.max() > 1:
        n = lst.nodes[root]
        print("node", root, n.kind, "rot", n.rotation, "box", n.screenBox, "flags", n.flags, "img", n.image_id, "max", dd.max(), "count", int((dd > 1).sum()),
              "stroke", n.strokeWeight, "fill", n.image_fill.kind if hasattr(n.image_fill, 'kind') else None, "nglyph", len(n.glyphs))
