#!/usr/bin/env python3
"""Where a fuzz seed's HIP / oracle differences lie: bounding box, count per 32-row band, and the same with blur nodes dropped.
usage: fuzz_where.py seed..."""
import os, sys, copy, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import ref_scenes as RS
from figdraw_amd.context import HipContext
from figdraw_amd.scene import FigKind
from figdraw_amd.scenes import load_glyph_fixture
from oracle import oracle as O
imgs = load_glyph_fixture(os.path.join(ROOT, 'tests', 'golden', 'glyphs_ubuntu20.npz'))
for seed in map(int, sys.argv[1:]):
    rnd = random.Random(seed * 7919)
    w, h = rnd.randrange(40, 1400), rnd.randrange(40, 900)
    atlas = seed % 2 == 0
    n = rnd.randrange(5, 90); clips = rnd.random() < 0.6; blur = rnd.random() < 0.5
    sc = RS.random_scene(seed, float(w), float(h), n=n, clips=clips, blur=blur, images=imgs if atlas else None)
    def run(scn):
        ctx = HipContext(atlas_size=1024, device=0); orc = O.Oracle(atlas_size=1024, threads=16)
        if atlas:
            for k, v in RS.used_images(scn, imgs).items():
                ctx.put_image(k, v); orc.put_image(k, v)
        ctx.render_frame(scn, w, h); orc.render_frame(scn, w, h)
        st = ctx.frame_stats()
        d = np.abs(ctx.read_pixels().astype(int) - orc.read_pixels().astype(int)).max(axis=2)
        ctx.close()
        return d, st
    d, st = run(sc)
    ys, xs = np.nonzero(d)
    kinds = {}
    for nd in sc.layers[0].nodes: kinds[nd.kind.name] = kinds.get(nd.kind.name, 0) + 1
    print(f"seed {seed} {w}x{h} n={n} clips={clips} blur={blur} draws={st.n_draws} phases={st.n_phases} blurs={st.n_blurs}: differing {len(ys)} max {d.max()} bbox x {xs.min()}..{xs.max()} y {ys.min()}..{ys.max()}  kinds {kinds}")
    for nd in sc.layers[0].nodes:
        if nd.kind == FigKind.nkBackdropBlur:
            print("   blur node box", nd.screenBox, "radius", getattr(nd, 'blur', None), getattr(nd, 'backdropBlur', None))
    sc2 = copy.deepcopy(sc)
    lst = sc2.layers[0]
    for nd in lst.nodes:
        if nd.kind == FigKind.nkBackdropBlur: nd.kind = FigKind.nkFrame
    d2, _ = run(sc2)
    print("   with blur nodes turned into frames: differing", int((d2 > 0).sum()), "max", d2.max())
