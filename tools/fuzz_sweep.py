#!/usr/bin/env python3
"""One-off sweep of RS.random_scene seeds: HIP vs oracle, report every seed that exceeds the parity bar.
usage: fuzz_sweep.py LO HI [mx]   (mx: frame widths are multiples of 4 and every blur runs on the matrix-pipe passes)"""
import os, sys
MX = len(sys.argv) > 3 and sys.argv[3] == "mx"
if MX: os.environ.setdefault("FDH_FORCE_BLUR_PATH", "3")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, random
import ref_scenes as RS
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import load_glyph_fixture
from oracle import oracle as O
imgs = load_glyph_fixture(os.path.join(ROOT, 'tests', 'golden', 'glyphs_ubuntu20.npz'))
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    rnd = random.Random(seed * 7919)
    w, h = rnd.randrange(40, 1400), rnd.randrange(40, 900)
    if MX: w = (w + 3) & ~3
    atlas = seed % 2 == 0
    sc = RS.random_scene(seed, float(w), float(h), n=rnd.randrange(5, 90), clips=rnd.random() < 0.6, blur=MX or rnd.random() < 0.5, images=imgs if atlas else None)
    ctx = HipContext(atlas_size=1024, device=0); orc = O.Oracle(atlas_size=1024, threads=16)
    if atlas:
        for k, v in RS.used_images(sc, imgs).items():
            ctx.put_image(k, v); orc.put_image(k, v)
    ctx.render_frame(sc, w, h); orc.render_frame(sc, w, h)
    d = np.abs(ctx.read_pixels().astype(int) - orc.read_pixels().astype(int)).max(axis=2)
    n0 = int((d > 0).sum())
    if d.max() > 1 or n0 > 0.005 * w * h:
        bad += 1
        print("MISMATCH seed", seed, w, h, "atlas" if atlas else "sdf", "max", d.max(), "n>0", n0, "n>1", int((d > 1).sum()))
    ctx.close()
print("swept", hi - lo, "seeds,", bad, "mismatches")
