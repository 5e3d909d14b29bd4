#!/usr/bin/env python3
"""GPU-box helper: one hash per RS.random_scene seed of the HIP frame (atlas scenes on even seeds).  Run it under two settings of an
environment switch and diff the outputs: a switch that must not change pixels shows no difference.  usage: fuzz_hashes.py LO HI"""
import hashlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ref_scenes as RS
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import load_glyph_fixture
imgs = load_glyph_fixture(os.path.join(ROOT, 'tests', 'golden', 'glyphs_ubuntu20.npz'))
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rnd = random.Random(seed * 7919)
    w, h = rnd.randrange(40, 1400), rnd.randrange(40, 900)
    sc = RS.random_scene(seed, float(w), float(h), n=rnd.randrange(5, 90), clips=rnd.random() < 0.6, blur=rnd.random() < 0.5, images=imgs)
    ctx = HipContext(atlas_size=1024, device=0)
    for k, v in RS.used_images(sc, imgs).items():
        ctx.put_image(k, v)
    ctx.render_frame(sc, w, h)
    print(seed, w, h, hashlib.sha256(ctx.read_pixels().tobytes()).hexdigest()[:16], flush=True)
    ctx.close()
