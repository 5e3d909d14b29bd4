#!/usr/bin/env python3
"""GPU-box helper: the HIP frame of tests/ref_scenes.py's atlas golden scenes -> gpurun_out/hip_<name>.npy (for analysing the
differences to the SwiftShader goldens off the box)."""
import os, sys
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import ref_scenes as RS
from conftest import GOLDEN
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import load_glyph_fixture
for name in sorted(RS.ATLAS_SCENES):
    fn, w, h = RS.ATLAS_SCENES[name]
    imgs = load_glyph_fixture(os.path.join(GOLDEN, "glyphs_ubuntu20.npz"))
    sc = fn(float(w), float(h), imgs)
    ctx = HipContext(atlas_size=RS.ATLAS_GOLDEN_SIZE, device=0)
    for k, img in RS.used_images(sc, imgs).items():
        ctx.put_image(k, img)
    ctx.render_frame(sc, w, h)
    np.save(os.path.join(root, "gpurun_out", f"hip_{name}.npy"), ctx.read_pixels())
    ctx.close()
