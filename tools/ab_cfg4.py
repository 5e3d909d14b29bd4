#!/usr/bin/env python3
"""GPU-box helper: config 4 (T10k: 5000 coverage glyphs + 5000 MSDF quads at 4K) frame and composite time for several builds.
usage: ab_cfg4.py [reps] name...   (name = build/libfigdraw_hip_<name>.so, or `tree`)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONE = r"""
import sys, os; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np
import ref_scenes as RS
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import load_glyph_fixture, make_glyph_scene
from oracle import oracle as O
imgs = load_glyph_fixture(os.path.join(%r, 'tests', 'golden', 'glyphs_ubuntu20.npz'))
c = HipContext(atlas_size=1024, device=0)
sc = make_glyph_scene(3840, 2160, imgs)
used = RS.used_images(sc, imgs)
for k in sorted(used): c.put_image(k, used[k])
c.render_frame(sc, 3840, 2160); c.replay(20); c.profile(40); s = c.frame_stats(); c.replay(100); t = c.frame_stats().ms_total
msg = ''
if len(sys.argv) > 2:
    o = O.Oracle(atlas_size=1024, threads=16)
    for k in sorted(used): o.put_image(k, used[k])
    o.render_frame(sc, 3840, 2160)
    d = np.abs(c.read_pixels().astype(int) - o.read_pixels().astype(int)).max(axis=2)
    msg = '  vs oracle: max %%d LSB, %%d px differ' %% (d.max(), (d > 0).sum())
print('%%-8s composite %%6.2f  bin %%5.2f  frame %%6.2f us%%s' %% (sys.argv[1], 1e3 * s.ms_composite, 1e3 * s.ms_bin, 1e3 * t, msg))
""" % (ROOT, ROOT, ROOT)
args = sys.argv[1:]
reps = int(args.pop(0)) if args and args[0].isdigit() else 2
for rep in range(reps):
    for n in args:
        lib = os.path.join(ROOT, "figdraw_amd", "libfigdraw_hip.so") if n == "tree" else os.path.join(ROOT, "build", f"libfigdraw_hip_{n}.so")
        subprocess.run([sys.executable, "-c", ONE, n] + (["parity"] if rep == 0 else []), env=dict(os.environ, FIGDRAW_HIP_LIB=lib))
