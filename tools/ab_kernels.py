#!/usr/bin/env python3
"""GPU-box helper: per-kernel times of the bench frame (fdh_profile) for several builds of the library, round-robin.
usage: ab_kernels.py [reps] name...   (name = build/libfigdraw_hip_<name>.so, or `tree` for the working tree's library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ONE = r"""
import sys; sys.path.insert(0, %r)
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
w, h = 3840, 2160
c = HipContext(device=0); c.render_frame(make_render_tree_100(w, h, frame=0, full_frame_blur=True), w, h); c.replay(30); c.profile(60); s = c.frame_stats()
c.replay(200); t = c.frame_stats().ms_total
print("%%-8s composite_main %%6.2f  composite_all %%6.2f  blur_h %%6.2f  blur_v %%6.2f  blur_fused %%6.2f  bin %%5.2f  frame %%6.2f us" %% (sys.argv[1], 1e3 * s.ms_composite_main, 1e3 * s.ms_composite, 1e3 * s.ms_blur_h, 1e3 * s.ms_blur_v, 1e3 * s.ms_blur_fused, 1e3 * s.ms_bin, 1e3 * t))
""" % ROOT
args = sys.argv[1:]
reps = int(args.pop(0)) if args and args[0].isdigit() else 2
for _ in range(reps):
    for n in args:
        lib = os.path.join(ROOT, "figdraw_amd", "libfigdraw_hip.so") if n == "tree" else os.path.join(ROOT, "build", f"libfigdraw_hip_{n}.so")
        subprocess.run([sys.executable, "-c", ONE, n], env=dict(os.environ, FIGDRAW_HIP_LIB=lib))
