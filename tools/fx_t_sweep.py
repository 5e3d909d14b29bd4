#!/usr/bin/env python3
"""GPU-box helper: the fused full-frame blur kernel's time against blocks per wave (FDH_FX_T, read once per process: a child per value).
usage: python3 tools/fx_t_sweep.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys; sys.path.insert(0, ROOTDIR)\n"
        "from figdraw_amd.context import HipContext\n"
        "from figdraw_amd.scenes import make_render_tree_100\n"
        "c = HipContext(device=0); c.set_blur_route(1)\n"
        "c.render_frame(make_render_tree_100(3840, 2160, 0, full_frame_blur=True), 3840, 2160); c.replay(20); c.profile(40)\n"
        "st = c.frame_stats(); print('T', TVAL, 'frame %.1f us, blur_h %.1f blur_v %.1f' % (1e3 * st.ms_total, 1e3 * st.ms_blur_h, 1e3 * st.ms_blur_v))\n").replace('ROOTDIR', repr(ROOT))
for t in (0, 3, 4, 5, 6, 7, 8, 10):
    env = dict(os.environ)
    if t: env["FDH_FX_T"] = str(t)
    subprocess.call([sys.executable, "-c", code.replace('TVAL', str(t))], env=env)
