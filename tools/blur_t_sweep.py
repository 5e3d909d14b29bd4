#!/usr/bin/env python3
"""GPU box: the two passes of the bench frame's full-frame blur (k_blur_mx) by blocks per wave (FDH_MX_T, read once per process).
usage: python3 tools/blur_t_sweep.py [W H]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
code = f"""
import sys; sys.path.insert(0, {ROOT!r})
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
c = HipContext(device=0); c.set_blur_route(0)
c.render_frame(make_render_tree_100({w}, {h}, 0, full_frame_blur=True), {w}, {h}); c.replay(10); c.profile(40)
st = c.frame_stats()
print('H %.2f us  V %.2f us  (frame %.1f us)' % (1e3 * st.ms_blur_big_h, 1e3 * st.ms_blur_big_v, 1e3 * (st.ms_bin + st.ms_composite + st.ms_blur_h + st.ms_blur_v)))
"""
for t in (0, 2, 3, 4, 5, 6, 8, 12):
    env = dict(os.environ)
    if t: env["FDH_MX_T"] = str(t)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"FDH_MX_T={t or 'auto'}: {r.stdout.strip() or r.stderr[-300:]}", flush=True)
