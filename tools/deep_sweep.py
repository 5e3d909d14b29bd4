#!/usr/bin/env python3
"""GPU-box helper (round 6): the full-frame compositor launch of a bench-tree frame against the deep threshold.
One child process per FDH_DEEP_MIN (the library reads it once).  usage: python3 tools/narrow_sweep.py [width height [blur]] [-- t0 t1 ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
ths = [0, 8, 12, 16, 20, 24, 32, 40, 48]
if "--" in args:
    i = args.index("--")
    ths = [int(v) for v in args[i + 1:]]
    args = args[:i]
if args and args[0] == "--child":
    sys.path.insert(0, ROOT)
    from figdraw_amd.context import HipContext
    from figdraw_amd.scenes import make_render_tree_100

    w, h, blur = int(args[1]), int(args[2]), args[3] == "1"
    ctx = HipContext(device=0)
    sc = make_render_tree_100(w, h, frame=0, full_frame_blur=blur)
    for _ in range(4):
        ctx.render_frame(sc, w, h)
        ctx.sync()
    ctx.replay(20)
    ctx.replay(200)
    frame = ctx.frame_stats().ms_total
    ctx.profile(5)
    ctx.profile(40)
    st = ctx.frame_stats()
    print(f"RESULT {os.environ.get('FDH_DEEP_MIN', 'default'):>7} deep bins {st.deep_bins:6.0f} of {st.n_bins:5d}   composite_main {st.ms_composite_main * 1e3:6.2f} us   "
          f"bin {st.ms_bin * 1e3:5.2f}   frame (replay) {frame * 1e3:6.2f} us")
    sys.exit(0)
w, h = (int(args[0]), int(args[1])) if len(args) >= 2 else (1920, 1080)
blur = "1" if len(args) >= 3 and args[2] != "0" else "0"
print(f"# bench tree {w}x{h}{' + full-frame blur' if blur == '1' else ''}: FDH_DEEP_MIN sweep")
for t in ths:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(w), str(h), blur], env={**os.environ, "FDH_DEEP_MIN": str(t)}, capture_output=True, text=True)
    ln = [x for x in r.stdout.splitlines() if x.startswith("RESULT")]
    print(ln[-1][7:] if ln else ("FAILED " + r.stderr[-500:]))
