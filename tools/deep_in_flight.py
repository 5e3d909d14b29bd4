#!/usr/bin/env python3
"""GPU-box helper (round 6): does the deep-strip launch cost THROUGHPUT when several contexts keep the GPU full?  BASELINE config 2's tree
(1920 x 1080, or the size given) through fdh_render_frame on 1 and on 4 contexts, frames of the animation in rotation, the loop in C
(tools/call_player.c) -- with the deep strips off (FDH_DEEP_MIN=0) and at the library's default.  One child process per setting.
usage: python3 tools/deep_in_flight.py [width height]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from figdraw_amd import call_stream as CS
    from figdraw_amd.context import HipContext
    from figdraw_amd.scenes import make_render_tree_100

    w, h = int(sys.argv[2]), int(sys.argv[3])
    scenes = [make_render_tree_100(w, h, frame=f).to_c() for f in range(8)]
    P = CS.Player()
    for nctx in (1, 4):
        ctxs = [HipContext(device=0) for _ in range(nctx)]
        P.play_scenes(ctxs, scenes, 60, w, h)
        ts = sorted(P.play_scenes(ctxs, scenes, 200, w, h) for _ in range(5))
        st = ctxs[0].frame_stats()
        print(f"RESULT FDH_DEEP_MIN={os.environ.get('FDH_DEEP_MIN', 'default'):>7} STRIP_MIN={os.environ.get('FDH_DEEP_STRIP_MIN', 'default'):>7}  contexts {nctx}:  {ts[2] / 200 * 1e6:6.1f} us per frame (median of 5 x 200 frames; best {ts[0] / 200 * 1e6:6.1f})   "
              f"= {w * h * 200 / ts[2] / 1e9:6.1f} Gpixel/s   deep bins {st.deep_bins:.0f}")
        for c in ctxs:
            c.close()
    sys.exit(0)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
print(f"# bench tree {w}x{h} through fdh_render_frame, 1 and 4 contexts in flight")
for rep in range(2):
    combos = [{"FDH_DEEP_MIN": "0"}, {}]
    for spec in os.environ.get("COMBOS", "").split():  # e.g. COMBOS="40:24 32:20": FDH_DEEP_MIN:FDH_DEEP_STRIP_MIN
        a, b = spec.split(":")
        combos.append({"FDH_DEEP_MIN": a, "FDH_DEEP_STRIP_MIN": b})
    for env in combos:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(w), str(h)], env={**os.environ, **env}, capture_output=True, text=True)
        for ln in r.stdout.splitlines():
            if ln.startswith("RESULT"):
                print(ln[7:])
        if r.returncode:
            print("FAILED", r.stderr[-800:])
