#!/usr/bin/env python3
"""GPU-box / container helper: FDH_WALK_TRACE=1 prints when each chunk of a forked sibling group ran on which slot (record-only context)."""
import os, sys
os.environ["FDH_WALK_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
ctx = HipContext(record_only=True)
ctx.set_walk_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
sc = make_render_tree_100(3840, 2160, 3, full_frame_blur=True)
for k in range(30):
    if k == 27: print("--- frames 27..29", file=sys.stderr)
    if k < 27:
        devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(2); os.dup2(devnull, 2)
    ctx.render_frame(sc, 3840, 2160)
    if k < 27:
        os.dup2(saved, 2); os.close(devnull); os.close(saved)
