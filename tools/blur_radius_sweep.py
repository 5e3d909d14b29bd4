#!/usr/bin/env python3
"""GPU-box helper: full-frame backdrop blur at 4K for several radii, on each blur build (FDH_FORCE_BLUR_PATH in a child
process): H + V time per frame from the context's own event spans.  usage: python3 tools/blur_radius_sweep.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
RADII = [4.0, 9.0, 18.0, 28.0, 40.0, 64.0]
if len(sys.argv) > 1:
    from figdraw_amd.context import HipContext
    from figdraw_amd.scene import Fig, FigKind, RenderList, Renders, rect, rgba
    ctx = HipContext(device=0); w, h = 3840, 2160
    out = []
    for r in RADII:
        lst = RenderList()
        lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, w, h), fill=rgba(250, 250, 250, 255)))
        for i in range(60):
            lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect((i * 531) % w, (i * 377) % h, 300, 200), fill=rgba(40 * (i % 6), 200 - 30 * (i % 5), 90, 255)))
        lst.addRoot(Fig(kind=FigKind.nkBackdropBlur, screenBox=rect(0, 0, w, h), fill=rgba(0, 0, 0, 0), blur=r))
        sc = Renders(); sc.setLayer(0, lst)
        ctx.render_frame(sc, w, h); ctx.replay(5); ctx.profile(20); st = ctx.frame_stats()
        out.append("r=%g: H %.1f V %.1f us" % (r, st.ms_blur_h * 1e3, st.ms_blur_v * 1e3))
    print(sys.argv[1], "; ".join(out)); sys.exit(0)
for path, name in ((2, "packed-FMA"), (3, "matrix pipe")):
    subprocess.check_call([sys.executable, __file__, name], env={**os.environ, "FDH_FORCE_BLUR_PATH": str(path)})
