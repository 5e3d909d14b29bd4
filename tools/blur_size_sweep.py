#!/usr/bin/env python3
"""GPU-box helper: full-frame backdrop blur (radius 18) at several frame sizes on each blur build (FDH_FORCE_BLUR_PATH in a
child process): H + V time from the context's event spans -- where the region-size threshold between the builds belongs."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
SIZES = [(360, 240), (640, 360), (960, 540), (1280, 720), (1920, 1080), (2560, 1440)]
if len(sys.argv) > 1:
    from figdraw_amd.context import HipContext
    from figdraw_amd.scene import Fig, FigKind, RenderList, Renders, rect, rgba
    ctx = HipContext(device=0)
    out = []
    for w, h in SIZES:
        lst = RenderList()
        lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, w, h), fill=rgba(250, 250, 250, 255)))
        for i in range(20):
            lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect((i * 131) % w, (i * 77) % h, 100, 60), fill=rgba(40 * (i % 6), 200 - 30 * (i % 5), 90, 255)))
        lst.addRoot(Fig(kind=FigKind.nkBackdropBlur, screenBox=rect(0, 0, w, h), fill=rgba(0, 0, 0, 0), blur=18.0))
        sc = Renders(); sc.setLayer(0, lst)
        ctx.render_frame(sc, w, h); ctx.replay(5); ctx.profile(20); st = ctx.frame_stats()
        out.append("%dx%d: %.1f" % (w, h, (st.ms_blur_h + st.ms_blur_v) * 1e3))
    print("%-22s H+V us: %s" % (sys.argv[1], "; ".join(out))); sys.exit(0)
for path, name in ((1, "2 outputs/thread"), (2, "8-12 outputs/thread"), (3, "matrix pipe")):
    subprocess.check_call([sys.executable, __file__, name], env={**os.environ, "FDH_FORCE_BLUR_PATH": str(path)})
