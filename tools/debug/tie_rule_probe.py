#!/usr/bin/env python3
"""Container-only probe (needs /root/reference + the SwiftShader in the kaleido package): which pixel centres lying EXACTLY on an edge
of a quad does the reference's GL path draw?  (Round 4's verdict: all 26 outliers of the `rotated_tree` golden are such centres on ONE
lower edge -- is the fill rule applied in GL's y-up window space?)

  part 1  one opaque parallelogram with integer vertices whose four edges each pass through pixel centres (edge vectors m x (odd,
          odd)), drawn through drawFilledQuad on white, and its mirror images in x, y and both: every tie pixel classified by where
          the interior lies and by whether SwiftShader drew it.
  part 2  the quad of `rotated_tree` that owns the 26 outliers -- vertices (238, 478), (553, 653), (669, 443), (355, 268), its
          lower-left edge 35 x (9, 5) -- drawn alone in frames of several sizes.

Findings (2026-10, recorded in tests/ref_scenes.py and DESIGN.md section 5):
  1. SwiftShader draws the ties of LEFT edges (interior to the right), upper and lower alike, and none of right edges: the
     top-left rule the oracle and the kernels apply, in either y orientation (a y flip keeps left edges left).  96 of 96 ties.
  2. The SAME quad keeps or loses all its tie pixels with the FRAME SIZE: 900 x 600, 900 x 640, 800 x 600, 1024 x 600: none of
     them drawn; 900 x 700, 900 x 601: all drawn.  The vertices reach the rasteriser through the float32 projection
     (ortho(0, W, H, 0), glcontext.nim:1951-1989) and the viewport transform; 478 -> 1 - 2 * 478 / 600 -> back is not 478 to the
     last bit, and the rasteriser's sub-pixel snap (1/256 px) keeps the residue: the edge misses the centres by < 0.004 px on one
     side or the other.  Which side is a property of W, H and the implementation's arithmetic -- no fill rule reproduces it, and
     another conformant rasteriser (LLVMpipe: other snap, other setup) need not agree with SwiftShader there either."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_swiftshader as SS  # noqa: E402

W, H = 256, 256


def quad(mx, my):
    # edge vectors (45, 25), (-35, 63): 5 x (9, 5) and 7 x (-5, 9) -- both components odd after dividing by the gcd
    p = [(60, 50), (105, 75), (70, 138), (25, 113)]
    if mx:
        p = [(W - x, y) for x, y in p]
    if my:
        p = [(x, H - y) for x, y in p]
    return p


def ties(p):
    out = []
    n = len(p)
    cx = sum(x for x, _ in p) / 4.0
    cy = sum(y for _, y in p) / 4.0
    for k in range(n):
        (x0, y0), (x1, y1) = p[k], p[(k + 1) % n]
        for y in range(min(y0, y1) - 1, max(y0, y1) + 1):
            for x in range(min(x0, x1) - 1, max(x0, x1) + 1):
                px, py = 2 * x + 1, 2 * y + 1
                e = (x1 - x0) * (py - 2 * y0) - (y1 - y0) * (px - 2 * x0)
                if e != 0:
                    continue
                t = ((px / 2 - x0) * (x1 - x0) + (py / 2 - y0) * (y1 - y0)) / float((x1 - x0) ** 2 + (y1 - y0) ** 2)
                if not (0.0 < t < 1.0):
                    continue
                # where the interior lies relative to this pixel: the side of the edge the quad's centre is on, along the pixel's row
                ex = x0 + (py / 2 - y0) * (x1 - x0) / float(y1 - y0)
                cxe = x0 + (cy - y0) * (x1 - x0) / float(y1 - y0)
                horiz = "R" if cx > cxe else "L"
                # interior above (smaller y) or below the edge at this x?
                ey = y0 + (px / 2 - x0) * (y1 - y0) / float(x1 - x0)
                vert = "below" if cy > ey else "above"
                out.append((x, y, horiz, vert))
    return out


def main():
    if not SS.available():
        print("needs /root/reference and SwiftShader")
        return 1
    tally = {}
    for mx in (0, 1):
        for my in (0, 1):
            p = quad(mx, my)
            verts = [c for xy in p for c in xy]
            col = (255, 0, 0, 255)
            calls = [("begin_frame", True, (1.0, 1.0, 1.0, 1.0)), ("draw_filled_quad", verts, [col] * 4), ("end_frame",)]
            img = SS.replay(calls, W, H)
            for x, y, horiz, vert in ties(p):
                drawn = bool(img[y, x, 1] < 128)  # red over white: green drops
                key = (f"interior {horiz}", f"interior {vert}")
                tally.setdefault(key, [0, 0])[0 if drawn else 1] += 1
    print("part 1: edge class of the tie pixel   drawn  not drawn")
    for key in sorted(tally):
        print(f"  {key[0]:12s} {key[1]:16s}  {tally[key][0]:5d}  {tally[key][1]:9d}")
    print("part 2: the rotated_tree quad alone, ties of its lower-left edge (238, 478) - (553, 653) inside the frame")
    p = [(238, 478), (553, 653), (669, 443), (355, 268)]  # BL, BR, TR, TL: the order drawRoundedRectSdf emits (glcontext.nim:1498-1503)
    verts = [c for xy in p for c in xy]
    for fw, fh in [(900, 600), (900, 640), (800, 600), (1024, 600), (900, 601), (900, 700)]:
        img = SS.replay([("begin_frame", True, (1.0, 1.0, 1.0, 1.0)), ("draw_filled_quad", verts, [(255, 0, 0, 255)] * 4), ("end_frame",)], fw, fh)
        drawn = und = 0
        for k in range(1, 35):
            x, y = 238 + 9 * k - 5, 478 + 5 * k - 3  # centre (x + 0.5, y + 0.5) = (238, 478) + (k - 0.5) (9, 5)
            if y >= fh:
                continue
            if img[y, x, 1] < 128:
                drawn += 1
            else:
                und += 1
        print(f"  frame {fw:4d} x {fh:3d}: drawn {drawn:2d}, not drawn {und:2d}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
