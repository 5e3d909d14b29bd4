#!/usr/bin/env python3
"""Builds tests/golden/outlines_ubuntu20.npz (container only: reads the reference's data/Ubuntu.ttf with fontTools).

For ASCII 33..126 at 20 px: the glyph's outline as quadratic segments in pixel units of a w x h image (y down, the glyph's
bounding box placed at a fractional offset so that edges do not sit on pixel boundaries), in the format fdh_put_glyph_outline
takes: n x 6 float32 {x0, y0, cx, cy, x1, y1}, cx = NaN for a straight line.

  segs_<code>  (n, 6) float32      size_<code>  (2,) int32 = (w, h)
Data only: point coordinates read from the font file, no code of the reference."""
import os
import sys

import numpy as np
from fontTools.pens.recordingPen import DecomposingRecordingPen
from fontTools.ttLib import TTFont

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FONT = "/root/reference/data/Ubuntu.ttf"
PX = 20.0


def segments(font, gs, name, scale):
    pen = DecomposingRecordingPen(gs)
    gs[name].draw(pen)
    segs, cur, start = [], None, None
    for op, args in pen.value:
        if op == "moveTo":
            cur = start = args[0]
        elif op == "lineTo":
            segs.append((cur, None, args[0]))
            cur = args[0]
        elif op == "qCurveTo":  # TrueType: off-curve points with implied on-curve midpoints, last = on-curve (or None)
            pts = list(args)
            if pts[-1] is None:  # closed contour of off-curve points only
                pts = pts[:-1]
                first = ((pts[-1][0] + pts[0][0]) / 2, (pts[-1][1] + pts[0][1]) / 2)
                cur = start = first
                pts.append(first)
            for i in range(len(pts) - 1):
                c = pts[i]
                end = pts[i + 1] if i == len(pts) - 2 else ((pts[i][0] + pts[i + 1][0]) / 2, (pts[i][1] + pts[i + 1][1]) / 2)
                segs.append((cur, c, end))
                cur = end
        elif op in ("closePath", "endPath"):
            if cur is not None and start is not None and cur != start:
                segs.append((cur, None, start))
            cur = start = None
    return [((a[0] * scale, a[1] * scale), None if c is None else (c[0] * scale, c[1] * scale), (b[0] * scale, b[1] * scale)) for a, c, b in segs]


def main():
    font = TTFont(FONT)
    gs = font.getGlyphSet()
    cmap = font.getBestCmap()
    scale = PX / font["head"].unitsPerEm
    arrays = {}
    for code in range(33, 127):
        sg = segments(font, gs, cmap[code], scale)
        if not sg:
            continue
        pts = [p for s in sg for p in s if p is not None]
        x0, x1 = min(p[0] for p in pts), max(p[0] for p in pts)
        y0, y1 = min(p[1] for p in pts), max(p[1] for p in pts)
        ox, oy = 1.0 + 0.3 * ((code * 7) % 3) - x0, 1.0 + 0.25 * ((code * 5) % 4) + y1  # y flips: font units are y-up
        w, h = int(np.ceil(x1 + ox)) + 2, int(np.ceil(oy - y0)) + 2
        out = np.zeros((len(sg), 6), np.float32)
        for i, (a, c, b) in enumerate(sg):
            out[i] = (a[0] + ox, oy - a[1], np.nan if c is None else c[0] + ox, np.nan if c is None else oy - c[1], b[0] + ox, oy - b[1])
        arrays[f"segs_{code}"] = out
        arrays[f"size_{code}"] = np.array([w, h], np.int32)
    path = os.path.join(ROOT, "tests", "golden", "outlines_ubuntu20.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes;", len(arrays) // 2, "glyphs")


if __name__ == "__main__":
    main()
