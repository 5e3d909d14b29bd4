"""Workload generators for the BASELINE.json configs.

`make_render_tree_100` restates examples/renderlist_100_common.nim:11-251 (the
"300 boxes with shadows" demo): per copy one red elliptical-corner rect with a
stroke, one green rect with a drop shadow (3-stop gradient on even copies) and
one blue rect with a stroke and an inset shadow; then an elliptical orange
pill, a 360x240 backdrop-blur node and a yellow overlay.

The reference draws the two per-copy base positions from Nim's
`initRand(12345)` (xoroshiro128+), which is not reproducible without a Nim
runtime; they are drawn from PCG32(seed 12345) instead.  All
other parameters are the closed-form float32 expressions of the reference.
"""
from __future__ import annotations

import math

import numpy as np

from .scene import (Fig, FigFlags, FigKind, FillGradientAxis, RenderList, Renders, RenderShadow, RenderStroke,
                    ShadowStyle, fill, linear, rect, rgba)

f32 = np.float32


class PCG32:
    """Minimal PCG-XSH-RR 64/32 (O'Neill 2014), used only to place the copies."""

    def __init__(self, seed: int, seq: int = 54):
        self.state = 0
        self.inc = ((seq << 1) | 1) & 0xFFFFFFFFFFFFFFFF
        self.next_u32()
        self.state = (self.state + seed) & 0xFFFFFFFFFFFFFFFF
        self.next_u32()

    def next_u32(self) -> int:
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & 0xFFFFFFFFFFFFFFFF
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self, hi: float) -> np.float32:
        return f32(f32(self.next_u32() >> 8) * f32(1.0 / 16777216.0) * f32(hi))


def _sin(x):
    return f32(math.sin(float(f32(x))))


def _cos(x):
    return f32(math.cos(float(f32(x))))


def _u16(x) -> int:  # Nim uint16(float) truncates
    return int(f32(x)) & 0xFFFF


def make_render_tree_100(w: float, h: float, frame: int = 0, copies: int = 100, full_frame_blur: bool = False,
                         seed: int = 12345, full_frame_blur_radius: float = 18.0) -> Renders:
    w, h = f32(w), f32(h)
    lst = RenderList()
    t = f32(f32(frame) * f32(0.02))
    lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 155)))

    redStartX, redStartY = f32(60), f32(60)
    greenStartX, greenStartY = f32(320), f32(120)
    blueStartX, blueStartY = f32(180), f32(300)
    maxW, maxH = f32(260), f32(180)
    maxX = max(f32(0), f32(w - (greenStartX + maxW)))
    maxY = max(f32(0), f32(h - (blueStartY + maxH)))
    rng = PCG32(seed)

    for i in range(copies):
        fi = f32(i)
        baseX = rng.uniform(maxX)
        baseY = rng.uniform(maxY)
        jitterX = f32(_sin(t + fi * f32(0.15)) * f32(20))
        jitterY = f32(_cos(t * f32(0.9) + fi * f32(0.2)) * f32(20))
        offsetX = min(max(f32(baseX + jitterX), f32(0)), maxX)
        offsetY = min(max(f32(baseY + jitterY), f32(0)), maxY)
        sizePulseW = f32(f32(0.5) + f32(0.5) * _sin(t * f32(0.8) + fi * f32(0.07)))
        sizePulseH = f32(f32(0.5) + f32(0.5) * _cos(t * f32(0.65) + fi * f32(0.09)))
        redW = f32(f32(160) + f32(100) * sizePulseW)
        redH = f32(f32(110) + f32(70) * sizePulseH)
        greenW = f32(f32(160) + f32(100) * sizePulseH)
        greenH = f32(f32(110) + f32(70) * sizePulseW)
        blueW = f32(f32(160) + f32(100) * (f32(1) - sizePulseW))
        blueH = f32(f32(110) + f32(70) * (f32(1) - sizePulseH))
        cornerPulse = f32(f32(0.5) + f32(0.5) * _sin(t * f32(1.25) + fi * f32(0.11)))
        c0 = f32(f32(4) + f32(26) * cornerPulse)
        c1 = f32(f32(6) + f32(22) * (f32(1) - cornerPulse))
        c2 = f32(f32(8) + f32(18) * (f32(0.5) + f32(0.5) * _sin(t * f32(0.7) + fi * f32(0.05))))
        c3 = f32(f32(10) + f32(16) * (f32(0.5) + f32(0.5) * _cos(t * f32(0.8) + fi * f32(0.06))))
        greenCornerPulse = f32(f32(0.5) + f32(0.5) * _cos(t * f32(0.95) + fi * f32(0.08)))
        g0 = f32(f32(6) + f32(22) * greenCornerPulse)
        g1 = f32(f32(8) + f32(18) * (f32(1) - greenCornerPulse))
        g2 = f32(f32(10) + f32(16) * (f32(0.5) + f32(0.5) * _cos(t * f32(0.75) + fi * f32(0.04))))
        g3 = f32(f32(12) + f32(14) * (f32(0.5) + f32(0.5) * _sin(t * f32(0.85) + fi * f32(0.05))))
        shadowPulse = f32(f32(0.5) + f32(0.5) * _sin(t * f32(1.1) + fi * f32(0.05)))
        shadowBlur = max(f32(0), f32(f32(6) + f32(18) * shadowPulse))
        shadowSpread = max(f32(0), f32(f32(4) + f32(20) * (f32(1) - shadowPulse)))
        shadowX = f32(f32(6) + f32(10) * _sin(t * f32(0.9) + fi * f32(0.03)))
        shadowY = f32(f32(6) + f32(10) * _cos(t * f32(0.9) + fi * f32(0.03)))
        insetPulse = f32(f32(0.5) + f32(0.5) * _sin(t * f32(1.05) + fi * f32(0.06)))
        insetBlur = max(f32(0), f32(f32(8) + f32(10) * insetPulse))
        insetSpread = max(f32(0), f32(f32(2) + f32(10) * (f32(1) - insetPulse)))
        insetX = f32(f32(6) * _sin(t * f32(0.85) + fi * f32(0.04)))
        insetY = f32(f32(6) * _cos(t * f32(0.8) + fi * f32(0.04)))
        useGreenGradient = (i % 2) == 0
        useBlueGradient = (i % 3) == 0

        lst.addRoot(Fig(
            kind=FigKind.nkRectangle,
            corners=[_u16(c0), _u16(c1), _u16(c2), _u16(c3)],
            cornerRadiiY=[_u16(c0), _u16(f32(c1 * f32(2))), _u16(c2), _u16(f32(c3 * f32(2)))],
            flags=FigFlags.NfEllipticalCorners,
            screenBox=rect(f32(redStartX + offsetX), f32(redStartY + offsetY), redW, redH),
            fill=rgba(220, 40, 40, 155),
            stroke=RenderStroke(weight=5.0, fill=fill(rgba(0, 0, 0, 155))),
        ))
        lst.addRoot(Fig(
            kind=FigKind.nkRectangle,
            screenBox=rect(f32(greenStartX + offsetX), f32(greenStartY + offsetY), greenW, greenH),
            corners=[_u16(g0), _u16(g1), _u16(g2), _u16(g3)],
            fill=(linear(rgba(18, 112, 64, 255), rgba(40, 180, 90, 255), rgba(78, 224, 188, 255),
                         axis=FillGradientAxis.fgaX if (i % 4) < 2 else FillGradientAxis.fgaDiagTLBR, midPos=128)
                  if useGreenGradient else fill(rgba(40, 180, 90, 155))),
            shadows=[RenderShadow(style=ShadowStyle.DropShadow, blur=shadowBlur, spread=shadowSpread, x=shadowX,
                                  y=shadowY, fill=fill(rgba(0, 0, 0, 155)))],
        ))
        lst.addRoot(Fig(
            kind=FigKind.nkRectangle,
            screenBox=rect(f32(blueStartX + offsetX), f32(blueStartY + offsetY), blueW, blueH),
            fill=(linear(rgba(44, 72, 186, 255), rgba(60, 90, 220, 255), rgba(118, 168, 255, 255),
                         axis=FillGradientAxis.fgaY if (i % 2) == 0 else FillGradientAxis.fgaDiagBLTR, midPos=132)
                  if useBlueGradient else fill(rgba(60, 90, 220, 155))),
            stroke=RenderStroke(weight=4.0, fill=fill(rgba(255, 255, 255, 210))),
            shadows=[RenderShadow(
                style=ShadowStyle.InnerShadow, blur=insetBlur, spread=insetSpread, x=insetX, y=insetY,
                fill=(linear(rgba(25, 25, 40, 100), rgba(65, 65, 95, 180), axis=FillGradientAxis.fgaDiagBLTR)
                      if useBlueGradient else fill(rgba(40, 40, 60, 150))))],
        ))

    lst.addRoot(Fig(
        kind=FigKind.nkRectangle,
        screenBox=rect(max(f32(20), f32(w - f32(200))), 20, 180, 100),
        fill=rgba(238, 140, 30, 220),
        corners=[90, 90, 90, 90], cornerRadiiY=[50, 50, 50, 50], flags=FigFlags.NfEllipticalCorners,
        stroke=RenderStroke(weight=4.0, fill=fill(rgba(90, 45, 0, 220))),
    ))

    if full_frame_blur:
        # one full-frame nkBackdropBlur(18) after the rects, before the overlay
        lst.addRoot(Fig(kind=FigKind.nkBackdropBlur, screenBox=rect(0, 0, w, h), fill=rgba(0, 0, 0, 0), blur=full_frame_blur_radius))

    yellowW, yellowH, yellowMargin = f32(360), f32(240), f32(20)
    yellowTravelX = max(f32(0), f32(w - yellowW - yellowMargin * f32(2)))
    yellowTravelY = max(f32(0), f32(h - yellowH - yellowMargin * f32(2)))
    yellowX = f32(yellowMargin + yellowTravelX * (f32(0.5) + f32(0.5) * _sin(t * f32(0.33))))
    yellowY = f32(yellowMargin + yellowTravelY * (f32(0.5) + f32(0.5) * _cos(t * f32(0.41))))
    yellowCorner = f32(f32(20) + f32(12) * (f32(0.5) + f32(0.5) * _sin(t * f32(0.7))))
    yc = _u16(yellowCorner)
    lst.addRoot(Fig(kind=FigKind.nkBackdropBlur, corners=[yc] * 4, screenBox=rect(yellowX, yellowY, yellowW, yellowH),
                    fill=rgba(0, 0, 0, 0), blur=18.0))
    lst.addRoot(Fig(kind=FigKind.nkRectangle, corners=[yc] * 4, screenBox=rect(yellowX, yellowY, yellowW, yellowH),
                    fill=rgba(255, 225, 55, 120), stroke=RenderStroke(weight=6.0, fill=fill(rgba(95, 72, 0, 185)))))

    out = Renders()
    out.setLayer(0, lst)
    return out


def make_rotated_tree(w: float, h: float, frame: int = 0, copies: int = 100) -> Renders:
    """The renderlist_100 tree with every rectangle rotated about its centre (Fig.rotation, figrender.nim:1768-1777): each draw
    becomes a rotated quad -- per-vertex ceil, two-triangle rasterisation, the compositor's one-pixel-slot build.  A workload
    of its own for tools/perf_configs.py (config 9); no backdrop blur (the blur kernels do not care about rotation)."""
    out = make_render_tree_100(w, h, frame, copies=copies)
    lst = out.layers[0]
    keep = []
    for i, n in enumerate(lst.nodes):
        if n.kind == FigKind.nkBackdropBlur:
            continue
        if i > 0:
            n.rotation = float(((i * 37) % 61) - 30)
        keep.append(n)
    lst.nodes = keep
    lst.rootIds = list(range(len(keep)))
    return out


def make_curves_scene(w: float, h: float, n: int = 1500, seed: int = 7, only_kind: int = -1, rotation: float = 0.0) -> Renders:
    """n stroked curves and lines (nkDrawable: quadratic beziers -> drawQuadraticBezierSdf, modes 18 - 20; lines -> rotated boxes;
    arcs with joins -> filled quads) scattered over the frame: config 10 of tools/perf_configs.py."""
    from .scene import StrokeCap, StrokeJoin, drawableArc, drawableBezier, drawableLine

    rng = PCG32(seed)
    lst = RenderList()
    lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, w, h), fill=rgba(250, 250, 246, 255)))
    for i in range(n):
        x, y = float(rng.uniform(f32(w - 260))), float(rng.uniform(f32(h - 200)))
        col = fill(rgba(int(rng.next_u32() & 255), int(rng.next_u32() & 255), int(rng.next_u32() & 255), 200 + (i % 56)))
        weight = 2.0 + float(i % 7)
        kind = i % 4 if only_kind < 0 else only_kind
        if kind == 0:
            ops = [drawableBezier([(5, 10), (120 + (i % 40), 170), (235, 20 + (i % 90))])]
        elif kind == 1:
            ops = [drawableBezier([(10, 150), (60, -40), (170, 220), (240, 60)])]  # a cubic: adaptive quadratic spans
        elif kind == 2:
            ops = [drawableLine((8, 12 + (i % 60)), (230, 150 - (i % 80)))]
        else:
            ops = [drawableArc((120, 90), 70.0, 0.2 * (i % 9), 3.2, steps=6)]
        lst.addRoot(Fig(kind=FigKind.nkDrawable, screenBox=rect(x, y, 250, 190),
                        drawStroke=RenderStroke(weight=weight, fill=col, cap=[StrokeCap.scButt, StrokeCap.scRound, StrokeCap.scSquare][i % 3],
                                                join=[StrokeJoin.sjBevel, StrokeJoin.sjMiter, StrokeJoin.sjRound][(i // 3) % 3]),
                        drawOps=ops, rotation=rotation * (1 + i % 3)))
    out = Renders()
    out.setLayer(0, lst)
    return out


# ---------------------------------------------------------------------------------------------------------------------
def load_glyph_fixture(path):
    """tests/golden/glyphs_ubuntu20.npz -> {image_id: (h, w, 4) uint8}.  Ids: 1000+code = coverage glyph (20 px
    Ubuntu, premultiplied white), 2000+code = 32x32 MSDF/MTSDF texture, 3000 = the reference's data/img1.png."""
    z = np.load(path)
    images = {}
    for code in range(33, 127):
        images[1000 + code] = z[f"cov_{code}"]
        images[2000 + code] = z[f"msdf_{code}"]
    images[3000] = z["img1_premul"]
    return images


def make_glyph_scene(w: float, h: float, images, cols: int = 100, rows: int = 100, pitch=(38.0, 21.0), origin=(8.0, 6.0),
                     msdf_size: float = 48.0, rotation: float = 0.0) -> Renders:
    """BASELINE.json configs[3] "T10k@4K" (SURVEY.md 8d #4): cols x rows glyph quads (ASCII 33..126 cycling) over a
    3-stop gradient background; even cells are coverage glyphs from the atlas drawn 1:1 at integer positions with a
    2-stop vertical tint (mode 0, what renderText emits: figrender.nim:456-496), odd cells are MSDF images drawn
    magnified (mode 13, pxRange 4, threshold 0.5: renderMsdfImage figrender.nim:1686-1708)."""
    from .scene import Glyph

    lst = RenderList()
    lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, w, h),
                    fill=linear(rgba(250, 250, 255, 255), rgba(225, 235, 250, 255), rgba(255, 240, 225, 255),
                                axis=FillGradientAxis.fgaDiagTLBR, midPos=110)))
    top, bottom = rgba(20, 30, 120, 255), rgba(160, 20, 40, 255)
    tint = [bottom, bottom, top, top]  # BL, BR, TR, TL of a Y gradient (gradientColors figbackend.nim:169-173)
    for r in range(rows):
        y = origin[1] + r * pitch[1]
        glyphs = []
        for c in range(cols):
            i = r * cols + c
            code = 33 + i % 94
            x = origin[0] + c * pitch[0]
            if i % 2 == 0:
                gh = images[1000 + code].shape[0]
                glyphs.append(Glyph(image_id=1000 + code, x=float(x), y=float(20 - gh), colors=tint))
        lst.addRoot(Fig(kind=FigKind.nkText, screenBox=rect(0, y, w, pitch[1]), glyphs=glyphs, rotation=rotation))  # (rotation: config 11 of tools/perf_configs.py)
        for c in range(cols):
            i = r * cols + c
            if i % 2 == 1:
                code = 33 + i % 94
                x = origin[0] + c * pitch[0]
                lst.addRoot(Fig(kind=FigKind.nkMsdfImage, screenBox=rect(x - 6.0, y - 14.0, msdf_size, msdf_size),
                                image_id=2000 + code, image_fill=fill(rgba(10, 90, 40, 230)), pxRange=4.0, sdThreshold=0.5, rotation=rotation * 3.0))
    out = Renders()
    out.setLayer(0, lst)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The reference's own two benchmark workloads (the only scenes it times itself), restated node for node in float32.
def make_non_clip_benchmark(w: float = 1200.0, h: float = 800.0, rows: int = 180, cols: int = 10) -> Renders:
    """examples/windy_non_clip_benchmark.nim:82-108 `makeNonClipRenderTree`: a background and rows x cols rounded cells, all
    roots, no clips (defaults :9-17: 180 x 10 cells in a 1200 x 800 window; 20 warm-up + 120 timed frames).  Most rows lie below
    the window: the reference submits them all the same."""
    w, h = f32(w), f32(h)
    margin, gap = f32(18.0), f32(5.0)
    cellW = f32(f32(f32(w - f32(margin * f32(2.0))) - f32(gap * f32(cols - 1))) / f32(cols))
    cellH = f32(18.0)
    lst = RenderList()
    lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(0.0, 0.0, w, h), fill=rgba(248, 249, 251, 255)))
    for row in range(rows):
        y = f32(margin + f32(f32(row) * f32(cellH + gap)))
        for col in range(cols):
            x = f32(margin + f32(f32(col) * f32(cellW + gap)))
            shade = (220 + (row * 3 + col * 7) % 35) & 255
            accent = (80 + (row * 11 + col * 13) % 90) & 255
            lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(x, y, cellW, cellH), corners=[4] * 4,
                            fill=rgba(shade, (245 - (col % 5) * 5) & 255, accent, 255)))
    out = Renders()
    out.setLayer(0, lst)
    return out


def make_clip_mask_benchmark(kind: str, w: float = 1200.0, h: float = 800.0, rows: int = 180, cols: int = 6) -> Renders:
    """examples/windy_clip_mask_benchmark.nim:147-186 `makeTableRenderTree` + `addCellContent` :104-145: a clipping viewport
    (NfClipContent, radius 10) holding rows x cols cells, each clipping its three overflowing children with its own
    NfClipContent (kind = "sub_clip": a second mask level per cell) or NfRectMaskContent (kind = "rect_mask": the analytic
    rect mask).  Defaults :9-14: 180 x 6 cells, 1200 x 800, scrolled by 37 px."""
    assert kind in ("sub_clip", "rect_mask")
    w, h = f32(w), f32(h)
    margin, gap = f32(22.0), f32(4.0)
    vx, vy, vw, vh = margin, margin, f32(w - f32(margin * f32(2.0))), f32(h - f32(margin * f32(2.0)))
    cellH = f32(22.0)
    cellW = f32(f32(vw - f32(gap * f32(cols + 1))) / f32(cols))
    scrollY = f32(37.0)
    cell_flag = FigFlags.NfClipContent if kind == "sub_clip" else FigFlags.NfRectMaskContent
    lst = RenderList()
    lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(0.0, 0.0, w, h), fill=rgba(248, 249, 251, 255)))
    viewport = lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(vx, vy, vw, vh), fill=rgba(232, 235, 240, 255),
                               flags=FigFlags.NfClipContent, corners=[10] * 4))
    for row in range(rows):
        y = f32(f32(f32(vy + gap) + f32(f32(row) * f32(cellH + gap))) - scrollY)
        for col in range(cols):
            x = f32(f32(vx + gap) + f32(f32(col) * f32(cellW + gap)))
            cell_color = rgba(255, 255, 255, 255) if (row + col) % 2 == 0 else rgba(242, 246, 250, 255)
            cell = lst.addChild(viewport, Fig(kind=FigKind.nkRectangle, screenBox=rect(x, y, cellW, cellH), fill=cell_color,
                                              flags=cell_flag, corners=[4] * 4))
            tone = (42 + (row * 7 + col * 17) % 72) & 255
            accent = rgba(36, (120 + (row * 5) % 80) & 255, 235, 255)
            spill = rgba(tone, (170 - (col * 11) % 70) & 255, 220, 255)
            muted = rgba((190 + (row + col) % 30) & 255, 210, 220, 255)
            lst.addChild(cell, Fig(kind=FigKind.nkRectangle, fill=accent, corners=[2] * 4,
                                   screenBox=rect(f32(x - f32(12.0)), f32(y + f32(4.0)), f32(cellW + f32(24.0)), 5.0)))
            lst.addChild(cell, Fig(kind=FigKind.nkRectangle, fill=spill, corners=[3] * 4,
                                   screenBox=rect(f32(x + f32(cellW * f32(0.38))), f32(y - f32(5.0)), f32(cellW * f32(0.74)), f32(cellH + f32(10.0)))))
            lst.addChild(cell, Fig(kind=FigKind.nkRectangle, fill=muted, corners=[2] * 4,
                                   screenBox=rect(f32(x + f32(7.0)), f32(f32(y + cellH) - f32(7.0)), f32(cellW - f32(14.0)), 8.0)))
    out = Renders()
    out.setLayer(0, lst)
    return out
