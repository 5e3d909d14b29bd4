"""Scenes of the reference's own pixel tests, restated with the reference's names,
plus extra scenes that exercise the rest of SURVEY.md §8(a) (no reference PNG
exists for those; their goldens come from the reference shaders on SwiftShader).

Coordinates are written as explicit numbers (the
reference's `w * 0.30'f32`-style expressions only reproduce its golden PNG when
evaluated in float64).
"""
from __future__ import annotations

from figdraw_amd.scene import (Fig, FigFlags, FigKind, FillGradientAxis, RenderList, Renders, RenderShadow,
                               RenderStroke, ShadowStyle, StrokeCap, StrokeJoin, drawableArc, drawableBezier,
                               drawableCircle, drawableEllipse, drawableLine, drawableRect, figCircle, figLine, fill,
                               linear, rect, rgba)

DROP, INNER = ShadowStyle.DropShadow, ShadowStyle.InnerShadow
RECT = FigKind.nkRectangle


def rgb_boxes_sdf(w=800.0, h=600.0) -> Renders:
    """tests/trender_rgb_boxes_sdf.nim:13-101 (golden: tests/expected/render_rgb_boxes_sdf.png)."""
    lst = RenderList()
    root = lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    lst.addChild(root, Fig(kind=RECT, corners=[10, 20, 30, 40], screenBox=rect(60, 60, 220, 140),
                           fill=rgba(220, 40, 40, 255), stroke=RenderStroke(weight=5.0, fill=fill(rgba(0, 0, 0, 255)))))
    lst.addChild(root, Fig(
        kind=RECT, screenBox=rect(320, 120, 220, 140),
        fill=linear(rgba(24, 128, 72, 255), rgba(40, 180, 90, 255), rgba(54, 206, 170, 255),
                    axis=FillGradientAxis.fgaX, midPos=140),
        shadows=[RenderShadow(style=DROP, blur=10, spread=10, x=10, y=10, fill=fill(rgba(0, 0, 0, 55)))]))
    lst.addChild(root, Fig(
        kind=RECT, screenBox=rect(180, 300, 220, 140), fill=rgba(60, 90, 220, 255),
        shadows=[
            RenderShadow(style=INNER, blur=12, spread=0, x=-6, y=-6,
                         fill=linear(rgba(25, 25, 25, 90), rgba(65, 65, 65, 175), axis=FillGradientAxis.fgaDiagTLBR)),
            RenderShadow(style=INNER, blur=12, spread=0, x=6, y=6,
                         fill=linear(rgba(255, 255, 255, 255), rgba(205, 205, 205, 115), axis=FillGradientAxis.fgaDiagTLBR)),
        ]))
    out = Renders()
    out.layers[0] = lst
    return out


def rgb_boxes(w=800.0, h=600.0) -> Renders:
    """tests/trender_rgb_boxes.nim:12-100 (golden: tests/expected/render_rgb_boxes.png): the solid-colour twin of rgb_boxes_sdf."""
    lst = RenderList()
    root = lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    lst.addChild(root, Fig(kind=RECT, corners=[10, 20, 30, 40], screenBox=rect(60, 60, 220, 140),
                           fill=rgba(220, 40, 40, 255), stroke=RenderStroke(weight=5.0, fill=fill(rgba(0, 0, 0, 255)))))
    lst.addChild(root, Fig(kind=RECT, screenBox=rect(320, 120, 220, 140), fill=rgba(40, 180, 90, 255),
                           shadows=[RenderShadow(style=DROP, blur=10, spread=10, x=10, y=10, fill=fill(rgba(0, 0, 0, 55)))]))
    lst.addChild(root, Fig(kind=RECT, screenBox=rect(180, 300, 220, 140), fill=rgba(60, 90, 220, 255),
                           shadows=[RenderShadow(style=INNER, blur=12, spread=0, x=-6, y=-6, fill=fill(rgba(55, 55, 55, 155))),
                                    RenderShadow(style=INNER, blur=12, spread=0, x=6, y=6, fill=fill(rgba(255, 255, 255, 255)))]))
    out = Renders()
    out.layers[0] = lst
    return out


def oneframe(w=240.0, h=160.0) -> Renders:
    """tests/tfigrender_oneframe_screenshot.nim:20-42."""
    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(32, 24, 120, 80), fill=rgba(220, 40, 40, 255)))
    out = Renders()
    out.layers[0] = lst
    return out


def linear_gradient(w=800.0, h=600.0) -> Renders:
    """tests/trender_linear_gradient.nim:13-99 (golden: tests/expected/render_linear_gradient.png)."""
    lst = RenderList()
    root = lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    lst.addChild(root, Fig(kind=RECT, screenBox=rect(80, 80, 440, 120), corners=[12] * 4,
                           fill=linear(rgba(220, 40, 40, 255), rgba(40, 200, 90, 255), rgba(50, 90, 225, 255),
                                       axis=FillGradientAxis.fgaX, midPos=128)))
    lst.addChild(root, Fig(kind=RECT, screenBox=rect(80, 240, 220, 220), corners=[10] * 4,
                           fill=linear(rgba(240, 210, 40, 255), rgba(110, 60, 210, 255), axis=FillGradientAxis.fgaY)))
    lst.addChild(root, Fig(kind=RECT, screenBox=rect(340, 250, 240, 180), fill=rgba(0, 0, 0, 0),
                           stroke=RenderStroke(weight=20, fill=linear(rgba(245, 70, 70, 255), rgba(70, 115, 245, 255),
                                                                      axis=FillGradientAxis.fgaX))))
    lst.addChild(root, Fig(kind=RECT, screenBox=rect(610, 300, 150, 200), fill=rgba(245, 245, 245, 255),
                           shadows=[RenderShadow(style=DROP, blur=6, spread=14, x=0, y=0,
                                                 fill=linear(rgba(255, 70, 70, 170), rgba(70, 110, 255, 170),
                                                             axis=FillGradientAxis.fgaX))]))
    out = Renders()
    out.layers[0] = lst
    return out


def _add_root_rect(lst, box, color, z, clip=False, rectMask=False, corners=10):
    flags = FigFlags(0)
    if clip:
        flags |= FigFlags.NfClipContent
    if rectMask:
        flags |= FigFlags.NfRectMaskContent
    return lst.addRoot(Fig(kind=RECT, zlevel=z, screenBox=box, fill=color, corners=[corners] * 4, flags=flags))


def _add_rect(lst, parent, box, color, z, clip=False, rectMask=False, corners=10):
    flags = FigFlags(0)
    if clip:
        flags |= FigFlags.NfClipContent
    if rectMask:
        flags |= FigFlags.NfRectMaskContent
    lst.addChild(parent, Fig(kind=RECT, zlevel=z, screenBox=box, fill=color, corners=[corners] * 4, flags=flags))


def layers_clip(w=800.0, h=375.0, rectMask=False) -> Renders:
    """tests/trender_layers_clip.nim:76-171 (golden: tests/expected/render_layers_clip.png, both variants)."""
    bg, container, button = rgba(255, 255, 255, 255), rgba(208, 208, 208, 255), rgba(43, 159, 234, 255)
    containerW, containerH, containerY = w * 0.30, w * 0.40, h * 0.10
    containerLeftX, containerRightX = w * 0.03, w * 0.50
    buttonX, buttonW, buttonH = containerW * 0.10, containerW * 1.30, containerH * 0.20
    buttonY1, buttonY2, buttonY3 = containerH * 0.15, containerH * 0.45, containerH * 0.75
    bgList, layer0, low, top = RenderList(), RenderList(), RenderList(), RenderList()
    bgList.addRoot(Fig(kind=RECT, zlevel=-20, screenBox=rect(0, 0, w, h), fill=bg))
    left = _add_root_rect(layer0, rect(containerLeftX, containerY, containerW, containerH), container, 0)
    right = _add_root_rect(layer0, rect(containerRightX, containerY, containerW, containerH), container, 0,
                           clip=not rectMask, rectMask=rectMask)
    _add_rect(layer0, left, rect(containerLeftX + buttonX, containerY + buttonY2, buttonW, buttonH), button, 0)
    _add_rect(layer0, right, rect(containerRightX + buttonX, containerY + buttonY2, buttonW, buttonH), button, 0)
    _add_root_rect(low, rect(containerLeftX + buttonX, containerY + buttonY3, buttonW, buttonH), button, -5)
    _add_root_rect(top, rect(containerLeftX + buttonX, containerY + buttonY1, buttonW, buttonH), button, 20)
    _add_root_rect(low, rect(containerRightX + buttonX, containerY + buttonY3, buttonW, buttonH), button, -5)
    _add_root_rect(top, rect(containerRightX + buttonX, containerY + buttonY1, buttonW, buttonH), button, 20)
    out = Renders()
    out.layers[-20] = bgList
    out.layers[0] = layer0
    out.layers[-5] = low
    out.layers[20] = top
    out.sort()
    return out


def rect_mask_mixed_batch(w=480.0, h=180.0) -> Renders:
    """tests/trender_layers_clip.nim:179-221."""
    lst = RenderList()
    _add_root_rect(lst, rect(0, 0, w, h), rgba(255, 255, 255, 255), 0, corners=0)
    _add_root_rect(lst, rect(32, 48, 96, 80), rgba(230, 70, 52, 255), 0, corners=0)
    m = _add_root_rect(lst, rect(180, 48, 80, 80), rgba(218, 218, 218, 255), 0, rectMask=True, corners=0)
    _add_rect(lst, m, rect(150, 72, 150, 34), rgba(56, 168, 88, 255), 0, corners=0)
    _add_root_rect(lst, rect(310, 48, 96, 80), rgba(54, 118, 230, 255), 0, corners=0)
    out = Renders()
    out.layers[0] = lst
    return out


def line_rect(w=800.0, h=600.0) -> Renders:
    """tests/trender_extras.nim:17-37 (golden: tests/expected/render_line_rect.png): a line is a rotated box, so this
    PNG also pins the direction of vmath's rotateZ."""
    lst = RenderList()
    root = lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    lst.addChild(root, figLine((90.0, 120.0), (710.0, 470.0), rgba(0, 0, 0, 255), 48.0))
    out = Renders()
    out.layers[0] = lst
    return out


def circle_rect(w=800.0, h=600.0) -> Renders:
    """tests/trender_extras.nim:39-58 (golden: tests/expected/render_circle_rect.png)."""
    lst = RenderList()
    root = lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    lst.addChild(root, figCircle((400.0, 300.0), rgba(0, 0, 0, 255), 110.0))
    out = Renders()
    out.layers[0] = lst
    return out


# ----------------------------------------------------------------------- extra coverage (SwiftShader goldens)
def drawables(w=420.0, h=300.0) -> Renders:
    """nkDrawable ops: lines with butt / round / square caps, a quadratic bezier (one SDF op, modes 18-20), an adaptive
    cubic, fixed-step arcs with bevel and miter joins (drawFilledQuad), an ellipse, a rounded rect with drawAa, and a
    3-stop gradient stroke on a curve."""
    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(252, 252, 248, 255)))
    D = FigKind.nkDrawable
    black = fill(rgba(20, 20, 20, 255))
    lst.addRoot(Fig(kind=D, screenBox=rect(10, 10, 130, 80), drawStroke=RenderStroke(weight=9.0, fill=black),
                    drawOps=[drawableLine((5, 10), (120, 30))]))
    lst.addRoot(Fig(kind=D, screenBox=rect(10, 40, 130, 80), drawStroke=RenderStroke(weight=9.0, fill=fill(rgba(200, 30, 30, 200)), cap=StrokeCap.scRound),
                    drawOps=[drawableLine((5, 10), (120, 30))]))
    lst.addRoot(Fig(kind=D, screenBox=rect(10, 70, 130, 80), drawStroke=RenderStroke(weight=9.0, fill=fill(rgba(30, 30, 200, 255)), cap=StrokeCap.scSquare),
                    drawOps=[drawableLine((15, 10), (110, 30))]))
    lst.addRoot(Fig(kind=D, screenBox=rect(150, 10, 120, 90), drawStroke=RenderStroke(weight=6.0, fill=fill(rgba(10, 120, 60, 255))),
                    drawOps=[drawableBezier([(5, 5), (60, 110), (115, 10)])]))
    lst.addRoot(Fig(kind=D, screenBox=rect(150, 60, 120, 90), drawStroke=RenderStroke(weight=5.0, fill=fill(rgba(150, 20, 160, 255)), cap=StrokeCap.scButt),
                    drawOps=[drawableBezier([(5, 5), (60, 80), (115, 10)])]))
    lst.addRoot(Fig(kind=D, screenBox=rect(280, 10, 130, 120), drawStroke=RenderStroke(weight=4.0, fill=linear(rgba(240, 120, 20, 255), rgba(30, 160, 220, 255), rgba(90, 20, 180, 255), axis=FillGradientAxis.fgaX, midPos=100)),
                    drawOps=[drawableBezier([(5, 60), (40, -60), (80, 180), (125, 50)])]))
    lst.addRoot(Fig(kind=D, screenBox=rect(10, 150, 130, 130), drawStroke=RenderStroke(weight=10.0, fill=fill(rgba(0, 90, 160, 230)), cap=StrokeCap.scButt, join=StrokeJoin.sjBevel),
                    drawOps=[drawableArc((65, 65), 50.0, 0.3, 3.6, steps=5)]))
    lst.addRoot(Fig(kind=D, screenBox=rect(150, 160, 120, 120), drawStroke=RenderStroke(weight=8.0, fill=fill(rgba(160, 90, 0, 255)), cap=StrokeCap.scSquare, join=StrokeJoin.sjMiter),
                    drawOps=[drawableArc((60, 60), 45.0, 3.4, -2.6, steps=3)]))
    lst.addRoot(Fig(kind=D, screenBox=rect(280, 140, 130, 70), fill=fill(rgba(20, 40, 80, 200)), drawStroke=RenderStroke(weight=3.0, fill=fill(rgba(255, 60, 0, 255))),
                    drawOps=[drawableEllipse((65, 35), (56.25, 28.5)), drawableCircle((20, 20), 11.0)]))
    lst.addRoot(Fig(kind=D, screenBox=rect(280, 220, 130, 70), fill=fill(rgba(250, 200, 40, 255)), drawAa=0.6,
                    drawStroke=RenderStroke(weight=2.0, fill=black), drawOps=[drawableRect((6.5, 8.25, 110, 50), (14, 0, 22, 6))]))
    lst.addRoot(Fig(kind=D, screenBox=rect(150, 230, 120, 60), drawStroke=RenderStroke(weight=3.0, fill=black, cap=StrokeCap.scButt, join=StrokeJoin.sjMiter),
                    drawOps=[drawableBezier([(5, 50), (60, 5)], steps=0), drawableBezier([(60, 5), (115, 50)])]))
    out = Renders()
    out.layers[0] = lst
    return out



def elliptical_and_fractional(w=320.0, h=240.0) -> Renders:
    """Elliptical corners, fractional rects (ceil-snapped quad with un-snapped half extents),
    translucent fills and strokes, spread-only and blur-only shadows."""
    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(250, 250, 245, 255)))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(20.3, 15.6, 120.5, 70.25), fill=rgba(30, 120, 220, 155),
                    corners=[30, 10, 8, 40], cornerRadiiY=[30, 20, 8, 80], flags=FigFlags.NfEllipticalCorners,
                    stroke=RenderStroke(weight=5.0, fill=fill(rgba(10, 10, 10, 200)))))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(170.5, 20.5, 110.0, 60.0), fill=rgba(240, 200, 40, 255),
                    corners=[90, 90, 90, 90], cornerRadiiY=[50, 50, 50, 50], flags=FigFlags.NfEllipticalCorners,
                    stroke=RenderStroke(weight=4.0, fill=fill(rgba(90, 45, 0, 220))),
                    shadows=[RenderShadow(style=DROP, blur=0, spread=6, x=3, y=4, fill=fill(rgba(0, 0, 0, 120))),
                             RenderShadow(style=DROP, blur=9, spread=0, x=-6, y=5, fill=fill(rgba(200, 0, 0, 90)))]))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(30.75, 120.2, 130.3, 90.9), corners=[16, 0, 24, 7],
                    fill=linear(rgba(44, 72, 186, 255), rgba(60, 90, 220, 200), rgba(118, 168, 255, 255),
                                axis=FillGradientAxis.fgaDiagBLTR, midPos=132),
                    stroke=RenderStroke(weight=3.5, fill=linear(rgba(255, 255, 255, 210), rgba(0, 0, 0, 210),
                                                                axis=FillGradientAxis.fgaY)),
                    shadows=[RenderShadow(style=INNER, blur=11, spread=5, x=4.5, y=-3.5,
                                          fill=linear(rgba(25, 25, 40, 100), rgba(65, 65, 95, 180),
                                                      axis=FillGradientAxis.fgaDiagBLTR)),
                             RenderShadow(style=INNER, blur=0, spread=3, x=0, y=0, fill=fill(rgba(255, 0, 255, 80)))]))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(190.2, 110.7, 100.6, 100.1), corners=[50, 50, 50, 50],
                    fill=linear(rgba(18, 112, 64, 255), rgba(40, 180, 90, 120), rgba(78, 224, 188, 255),
                                axis=FillGradientAxis.fgaDiagTLBR, midPos=60),
                    shadows=[RenderShadow(style=DROP, blur=14, spread=3, x=5, y=7,
                                          fill=linear(rgba(255, 70, 70, 170), rgba(70, 110, 255, 170),
                                                      axis=FillGradientAxis.fgaY))]))
    out = Renders()
    out.layers[0] = lst
    return out


def nested_clips(w=320.0, h=240.0) -> Renders:
    """NfClipContent nested two deep (mask = (alpha*parent)^2 per level), a clipped node with its own
    shadow (drawn unclipped, before beginMask) and children overflowing both clips."""
    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    outer = lst.addRoot(Fig(kind=RECT, screenBox=rect(30.5, 20.25, 220, 170), fill=rgba(220, 220, 230, 255),
                            corners=[40, 12, 25, 60], flags=FigFlags.NfClipContent,
                            shadows=[RenderShadow(style=DROP, blur=8, spread=2, x=4, y=4, fill=fill(rgba(0, 0, 0, 90)))]))
    lst.addChild(outer, Fig(kind=RECT, screenBox=rect(10, 100, 300, 60), fill=rgba(43, 159, 234, 200), corners=[10] * 4))
    inner = lst.addChild(outer, Fig(kind=RECT, screenBox=rect(120.75, 40.5, 180, 120), fill=rgba(250, 180, 60, 255),
                                    corners=[30] * 4, flags=FigFlags.NfClipContent,
                                    stroke=RenderStroke(weight=3, fill=fill(rgba(120, 60, 0, 255)))))
    lst.addChild(inner, Fig(kind=RECT, screenBox=rect(100, 30, 250, 50), fill=rgba(200, 30, 60, 230), corners=[20] * 4))
    lst.addChild(inner, Fig(kind=RECT, screenBox=rect(150, 90, 60, 120), fill=rgba(20, 160, 70, 255)))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(5, 200, 120, 30), fill=rgba(90, 20, 160, 255), corners=[8] * 4))
    out = Renders()
    out.layers[0] = lst
    return out


def deep_clips(w=400.0, h=300.0, depth=24) -> Renders:
    """NfClipContent nested `depth` levels deep -- the reference draws one mask plane per level and has no limit
    (glcontext.nim:1886-1914); the compositor keeps 16 levels in LDS and spills deeper ones to a global plane.  Every level
    shrinks by a few pixels with alternating corner radii and carries a translucent band that overflows it; the deepest
    clip holds a child that overflows all of them.  Alpha multiplies down the stack ((alpha * parent)^2 per level), so the
    inner levels are drawn with opaque fills to stay visible."""
    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    parent = None
    for k in range(depth):
        inset = 4.0 * k
        box = rect(10.25 + inset, 8.5 + inset * 0.7, w - 20.5 - 2 * inset, h - 17.0 - 1.4 * inset)
        node = Fig(kind=RECT, screenBox=box, fill=rgba((40 * k) % 256, (200 - 7 * k) % 256, (90 + 23 * k) % 256, 255),
                   corners=[(6 + 5 * k) % 30, (3 * k) % 25, 12, (9 + 2 * k) % 40], flags=FigFlags.NfClipContent)
        idx = lst.addRoot(node) if parent is None else lst.addChild(parent, node)
        lst.addChild(idx, Fig(kind=RECT, screenBox=rect(0, 20 + 9 * k, w, 6), fill=rgba(255, 255, 255, 140)))
        parent = idx
    lst.addChild(parent, Fig(kind=RECT, screenBox=rect(-20, h / 2 - 40, w + 40, 80), fill=rgba(220, 30, 60, 255), corners=[20] * 4))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(w - 90, h - 50, 80, 40), fill=rgba(30, 30, 30, 200), corners=[8] * 4))
    out = Renders()
    out.layers[0] = lst
    return out


def rect_mask_nested(w=320.0, h=240.0) -> Renders:
    """NfRectMaskContent with rounded corners; a second rect mask nested inside falls back to a real mask
    (glcontext.nim:1932-1943); plus a NfClipContent child inside a rect mask."""
    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    a = lst.addRoot(Fig(kind=RECT, screenBox=rect(20.5, 20.5, 200, 150), fill=rgba(225, 225, 225, 255),
                        corners=[30, 8, 16, 44], flags=FigFlags.NfRectMaskContent))
    lst.addChild(a, Fig(kind=RECT, screenBox=rect(0, 60, 320, 40), fill=rgba(56, 168, 88, 220)))
    b = lst.addChild(a, Fig(kind=RECT, screenBox=rect(100, 30, 160, 110), fill=rgba(54, 118, 230, 255),
                            corners=[25] * 4, flags=FigFlags.NfRectMaskContent))
    lst.addChild(b, Fig(kind=RECT, screenBox=rect(80, 100, 220, 30), fill=rgba(230, 70, 52, 255), corners=[6] * 4))
    c = lst.addChild(a, Fig(kind=RECT, screenBox=rect(30, 110, 90, 50), fill=rgba(250, 210, 80, 255),
                            corners=[14] * 4, flags=FigFlags.NfClipContent))
    lst.addChild(c, Fig(kind=RECT, screenBox=rect(10, 130, 140, 20), fill=rgba(30, 30, 30, 255)))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(230, 180, 70, 40), fill=rgba(120, 20, 160, 255), corners=[8] * 4))
    out = Renders()
    out.layers[0] = lst
    return out


def backdrop_blur(w=320.0, h=240.0) -> Renders:
    """nkBackdropBlur nodes (a rounded one with a tint overlay and a later, overlapping one: each is a global
    barrier in painter's order) over a busy background; radius 18 -> step 2.25 px."""
    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    cols = [rgba(220, 40, 40, 255), rgba(40, 180, 90, 255), rgba(60, 90, 220, 255), rgba(240, 200, 40, 255)]
    for i in range(12):
        lst.addRoot(Fig(kind=RECT, screenBox=rect(10 + 25 * i, 10 + 17 * (i % 5), 40, 170 - 9 * i), fill=cols[i % 4],
                        corners=[6 + i] * 4, stroke=RenderStroke(weight=2, fill=fill(rgba(0, 0, 0, 255)))))
    lst.addRoot(Fig(kind=FigKind.nkBackdropBlur, screenBox=rect(40.5, 50.25, 180, 120), corners=[28] * 4,
                    fill=rgba(255, 255, 255, 60), blur=18.0))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(60, 70, 60, 30), fill=rgba(0, 0, 0, 255), corners=[5] * 4))
    lst.addRoot(Fig(kind=FigKind.nkBackdropBlur, screenBox=rect(150, 20, 150, 200), corners=[10, 40, 10, 40],
                    cornerRadiiY=[20, 40, 5, 70], flags=FigFlags.NfEllipticalCorners, fill=rgba(0, 0, 0, 0), blur=5.0))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(250, 190, 60, 40), fill=rgba(255, 225, 55, 120), corners=[10] * 4,
                    stroke=RenderStroke(weight=6.0, fill=fill(rgba(95, 72, 0, 185)))))
    out = Renders()
    out.layers[0] = lst
    return out


def rotation_and_transform(w=320.0, h=240.0) -> Renders:
    """Rotated nodes (two-triangle rasterisation of per-vertex ceil'd quads),
    nkTransform translation + matrix, and a rect mask created under a rotated transform."""
    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(30, 30, 120, 70), rotation=17.0, fill=rgba(220, 40, 40, 200),
                    corners=[10, 20, 30, 5], stroke=RenderStroke(weight=4, fill=fill(rgba(0, 0, 0, 255))),
                    shadows=[RenderShadow(style=DROP, blur=8, spread=4, x=6, y=6, fill=fill(rgba(0, 0, 0, 100)))]))
    t = lst.addRoot(Fig(kind=FigKind.nkTransform, translation=(150.0, 20.0), useMatrix=True,
                        matrix=[1.2, 0.25, 0, 0, -0.3, 0.9, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]))
    lst.addChild(t, Fig(kind=RECT, screenBox=rect(10, 10, 100, 60),
                        fill=linear(rgba(24, 128, 72, 255), rgba(40, 180, 90, 255), rgba(54, 206, 170, 255),
                                    axis=FillGradientAxis.fgaX, midPos=140), corners=[14] * 4,
                        shadows=[RenderShadow(style=INNER, blur=8, spread=1, x=3, y=3, fill=fill(rgba(0, 0, 0, 160)))]))
    m = lst.addRoot(Fig(kind=RECT, screenBox=rect(40, 130, 140, 80), rotation=-25.0, fill=rgba(210, 210, 225, 255),
                        corners=[20] * 4, flags=FigFlags.NfRectMaskContent))
    lst.addChild(m, Fig(kind=RECT, screenBox=rect(20, 150, 200, 30), fill=rgba(43, 159, 234, 255), corners=[8] * 4))
    lst.addRoot(Fig(kind=RECT, screenBox=rect(210, 140, 80, 60), rotation=45.0,
                    fill=linear(rgba(240, 210, 40, 255), rgba(110, 60, 210, 255), axis=FillGradientAxis.fgaDiagTLBR)))
    out = Renders()
    out.layers[0] = lst
    return out


REFERENCE_PNG_SCENES = {
    # name: (builder, width, height, reference golden png under tests/expected/)
    "rgb_boxes_sdf": (rgb_boxes_sdf, 800, 600, "render_rgb_boxes_sdf.png"),
    "rgb_boxes": (rgb_boxes, 800, 600, "render_rgb_boxes.png"),
    "linear_gradient": (linear_gradient, 800, 600, "render_linear_gradient.png"),
    "layers_clip": (lambda w, h: layers_clip(w, h, False), 800, 375, "render_layers_clip.png"),
    "layers_rect_mask": (lambda w, h: layers_clip(w, h, True), 800, 375, "render_layers_clip.png"),
    "line_rect": (line_rect, 800, 600, "render_line_rect.png"),
    "circle_rect": (circle_rect, 800, 600, "render_circle_rect.png"),
}

def rotated_tree(w=900.0, h=600.0) -> Renders:
    """Thirty rotated rectangles of the renderlist_100 tree (elliptical corners, strokes, drop and inner shadows, 2- and 3-stop
    gradients; -30 .. 30 degrees): the workload of tools/perf_configs.py config 9 in small -- rotated quads four pixels per lane,
    strips outside a quad, saturated cores of rotated boxes."""
    from figdraw_amd.scenes import make_rotated_tree

    out = make_rotated_tree(w, h, 2, copies=10)
    # circular corners only: SwiftShader interpolates a per-quad constant with a relative error of ~1e-4 on rotated triangles, and an
    # elliptical corner's radii travel PACKED in one float (x + 4096 y, atlas.frag:88-94) -- at some angles the decoded radii are
    # garbage (a rotated elliptical stroke vanishes at 10 and -17 degrees, is exact at 25; circular radii survive the error).  A
    # harness artefact like the sdfMode one, not the shader's: elliptical corners under rotation are covered
    # against the oracle (test_rotated_tree_matches_oracle), and unrotated against SwiftShader (elliptical_and_fractional).
    for n in out.layers[0].nodes:
        n.flags &= ~FigFlags.NfEllipticalCorners
    return out


def curves(w=640.0, h=420.0) -> Renders:
    """Forty stroked nkDrawable curves, lines and arcs (config 10 in small): quadratic-bezier spans, rotated boxes, join quads."""
    from figdraw_amd.scenes import make_curves_scene

    return make_curves_scene(w, h, n=40, seed=3)


SWIFTSHADER_SCENES = {
    "oneframe": (oneframe, 240, 160),
    "rect_mask_mixed_batch": (rect_mask_mixed_batch, 480, 180),
    "elliptical_and_fractional": (elliptical_and_fractional, 320, 240),
    "nested_clips": (nested_clips, 320, 240),
    "rect_mask_nested": (rect_mask_nested, 320, 240),
    "backdrop_blur": (backdrop_blur, 320, 240),
    "rotation_and_transform": (rotation_and_transform, 320, 240),
    "drawables": (drawables, 420, 300),
}


# Goldens whose tests COUNT the pixels beyond the bar instead of bounding every pixel, because the reference's shaders on SwiftShader
# and a float evaluation legitimately differ at isolated pixels there: (name: builder, width, height, pixels allowed beyond the bar)
#   rotated_tree: pixel centres lying EXACTLY on an outer edge of a rotated quad (integer vertices: a slope-5/9 edge passes through a
#     centre every 9 pixels).  Not a fill-rule difference -- tools/debug/tie_rule_probe.py: SwiftShader draws the ties of left edges
#     and only those (96 of 96 over every edge class and mirror image), the top-left rule the oracle and the kernels apply on exact
#     integer edge functions; y orientation cannot matter, a y flip keeps left edges left.  But the SAME quad -- (238, 478), (553, 653),
#     (669, 443), (355, 268), the one that owns 25 of the 26 -- loses ALL its tie pixels in frames of 900 x 600, 900 x 640, 800 x 600
#     and keeps ALL of them at 900 x 700 and 900 x 601: its vertices reach the rasteriser through the float32 projection and viewport
#     transform, come back off the integers in the last bits, and the 1/256-px snap keeps the residue (the 26 centres lie 0.000 -
#     0.004 px from the edge, manifest.json).  Which side they fall is a property of (W, H) and of the implementation's arithmetic;
#     no rule reproduces it and another conformant rasteriser need not agree.  Counted: 26 measured, 28 allowed.
#   curves: sdBezier's closed-form cubic cancels catastrophically at isolated pixels ("Rotated quads and
#     curves"): two conformant evaluations of it differ there by anything.
# (allowances = twice what is measured, not measured + 2: the counts hang on the last bit of float code that another compiler or ROCm
# release may generate differently -- ADVICE r5 --; what must NOT move is where the outliers are, and the rotated scene's test asserts
# that: every one within 0.004 px of a quad edge)
OUTLIER_SCENES = {
    "rotated_tree": (rotated_tree, 900, 600, 52),  # 26 measured (oracle and HIP alike)
    "curves": (curves, 640, 420, 8),               # 1 measured
}


def quads_of_call_stream(calls):
    """the pixel-grid quads (ceil'd vertices, glcontext.nim:1498-1509) of every draw_rounded_rect_sdf of a recorded BackendContext call stream,
    under its translate / rotate / scale calls (vmath's rotateZ: x' = cos x + sin y, y' = -sin x + cos y -- the convention
    tests/expected/render_line_rect.png pins)"""
    import math

    import numpy as np

    M, stack, out = np.eye(3), [], []
    for c in calls:
        if c[0] == "save_transform":
            stack.append(M.copy())
        elif c[0] == "restore_transform":
            M = stack.pop()
        elif c[0] == "translate":
            M = M @ np.array([[1, 0, c[1]], [0, 1, c[2]], [0, 0, 1.0]])
        elif c[0] == "scale":
            M = M @ np.diag([c[1], c[2] if len(c) > 2 and c[2] is not None else c[1], 1.0])
        elif c[0] == "rotate":
            cs, sn = math.cos(c[1]), math.sin(c[1])
            M = M @ np.array([[cs, sn, 0], [-sn, cs, 0], [0, 0, 1.0]])
        elif c[0] == "draw_rounded_rect_sdf":
            x, y, rw, rh = c[1]
            out.append([tuple(np.ceil((M @ np.array([px, py, 1.0]))[:2])) for px, py in ((x, y + rh), (x + rw, y + rh), (x + rw, y), (x, y))])
    return out


def worst_distance_to_a_quad_edge(pixels, quads):
    """the largest, over the given (x, y) pixels, of the distance from the pixel's centre to the nearest edge of any quad"""
    import numpy as np

    def seg(p, a, b):
        p, a, b = np.array(p), np.array(a), np.array(b)
        ab = b - a
        t = np.clip(np.dot(p - a, ab) / max(float(np.dot(ab, ab)), 1e-9), 0.0, 1.0)
        return float(np.linalg.norm(p - (a + t * ab)))
    return max((min(seg((x + 0.5, y + 0.5), q[i], q[(i + 1) % 4]) for q in quads for i in range(4)) for x, y in pixels), default=0.0)


# ----------------------------------------------------------------------- atlas scenes (need images: see ATLAS_SCENES)
def glyphs_small(w=330.0, h=90.0, images=None) -> Renders:
    """An 8x3 corner of the T10k glyph workload: coverage glyphs 1:1 (mode 0, gradient tint) + magnified MSDF (mode 13)."""
    from figdraw_amd.scenes import make_glyph_scene

    return make_glyph_scene(w, h, images, cols=8, rows=3)


def images_and_msdf_variants(w=360.0, h=260.0, images=None) -> Renders:
    """nkImage magnified / 1:1 / minified (mip chain, LINEAR_MIPMAP_LINEAR) with tints and NfInvertY, MTSDF, annular
    MSDF stroke, a rotated MSDF image, glyphs under NfInvertY text, and an image inside a clip."""
    from figdraw_amd.scene import Glyph

    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(160, 160, 160, 255)))
    # tests/trender_image.nim:13-97 draws data/img1.png (100x100) at 160x160 on grey 160
    lst.addRoot(Fig(kind=FigKind.nkImage, screenBox=rect(10, 10, 160, 160), image_id=3000))
    lst.addRoot(Fig(kind=FigKind.nkImage, screenBox=rect(180, 10, 100, 100), image_id=3000, image_fill=fill(rgba(255, 200, 200, 200))))
    lst.addRoot(Fig(kind=FigKind.nkImage, screenBox=rect(290, 10, 60, 60), image_id=3000))                     # minified 0.6x
    lst.addRoot(Fig(kind=FigKind.nkImage, screenBox=rect(290, 80, 27, 31), image_id=3000, flags=FigFlags.NfInvertY))  # ~0.29x, flipped
    lst.addRoot(Fig(kind=FigKind.nkMtsdfImage, screenBox=rect(180, 120, 96, 96), image_id=2000 + ord("R"),
                    image_fill=fill(rgba(20, 20, 160, 255)), pxRange=4.0, sdThreshold=0.5))
    lst.addRoot(Fig(kind=FigKind.nkMsdfImage, screenBox=rect(20, 175, 80, 80), image_id=2000 + ord("g"),
                    image_fill=fill(rgba(200, 40, 20, 255)), pxRange=4.0, sdThreshold=0.45, strokeWeight=3.0))
    lst.addRoot(Fig(kind=FigKind.nkMsdfImage, screenBox=rect(110, 180, 64, 64), rotation=30.0, image_id=2000 + ord("&"),
                    image_fill=fill(rgba(0, 0, 0, 255)), pxRange=4.0, sdThreshold=0.5))
    tint = [rgba(255, 255, 0, 255), rgba(0, 255, 255, 255), rgba(255, 0, 255, 255), rgba(255, 255, 255, 255)]
    glyphs = [Glyph(image_id=1000 + ord(ch), x=float(4 + 13 * i), y=float(20 - images[1000 + ord(ch)].shape[0]), colors=tint)
              for i, ch in enumerate("figdraw!")]
    lst.addRoot(Fig(kind=FigKind.nkText, screenBox=rect(285, 130, 120, 24), glyphs=glyphs))
    lst.addRoot(Fig(kind=FigKind.nkText, screenBox=rect(285, 160, 120, 24), glyphs=glyphs, flags=FigFlags.NfInvertY))
    c = lst.addRoot(Fig(kind=RECT, screenBox=rect(285, 195, 70, 55), fill=rgba(255, 255, 255, 255), corners=[18] * 4,
                        flags=FigFlags.NfClipContent))
    lst.addChild(c, Fig(kind=FigKind.nkImage, screenBox=rect(270, 185, 100, 100), image_id=3000))
    out = Renders()
    out.layers[0] = lst
    return out


# Atlas size used for the goldens: 256.  SwiftShader samples with 16-bit NORMALISED texture coordinates, i.e. only
# 65536/atlasSize sub-texel steps; on a 1024 atlas that is 6 bits and MSDF edges (alpha slope = screenPxRange)
# land up to 7 LSB away from exact bilinear filtering, at 256 the same scenes agree within 2 LSB (measured, see
# DESIGN.md).  Texture-filter precision is implementation-defined in GL; the oracle and the HIP path filter in float.
def glyph_rows_rotated(w=330.0, h=110.0, images=None) -> Renders:
    """glyphs_small with its text rows turned by 7 degrees and its MSDF images by 21: every atlas quad a rotated quad (per-triangle
    LOD and fwidth, trilinear sampling at rho just above 1) -- the rotated atlas path of round 4."""
    from figdraw_amd.scenes import make_glyph_scene

    return make_glyph_scene(w, h, images, cols=8, rows=3, rotation=7.0, origin=(8.0, 16.0))


ATLAS_GOLDEN_SIZE = 256
# (oracle vs golden: max LSB, share of pixels beyond 1 LSB; HIP vs golden: pixels allowed beyond 2 LSB).  The rotated rows are sampled
# trilinearly at rho just above 1 from coordinates SwiftShader keeps in 16-bit fixed point AND derives its LOD from per-fragment
# differences of those: 47 of 36 300 pixels sit 2 LSB from the float evaluation, two of them 3 (manifest.json; with the oracle's
# sampler on the 16-bit grid, 35).
ATLAS_TOLERANCE = {"glyph_rows_rotated": (3, 0.002, 8)}
ATLAS_SCENES = {
    "glyph_rows_rotated": (glyph_rows_rotated, 330, 110),
    "glyphs_small": (glyphs_small, 330, 90),
    "images_and_msdf_variants": (images_and_msdf_variants, 360, 260),
}


def text_frontend(w=300.0, h=120.0, images=None) -> Renders:
    """renderText beyond the glyph loop (figrender.nim:417-497): selection rectangles (NfSelectText, node fill, width
    forced to >= 1, h <= 0 skipped), underline / strikethrough runs, glyph x positions the RENDERER snaps (sub-pixel
    shift, or one of the ten glyph variants when those are enabled -- here the variants are other letters so the
    choice is visible), under NfInvertY as well."""
    from figdraw_amd.scene import GLYPH_VARIANT_STEPS, Glyph, TextRect, text_decoration_rects

    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(250, 250, 245, 255)))
    word = "figdraw"
    black = [rgba(20, 20, 20, 255)] * 4
    variants = [1000 + ord(c) for c in "abcdefghij"]
    for row, (flags, y0) in enumerate(((FigFlags.NfSelectText, 8.0), (FigFlags.NfSelectText | FigFlags.NfInvertY, 62.0))):
        glyphs, x = [], 6.3
        for ch in word:
            gh = images[1000 + ord(ch)].shape[0]
            glyphs.append(Glyph(image_id=1000 + ord(ch), x=x, y=24.0 - gh, colors=black, subpixel_shift=-1.0,
                                variant_ids=variants if row == 1 else None))
            x += 13.37
        rects = [TextRect(20.25, 2.0, 55.5, 26.0), TextRect(90.0, 2.0, 0.2, 26.0), TextRect(120.0, 2.0, 30.0, 0.0)]
        rects += text_decoration_rects(6.0, x, 4.0, 26.0, 20.0, underline=True, strikethrough=(row == 1), color=fill(rgba(200, 30, 30, 255)))
        lst.addRoot(Fig(kind=FigKind.nkText, screenBox=rect(10.5, y0, 200, 40), flags=flags, fill=rgba(90, 140, 255, 120), glyphs=glyphs,
                        textRects=rects))
    out = Renders()
    out.layers[0] = lst
    return out


def blur_sweep(w=700.0, h=420.0, radii=(1.0, 3.0, 6.5, 9.0, 12.0, 18.0, 24.0, 30.0, 40.0, 64.0)) -> Renders:
    """One backdrop-blur node per radius (tap reach 2 .. 66 px: every FIR width the blur kernels are specialised for), over a
    busy background, the nodes hanging over all four frame edges (clamp-to-edge taps) and overlapping each other; the last
    one covers the whole frame."""
    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(250, 250, 250, 255)))
    cols = [rgba(220, 40, 40, 255), rgba(40, 180, 90, 255), rgba(60, 90, 220, 255), rgba(240, 200, 40, 255), rgba(10, 10, 10, 255)]
    for i in range(40):
        lst.addRoot(Fig(kind=RECT, screenBox=rect((i * 53) % int(w) - 20, (i * 37) % int(h) - 15, 30 + (i * 7) % 90, 20 + (i * 11) % 70),
                        fill=cols[i % 5], corners=[(i * 3) % 17] * 4))
    n = len(radii)
    for i, r in enumerate(radii):
        fx, fy = (i % 4) / 3.0, (i // 4) / max(1, (n - 1) // 4)
        bw, bh = 0.45 * w, 0.5 * h
        lst.addRoot(Fig(kind=FigKind.nkBackdropBlur, screenBox=rect(-0.1 * w + fx * (1.2 * w - bw) + 0.25, -0.1 * h + fy * (1.2 * h - bh) + 0.5, bw, bh),
                        corners=[4 + 5 * i] * 4, fill=rgba(255, 255, 255, 20 if i % 2 else 0), blur=r))
        lst.addRoot(Fig(kind=RECT, screenBox=rect(fx * (w - 40), fy * (h - 30), 40, 30), fill=cols[i % 5], corners=[3] * 4))
    lst.addRoot(Fig(kind=FigKind.nkBackdropBlur, screenBox=rect(0, 0, w, h), fill=rgba(0, 0, 0, 0), blur=14.0))
    out = Renders()
    out.layers[0] = lst
    return out


def random_scene(seed: int, w: float, h: float, n: int = 40, clips: bool = True, blur: bool = True, images=None) -> Renders:
    """Seeded random mix of everything the SDF path has: opaque and translucent fills (solid, 2- and 3-stop on all
    axes), circular and elliptical corners, strokes, drop and inner shadows, nested NfClipContent / NfRectMaskContent
    containers, rotations and a backdrop blur -- at whatever (odd) frame size the caller picks.  Used to compare the
    HIP path with the oracle where no golden exists (strip masks, saturated cores, occlusion culling, edge tiles)."""
    import random

    rnd = random.Random(seed)
    lst = RenderList()

    def col(opaque=None):
        a = 255 if (opaque if opaque is not None else rnd.random() < 0.4) else rnd.randrange(20, 255)
        return rgba(rnd.randrange(256), rnd.randrange(256), rnd.randrange(256), a)

    def some_fill():
        k = rnd.random()
        if k < 0.55:
            return fill(col())
        opaque = rnd.random() < 0.5
        axis = rnd.choice(list(FillGradientAxis))
        if k < 0.8:
            return linear(col(opaque), col(opaque), axis=axis)
        return linear(col(opaque), col(opaque), col(opaque), axis=axis, midPos=rnd.randrange(20, 236))

    def some_atlas_node(parent=None):
        """coverage glyphs (1:1, tinted), images magnified / minified / flipped, MSDF / MTSDF incl. annular, some rotated"""
        from figdraw_amd.scene import Glyph

        k = rnd.random()
        x, y = rnd.uniform(0, 0.8 * w), rnd.uniform(0, 0.8 * h)
        if k < 0.35:
            word = "".join(rnd.choice("figdrawABC&!?gjpq") for _ in range(rnd.randrange(2, 9)))
            tint = [col(True) for _ in range(4)] if rnd.random() < 0.5 else [col(True)] * 4
            gl = [Glyph(image_id=1000 + ord(c), x=float(round(13.2 * i)) + (0.0 if rnd.random() < 0.7 else 0.37),
                        y=float(20 - images[1000 + ord(c)].shape[0]), colors=tint) for i, c in enumerate(word)]
            f = Fig(kind=FigKind.nkText, screenBox=rect(round(x), round(y), 120, 24), glyphs=gl,
                    flags=FigFlags.NfInvertY if rnd.random() < 0.2 else FigFlags(0))
        elif k < 0.6:
            sz = rnd.choice([100.0, 160.0, 230.5, 60.0, 33.0])
            f = Fig(kind=FigKind.nkImage, screenBox=rect(x, y, sz, sz * rnd.choice([1.0, 0.8])), image_id=3000,
                    image_fill=fill(col()) if rnd.random() < 0.5 else fill(rgba(255, 255, 255, 255)),
                    flags=FigFlags.NfInvertY if rnd.random() < 0.3 else FigFlags(0))
        else:
            ch = rnd.choice("R&gA?")
            sz = rnd.uniform(24, 140)
            f = Fig(kind=rnd.choice([FigKind.nkMsdfImage, FigKind.nkMtsdfImage]), screenBox=rect(x, y, sz, sz), image_id=2000 + ord(ch),
                    image_fill=some_fill(), pxRange=4.0, sdThreshold=rnd.choice([0.5, 0.45]),
                    strokeWeight=rnd.choice([0.0, 0.0, 2.5]))
        if rnd.random() < 0.15:
            f.rotation = rnd.uniform(-50, 50)
        return lst.addRoot(f) if parent is None else lst.addChild(parent, f)

    def some_rect(parent=None, depth=0):
        bw, bh = rnd.uniform(8, 0.7 * w), rnd.uniform(8, 0.7 * h)
        x, y = rnd.uniform(-0.1 * w, w - 0.5 * bw), rnd.uniform(-0.1 * h, h - 0.5 * bh)
        if rnd.random() < 0.5:  # fractional boxes exercise the ceil snapping
            x, y, bw, bh = round(x), round(y), round(bw), round(bh)
        mr = int(min(bw, bh) / 2)
        corners = [rnd.randrange(0, max(mr, 1)) if rnd.random() < 0.7 else 0 for _ in range(4)]
        f = Fig(kind=RECT, screenBox=rect(x, y, bw, bh), fill=some_fill(), corners=corners)
        if rnd.random() < 0.25:
            f.cornerRadiiY = [rnd.randrange(0, max(mr, 1)) for _ in range(4)]
            f.flags |= FigFlags.NfEllipticalCorners
        if rnd.random() < 0.4:
            f.stroke = RenderStroke(weight=rnd.choice([1.0, 2.5, 6.0, 14.0]), fill=some_fill())
        if rnd.random() < 0.35:
            st = rnd.choice([ShadowStyle.DropShadow, ShadowStyle.InnerShadow])
            f.shadows = [RenderShadow(style=st, blur=rnd.uniform(0, 24), spread=rnd.uniform(0, 12), x=rnd.uniform(-12, 12),
                                      y=rnd.uniform(-12, 12), fill=some_fill())]
        if rnd.random() < 0.1:
            f.rotation = rnd.uniform(-40, 40)
        idx = lst.addRoot(f) if parent is None else lst.addChild(parent, f)
        if clips and depth < 2 and rnd.random() < 0.2:
            f.flags |= rnd.choice([FigFlags.NfClipContent, FigFlags.NfRectMaskContent])
            for _ in range(rnd.randrange(1, 4)):
                if images is not None and rnd.random() < 0.4:
                    some_atlas_node(idx)
                else:
                    some_rect(idx, depth + 1)
        return idx

    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=fill(col(rnd.random() < 0.5))))
    for i in range(n):
        if images is not None and rnd.random() < 0.45:
            some_atlas_node()
        else:
            some_rect()
        if blur and i == n // 2:
            bw, bh = rnd.uniform(0.2 * w, 0.8 * w), rnd.uniform(0.2 * h, 0.8 * h)
            c = rnd.randrange(0, 30)
            lst.addRoot(Fig(kind=FigKind.nkBackdropBlur, screenBox=rect(rnd.uniform(0, w - bw), rnd.uniform(0, h - bh), bw, bh),
                            corners=[c] * 4, fill=rgba(0, 0, 0, 0), blur=rnd.choice([3.0, 9.0, 18.0, 40.0])))
    out = Renders()
    out.layers[0] = lst
    return out


# Hostile content for the backdrop blur (glsl/blur.frag:11-32): the product's matrix-pipe passes multiply each tap as one f16, so their
# error is largest where neighbouring texels are uncorrelated.  Opaque white noise and a 0 / 255 checkerboard of 1-pixel cells, at a size
# that takes the matrix-pipe kernels without being forced to (>= 384 K pixels, fdh_context.cpp Context::prepare).  The sources are rebuilt
# from this recipe; tests/golden/ss_blur_big_<kind>_r<radius>.png holds what the reference's blur.frag makes of them on SwiftShader.
# Powers of two on purpose: SwiftShader samples on a 16-bit normalised coordinate grid (oracle.texcoord_model), on which the texel centres
# of a 1024 x 512 surface lie exactly.  At 896 x 448 they do not, and with radius 64 -- tap step 8 px: every tap ON a texel centre, the
# exact result is the checkerboard itself -- each tap leaks up to 0.7 % of its neighbour: SwiftShader's frame then reads 2..3 / 252..253
# where blur.frag's own arithmetic gives 0 / 255, a property of that rasteriser's sampler, not of the shader.
HOSTILE_BLUR_SIZE = (1024, 512)
HOSTILE_BLUR_KINDS = ("noise", "checker")
HOSTILE_BLUR_RADII = (5.0, 18.0, 64.0)
HOSTILE_BLUR_KEY = 0x626C7572


def hostile_blur_source(kind):
    import numpy as np

    w, h = HOSTILE_BLUR_SIZE
    if kind == "noise":
        src = np.random.default_rng(20260).integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    elif kind == "checker":
        yy, xx = np.mgrid[0:h, 0:w]
        src = np.repeat((((xx + yy) & 1) * 255).astype(np.uint8)[:, :, None], 4, axis=2)
    else:
        raise ValueError(kind)
    src[:, :, 3] = 255  # opaque: drawn 1:1 it REPLACES the frame, so the frame the blur node snapshots is the source itself
    return np.ascontiguousarray(src)


def hostile_blur_scene(radius):
    """the source as a 1:1 image over the whole frame, then one transparent full-frame nkBackdropBlur: the frame is blur(source)"""
    w, h = HOSTILE_BLUR_SIZE
    lst = RenderList()
    lst.addRoot(Fig(kind=FigKind.nkImage, screenBox=rect(0, 0, w, h), image_id=HOSTILE_BLUR_KEY))
    lst.addRoot(Fig(kind=FigKind.nkBackdropBlur, screenBox=rect(0, 0, w, h), fill=rgba(0, 0, 0, 0), blur=radius))
    out = Renders()
    out.layers[0] = lst
    return out


# ---- the reference's invert tests (tests/trender_image_msdf_invert.nim, tests/trender_text_invert.nim): a node drawn as it is, the same
# node under a parent that mirrors y (nkTransform: translation (0, h), matrix scale(1, -1, 1)), and the same with NfInvertY, which must
# stand upright again.  The reference asserts ROW PROFILES, not pixels: restated as known answers in test_oracle.py (oracle) and
# test_hip_parity.py (HIP).
INVERT_BITMAP_KEY, INVERT_MSDF_KEY = 0x696E7631, 0x696E7632
INVERT_RECTS = {"image_base": (40, 50, 180, 180), "image_no_invert": (260, 50, 180, 180), "image_invert": (480, 50, 180, 180),
                "msdf_base": (40, 270, 180, 180), "msdf_no_invert": (260, 270, 180, 180), "msdf_invert": (480, 270, 180, 180)}


def invert_test_images():
    """makeAsymmetricImage / makeSyntheticMsdfField (trender_image_msdf_invert.nim:11-33): 24 x 24, the top third one colour, the rest another"""
    import numpy as np

    bitmap = np.zeros((24, 24, 4), np.uint8); bitmap[:8] = (0, 0, 0, 255); bitmap[8:] = (255, 230, 0, 255)
    field_ = np.zeros((24, 24, 4), np.uint8); field_[:8] = (255, 255, 255, 255); field_[8:] = (0, 0, 0, 255)
    return {INVERT_BITMAP_KEY: bitmap, INVERT_MSDF_KEY: field_}


def _mirrored_input(r, h):  # mirroredInputRect (:98-99)
    return rect(r[0], h - r[1] - r[3], r[2], r[3])


def image_msdf_invert(w=720.0, h=520.0) -> Renders:
    """trender_image_msdf_invert.nim:101-201"""
    R = INVERT_RECTS
    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    lst.addRoot(Fig(kind=FigKind.nkImage, screenBox=rect(*R["image_base"]), image_id=INVERT_BITMAP_KEY))
    lst.addRoot(Fig(kind=FigKind.nkMsdfImage, screenBox=rect(*R["msdf_base"]), image_id=INVERT_MSDF_KEY, image_fill=fill(rgba(0, 0, 0, 255)),
                    pxRange=4.0, sdThreshold=0.5))
    m = lst.addRoot(Fig(kind=FigKind.nkTransform, translation=(0.0, h), useMatrix=True, matrix=[1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]))
    lst.addChild(m, Fig(kind=FigKind.nkImage, screenBox=_mirrored_input(R["image_no_invert"], h), image_id=INVERT_BITMAP_KEY))
    lst.addChild(m, Fig(kind=FigKind.nkImage, screenBox=_mirrored_input(R["image_invert"], h), image_id=INVERT_BITMAP_KEY, flags=FigFlags.NfInvertY))
    lst.addChild(m, Fig(kind=FigKind.nkMsdfImage, screenBox=_mirrored_input(R["msdf_no_invert"], h), image_id=INVERT_MSDF_KEY,
                        image_fill=fill(rgba(0, 0, 0, 255)), pxRange=4.0, sdThreshold=0.5))
    lst.addChild(m, Fig(kind=FigKind.nkMsdfImage, screenBox=_mirrored_input(R["msdf_invert"], h), image_id=INVERT_MSDF_KEY,
                        image_fill=fill(rgba(0, 0, 0, 255)), pxRange=4.0, sdThreshold=0.5, flags=FigFlags.NfInvertY))
    out = Renders()
    out.layers[0] = lst
    return out


def text_invert(w=640.0, h=360.0, images=None) -> Renders:
    """trender_text_invert.nim:818-892 in structure: a selected text node, and the same node with NfInvertY under a parent that mirrors y.
    The reference typesets one 72-px "g" with pixie (third party, absent): here the row is "gjpy" out of the 20-px glyph fixture, placed
    by hand, with the selection rectangle the layout would have produced handed over as a TextRect."""
    from figdraw_amd.scene import Glyph, TextRect

    sel = fill(rgba(255, 210, 70, 210))
    glyphs = [Glyph(image_id=1000 + ord(ch), x=float(6 + 14 * i), y=float(30 - images[1000 + ord(ch)].shape[0] + (5 if ch in "gjpy" else 0)))
              for i, ch in enumerate("gjpy")]
    rects = [TextRect(kind=0, x=4.0, y=6.0, w=60.0, h=32.0)]
    lst = RenderList()
    lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(255, 255, 255, 255)))
    lst.addRoot(Fig(kind=FigKind.nkText, flags=FigFlags.NfSelectText, screenBox=rect(96, 120, 220, 140), fill=sel, glyphs=glyphs, textRects=rects))
    m = lst.addRoot(Fig(kind=FigKind.nkTransform, translation=(0.0, h), useMatrix=True, matrix=[1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]))
    lst.addChild(m, Fig(kind=FigKind.nkText, flags=FigFlags.NfInvertY | FigFlags.NfSelectText, screenBox=_mirrored_input((352, 120, 220, 140), h),
                        fill=sel, glyphs=glyphs, textRects=rects))
    out = Renders()
    out.layers[0] = lst
    return out


def row_profile(img, r):
    """rowProfile (trender_image_msdf_invert.nim:35-51): per row of the rectangle, the sum over its pixels of (255 - r) + (255 - g) + (255 - b)"""
    import numpy as np

    x0, y0 = max(0, int(r[0])), max(0, int(r[1]))
    x1, y1 = min(img.shape[1] - 1, int(r[0] + r[2]) - 1), min(img.shape[0] - 1, int(r[1] + r[3]) - 1)
    return (255 * 3 - img[y0:y1 + 1, x0:x1 + 1, :3].astype(np.int64).sum(axis=2)).sum(axis=1)


def check_image_msdf_invert(img):
    """the assertions of trender_image_msdf_invert.nim:224-262"""
    import numpy as np

    P = {k: row_profile(img, r) for k, r in INVERT_RECTS.items()}
    assert all(len(v) > 0 for v in P.values())
    assert P["image_base"].max() - P["image_base"].min() > 500 and P["msdf_base"].max() - P["msdf_base"].min() > 500
    d = lambda a, b: int(np.abs(a - b).sum())
    for kind in ("image", "msdf"):
        base, no_inv, inv = P[kind + "_base"], P[kind + "_no_invert"], P[kind + "_invert"]
        assert d(base, no_inv[::-1]) < d(base, no_inv), kind      # under the mirroring parent the node stands on its head ...
        assert d(base, inv) <= d(base, inv[::-1]), kind           # ... and NfInvertY puts it upright again
    return {k: int(v.max() - v.min()) for k, v in P.items()}


def check_text_invert(img):
    """the assertions of trender_text_invert.nim:918-943 (ink / highlight classifiers :20-24)"""
    import numpy as np

    def bounds(mask, x0, y0, w, h):
        sub = mask[y0:y0 + h, x0:x0 + w]
        ys, xs = np.nonzero(sub)
        assert len(ys) > 0
        return xs.min() + x0, ys.min() + y0, xs.max() + x0, ys.max() + y0
    r, g, b, a = (img[:, :, k].astype(int) for k in range(4))
    ink = (a >= 20) & ((r < 220) | (g < 220) | (b < 220))
    hl = (a >= 20) & (r >= 180) & (g >= 150) & (b <= 140)
    lb, rb = bounds(ink, 32, 40, 260, 260), bounds(ink, 300, 40, 260, 260)
    lh, rh = bounds(hl, 32, 40, 260, 260), bounds(hl, 300, 40, 260, 260)
    assert abs((lb[3] - lb[1]) - (rb[3] - rb[1])) <= 4 and abs(rb[1] - lb[1]) <= 4, (lb, rb)
    assert abs((lh[3] - lh[1]) - (rh[3] - rh[1])) <= 2 and abs(rh[1] - lh[1]) <= 2, (lh, rh)
    prof = lambda bb: ink[bb[1]:bb[3] + 1, bb[0]:bb[2] + 1].sum(axis=1)
    lp, rp = prof(lb), prof(rb)
    n = min(len(lp), len(rp))
    assert int(np.abs(lp[:n] - rp[:n]).sum()) <= int(np.abs(lp[:n] - rp[:n][::-1]).sum())
    return lb, rb, lh, rh


FLIPPY_IMAGE_KEY = 0x696D6731  # any key: the reference hashes the file name (figbasics.nim imgId)


def image_flippy(w=800.0, h=600.0) -> Renders:
    """tests/trender_image.nim:13-38: grey 160 background + nkImage rect(60,60,160,160) showing data/img1.flippy
    (100x100, 8 stored mips).  Reference output: tests/expected/render_image.png (atlas 2048)."""
    lst = RenderList()
    root = lst.addRoot(Fig(kind=RECT, screenBox=rect(0, 0, w, h), fill=rgba(160, 160, 160, 255)))
    lst.addChild(root, Fig(kind=FigKind.nkImage, screenBox=rect(60, 60, 160, 160), image_id=FLIPPY_IMAGE_KEY))
    out = Renders()
    out.layers[0] = lst
    return out


def used_images(renders, images):
    """The subset of `images` a scene references, in sorted-key order (the upload order every backend uses)."""
    ids = set()
    for lst in renders.layers.values():
        for n in lst.nodes:
            if n.image_id:
                ids.add(n.image_id)
            for g in n.glyphs:
                ids.add(g.image_id)
    return {k: images[k] for k in sorted(ids)}
