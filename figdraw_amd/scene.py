"""Scene model: the host-side mirror of figdraw's `Renders -> RenderList -> Fig`.

Names follow the reference so that scenes read like the reference's own tests:
  Fig / RenderList / Renders      src/figdraw/fignodes.nim:44-92
  FigKind / FigFlags / RenderShadow / RenderStroke / corner arrays
                                  src/figdraw/figbasics.nim:31-113
  Fill, fill(), linear()          src/figdraw/common/filltypes.nim:11-58
  add_root / add_child            src/figdraw/fignodes.nim:316-374

`Renders.to_c()` marshals the tree into the POD structs of
`include/figdraw_hip.h` (FdhFig, FdhLayer, FdhScene) that both the HIP library
and the test oracle consume.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

ShadowCount = 4  # figbasics.nim:12


class FigKind(enum.IntEnum):  # figbasics.nim:37-48
    nkFrame = 0
    nkText = 1
    nkRectangle = 2
    nkDrawable = 3
    nkScrollBar = 4
    nkImage = 5
    nkMsdfImage = 6
    nkMtsdfImage = 7
    nkBackdropBlur = 8
    nkTransform = 9


class FigFlags(enum.IntFlag):  # figbasics.nim:50-58 (bit = enum ordinal)
    NfClipContent = 1 << 0
    NfDisableRender = 1 << 1
    NfRootWindow = 1 << 2
    NfInactive = 1 << 3
    NfSelectText = 1 << 4
    NfInvertY = 1 << 5
    NfRectMaskContent = 1 << 6
    NfEllipticalCorners = 1 << 7


class ShadowStyle(enum.IntEnum):  # figbasics.nim:60-64
    NoShadow = 0
    DropShadow = 1
    InnerShadow = 2


class FillKind(enum.IntEnum):  # filltypes.nim:18-21
    flColor = 0
    flLinear2 = 1
    flLinear3 = 2


class FillGradientAxis(enum.IntEnum):  # filltypes.nim:12-16
    fgaX = 0
    fgaY = 1
    fgaDiagTLBR = 2
    fgaDiagBLTR = 3


class DrawableKind(enum.IntEnum):  # fignodes.nim:13-19
    dkLine = 0
    dkCircle = 1
    dkRectangle = 2
    dkBezier = 3
    dkArc = 4
    dkEllipse = 5


class StrokeCap(enum.IntEnum):  # figbasics.nim:66-70
    scAuto = 0
    scRound = 1
    scButt = 2
    scSquare = 3


class StrokeJoin(enum.IntEnum):  # figbasics.nim:72-76
    sjAuto = 0
    sjRound = 1
    sjBevel = 2
    sjMiter = 3


class SdfMode(enum.IntEnum):  # figbackend.nim:36-52
    sdfModeAtlas = 0
    sdfModeClipAA = 3
    sdfModeDropShadow = 7
    sdfModeDropShadowAA = 8
    sdfModeInsetShadow = 9
    sdfModeInsetShadowAnnular = 10
    sdfModeAnnular = 11
    sdfModeAnnularAA = 12
    sdfModeMsdf = 13
    sdfModeMtsdf = 14
    sdfModeMsdfAnnular = 15
    sdfModeMtsdfAnnular = 16
    sdfModeBackdropBlur = 17
    sdfModeBezierStrokeAA = 18
    sdfModeBezierStrokeButtAA = 19
    sdfModeBezierStrokeSquareAA = 20


RGBA = Tuple[int, int, int, int]


def rgba(r: int, g: int, b: int, a: int = 255) -> RGBA:
    return (int(r) & 255, int(g) & 255, int(b) & 255, int(a) & 255)


@dataclass(frozen=True)
class Fill:
    kind: FillKind = FillKind.flColor
    axis: FillGradientAxis = FillGradientAxis.fgaX
    start: RGBA = (0, 0, 0, 0)
    mid: RGBA = (0, 0, 0, 0)
    stop: RGBA = (0, 0, 0, 0)
    mid_pos: int = 128


def fill(color: RGBA) -> Fill:  # filltypes.nim:47-48
    return Fill(FillKind.flColor, FillGradientAxis.fgaX, tuple(color))


def linear(start: RGBA, *rest, axis: FillGradientAxis = FillGradientAxis.fgaX, midPos: int = 128) -> Fill:
    """linear(start, stop, axis=) or linear(start, mid, stop, axis=, midPos=) -- filltypes.nim:50-58."""
    if len(rest) == 1:
        return Fill(FillKind.flLinear2, axis, tuple(start), (0, 0, 0, 0), tuple(rest[0]))
    if len(rest) == 2:
        return Fill(FillKind.flLinear3, axis, tuple(start), tuple(rest[0]), tuple(rest[1]), int(midPos))
    raise TypeError("linear(start, stop) or linear(start, mid, stop)")


def fill_from_json(d) -> Fill:
    """A Fill from the dict form used in recorded call streams ({kind, axis, start, mid, stop, mid_pos})."""
    if isinstance(d, Fill):
        return d
    return Fill(FillKind(d["kind"]), FillGradientAxis(d["axis"]), tuple(d["start"]), tuple(d["mid"]), tuple(d["stop"]), int(d["mid_pos"]))


def _as_fill(v) -> Fill:
    if isinstance(v, Fill):
        return v
    return fill(tuple(v))


@dataclass
class RenderShadow:  # figbasics.nim:78-84
    style: ShadowStyle = ShadowStyle.NoShadow
    fill: Fill = field(default_factory=Fill)
    blur: float = 0.0
    spread: float = 0.0
    x: float = 0.0
    y: float = 0.0


@dataclass
class RenderStroke:  # figbasics.nim:86-90
    weight: float = 0.0
    fill: Fill = field(default_factory=Fill)
    cap: int = 0
    join: int = 0


@dataclass
class Glyph:
    """A pre-shaped glyph quad (text layout / rasterisation is CPU pre-processing
    in the reference -- figrender.nim:456-496 only consumes key, position, colours)."""
    image_id: int
    x: float
    y: float
    colors: Sequence[RGBA] = ((0, 0, 0, 255),) * 4  # BL, BR, TR, TL
    subpixel_shift: float = 0.0  # >= 0 explicit; < 0: derived from x by the renderer (figrender.nim:464-471)
    variant_ids: Optional[Sequence[int]] = None  # GLYPH_VARIANT_STEPS atlas keys (sub-pixel glyph variants), optional


GLYPH_VARIANT_STEPS = 10  # common/fontglyphs.nim:43


@dataclass
class TextRect:
    """A rectangle renderText draws before the glyphs (figrender.nim:355-452), text-local UI units.
    kind 0 = selection (node fill, needs NfSelectText), kind 1 = underline / strikethrough (own fill)."""
    x: float
    y: float
    w: float
    h: float
    kind: int = 0
    fill: Fill = field(default_factory=Fill)

    def __post_init__(self):
        self.fill = _as_fill(self.fill)


def text_decoration_rects(min_x, max_x, min_y, max_y, font_size, underline=False, strikethrough=False, color=None):
    """renderTextDecorations (figrender.nim:371-415) for one (span, line) run whose glyph rectangles span
    [min_x,max_x] x [min_y,max_y]: thickness = max(round(size/16), 1)."""
    import math

    out = []
    if not (min_x < max_x and min_y < max_y):
        return out
    t = max(math.floor(font_size / 16.0 + 0.5), 1.0)
    f = color if color is not None else fill((0, 0, 0, 255))
    if underline:
        out.append(TextRect(min_x, max_y - t * 1.5, max_x - min_x, t, kind=1, fill=f))
    if strikethrough:
        out.append(TextRect(min_x, min_y + (max_y - min_y) * 0.5 - t * 0.5, max_x - min_x, t, kind=1, fill=f))
    return out


@dataclass
class DrawableOp:  # fignodes.nim:21-42
    kind: DrawableKind
    v: Sequence[float] = ()
    corners: Sequence[int] = (0, 0, 0, 0)
    controls: Sequence[Tuple[float, float]] = ()
    steps: int = 0


def drawableLine(a, b) -> DrawableOp:  # fignodes.nim:234-238
    return DrawableOp(DrawableKind.dkLine, (a[0], a[1], b[0], b[1]))


def drawableCircle(center, radius) -> DrawableOp:
    return DrawableOp(DrawableKind.dkCircle, (center[0], center[1], radius))


def drawableEllipse(center, radii) -> DrawableOp:
    return DrawableOp(DrawableKind.dkEllipse, (center[0], center[1], radii[0], radii[1]))


def drawableRect(box, corners=(0, 0, 0, 0)) -> DrawableOp:
    return DrawableOp(DrawableKind.dkRectangle, tuple(box), corners=tuple(corners))


def drawableBezier(controls, steps: int = 0) -> DrawableOp:  # `steps = 0` inherits drawSteps or is adaptive
    return DrawableOp(DrawableKind.dkBezier, controls=[tuple(c) for c in controls], steps=steps)


def drawableArc(center, radius, startAngle, sweepAngle, steps: int = 0) -> DrawableOp:
    return DrawableOp(DrawableKind.dkArc, (center[0], center[1], radius, startAngle, sweepAngle), steps=steps)


@dataclass
class Fig:  # fignodes.nim:54-92
    kind: FigKind = FigKind.nkFrame
    zlevel: int = 0
    parent: int = -1
    flags: FigFlags = FigFlags(0)
    childCount: int = 0
    screenBox: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 0.0)
    rotation: float = 0.0
    fill: Fill = field(default_factory=Fill)
    corners: Sequence[int] = (0, 0, 0, 0)        # TL, TR, BL, BR
    cornerRadiiY: Sequence[int] = (0, 0, 0, 0)
    shadows: List[RenderShadow] = field(default_factory=list)
    stroke: RenderStroke = field(default_factory=RenderStroke)
    image_id: int = 0
    image_fill: Fill = field(default_factory=lambda: fill((255, 255, 255, 255)))
    pxRange: float = 0.0
    sdThreshold: float = 0.0
    strokeWeight: float = 0.0
    blur: float = 0.0
    translation: Tuple[float, float] = (0.0, 0.0)
    matrix: Optional[Sequence[float]] = None  # 16 floats, column-major
    useMatrix: bool = False
    glyphs: List[Glyph] = field(default_factory=list)
    textRects: List[TextRect] = field(default_factory=list)
    drawStroke: RenderStroke = field(default_factory=RenderStroke)
    drawSteps: int = 0
    drawAa: float = 0.0
    drawOps: List[DrawableOp] = field(default_factory=list)

    def __post_init__(self):
        self.fill = _as_fill(self.fill)
        self.stroke.fill = _as_fill(self.stroke.fill)
        self.drawStroke.fill = _as_fill(self.drawStroke.fill)
        for s in self.shadows:
            s.fill = _as_fill(s.fill)


def rect(x, y, w, h):
    return (float(x), float(y), float(w), float(h))


@dataclass
class RenderList:  # fignodes.nim:44-46
    nodes: List[Fig] = field(default_factory=list)
    rootIds: List[int] = field(default_factory=list)

    def addRoot(self, root: Fig) -> int:  # fignodes.nim:316-330
        idx = len(self.nodes)
        assert idx <= 32767
        root.parent = -1
        self.nodes.append(root)
        self.rootIds.append(idx)
        return idx

    def addChild(self, parentIdx: int, child: Fig) -> int:  # fignodes.nim:352-374
        assert 0 <= parentIdx < len(self.nodes)
        idx = len(self.nodes)
        assert idx <= 32767
        self.nodes[parentIdx].childCount += 1
        child.parent = parentIdx
        self.nodes.append(child)
        return idx


    # ---- physical inserts: nodes are shifted, parent / root indexes rewritten, child counts recomputed
    def child_indices(self, parentIdx: int):  # iterator childIndex fignodes.nim:165-177
        out, idx = [], parentIdx + 1
        cnt = self.nodes[parentIdx].childCount
        while len(out) < cnt and idx < len(self.nodes):
            if self.nodes[idx].parent == parentIdx:
                out.append(idx)
            idx += 1
        return out

    def _child_insert_index(self, parentIdx: int, childPos: int) -> int:  # fignodes.nim:179-193
        cc = self.nodes[parentIdx].childCount
        assert 0 <= childPos <= cc
        if childPos == cc:
            return len(self.nodes)
        return self.child_indices(parentIdx)[childPos]

    def _shift_indexes(self, insertIdx: int, count: int):  # fignodes.nim:134-144
        for n in self.nodes:
            if n.parent >= insertIdx:
                n.parent += count
        self.rootIds = [r + count if r >= insertIdx else r for r in self.rootIds]

    def _recompute_child_counts(self):
        for n in self.nodes:
            n.childCount = 0
        for n in self.nodes:
            if n.parent >= 0:
                self.nodes[n.parent].childCount += 1

    def insertRoot(self, root: Fig, rootPos: int) -> int:  # fignodes.nim:332-350
        assert 0 <= rootPos <= len(self.rootIds)
        insertIdx = len(self.nodes) if rootPos == len(self.rootIds) else self.rootIds[rootPos]
        self._shift_indexes(insertIdx, 1)
        root.parent = -1
        self.nodes.insert(insertIdx, root)
        self.rootIds.insert(rootPos, insertIdx)
        self._recompute_child_counts()
        return insertIdx

    def insertChild(self, parentIdx: int, child: Fig, childPos: int) -> int:  # fignodes.nim:376-400
        insertIdx = self._child_insert_index(parentIdx, childPos)
        self._shift_indexes(insertIdx, 1)
        child.parent = parentIdx + 1 if parentIdx >= insertIdx else parentIdx
        self.nodes.insert(insertIdx, child)
        self._recompute_child_counts()
        return insertIdx


class Renders:  # fignodes.nim:48-49 -- OrderedTable[ZLevel, RenderList]; insertion order is render order
    def __init__(self):
        self.layers: Dict[int, RenderList] = {}

    def __getitem__(self, lvl: int) -> RenderList:
        if lvl not in self.layers:
            self.layers[lvl] = RenderList()
        return self.layers[lvl]

    def setLayer(self, lvl: int, lst: RenderList):
        self.layers[lvl] = lst

    def addRoot(self, lvl: int, root: Fig) -> int:  # fignodes.nim:468-474
        root.zlevel = lvl
        return self[lvl].addRoot(root)

    def addChild(self, lvl: int, parentIdx: int, child: Fig) -> int:
        child.zlevel = lvl
        return self[lvl].addChild(parentIdx, child)

    def sort(self):  # tests sort layers by z (trender_layers_clip.nim:168-171)
        self.layers = dict(sorted(self.layers.items(), key=lambda kv: kv[0]))

    # ------------------------------------------------------------------ marshalling
    def to_c(self) -> "CScene":
        return CScene(self)


# ---------------------------------------------------------------------- C structs (include/figdraw_hip.h)
class CColor(C.Structure):
    _fields_ = [("r", C.c_uint8), ("g", C.c_uint8), ("b", C.c_uint8), ("a", C.c_uint8)]


class CFill(C.Structure):
    _fields_ = [("kind", C.c_int32), ("axis", C.c_int32), ("start", CColor), ("mid", CColor), ("stop", CColor),
                ("mid_pos", C.c_uint8), ("_pad", C.c_uint8 * 3)]


class CShadow(C.Structure):
    _fields_ = [("style", C.c_int32), ("fill", CFill), ("blur", C.c_float), ("spread", C.c_float),
                ("x", C.c_float), ("y", C.c_float)]


class CStroke(C.Structure):
    _fields_ = [("weight", C.c_float), ("fill", CFill), ("cap", C.c_int32), ("join", C.c_int32)]


class CFig(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("flags", C.c_uint32), ("parent", C.c_int32), ("child_count", C.c_int32),
        ("zlevel", C.c_int32), ("box", C.c_float * 4), ("rotation", C.c_float), ("fill", CFill),
        ("corners", C.c_uint16 * 4), ("corner_radii_y", C.c_uint16 * 4), ("shadows", CShadow * 4),
        ("stroke", CStroke), ("image_id", C.c_int64), ("image_fill", CFill), ("px_range", C.c_float),
        ("sd_threshold", C.c_float), ("stroke_weight", C.c_float), ("blur", C.c_float),
        ("translation", C.c_float * 2), ("matrix", C.c_float * 16), ("use_matrix", C.c_int32),
        ("glyph_first", C.c_int32), ("glyph_count", C.c_int32),
        ("draw_stroke", CStroke), ("draw_steps", C.c_uint16), ("_pad0", C.c_uint16), ("draw_aa", C.c_float),
        ("op_first", C.c_int32), ("op_count", C.c_int32),
        ("text_rect_first", C.c_int32), ("text_rect_count", C.c_int32),
    ]


class CDrawOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("steps", C.c_uint16), ("corners", C.c_uint16 * 4), ("_pad", C.c_uint16),
                ("v", C.c_float * 6), ("ctrl_first", C.c_int32), ("ctrl_count", C.c_int32)]


class CGlyph(C.Structure):
    _fields_ = [("image_id", C.c_int64), ("x", C.c_float), ("y", C.c_float), ("colors", CColor * 4),
                ("subpixel_shift", C.c_float)]


class CTextRect(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("w", C.c_float), ("h", C.c_float), ("fill", CFill), ("kind", C.c_int32)]


class CLayer(C.Structure):
    _fields_ = [("zlevel", C.c_int32), ("n_nodes", C.c_int32), ("n_roots", C.c_int32), ("_pad", C.c_int32),
                ("nodes", C.POINTER(CFig)), ("root_ids", C.POINTER(C.c_int32))]


class CSceneStruct(C.Structure):
    _fields_ = [("layers", C.POINTER(CLayer)), ("glyphs", C.POINTER(CGlyph)), ("n_layers", C.c_int32),
                ("n_glyphs", C.c_int32), ("ops", C.POINTER(CDrawOp)), ("controls", C.POINTER(C.c_float)),
                ("n_ops", C.c_int32), ("n_controls", C.c_int32), ("text_rects", C.POINTER(CTextRect)),
                ("n_text_rects", C.c_int32), ("_pad", C.c_int32), ("glyph_variant_ids", C.POINTER(C.c_int64))]


def _ccolor(c: RGBA) -> CColor:
    return CColor(int(c[0]), int(c[1]), int(c[2]), int(c[3]))


def cfill(f: Fill) -> CFill:
    out = CFill()
    out.kind = int(f.kind)
    out.axis = int(f.axis)
    out.start = _ccolor(f.start)
    out.mid = _ccolor(f.mid)
    out.stop = _ccolor(f.stop)
    out.mid_pos = int(f.mid_pos)
    return out


_IDENT16 = (1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0)


class CScene:
    """Owns the ctypes arrays behind an FdhScene; keep it alive while the C side reads it."""

    def __init__(self, renders: Renders):
        glyphs: List[Glyph] = []
        trects: List[TextRect] = []
        ops: List[DrawableOp] = []
        self._keep = []
        layers = (CLayer * max(1, len(renders.layers)))()
        for li, (z, lst) in enumerate(renders.layers.items()):
            nodes = (CFig * max(1, len(lst.nodes)))()
            for i, n in enumerate(lst.nodes):
                cn = nodes[i]
                cn.kind = int(n.kind)
                cn.flags = int(n.flags)
                cn.parent = int(n.parent)
                cn.child_count = int(n.childCount)
                cn.zlevel = int(n.zlevel)
                cn.box = (C.c_float * 4)(*[float(v) for v in n.screenBox])
                cn.rotation = float(n.rotation)
                cn.fill = cfill(n.fill)
                cn.corners = (C.c_uint16 * 4)(*[int(v) for v in n.corners])
                cn.corner_radii_y = (C.c_uint16 * 4)(*[int(v) for v in n.cornerRadiiY])
                for si in range(ShadowCount):
                    if si < len(n.shadows):
                        s = n.shadows[si]
                        cs = cn.shadows[si]
                        cs.style = int(s.style)
                        cs.fill = cfill(s.fill)
                        cs.blur, cs.spread, cs.x, cs.y = float(s.blur), float(s.spread), float(s.x), float(s.y)
                cn.stroke.weight = float(n.stroke.weight)
                cn.stroke.fill = cfill(n.stroke.fill)
                cn.stroke.cap = int(n.stroke.cap)
                cn.stroke.join = int(n.stroke.join)
                cn.image_id = int(n.image_id)
                cn.image_fill = cfill(n.image_fill)
                cn.px_range, cn.sd_threshold, cn.stroke_weight = float(n.pxRange), float(n.sdThreshold), float(n.strokeWeight)
                cn.blur = float(n.blur)
                cn.translation = (C.c_float * 2)(float(n.translation[0]), float(n.translation[1]))
                cn.matrix = (C.c_float * 16)(*[float(v) for v in (n.matrix if n.matrix is not None else _IDENT16)])
                cn.use_matrix = 1 if n.useMatrix else 0
                cn.glyph_first = len(glyphs)
                cn.glyph_count = len(n.glyphs)
                glyphs.extend(n.glyphs)
                cn.text_rect_first = len(trects)
                cn.text_rect_count = len(n.textRects)
                trects.extend(n.textRects)
                cn.draw_stroke.weight = float(n.drawStroke.weight)
                cn.draw_stroke.fill = cfill(n.drawStroke.fill)
                cn.draw_stroke.cap = int(n.drawStroke.cap)
                cn.draw_stroke.join = int(n.drawStroke.join)
                cn.draw_steps = int(n.drawSteps)
                cn.draw_aa = float(n.drawAa)
                cn.op_first = len(ops)
                cn.op_count = len(n.drawOps)
                ops.extend(n.drawOps)
            roots = (C.c_int32 * max(1, len(lst.rootIds)))(*lst.rootIds)
            self._keep += [nodes, roots]
            L = layers[li]
            L.zlevel = int(z)
            L.n_nodes = len(lst.nodes)
            L.n_roots = len(lst.rootIds)
            L.nodes = C.cast(nodes, C.POINTER(CFig))
            L.root_ids = C.cast(roots, C.POINTER(C.c_int32))
        cg = (CGlyph * max(1, len(glyphs)))()
        for i, g in enumerate(glyphs):
            cg[i].image_id = int(g.image_id)
            cg[i].x, cg[i].y = float(g.x), float(g.y)
            for k in range(4):
                cg[i].colors[k] = _ccolor(g.colors[k])
            cg[i].subpixel_shift = float(g.subpixel_shift)
        ctr = (CTextRect * max(1, len(trects)))()
        for i, t in enumerate(trects):
            ctr[i].x, ctr[i].y, ctr[i].w, ctr[i].h = float(t.x), float(t.y), float(t.w), float(t.h)
            ctr[i].fill = cfill(t.fill)
            ctr[i].kind = int(t.kind)
        cvar = None
        if any(g.variant_ids is not None for g in glyphs):
            cvar = (C.c_int64 * (GLYPH_VARIANT_STEPS * max(1, len(glyphs))))()
            for i, g in enumerate(glyphs):
                ids = list(g.variant_ids) if g.variant_ids is not None else [g.image_id] * GLYPH_VARIANT_STEPS
                for k in range(GLYPH_VARIANT_STEPS):
                    cvar[i * GLYPH_VARIANT_STEPS + k] = int(ids[k])
        cops = (CDrawOp * max(1, len(ops)))()
        ctrl: List[float] = []
        for i, op in enumerate(ops):
            cops[i].kind = int(op.kind)
            cops[i].steps = int(op.steps)
            cops[i].corners = (C.c_uint16 * 4)(*[int(v) for v in op.corners])
            vv = [float(x) for x in op.v] + [0.0] * (6 - len(op.v))
            cops[i].v = (C.c_float * 6)(*vv)
            cops[i].ctrl_first = len(ctrl) // 2
            cops[i].ctrl_count = len(op.controls)
            for (x, y) in op.controls:
                ctrl += [float(x), float(y)]
        cctrl = (C.c_float * max(1, len(ctrl)))(*ctrl)
        self._keep += [layers, cg, cops, cctrl, ctr, cvar]
        self.struct = CSceneStruct()
        self.struct.layers = C.cast(layers, C.POINTER(CLayer))
        self.struct.glyphs = C.cast(cg, C.POINTER(CGlyph))
        self.struct.n_layers = len(renders.layers)
        self.struct.n_glyphs = len(glyphs)
        self.struct.ops = C.cast(cops, C.POINTER(CDrawOp))
        self.struct.controls = C.cast(cctrl, C.POINTER(C.c_float))
        self.struct.n_ops = len(ops)
        self.struct.n_controls = len(ctrl) // 2
        self.struct.text_rects = C.cast(ctr, C.POINTER(CTextRect))
        self.struct.n_text_rects = len(trects)
        self.struct.glyph_variant_ids = C.cast(cvar, C.POINTER(C.c_int64)) if cvar is not None else None

    def byref(self):
        return C.byref(self.struct)


def figLine(a, b, fill_, weight: float, zlevel: int = 0) -> Fig:
    """figextras.nim:3-19: a stroked line as an nkDrawable whose box bounds the segment + half the weight."""
    hw = max(0.0, float(weight)) / 2.0
    x, y = min(a[0], b[0]) - hw, min(a[1], b[1]) - hw
    box = (x, y, abs(b[0] - a[0]) + hw * 2.0, abs(b[1] - a[1]) + hw * 2.0)
    f = Fig(kind=FigKind.nkDrawable, zlevel=zlevel, screenBox=box, fill=_as_fill(fill_),
            drawStroke=RenderStroke(weight=float(weight), fill=_as_fill(fill_)))
    f.drawOps.append(drawableLine((a[0] - x, a[1] - y), (b[0] - x, b[1] - y)))
    return f


def figCircle(center, fill_, radius: float, zlevel: int = 0) -> Fig:
    """figextras.nim:32-44."""
    r = max(0.0, float(radius))
    f = Fig(kind=FigKind.nkDrawable, zlevel=zlevel, fill=_as_fill(fill_), screenBox=(center[0] - r, center[1] - r, 2 * r, 2 * r))
    f.drawOps.append(drawableCircle((r, r), r))
    return f
