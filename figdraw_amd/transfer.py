"""Host-side mirror of the reference's `common/transfer.nim` (44-191): an application's own node tree -> `Renders`.

The reference walks ANY tree whose nodes carry `kind`, `screenBox`, `flags`, `zlevel`, `children` ... (a Nim generic with
`when compiles(...)` probes for every optional field) and files each node under its z-level: a child on its parent's level
becomes that parent's child in the level's `RenderList`, a child on another level becomes a ROOT of that level's list
(`convert`, transfer.nim:162-187); inactive children (`NfInactive`) are skipped with their subtrees; levels end up sorted by z
(`copyInto`, :189-201).  `to_render_fig` is `toRenderFig` (:44-160): the per-kind field copy with the same fallbacks (a colour
where a `Fill` is wanted, float or integer corner radii clamped to u16, `stroke.color` / `shadow.color` spellings of older
trees, legacy drawable `points` turned into rectangle ops).  Python's duck typing stands in for `when compiles`.

The result feeds `HipContext.render_frame` / `fdh_render_frame` like any other `Renders`; `to_tree` is the reference's
`toTree` (:31-42), the shape its tests assert on (`tests/ttransfer.nim`, restated in `tests/test_transfer.py`)."""
from dataclasses import dataclass, field
from typing import Any, List

from .scene import (DrawableOp, Fig, FigFlags, FigKind, Fill, RenderList, Renders, RenderShadow, RenderStroke, ShadowStyle,
                    drawableRect, fill, rect)

_MISSING = object()


def _get(obj: Any, *names, default=_MISSING):
    """first attribute of `obj` that exists (the `when compiles(current.a) ... elif compiles(current.b)` chains)"""
    for n in names:
        cur = obj
        ok = True
        for part in n.split("."):
            if isinstance(cur, dict):
                if part not in cur:
                    ok = False
                    break
                cur = cur[part]
            elif hasattr(cur, part):
                cur = getattr(cur, part)
            else:
                ok = False
                break
        if ok:
            return cur
    return default


def corner_to_u16(v) -> int:  # cornerToU16, transfer.nim:8-20
    if isinstance(v, float):
        if v <= 0.0:
            return 0
        # Nim's round(): half away from zero
        return min(int(v + 0.5), 65535)
    return 0 if v <= 0 else min(int(v), 65535)


def _as_fill(v, default: Fill) -> Fill:
    if v is _MISSING or v is None:
        return default
    if isinstance(v, Fill):
        return v
    return fill(tuple(int(c) for c in v))  # an RGBA colour


@dataclass
class RenderTree:  # transfer.nim:4-6
    id: int = 0
    children: List["RenderTree"] = field(default_factory=list)

    def __getitem__(self, idx: int) -> "RenderTree":  # `[]`, :22-25: an empty tree where there is no child
        return RenderTree() if not self.children else self.children[idx]


def to_tree(lst_or_nodes, idx: int = None) -> RenderTree:  # toTree, transfer.nim:31-42
    if isinstance(lst_or_nodes, RenderList):
        if idx is None:
            out = RenderTree()
            for r in lst_or_nodes.rootIds:
                out.children.append(to_tree(lst_or_nodes, r))
            return out
        out = RenderTree(id=idx)
        for ci in lst_or_nodes.child_indices(idx):
            out.children.append(to_tree(lst_or_nodes, ci))
        return out
    raise TypeError("to_tree takes a RenderList")


def to_render_fig(cur: Any) -> Fig:  # toRenderFig, transfer.nim:44-160
    kind = FigKind(_get(cur, "kind"))
    out = Fig(kind=kind)
    out.screenBox = tuple(float(v) for v in _get(cur, "screenBox", default=(0.0, 0.0, 0.0, 0.0)))
    out.flags = FigFlags(_get(cur, "flags", default=0))
    out.zlevel = int(_get(cur, "zlevel", default=0))
    out.rotation = float(_get(cur, "rotation", default=0.0))
    out.fill = _as_fill(_get(cur, "fill"), Fill())
    c = _get(cur, "corners")
    if c is not _MISSING:
        out.corners = tuple(corner_to_u16(v) for v in c)
    c = _get(cur, "cornerRadiiY")
    if c is not _MISSING:
        out.cornerRadiiY = tuple(corner_to_u16(v) for v in c)

    if kind == FigKind.nkRectangle:
        out.stroke = RenderStroke(weight=float(_get(cur, "stroke.weight", default=0.0)),
                                  fill=_as_fill(_get(cur, "stroke.fill", "stroke.color"), fill((0, 0, 0, 0))))
        shadows = _get(cur, "shadows", default=())
        for orig in list(shadows)[:4]:  # min(result.shadows.len, current.shadows.len): Fig holds four
            style = _get(orig, "style", default=ShadowStyle.NoShadow)  # (a tree that says `kind` for it gets NoShadow, as in :77-80)
            out.shadows.append(RenderShadow(style=ShadowStyle(style), blur=float(_get(orig, "blur", default=0.0)),
                                            spread=float(_get(orig, "spread", default=0.0)), x=float(_get(orig, "x", default=0.0)),
                                            y=float(_get(orig, "y", default=0.0)),
                                            fill=_as_fill(_get(orig, "fill", "color"), fill((0, 0, 0, 0)))))
    elif kind == FigKind.nkImage:
        out.image_id = int(_get(cur, "image.id", "image_id", default=0))
        out.image_fill = _as_fill(_get(cur, "image.fill", "image.color", "image_fill"), fill((255, 255, 255, 255)))
    elif kind in (FigKind.nkMsdfImage, FigKind.nkMtsdfImage):
        pre = "msdfImage" if kind == FigKind.nkMsdfImage else "mtsdfImage"
        out.image_id = int(_get(cur, pre + ".id", "image_id", default=0))
        out.image_fill = _as_fill(_get(cur, pre + ".fill", pre + ".color", "image_fill"), fill((255, 255, 255, 255)))
        out.pxRange = float(_get(cur, pre + ".pxRange", "pxRange", default=0.0))
        out.sdThreshold = float(_get(cur, pre + ".sdThreshold", "sdThreshold", default=0.0))
        out.strokeWeight = float(_get(cur, pre + ".strokeWeight", "strokeWeight", default=0.0))
    elif kind == FigKind.nkBackdropBlur:
        out.blur = float(_get(cur, "backdropBlur.blur", "blur", default=0.0))
    elif kind == FigKind.nkTransform:
        t = _get(cur, "transform.translation", "translation")
        if t is not _MISSING:
            out.translation = (float(t[0]), float(t[1]))
        m = _get(cur, "transform.matrix", "matrix", "transformMatrix")
        explicit = _get(cur, "transform.useMatrix", "useMatrix")
        if m is not _MISSING and m is not None:
            out.matrix = tuple(float(v) for v in m)
            out.useMatrix = bool(explicit) if explicit is not _MISSING else True  # :137-142: a bare matrix field means "use it"
    elif kind == FigKind.nkText:
        out.glyphs = list(_get(cur, "glyphs", default=[]))
        out.textRects = list(_get(cur, "textRects", default=[]))
    elif kind == FigKind.nkDrawable:
        st = _get(cur, "drawStroke", "stroke")
        if st is not _MISSING:
            out.drawStroke = RenderStroke(weight=float(_get(st, "weight", default=0.0)), fill=_as_fill(_get(st, "fill", "color"), fill((0, 0, 0, 0))),
                                          **{k: _get(st, k) for k in ("cap", "join") if _get(st, k) is not _MISSING})
        out.drawSteps = int(_get(cur, "drawSteps", default=0))
        out.drawAa = float(_get(cur, "drawAa", default=0.0))
        ops = _get(cur, "drawOps")
        if ops is not _MISSING and ops:
            out.drawOps = [op for op in ops]
        else:
            pts = _get(cur, "points", default=())
            sb = out.screenBox
            out.drawOps = [drawableRect(rect(p[0], p[1], sb[2], sb[3])) for p in pts]  # legacy trees: :154-157
    return out


def _convert(renders: Renders, cur: Any, parent_idx: int, parent_z: int):  # convert, transfer.nim:162-187
    fig = to_render_fig(cur)
    z = fig.zlevel
    if z not in renders.layers:
        renders.layers[z] = RenderList()
    lst = renders.layers[z]
    if parent_idx < 0 or parent_z != z:
        idx = lst.addRoot(fig)
    else:
        idx = lst.addChild(parent_idx, fig)
    for child in _get(cur, "children", default=()):
        if FigFlags(_get(child, "flags", default=0)) & FigFlags.NfInactive:
            continue
        child_z = int(_get(child, "zlevel", default=0))
        _convert(renders, child, idx if child_z == z else -1, z)


def copy_into(root: Any) -> Renders:  # copyInto, transfer.nim:189-201
    out = Renders()
    _convert(out, root, -1, int(_get(root, "zlevel", default=0)))
    out.sort()
    return out
