"""figdraw_amd/transfer.py against the reference's own cases (tests/ttransfer.nim): an application tree with nodes on several
z-levels -> per-level RenderLists, and the per-kind field copy of toRenderFig with its fallbacks."""
from types import SimpleNamespace as NS

from figdraw_amd import scene as S
from figdraw_amd.transfer import copy_into, corner_to_u16, to_render_fig, to_tree


def node(name, z=0, children=(), kind=S.FigKind.nkRectangle, **kw):
    """a stand-in for tests/ui_test_nodes.nim's FigTest: float corner radii, a Color fill, `stroke`, two `Shadow`s with `kind`"""
    return NS(name=name, kind=kind, zlevel=z, flags=kw.pop("flags", 0), screenBox=(0.0, 0.0, 10.0, 10.0), rotation=0.0,
              fill=(0, 0, 0, 0), corners=[0.0] * 4, cornerRadiiY=[0.0] * 4, stroke=S.RenderStroke(), shadows=kw.pop("shadows", []), children=list(children), **kw)


def test_basic_single_layer():
    """ttransfer.nim:47-66"""
    root = node("root", children=[node("body", children=[node("child1"), node("child2"), node("child3")]), node("body2")])
    renders = copy_into(root)
    lst = renders[0]
    assert lst.rootIds == [0]
    assert len(lst.nodes) == 6
    tree = to_tree(lst)
    assert len(tree.children) == 1 and len(tree[0].children) == 2 and len(tree[0][0].children) == 3


def test_three_layers_out_of_order():
    """ttransfer.nim:68-110: the tree of `draw(fig: TestFig)` (:24-43); children on another level become roots of that level"""
    root = node("root", z=20, children=[
        node("body", z=20, children=[node("child0", z=20, children=[node("child01", z=20)])]),
        node("child1", z=30, children=[node("child11", z=30), node("child12", z=30),
                                       node("child13", z=-10, children=[node("child131", z=-10)])]),
        node("body2", z=20, children=[node("child21", z=-10)]),
    ])
    renders = copy_into(root)
    assert list(renders.layers) == [-10, 20, 30]  # sorted by z
    assert len(renders[-10].nodes) == 3
    assert len(renders[20].nodes) == 5
    assert len(renders[30].nodes) == 3
    res20 = to_tree(renders[20])
    assert len(res20.children) == 1
    assert len(res20[0].children) == 2
    assert len(res20[0][0].children) == 1
    assert len(res20[0][0][0].children) == 1
    res30 = to_tree(renders[30])
    assert len(res30.children) == 1
    assert len(res30[0].children) == 2
    # (not asserted by the reference, but what `convert` does) the -10 level: child13 with its child, and child21, as roots
    assert renders[-10].rootIds == [0, 2] and renders[-10].nodes[1].parent == 0


def test_inactive_children_are_skipped_with_their_subtrees():
    """transfer.nim:177-179"""
    root = node("root", children=[node("a", flags=int(S.FigFlags.NfInactive), children=[node("a1")]), node("b")])
    assert len(copy_into(root)[0].nodes) == 2


def test_keeps_backdrop_blur_style():
    """ttransfer.nim:112-121"""
    n = S.Fig(kind=S.FigKind.nkBackdropBlur, blur=14.0, fill=S.fill(S.rgba(255, 255, 255, 64)), corners=(12, 12, 12, 12))
    out = to_render_fig(n)
    assert out.kind == S.FigKind.nkBackdropBlur and out.blur == 14.0 and tuple(out.corners) == (12, 12, 12, 12)
    assert out.fill.start == (255, 255, 255, 64)


def test_keeps_elliptical_corner_axes_and_flag():
    """ttransfer.nim:123-132: float radii of an application node -> u16"""
    n = node("r", flags=int(S.FigFlags.NfEllipticalCorners))
    n.corners = [12.0, 10.0, 8.0, 6.0]
    n.cornerRadiiY = [3.0, 4.0, 5.0, 6.0]
    out = to_render_fig(n)
    assert tuple(out.corners) == (12, 10, 8, 6) and tuple(out.cornerRadiiY) == (3, 4, 5, 6)
    assert out.flags & S.FigFlags.NfEllipticalCorners
    assert [corner_to_u16(v) for v in (-3, 0, 70000, 2.5, 3.5, -1.0, 1e9)] == [0, 0, 65535, 3, 4, 0, 65535]  # transfer.nim:8-20


def test_keeps_transform_style():
    """ttransfer.nim:134-145"""
    m = (2.0, 0, 0, 0, 0, 3.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0)
    out = to_render_fig(S.Fig(kind=S.FigKind.nkTransform, translation=(12.0, -8.0), matrix=m, useMatrix=True))
    assert out.kind == S.FigKind.nkTransform and out.translation == (12.0, -8.0) and out.useMatrix and tuple(out.matrix) == m
    # a tree that only has `transformMatrix` (transfer.nim:140-142)
    legacy = NS(kind=S.FigKind.nkTransform, screenBox=(0, 0, 0, 0), flags=0, zlevel=0, rotation=0.0, fill=(0, 0, 0, 0), transformMatrix=m, children=[])
    out = to_render_fig(legacy)
    assert out.useMatrix and tuple(out.matrix) == m


def test_keeps_drawable_ops_and_shared_paint():
    """ttransfer.nim:147-181"""
    d = S.Fig(kind=S.FigKind.nkDrawable, screenBox=S.rect(1, 2, 30, 20), fill=S.fill(S.rgba(10, 20, 30, 255)),
              drawStroke=S.RenderStroke(weight=2.5, fill=S.fill(S.rgba(200, 40, 70, 255))), drawSteps=36, drawAa=0.85,
              drawOps=[S.drawableLine((1, 2), (3, 4)), S.drawableCircle((8, 9), 5.0), S.drawableArc((12, 13), 7.0, 0.0, 1.0),
                       S.drawableEllipse((15, 16), (9, 4))])
    out = to_render_fig(d)
    assert out.kind == S.FigKind.nkDrawable and out.fill.start == (10, 20, 30, 255)
    assert out.drawStroke.weight == 2.5 and out.drawStroke.fill.start == (200, 40, 70, 255)
    assert out.drawSteps == 36 and out.drawAa == 0.85
    assert [op.kind for op in out.drawOps] == [S.DrawableKind.dkLine, S.DrawableKind.dkCircle, S.DrawableKind.dkArc, S.DrawableKind.dkEllipse]


def test_converts_legacy_drawable_points_to_rect_ops():
    """ttransfer.nim:183-195"""
    legacy = node("d", kind=S.FigKind.nkDrawable, points=[(2.0, 3.0)])
    legacy.screenBox = (0.0, 0.0, 7.0, 9.0)
    legacy.stroke = S.RenderStroke(weight=1.5, fill=S.fill(S.rgba(90, 100, 110, 255)))
    out = to_render_fig(legacy)
    assert out.drawStroke.weight == 1.5 and out.drawStroke.fill.start == (90, 100, 110, 255)
    assert len(out.drawOps) == 1 and out.drawOps[0].kind == S.DrawableKind.dkRectangle


def test_application_shadows_and_colours():
    """transfer.nim:62-92: `stroke.color` / `shadow.color` spellings, a colour where a Fill is wanted, at most four shadows"""
    sh = [NS(style=S.ShadowStyle.DropShadow, blur=4.0, spread=1.0, x=2.0, y=3.0, color=(1, 2, 3, 4)) for _ in range(5)]
    n = node("r", shadows=sh)
    n.stroke = NS(weight=2.0, color=(9, 8, 7, 6))
    n.fill = (5, 6, 7, 8)
    out = to_render_fig(n)
    assert out.fill.start == (5, 6, 7, 8) and out.stroke.weight == 2.0 and out.stroke.fill.start == (9, 8, 7, 6)
    assert len(out.shadows) == 4 and out.shadows[0].fill.start == (1, 2, 3, 4) and out.shadows[0].style == S.ShadowStyle.DropShadow


def test_converted_tree_renders_like_the_hand_built_one():
    """the product of copy_into is an ordinary Renders: same call stream through the front-end as the same scene built by hand"""
    import pytest

    from figdraw_amd.context import HipContext

    app = node("root", children=[node("a"), node("b", z=5)])
    app.screenBox = (0.0, 0.0, 64.0, 48.0); app.fill = (255, 255, 255, 255)
    app.children[0].screenBox = (8.0, 8.0, 20.0, 12.0); app.children[0].fill = (200, 30, 30, 255); app.children[0].corners = [3.0] * 4
    app.children[1].screenBox = (20.0, 14.0, 30.0, 20.0); app.children[1].fill = (30, 30, 200, 128)
    got = copy_into(app)
    want = S.Renders()
    r = want.addRoot(0, S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(0, 0, 64, 48), fill=S.rgba(255, 255, 255, 255)))
    want.addChild(0, r, S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(8, 8, 20, 12), fill=S.rgba(200, 30, 30, 255), corners=(3, 3, 3, 3)))
    want.addRoot(5, S.Fig(kind=S.FigKind.nkRectangle, screenBox=S.rect(20, 14, 30, 20), fill=S.rgba(30, 30, 200, 128)))
    try:
        ctx = HipContext(record_only=True)
    except Exception as e:  # the library is always built in this suite; a record-only context needs no GPU
        pytest.fail(str(e))
    streams = []
    for sc in (got, want):
        ctx.record_begin()
        ctx.render_frame(sc, 64, 48)
        streams.append(ctx.record_calls())
    assert streams[0] == streams[1] and len(streams[0]) >= 5
