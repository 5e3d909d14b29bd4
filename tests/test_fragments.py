"""RenderFragments (figdraw_amd/fragments.py): the seven cases of the reference's tests/trenderfragments.nim:66-260,
restated.  Node identity rides on `rotation` as in the reference's `testFig`."""
import numpy as np

from figdraw_amd.fragments import RenderFragments
from figdraw_amd.scene import Fig, FigKind, RenderList, Renders, fill, rect, rgba
from oracle import oracle as O


def fig(i, z=0):
    return Fig(kind=FigKind.nkRectangle, zlevel=z, rotation=float(i))


def nid(n):
    return int(n.rotation)


def child_ids(fr, parent):
    return [nid(fr[c]) for c in fr.children(parent)]


def test_inserts_fragment_roots_without_changing_base_physical_indexes():  # :67-92
    fr = RenderFragments()
    root = fr.addRoot(0, fig(10))
    fr.addChild(0, root, fig(40))
    ch = RenderList()
    cr = ch.addRoot(fig(20))
    ch.addChild(cr, fig(21))
    ch.addRoot(fig(30))
    ins = fr.insertChildren(0, root, ch, 0)
    roots = list(fr.roots(0))
    assert [nid(n) for n in fr[0].nodes] == [10, 40]
    assert len(ins) == 2 and nid(fr[ins[0]]) == 20 and nid(fr[ins[1]]) == 30
    assert child_ids(fr, roots[0]) == [20, 30, 40]
    assert child_ids(fr, ins[0]) == [21]
    assert fr.effectiveChildCount(roots[0]) == 3
    assert fr[0].nodes[root].childCount == 1


def test_physical_inserts_keep_fragment_traversal_metadata_synchronized():  # :94-109
    fr = RenderFragments()
    root = fr.addRoot(0, fig(10))
    fr.addChild(0, root, fig(11))
    fr.addChild(0, root, fig(13))
    ch = RenderList()
    ch.addRoot(fig(20))
    fr.insertChildren(0, root, ch, 1)
    fr.insertChild(0, root, fig(12), 2)
    fr.insertRoot(0, fig(5), 0)
    roots = list(fr.roots(0))
    assert [nid(fr[r]) for r in roots] == [5, 10]
    assert child_ids(fr, roots[1]) == [11, 20, 12, 13]


def test_nested_cursor_insert_and_append_overloads():  # :111-127
    fr = RenderFragments()
    root = fr.addRoot(0, fig(10))
    ch = RenderList()
    cr = ch.addRoot(fig(20))
    ch.addChild(cr, fig(21))
    ins = fr.insertChildren(0, root, ch, 0)
    nested = RenderList()
    nested.addRoot(fig(22))
    fr.insertChildren(ins[0], nested, 1)
    app = fr.addChild(ins[0], fig(23))
    assert nid(fr[app]) == 23
    assert child_ids(fr, ins[0]) == [21, 22, 23]


def test_replaces_an_inserted_fragment_preserving_its_position():  # :129-152
    fr = RenderFragments()
    root = fr.addRoot(5, fig(10))
    fr.addChild(5, root, fig(40))
    initial = RenderList()
    initial.addRoot(fig(20))
    initial.addRoot(fig(30))
    ins = fr.insertChildren(5, root, initial, 0)
    upd = RenderList()
    ur = upd.addRoot(fig(50, 1))
    upd.addChild(ur, fig(51, 1))
    upd.addRoot(fig(60, 1))
    rep = fr.updateFragment(ins[0], upd)
    roots = list(fr.roots(5))
    assert len(rep) == 2
    assert child_ids(fr, roots[0]) == [50, 60, 40]
    assert child_ids(fr, rep[0]) == [51]
    assert fr[rep[0]].zlevel == 5 and fr[rep[1]].zlevel == 5
    assert [nid(n) for n in fr[5].nodes] == [10, 40]


def test_replaces_a_nested_fragment_through_its_cursor():  # :154-174
    fr = RenderFragments()
    root = fr.addRoot(0, fig(10))
    pl = RenderList()
    pl.addRoot(fig(20))
    parent = fr.insertChildren(0, root, pl, 0)[0]
    nl = RenderList()
    nl.addRoot(fig(30))
    nested = fr.insertChildren(parent, nl, 0)[0]
    upd = RenderList()
    upd.addRoot(fig(31))
    upd.addRoot(fig(32))
    rep = fr.updateFragment(nested, upd)
    assert len(rep) == 2
    assert child_ids(fr, parent) == [31, 32]


def test_renderer_traverses_transform_fragments():  # :176-201: rect (2,2) under translation (5,-4) -> (7,-2)
    from test_oracle import _xf_point

    fr = RenderFragments()
    root = fr.addRoot(0, Fig(kind=FigKind.nkTransform, translation=(5.0, -4.0)))
    ch = RenderList()
    ch.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(2, 2, 1, 1), fill=fill(rgba(255, 0, 0, 255))))
    fr.insertChildren(0, root, ch, 0)
    o = O.Oracle()
    o.record_begin()
    o.render_frame(fr.flatten(), 64, 64)
    calls = o.record_calls()
    draws = [i for i, c in enumerate(calls) if c[0] == "draw_rounded_rect_sdf"]
    assert len(draws) == 1
    x, y = _xf_point(calls, draws[0], *calls[draws[0]][1][:2])
    assert abs(x - 7.0) < 1e-4 and abs(y + 2.0) < 1e-4


def test_wraps_an_unchanged_renders_value():  # :203-213
    r = Renders()
    root = r.addRoot(2, fig(10))
    r.addChild(2, root, fig(11))
    fr = RenderFragments(r)
    roots = list(fr.roots(2))
    assert child_ids(fr, roots[0]) == [11]
    assert [nid(n) for n in r[2].nodes] == [10, 11]


def test_flatten_renders_the_same_pixels_as_the_equivalent_plain_tree():
    """A fragment tree and the plain Renders with the same logical order give the same frame (oracle)."""
    def content(add_root, add_child):
        bg = add_root(Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, 96, 64), fill=fill(rgba(240, 240, 240, 255))))
        panel = add_root(Fig(kind=FigKind.nkRectangle, screenBox=rect(8, 8, 60, 40), fill=fill(rgba(30, 90, 200, 200)), corners=[8] * 4))
        return bg, panel

    plain = Renders()
    _, p = content(lambda f: plain.addRoot(0, f), None)
    plain.addChild(0, p, Fig(kind=FigKind.nkRectangle, screenBox=rect(12, 12, 20, 10), fill=fill(rgba(255, 0, 0, 255))))
    plain.addChild(0, p, Fig(kind=FigKind.nkRectangle, screenBox=rect(20, 18, 30, 20), fill=fill(rgba(0, 255, 0, 128)), corners=[5] * 4))
    fr = RenderFragments()
    _, p2 = content(lambda f: fr.addRoot(0, f), None)
    fr.addChild(0, p2, Fig(kind=FigKind.nkRectangle, screenBox=rect(20, 18, 30, 20), fill=fill(rgba(0, 255, 0, 128)), corners=[5] * 4))
    ch = RenderList()
    ch.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(12, 12, 20, 10), fill=fill(rgba(255, 0, 0, 255))))
    fr.insertChildren(0, p2, ch, 0)
    a, b = O.Oracle(), O.Oracle()
    a.render_frame(plain, 96, 64)
    b.render_frame(fr.flatten(), 96, 64)
    assert np.array_equal(a.read_pixels(), b.read_pixels())


def test_device_fragments_send_only_what_changed():
    """DeviceFragments: the fragment tree retained in a context (record-only here: no GPU).  After every fragment operation the
    retained frame's draw records equal those of a full render of fragments.flatten(), and only the touched roots were walked."""
    import random

    from figdraw_amd.context import HipContext
    from figdraw_amd.fragments import DeviceFragments, RenderFragments
    from figdraw_amd.scene import Fig, FigKind, RenderList, Renders, rect, rgba

    rnd = random.Random(3)
    w, h = 400, 300

    def box(i, n=0):
        return Fig(kind=FigKind.nkRectangle, screenBox=rect(10 + 13 * (i % 20), 8 + 11 * (i // 20) + n, 40 + i % 7, 30), fill=rgba(20 * (i % 12), 255 - 9 * (i % 25), 40 + i, 255),
                   corners=[i % 9] * 4)

    base = Renders()
    lst = RenderList()
    parents = [lst.addRoot(box(i)) for i in range(24)]
    base.setLayer(0, lst)
    fr = RenderFragments(base)
    cursors = []
    for i in (3, 9, 17):  # three fragments hanging under base roots
        child = RenderList()
        r = child.addRoot(box(100 + i))
        child.addChild(r, box(200 + i))
        cursors += fr.insertChildren(0, parents[i], child, 0)
    ctx = HipContext(record_only=True)
    dev = DeviceFragments(ctx, fr)

    def check(max_walked):
        dev.render(w, h)
        ref = HipContext(record_only=True)
        ref.render_frame(fr.flatten(), w, h)
        assert ctx.record_digest() == ref.record_digest()
        ref.close()
        walked, reused = ctx.scene_stats()
        assert walked <= max_walked, (walked, reused)

    check(10 ** 6)  # first frame: everything
    check(0)        # nothing changed: nothing walked
    for step in range(10):
        if cursors and step % 2 == 0:  # updateFragment: one fragment's contents change -> one base root re-decomposed
            upd = RenderList()
            r = upd.addRoot(box(300 + step, n=step))
            for k in range(rnd.randrange(0, 3)):
                upd.addChild(r, box(400 + 10 * step + k))
            cur = cursors[rnd.randrange(len(cursors))]
            new = fr.updateFragment(cur, upd)
            cursors = [c for c in cursors if c != cur] + list(new or [])
            check(1)
        else:  # a new root in the base tree
            fr.insertRoot(0, box(500 + step), rnd.randrange(0, 5))
            # (cursors into the base layer shift with it: RenderFragments keeps them valid itself)
            check(2)
    ctx.close()
