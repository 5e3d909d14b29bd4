"""RenderFragments -- host-side mirror of figdraw's `renderfragments.nim`: a render tree whose base `Renders` stays
physically unchanged while independently replaceable sub-trees (fragments) are inserted under its nodes.

The renderer traverses it through `roots(lvl)` / `children(cursor)` (renderfragments.nim:277-313); `flatten()` runs
exactly that traversal and emits a plain `Renders`, which is what `HipContext.render_frame` / the oracle consume -- the
draw stream is the one `figrender.render` produces for the fragment tree.

The device-side half (persistent draw lists with sub-range replacement) is `DeviceFragments` below: it keeps
the flattened tree RETAINED in a HipContext (fdh_scene_*, include/figdraw_hip.h) and turns what changed between two frames
-- an updateFragment, an inserted or removed root -- into fdh_scene_replace_root / insert_root calls, so the library
re-decomposes only those roots and splices every other root's draw records from its cache.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from .scene import Fig, RenderList, Renders


class RenderFragment:  # renderfragments.nim:23-25 -- an independently replaceable render subtree
    def __init__(self, lst: RenderList, entries: "_Entries"):
        self.list = lst
        self.entries = entries


# a child entry is ("n", nodeIdx) or ("f", fragment, rootIdx)              (RenderChild :9-16)
@dataclass
class _Entries:  # RenderEntries :18-21
    childEntries: Dict[int, list] = field(default_factory=dict)
    rootEntries: list = field(default_factory=list)
    ready: bool = False

    def reset(self):
        self.childEntries.clear()
        self.rootEntries.clear()
        self.ready = False


@dataclass(frozen=True, eq=False)
class RenderCursor:  # :33-36 -- a Fig in a base layer or in an inserted fragment
    zlevel: int
    index: int
    fragment: Optional[RenderFragment] = None

    def __eq__(self, o):  # :40-41
        return isinstance(o, RenderCursor) and self.zlevel == o.zlevel and self.index == o.index and self.fragment is o.fragment

    def __hash__(self):
        return hash((self.zlevel, self.index, id(self.fragment)))


def _validate_root_ids(lst: RenderList):  # :64-76
    for r in lst.rootIds:
        assert 0 <= r < len(lst.nodes) and lst.nodes[r].parent < 0
    for i, n in enumerate(lst.nodes):
        if n.parent < 0:
            assert i in lst.rootIds


def _rebuild(lst: RenderList, e: _Entries):  # :83-93
    e.childEntries.clear()
    e.rootEntries.clear()
    for i, n in enumerate(lst.nodes):
        if n.parent < 0:
            e.rootEntries.append(("n", i))
        else:
            assert n.parent < len(lst.nodes)
            e.childEntries.setdefault(n.parent, []).append(("n", i))
    e.ready = True


def _ensure(lst: RenderList, e: _Entries):
    if not e.ready:
        _rebuild(lst, e)


def _shift_entry_indexes(e: _Entries, insertIdx: int, count: int):  # :99-120
    if not e.ready or count == 0:
        return
    bump = lambda c: ("n", c[1] + count) if c[0] == "n" and c[1] >= insertIdx else c  # noqa: E731
    e.childEntries = {(p + count if p >= insertIdx else p): [bump(c) for c in cs] for p, cs in e.childEntries.items()}
    e.rootEntries = [bump(c) for c in e.rootEntries]


def _relevel(lst: RenderList, lvl: int):  # :150-152
    for n in lst.nodes:
        n.zlevel = lvl


def _insert_fragment(lst: RenderList, e: _Entries, parentIdx: int, children: RenderList, childPos: int):  # :154-176
    _ensure(lst, e)
    assert 0 <= parentIdx < len(lst.nodes)
    assert childPos <= len(e.childEntries.get(parentIdx, []))
    _validate_root_ids(children)
    fe = _Entries()
    _rebuild(children, fe)
    if not fe.rootEntries:
        return None
    frag = RenderFragment(children, fe)
    for off, root in enumerate(fe.rootEntries):
        e.childEntries.setdefault(parentIdx, []).insert(childPos + off, ("f", frag, root[1]))
    return frag


def _append_children(lst: RenderList, e: _Entries, parentIdx: int, children: RenderList) -> List[int]:  # :178-218
    _ensure(lst, e)
    _validate_root_ids(children)
    if not children.nodes:
        return []
    base = len(lst.nodes)
    assert base + len(children.nodes) <= 32767
    for n in children.nodes:
        m = copy.copy(n)
        m.parent = parentIdx if n.parent < 0 else base + n.parent
        lst.nodes.append(m)
    out = []
    for r in children.rootIds:
        e.childEntries.setdefault(parentIdx, []).append(("n", base + r))
        lst.nodes[parentIdx].childCount += 1
        out.append(base + r)
    for sp, n in enumerate(children.nodes):
        if n.childCount > 0:
            e.childEntries[base + sp] = [("n", base + c) for c in children.child_indices(sp)]
    return out


def _insert_child_into(lst: RenderList, e: _Entries, parentIdx: int, child: Fig, childPos: int) -> int:  # :370-396
    _ensure(lst, e)
    assert childPos <= len(e.childEntries.get(parentIdx, []))
    physical = lst.nodes[parentIdx].childCount
    insertIdx = lst._child_insert_index(parentIdx, childPos) if childPos <= physical else len(lst.nodes)
    _shift_entry_indexes(e, insertIdx, 1)
    idx = lst.insertChild(parentIdx, child, min(childPos, physical))
    shifted_parent = parentIdx + 1 if parentIdx >= insertIdx else parentIdx
    e.childEntries.setdefault(shifted_parent, []).insert(childPos, ("n", idx))
    return idx


class RenderFragments:  # :27-31
    def __init__(self, renders: Optional[Renders] = None):  # newRenderFragments :226-233
        self.base = renders if renders is not None else Renders()
        self.layerEntries: Dict[int, _Entries] = {}

    # ---- plumbing
    def _state(self, lvl: int) -> _Entries:  # layerState :220-224
        lst = self.base[lvl]
        e = self.layerEntries.setdefault(lvl, _Entries())
        _ensure(lst, e)
        return e

    def clear(self):
        self.base.layers.clear()
        self.layerEntries.clear()

    def __contains__(self, lvl):
        return lvl in self.base.layers

    def __getitem__(self, key):  # `[]` :259-275: a layer's RenderList, or the Fig a cursor points at
        if isinstance(key, RenderCursor):
            return (key.fragment.list if key.fragment is not None else self.base.layers[key.zlevel]).nodes[key.index]
        self._state(key)
        return self.base.layers[key]

    def setLayer(self, lvl: int, lst: RenderList):  # :263-265
        self.base.setLayer(lvl, lst)
        self.layerEntries.setdefault(lvl, _Entries()).reset()

    def effectiveChildCount(self, *a) -> int:  # :245-257
        if len(a) == 1:
            parent = a[0]
            if parent.fragment is None:
                return self.effectiveChildCount(parent.zlevel, parent.index)
            _ensure(parent.fragment.list, parent.fragment.entries)
            return len(parent.fragment.entries.childEntries.get(parent.index, []))
        lvl, parentIdx = a
        return len(self._state(lvl).childEntries.get(parentIdx, []))

    # ---- traversal (:277-313)
    def roots(self, lvl: int):
        for c in list(self._state(lvl).rootEntries):
            yield RenderCursor(lvl, c[1]) if c[0] == "n" else RenderCursor(lvl, c[2], c[1])

    def children(self, parent: RenderCursor):
        if parent.fragment is None:
            entries, own = self._state(parent.zlevel), None
        else:
            _ensure(parent.fragment.list, parent.fragment.entries)
            entries, own = parent.fragment.entries, parent.fragment
        for c in list(entries.childEntries.get(parent.index, [])):
            yield RenderCursor(parent.zlevel, c[1], own) if c[0] == "n" else RenderCursor(parent.zlevel, c[2], c[1])

    # ---- building (:315-485).  Overloads as in the reference: (lvl, ...) addresses the base layer, a RenderCursor any node.
    def addRoot(self, *a) -> int:
        lvl, root = (a[0].zlevel, a[0]) if len(a) == 1 else a
        root.zlevel = lvl
        e = self._state(lvl)
        idx = self.base.layers[lvl].addRoot(root)
        e.rootEntries.append(("n", idx))
        return idx

    def insertRoot(self, *a) -> int:
        lvl, root, pos = (a[0].zlevel, a[0], a[1]) if len(a) == 2 else a
        e = self._state(lvl)
        lst = self.base.layers[lvl]
        insertIdx = len(lst.nodes) if pos == len(lst.rootIds) else lst.rootIds[pos]
        _shift_entry_indexes(e, insertIdx, 1)
        root.zlevel = lvl
        idx = lst.insertRoot(root, pos)
        e.rootEntries.insert(pos, ("n", idx))
        return idx

    def addChild(self, *a):
        if isinstance(a[0], RenderCursor):
            parent, child = a
            child.zlevel = parent.zlevel
            if parent.fragment is None:
                return RenderCursor(parent.zlevel, self.addChild(parent.zlevel, parent.index, child))
            f = parent.fragment
            _ensure(f.list, f.entries)
            idx = f.list.addChild(parent.index, child)
            f.entries.childEntries.setdefault(parent.index, []).append(("n", idx))
            return RenderCursor(parent.zlevel, idx, f)
        lvl, parentIdx, child = a
        child.zlevel = lvl
        e = self._state(lvl)
        idx = self.base.layers[lvl].addChild(parentIdx, child)
        e.childEntries.setdefault(parentIdx, []).append(("n", idx))
        return idx

    def insertChild(self, *a):
        if isinstance(a[0], RenderCursor):
            parent, child, pos = a
            child.zlevel = parent.zlevel
            if parent.fragment is None:
                return RenderCursor(parent.zlevel, self.insertChild(parent.zlevel, parent.index, child, pos))
            f = parent.fragment
            return RenderCursor(parent.zlevel, _insert_child_into(f.list, f.entries, parent.index, child, pos), f)
        lvl, parentIdx, child, pos = a
        child.zlevel = lvl
        e = self._state(lvl)
        return _insert_child_into(self.base.layers[lvl], e, parentIdx, child, pos)

    def insertChildren(self, *a) -> List[RenderCursor]:
        if isinstance(a[0], RenderCursor):
            parent, children, pos = a
            _relevel(children, parent.zlevel)
            if parent.fragment is None:
                return self.insertChildren(parent.zlevel, parent.index, children, pos)
            f = parent.fragment
            frag = _insert_fragment(f.list, f.entries, parent.index, children, pos)
            lvl = parent.zlevel
        else:
            lvl, parentIdx, children, pos = a
            _relevel(children, lvl)
            e = self._state(lvl)
            frag = _insert_fragment(self.base.layers[lvl], e, parentIdx, children, pos)
        if frag is None:
            return []
        return [RenderCursor(lvl, r[1], frag) for r in frag.entries.rootEntries]

    def addChildren(self, *a):
        if isinstance(a[0], RenderCursor):
            parent, children = a
            _relevel(children, parent.zlevel)
            if parent.fragment is None:
                return [RenderCursor(parent.zlevel, i) for i in self.addChildren(parent.zlevel, parent.index, children)]
            f = parent.fragment
            return [RenderCursor(parent.zlevel, i, f) for i in _append_children(f.list, f.entries, parent.index, children)]
        lvl, parentIdx, children = a
        _relevel(children, lvl)
        e = self._state(lvl)
        return _append_children(self.base.layers[lvl], e, parentIdx, children)

    # ---- replacing a fragment in place (:487-544)
    def updateFragment(self, cursor: RenderCursor, updated: RenderList) -> List[RenderCursor]:
        assert cursor.fragment is not None
        target = cursor.fragment
        _relevel(updated, cursor.zlevel)
        _validate_root_ids(updated)
        ue = _Entries()
        _rebuild(updated, ue)
        roots = [r[1] for r in ue.rootEntries]

        def replace(entries: _Entries):
            for p, cs in entries.childEntries.items():
                out, done = [], False
                for c in cs:
                    if c[0] == "f" and c[1] is target:
                        if not done:
                            out += [("f", target, r) for r in roots]
                            done = True
                    else:
                        out.append(c)
                entries.childEntries[p] = out
            for cs in list(entries.childEntries.values()):
                for c in cs:
                    if c[0] == "f" and c[1] is not target:
                        replace(c[1].entries)

        for e in self.layerEntries.values():
            replace(e)
        target.list = updated
        target.entries = ue
        return [RenderCursor(cursor.zlevel, r, target) for r in roots]

    # ---- what the renderer sees
    def flatten(self) -> Renders:
        """The tree in `roots` / `children` order as a plain Renders (nodes copied; child counts = effective counts)."""
        out = Renders()
        for lvl in self.base.layers:
            lst = RenderList()

            def emit(cur: RenderCursor, parent: int):
                n = copy.copy(self[cur])
                n.childCount = 0
                idx = lst.addRoot(n) if parent < 0 else lst.addChild(parent, n)
                for ch in self.children(cur):
                    emit(ch, idx)

            for r in self.roots(lvl):
                emit(r, -1)
            out.layers[lvl] = lst
        return out


class DeviceFragments:
    """A RenderFragments tree kept retained in a HipContext: `render(w, h)` sends only the roots that changed since the last
    call (per layer: a diff of the roots' contents), then renders.  Same pixels / same draw records as
    `ctx.render_frame(fragments.flatten(), w, h)`."""

    def __init__(self, ctx, fragments: "RenderFragments"):
        self.ctx = ctx
        self.fragments = fragments
        self._roots = None  # per layer: [(digest, subtree)] as last sent
        self._size = None

    @staticmethod
    def _subtrees(lst: RenderList):
        root_of = []
        for i, n in enumerate(lst.nodes):
            root_of.append(i if n.parent < 0 else root_of[n.parent])
        members = {r: [] for r in lst.rootIds}
        for i, r in enumerate(root_of):
            if r in members:
                members[r].append(i)
        out = []
        for r in lst.rootIds:
            idx = members[r]
            pos = {g: k for k, g in enumerate(idx)}
            sub = []
            for g in idx:
                f = copy.copy(lst.nodes[g])
                f.parent = -1 if g == r else pos[f.parent]
                sub.append(f)
            out.append((repr(sub), sub))  # (the dataclass repr covers every field: a content key)
        return out

    def render(self, w, h, **kw):
        import difflib

        flat = self.fragments.flatten()
        now = [self._subtrees(lst) for lst in flat.layers.values()]
        if self._roots is None or self._size != (w, h) or len(now) != len(self._roots):
            self.ctx.scene_retain(flat, w, h, **kw)
        else:
            for layer, (old, new) in enumerate(zip(self._roots, now)):
                sm = difflib.SequenceMatcher(a=[k for k, _ in old], b=[k for k, _ in new], autojunk=False)
                shift = 0  # slots already inserted / removed in front of the current opcode
                for tag, i0, i1, j0, j1 in sm.get_opcodes():
                    if tag == "equal":
                        continue
                    n_old, n_new = i1 - i0, j1 - j0
                    for k in range(min(n_old, n_new)):
                        self.ctx.scene_replace_root(layer, i0 + shift + k, new[j0 + k][1])
                    for k in range(n_new, n_old):  # surplus old roots go
                        self.ctx.scene_replace_root(layer, i0 + shift + n_new, [])
                    for k in range(n_old, n_new):  # surplus new roots come
                        self.ctx.scene_insert_root(layer, i0 + shift + k, new[j0 + k][1])
                    shift += n_new - n_old
            self.ctx.scene_render()
        self._roots, self._size = now, (w, h)
