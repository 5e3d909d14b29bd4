"""Recorded BackendContext call streams as binary, and the C player that issues them (tools/call_player.c).

`fdh_record_json` returns the calls a context received as [name, args...] entries.  `pack()` turns one frame's list into the
word stream tools/call_player.c reads; `Player` loads the C driver (built with gcc -std=c99 against include/figdraw_hip.h) and
plays streams through the library's per-call entry points with no Python between the calls -- the path the reference's
renderer takes through its backend (figbackend.nim:468-634), timed the way its benchmark does
(examples/windy_non_clip_benchmark.nim:113-147)."""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
PLAYER_SRC = os.path.join(ROOT, "tools", "call_player.c")
PLAYER_LIB = os.path.join(ROOT, "build", "libfdh_call_player.so")

OPS = {name: i + 1 for i, name in enumerate([
    "begin_frame", "end_frame", "save_transform", "restore_transform", "translate", "rotate", "scale", "apply_transform",
    "set_aa_factor", "draw_rounded_rect_sdf", "draw_image", "draw_msdf", "draw_backdrop_blur", "begin_mask", "end_mask", "pop_mask",
    "begin_rect_mask", "pop_rect_mask", "draw_quadratic_bezier_sdf", "draw_filled_quad", "draw_rect", "set_text_subpixel_shift", "draw_image_adj"])}


def _f(v):
    return struct.unpack("<I", struct.pack("<f", float(v)))[0]


def _col(c):
    return (int(c[0]) & 255) | ((int(c[1]) & 255) << 8) | ((int(c[2]) & 255) << 16) | ((int(c[3]) & 255) << 24)


def _key(k):
    k = int(k) & 0xFFFFFFFFFFFFFFFF
    return [k & 0xFFFFFFFF, k >> 32]


def pack(calls) -> np.ndarray:
    """one frame's recorded calls (begin_frame .. end_frame) -> uint32 word stream"""
    w = []
    for call in calls:
        name, a = call[0], call[1:]
        w.append(OPS[name])
        if name == "begin_frame":
            w += [int(a[0])] + [_f(v) for v in a[1]]
        elif name in ("translate", "scale"):
            w += [_f(a[0]), _f(a[1])]
        elif name in ("rotate", "set_aa_factor", "set_text_subpixel_shift"):
            w += [_f(a[0])]
        elif name == "apply_transform":
            w += [_f(v) for v in a[0]]
        elif name == "draw_rounded_rect_sdf":
            rect, cols, rx, ry, mode, factor, spread, shape, fill_mode, mid, stop, mid_pos = a
            w += [_f(v) for v in rect] + [_col(c) for c in cols] + [_f(v) for v in rx] + [_f(v) for v in ry]
            w += [int(mode), _f(factor), _f(spread), _f(shape[0]), _f(shape[1]), int(fill_mode), _col(mid), _col(stop), _f(mid_pos)]
        elif name == "draw_image":
            key, pos, cols, size, flip = a
            w += _key(key) + [_f(pos[0]), _f(pos[1])] + [_col(c) for c in cols] + [_f(size[0]), _f(size[1]), int(flip)]
        elif name == "draw_image_adj":
            key, pos, col, size = a
            w += _key(key) + [_f(pos[0]), _f(pos[1]), _col(col), _f(size[0]), _f(size[1])]
        elif name == "draw_msdf":
            key, pos, col, size, px_range, thr, stroke, mtsdf, flip = a
            w += _key(key) + [_f(pos[0]), _f(pos[1]), _col(col), _f(size[0]), _f(size[1]), _f(px_range), _f(thr), _f(stroke), int(mtsdf), int(flip)]
        elif name == "draw_backdrop_blur":
            rect, rx, ry, radius = a
            w += [_f(v) for v in rect] + [_f(v) for v in rx] + [_f(v) for v in ry] + [_f(radius)]
        elif name in ("begin_mask", "begin_rect_mask"):
            rect, rx, ry = a
            w += [_f(v) for v in rect] + [_f(v) for v in rx] + [_f(v) for v in ry]
        elif name == "draw_quadratic_bezier_sdf":
            rect, fill, p0, p1, p2, weight, cap = a
            w += [_f(v) for v in rect] + [int(fill["kind"]), int(fill["axis"]), _col(fill["start"]), _col(fill["mid"]), _col(fill["stop"]), int(fill["mid_pos"])]
            w += [_f(p0[0]), _f(p0[1]), _f(p1[0]), _f(p1[1]), _f(p2[0]), _f(p2[1]), _f(weight), int(cap)]
        elif name == "draw_filled_quad":
            verts, cols = a
            w += [_f(v) for v in verts] + [_col(c) for c in cols]
        elif name == "draw_rect":
            w += [_f(v) for v in a[0]] + [_col(a[1])]
        elif a:
            raise ValueError(f"call_stream.pack: unexpected arguments for {name}")
    return np.array(w, dtype=np.uint32)


def build_player(force: bool = False) -> str:
    """gcc -std=c99 the C driver against the installed header and library (a few hundred ms).  With FIGDRAW_HIP_LIB set (a variant library:
    A/B runs, instrumented builds) the player is linked against THAT file, under a name of its own -- linked against the product beside a
    variant's contexts, two libraries with two layouts met in one process."""
    from . import context

    lib = os.path.abspath(os.environ.get("FIGDRAW_HIP_LIB", context.LIB_PATH))
    out = PLAYER_LIB if lib == os.path.abspath(context.LIB_PATH) else PLAYER_LIB[:-3] + "_" + os.path.basename(lib)[:-3].replace("libfigdraw_hip_", "") + ".so"
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(PLAYER_SRC), os.path.getmtime(lib)):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tmp = f"{out}.{os.getpid()}.tmp"  # (several ranks may find it stale at once: each links its own file, the rename is atomic)
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-D_POSIX_C_SOURCE=200112L", "-pthread", "-fPIC", "-shared",
                           "-I", os.path.join(ROOT, "include"), PLAYER_SRC, "-o", tmp,
                           "-L", os.path.dirname(lib), "-l:" + os.path.basename(lib), "-Wl,-rpath," + os.path.dirname(lib)])
    os.replace(tmp, out)
    return out


class Player:
    def __init__(self):
        from . import context

        context.load()  # (the library the player is linked against, by path)
        self.P = C.CDLL(build_player())
        vp = C.c_void_p
        self.P.fdh_play_calls.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int]
        self.P.fdh_play_frames.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        self.P.fdh_play_scenes.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp), C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_double)]
        self.P.fdh_play_scenes_threads.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp), C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_double)]

    def play(self, ctx, stream: np.ndarray, w: int, h: int):
        """one frame through the per-call entry points of `ctx` (a HipContext)"""
        stream = np.ascontiguousarray(stream, dtype=np.uint32)
        ctx.W, ctx.H = int(w), int(h)
        ctx._ck(self.P.fdh_play_calls(ctx.h, stream.ctypes.data, stream.size, int(w), int(h)))

    def play_frames(self, ctxs, streams, frames: int, w: int, h: int) -> float:
        """frame k = streams[k % len(streams)] on ctxs[k % len(ctxs)], every context waited for at the end; seconds of wall time"""
        streams = [np.ascontiguousarray(s, dtype=np.uint32) for s in streams]
        hs = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        ps = (C.c_void_p * len(streams))(*[s.ctypes.data for s in streams])
        ns = (C.c_size_t * len(streams))(*[s.size for s in streams])
        sec = C.c_double()
        rc = self.P.fdh_play_frames(hs, len(ctxs), ps, ns, len(streams), int(frames), int(w), int(h), C.byref(sec))
        for c in ctxs:
            c.W, c.H = int(w), int(h)
        ctxs[0]._ck(rc)
        return sec.value

    def play_scenes(self, ctxs, cscenes, frames: int, w: float, h: float, threads: int = 1) -> float:
        """frame k = fdh_render_frame(cscenes[k % n]) on ctxs[k % len(ctxs)] from C; seconds of wall time.  cscenes: Renders.to_c() objects.
        threads > 1: context c is driven by host thread c % threads (fdh_play_scenes_threads)"""
        hs = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        ps = (C.c_void_p * len(cscenes))(*[C.addressof(cs.struct) for cs in cscenes])
        sec = C.c_double()
        if threads > 1:
            rc = self.P.fdh_play_scenes_threads(hs, len(ctxs), ps, len(cscenes), int(frames), float(w), float(h), int(threads), C.byref(sec))
        else:
            rc = self.P.fdh_play_scenes(hs, len(ctxs), ps, len(cscenes), int(frames), float(w), float(h), C.byref(sec))
        for c in ctxs:
            c.W, c.H = int(w), int(h)
        ctxs[0]._ck(rc)
        return sec.value
