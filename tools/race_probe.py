#!/usr/bin/env python3
"""GPU-box diagnostic for the cross-context blur anomaly: like tools/race_contexts.py, but it can keep
the contexts alive across iterations, churn device allocations / streams while frames are in flight, and on a mismatch it
fetches the horizontal pass's output as well and prints WHERE inside the 32 x 32 matrix-pipe blocks the wrong texels sit.

env: SIZE=1280x720  NC=4  COPIES=40  RADII=18
     NB=<n>        only the first n contexts carry the full-frame blur (default: all)
     MODE=fresh    contexts created and destroyed every iteration (what race_contexts.py does)
          persist  contexts created once; an iteration only enqueues replays and compares
          churn    persist + a scratch context (stream, surfaces) created, rendered and destroyed while the others are in flight
          malloc   persist + hipMalloc / hipFree of 64 MB blocks (no kernels) while the others are in flight
usage: python3 tools/race_probe.py [iterations]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100

w, h = [int(v) for v in os.environ.get("SIZE", "1280x720").split("x")]
COPIES = int(os.environ.get("COPIES", "40"))
NC = int(os.environ.get("NC", "4"))
MODE = os.environ.get("MODE", "fresh")
RADII = [float(v) for v in os.environ.get("RADII", "18").split(",")]
ROUNDS = int(os.environ.get("ROUNDS", "6"))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20

NB = int(os.environ.get("NB", str(NC)))  # contexts 0 .. NB-1 carry the full-frame (matrix-pipe) blur, the others none
scenes = [make_render_tree_100(w, h, frame=f, copies=COPIES, full_frame_blur=f < NB, full_frame_blur_radius=RADII[f % len(RADII)]) for f in range(NC)]
hip = HipContext(device=0)
SNAP = False  # (round 1 - 4: FDH_DEBUG_SNAP made the library copy the surface in-stream after phase 0; the switch was pruned in round 5)
alone, alone_h, alone_p0 = [], [], []
for sc in scenes:
    hip.render_frame(sc, w, h)
    alone.append(hip.read_pixels())
    alone_h.append(hip.debug_read_surface(1))
    if SNAP:
        alone_p0.append(hip.debug_read_surface(3))

_hiprt = None


def hiprt():
    global _hiprt
    if _hiprt is None:
        _hiprt = ctypes.CDLL("libamdhip64.so")
        _hiprt.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        _hiprt.hipFree.argtypes = [ctypes.c_void_p]
    return _hiprt


def describe(tag, got, want):
    d = got.astype(int) - want.astype(int)
    bad = np.abs(d).max(axis=2) > 0
    ys, xs = np.nonzero(bad)
    if len(ys) == 0:
        print(f"    {tag}: identical")
        return
    ch = [int((d[..., c] != 0).sum()) for c in range(4)]
    lo = [int(d[..., c].min()) for c in range(4)]
    hi = [int(d[..., c].max()) for c in range(4)]
    blocks = sorted(set(zip((xs // 32).tolist(), (ys // 32).tolist())))
    print(f"    {tag}: {len(ys)} px wrong, per channel {ch}, min {lo}, max {hi}, bbox x {xs.min()}..{xs.max()} y {ys.min()}..{ys.max()}, "
          f"{len(blocks)} blocks, first {blocks[:5]}")
    print(f"      x%32 histogram {np.bincount(xs % 32, minlength=32).tolist()}")
    print(f"      y%32 histogram {np.bincount(ys % 32, minlength=32).tolist()}")
    bx, by = blocks[0]
    sub = d[by * 32:(by + 1) * 32, bx * 32:(bx + 1) * 32, :]
    c = int(np.argmax([np.abs(sub[..., k]).sum() for k in range(4)]))
    print(f"      block ({bx},{by}) channel {c} differences (rows = y%32):")
    for r in range(sub.shape[0]):
        if np.any(sub[r, :, c]):
            print("       y%32=" + str(r).rjust(2), " ".join(str(int(v)).rjust(4) for v in sub[r, :, c]))


def check(ctxs, it, counts):
    for i, (c, want) in enumerate(zip(ctxs, alone)):
        c.sync()
        got = c.read_pixels()
        counts[1] += 1
        if not np.array_equal(got, want):
            counts[0] += 1
            if counts[0] <= int(os.environ.get("VERBOSE", "3")):
                print(f"  iter {it} ctx {i}: MISMATCH")
                describe("frame   ", got, want)
                hgot = c.debug_read_surface(1)
                describe("H output", hgot, alone_h[i])
                if SNAP:
                    describe("phase-0 surface (in-stream copy)", c.debug_read_surface(3), alone_p0[i])
                if counts[0] == 1 and os.environ.get("SAVE"):
                    np.savez_compressed(os.environ["SAVE"], got=got, want=want, hgot=hgot, hwant=alone_h[i])
            else:
                print(f"  iter {it} ctx {i}: mismatch ({int((got != want).any(axis=2).sum())} px)")


def dump_mx_bad():
    """FDH_MX_CHECK builds (round 2; the switch was pruned from the kernels in round 6 -- git history has it): texels a wave read from its
    LDS ring that differ from global memory.  A library without the entry point: nothing to dump."""
    L = hip.L
    if not hasattr(L, "fdh_debug_mx_bad"):
        return
    try:
        f = L.fdh_debug_mx_bad
    except AttributeError:
        return
    n = ctypes.c_uint(0)
    buf = (ctypes.c_uint * (4096 * 8))()
    f(ctypes.byref(n), buf, 1)
    if n.value == 0:
        return
    rec = np.frombuffer(buf, dtype=np.uint32).reshape(4096, 8)[:min(n.value, 4096)]
    print(f"  LDS != global: {n.value} texels; pass H {int((rec[:, 0] == 0).sum())} V {int((rec[:, 0] == 1).sum())}")
    for kind in (0, 1):
        r = rec[rec[:, 0] == kind]
        if len(r) == 0:
            continue
        print(f"   pass {'HV'[kind]}: k-step m histogram {np.bincount(r[:, 3], minlength=6).tolist()}, lane histogram {np.bincount(r[:, 4], minlength=64).tolist()}")
        print(f"     t histogram {np.bincount(r[:, 5], minlength=8).tolist()}, block-in-wave b histogram {np.bincount(r[:, 2] & 0xffff, minlength=5).tolist()}")
        x = r[:, 6] ^ r[:, 7]
        print("     bytes differing [b0,b1,b2,b3]:", [int(((x >> (8 * k)) & 255 != 0).sum()) for k in range(4)])
        for q in r[:12]:
            print(f"     wg {q[1]} b {q[2] & 0xffff}/{q[2] >> 16} m {q[3]} lane {q[4]} t {q[5]} got {q[6]:08x} want {q[7]:08x}")


def dump_edge_bad():
    """FDH_EDGE_CHECK builds (round 2; pruned in round 6 like FDH_MX_CHECK): packed edge path vs generic path, same strip, same draw"""
    L = hip.L
    try:
        f = L.fdh_debug_edge_bad
    except AttributeError:
        return
    n = ctypes.c_uint(0)
    buf = (ctypes.c_uint * (4096 * 8))()
    f(ctypes.byref(n), buf, 1)
    print(f"  packed edge path != generic path: {n.value} values")
    if n.value == 0:
        return
    r = np.frombuffer(buf, dtype=np.uint32).reshape(4096, 8)[:min(n.value, 4096)]
    big = np.abs(r[:, 6].copy().view(np.float32) - r[:, 7].copy().view(np.float32)) > 1.5  # (the two paths differ by 1 LSB at a few rounding ties: not the anomaly)
    print("   by draw mode:", {int(m): int((r[:, 0] == m).sum()) for m in np.unique(r[:, 0])}, " elliptical:", int((r[:, 5] & 1).sum()), " off by more than 1:", int(big.sum()))
    surf = {((c.frame_device_ptr()[0] >> 12) & 0x7fffffff): i for i, c in enumerate(LIVE)}
    print("   off-by-more-than-1 values by context:", {surf.get(int(k), hex(int(k))): int(((r[:, 5] >> 1) == k)[big].sum()) for k in np.unique(r[:, 5] >> 1)})
    r = r[big] if big.any() else r
    print("   by pixel slot:", np.bincount(r[:, 3] >> 2, minlength=4).tolist(), " by channel:", np.bincount(r[:, 3] & 3, minlength=4).tolist())
    print("   by 16-lane quarter:", np.bincount(r[:, 4] >> 4, minlength=4).tolist(), " lanes:", np.bincount(r[:, 4], minlength=64).tolist())
    S = r[:, 6].copy().view(np.float32); F = r[:, 7].copy().view(np.float32)
    for q, a, b in list(zip(r, S, F))[:16]:
        print(f"     mode {q[0]} wg {q[1]} draw {q[2]} pixel {q[3] >> 2} ch {q[3] & 3} lane {q[4]} ellip {q[5]} packed {a} generic {b}")


counts = [0, 0]
LIVE = []
if MODE == "fresh":
    for it in range(iters):
        ctxs = [HipContext(device=0) for _ in scenes]
        for c, sc in zip(ctxs, scenes):
            c.render_frame(sc, w, h)
        for _ in range(ROUNDS):
            for c in ctxs:
                c.replay_async(3)
        check(ctxs, it, counts)
        if it < 4:
            dump_mx_bad()
        for c in ctxs:
            c.close()
else:
    ctxs = [HipContext(device=0) for _ in scenes]
    LIVE[:] = ctxs
    for c, sc in zip(ctxs, scenes):
        c.render_frame(sc, w, h)
    for c in ctxs:
        c.sync()
    held = []
    for it in range(iters):
        for r in range(ROUNDS):
            for c in ctxs:
                c.replay_async(3)
            if MODE == "churn" and r % 2 == 0:
                x = HipContext(device=0)
                x.render_frame(scenes[0], w, h)
                x.sync()
                x.close()
            if MODE == "malloc":  # (hipFree waits for every stream: the blocks are released after the comparison)
                for _ in range(2):
                    p = ctypes.c_void_p()
                    hiprt().hipMalloc(ctypes.byref(p), 64 << 20)
                    held.append(p)
        check(ctxs, it, counts)
        if it < 4:
            dump_mx_bad()
            dump_edge_bad()
        while held:
            hiprt().hipFree(held.pop())
    for c in ctxs:
        c.close()
print(f"MODE={MODE} NC={NC} SIZE={w}x{h}: bad {counts[0]} of {counts[1]} context-runs")
