#!/usr/bin/env python3
"""GPU-box helper: contexts created, used and destroyed on several host threads at once (one context per thread at a time: a handle is
single-threaded, different handles may live on different threads).  Every frame must equal the one a lone context renders.
usage: thread_churn.py [threads] [iterations]"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import ref_scenes as RS  # noqa: E402
from figdraw_amd.context import HipContext  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 25
sizes = [(640, 360), (1280, 720), (333, 217), (1920, 1080)]
scenes = [(w, h, RS.random_scene(40 + i, float(w), float(h), n=30 + 5 * i, clips=i % 2 == 0, blur=i % 3 != 1)) for i, (w, h) in enumerate(sizes * 2)]
ref = HipContext(device=0, sync_submit=True)
want = []
for w, h, sc in scenes:
    ref.render_frame(sc, w, h)
    want.append(ref.read_pixels())
ref.close()
bad = []


def work(t):
    try:
        for it in range(N):
            ctx = HipContext(device=0, sync_submit=(it + t) % 3 == 0)
            for k in range(3):
                i = (t * 7 + it * 3 + k) % len(scenes)
                w, h, sc = scenes[i]
                ctx.render_frame(sc, w, h)
                if k != 1:  # (sometimes two frames back to back without a read in between)
                    got = ctx.read_pixels()
                    if not np.array_equal(got, want[i]):
                        d = (got != want[i]).any(axis=2)
                        ys, xs = np.nonzero(d)
                        wrong = got[d]
                        vals, cnts = np.unique(wrong.view(np.uint32).ravel(), return_counts=True)
                        top = sorted(zip(cnts.tolist(), [hex(v) for v in vals.tolist()]), reverse=True)[:4]
                        ys_, xs_ = np.nonzero(d)
                        runs = f"wrong values (count, RGBA as 0xAABBGGRR): {top}; wrong pixels per 64-px bin row {np.bincount(ys_ // 64).tolist()}"
                        try:
                            vu = ctx.verify_upload()
                        except Exception as e:  # noqa: BLE001
                            vu = repr(e)
                        try:  # the same GPU work again from the records the device holds, then the whole frame again
                            bd0 = ctx.bin_digest()
                            ctx.replay(1)
                            ctx.sync()
                            bd1 = ctx.bin_digest()
                            vu = f"{vu} bins as left {bd0} after the replay {bd1}"
                            ctx.sync()
                            again = int((ctx.read_pixels() != want[i]).any(axis=2).sum())
                            ctx.render_frame(sc, w, h)
                            anew = int((ctx.read_pixels() != want[i]).any(axis=2).sum())
                        except Exception as e:  # noqa: BLE001
                            again = anew = repr(e)
                        bad.append((t, it, i, k, runs, f"upload check {vu}", f"replayed: {again} px wrong, rendered again: {anew} px wrong", f"{int(d.sum())} px, rows {ys.min()}..{ys.max()}, cols {xs.min()}..{xs.max()}, max {int(np.abs(got.astype(int) - want[i].astype(int)).max())}, sync_submit {(it + t) % 3 == 0}"))
            ctx.close()
    except Exception as e:  # noqa: BLE001
        bad.append((t, "exception", repr(e)))


threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
for th in threads:
    th.start()
for th in threads:
    th.join()
print(f"thread churn: {T} threads x {N} contexts x 3 frames: {len(bad)} wrong", bad[:5])
sys.exit(1 if bad else 0)
