import sys, os, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    from figdraw_amd.context import HipContext
    from figdraw_amd.scenes import make_render_tree_100
    ctx = HipContext(device=0); w, h = 3840, 2160
    ctx.render_frame(make_render_tree_100(w, h, frame=3, full_frame_blur=True), w, h)
    np.save(sys.argv[1], ctx.read_pixels()); sys.exit(0)
subprocess.check_call([sys.executable, __file__, '/tmp/a.npy'], env={**os.environ, 'FDH_FORCE_BLUR_PATH': '2'})
subprocess.check_call([sys.executable, __file__, '/tmp/b.npy'])
a = np.load('/tmp/a.npy').astype(int); b = np.load('/tmp/b.npy').astype(int)
d = np.abs(a - b).max(axis=2); ys, xs = np.nonzero(d > 1)
print('differing >1:', len(ys), 'max', d.max())
if len(ys):
    print('bbox x', xs.min(), xs.max(), 'y', ys.min(), ys.max())
    print('by x%32:', np.bincount(xs % 32, minlength=32)); print('by y%32:', np.bincount(ys % 32, minlength=32))
    print('cols:', np.unique(xs // 32)[:20], 'rows:', np.unique(ys // 32)[:20])
    for i in range(0, len(ys), max(1, len(ys) // 8)): print(xs[i], ys[i], a[ys[i], xs[i]], b[ys[i], xs[i]])
blk = sorted(set(zip((xs // 32).tolist(), (ys // 32).tolist())))
print(len(blk), 'blocks:', blk[:60])
import collections
print('channel diffs:', [(int((np.abs(a - b)[..., c] > 1).sum())) for c in range(4)])
print('b values where differ, R:', np.unique(b[ys, xs, 0])[:10], 'a values R:', np.unique(a[ys, xs, 0])[:20])
