import sys, os, subprocess, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
code = r'''
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from test_hip_parity import _full_frame_blur_scene
from figdraw_amd.context import HipContext
ctx = HipContext(device=0)
ctx.render_frame(_full_frame_blur_scene(644, 388, 12.0, 1), 644, 388, color=(0.2,0.3,0.4,0.5))
np.save(sys.argv[1], ctx.read_pixels())
'''
subprocess.check_call([sys.executable,'-c',code,'/tmp/a.npy'])
subprocess.check_call([sys.executable,'-c',code,'/tmp/b.npy'], env=dict(os.environ, FDH_BLUR_FUSED='0'))
a=np.load('/tmp/a.npy'); b=np.load('/tmp/b.npy')
ys,xs=np.nonzero((a!=b).any(axis=2))
for y,x in zip(ys,xs): print(x,y,a[y,x],b[y,x])
