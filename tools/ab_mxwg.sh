#!/bin/bash
# GPU-box helper: waves per workgroup in the blur kernels (FDH_MX_WG_H / _V / FDH_FX_WG variants under build/), the 4K bench frame:
# two-pass route (H, V) and the fused kernel (frame time one at a time through replay)
for v in "" $*; do
  lib=""; [ -n "$v" ] && lib=$PWD/build/libfigdraw_hip_$v.so
  echo "variant ${v:-product}"
  for i in 1 2; do FIGDRAW_HIP_LIB=$lib python3 - <<'PY'
import os, sys, zlib; sys.path.insert(0, os.getcwd())
if not os.environ.get("FIGDRAW_HIP_LIB"): os.environ.pop("FIGDRAW_HIP_LIB", None)
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
sc = make_render_tree_100(3840, 2160, 0, full_frame_blur=True)
c = HipContext(device=0); c.set_blur_route(0)
c.render_frame(sc, 3840, 2160); c.replay(10); c.profile(60)
st = c.frame_stats()
crc = zlib.crc32(c.read_pixels().tobytes())
d = HipContext(device=0); d.set_blur_route(1)
d.render_frame(sc, 3840, 2160); d.replay(10); d.profile(60)
sf = d.frame_stats()
print('two-pass H %.2f us  V %.2f us   fused %.2f us  crc %08x %08x' % (1e3 * st.ms_blur_big_h, 1e3 * st.ms_blur_big_v, 1e3 * sf.ms_blur_fused, crc, zlib.crc32(d.read_pixels().tobytes())))
PY
  done
done
