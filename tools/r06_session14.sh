#!/bin/bash
# k_blur_small's tile shape: kernel time (rocprofv3, 300 replayed 4K bench frames) and the frame's checksum per variant library
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; o=$R/gpurun_out/s14; mkdir -p $o
for lib in figdraw_amd/libfigdraw_hip.so build/libfigdraw_hip_sm_rs0_v2.so build/libfigdraw_hip_sm_rs1_v2.so build/libfigdraw_hip_sm_rs0_v1.so figdraw_amd/libfigdraw_hip.so build/libfigdraw_hip_sm_rs0_v2.so; do
  n=$(basename $lib .so); export FIGDRAW_HIP_LIB=$R/$lib N=300
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$n -o t -- python3 $R/tools/one_frame.py > $o/log_$n.txt 2>&1 < /dev/null
  f=$(find $o/prof_$n -name "*kernel_stats.csv" | head -1)
  echo "== $n: $([ -n "$f" ] && grep k_blur_small "$f" | cut -d, -f2-4,6-7)"
  timeout 100 python3 - <<P 2>/dev/null | tail -1
import sys, hashlib; sys.path.insert(0, "$R")
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
c = HipContext(device=0); h = hashlib.md5()
for (w, hh, f) in ((3840, 2160, 3), (700, 500, 5), (1000, 700, 9)):
    c.render_frame(make_render_tree_100(w, hh, f, full_frame_blur=(w > 3000)), w, hh); h.update(c.read_pixels().tobytes())
print("   frames md5", h.hexdigest())
P
done
