#!/bin/bash
python3 - <<'PY' &
import os, sys
sys.path.insert(0, os.getcwd())
from figdraw_amd.context import HipContext
from figdraw_amd.scenes import make_render_tree_100
w, h = 3840, 2160
c = HipContext(device=0)
c.render_frame(make_render_tree_100(w, h, frame=0, full_frame_blur=True), w, h)
import time
t0 = time.time()
while time.time() - t0 < 8: c.replay(200)
c.sync()
PY
sleep 3
for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -3; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1; sleep 0.5; done
wait
