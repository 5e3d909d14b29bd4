"""figdraw_amd -- MI355X-native offscreen rasteriser for figdraw's per-pixel SDF path.

Only what the hot path needs lives here:
  csrc/      hand-written HIP kernels + the C-ABI library (libfigdraw_hip.so)
  scene.py   host-side mirror of figdraw's Renders/RenderList/Fig scene model
  context.py HipContext: the BackendContext-shaped binding over the C ABI
  scenes.py  workload generators (BASELINE.json configs)
"""
from . import scene  # noqa: F401

__version__ = "0.1.0"
