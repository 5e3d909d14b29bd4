import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_png(name):
    import numpy as np
    from PIL import Image

    return np.array(Image.open(os.path.join(GOLDEN, name)).convert("RGBA"))


def diff_stats(a, b):
    import numpy as np

    d = np.abs(a.astype(int) - b.astype(int)).max(axis=2)
    return int(d.max()), int((d > 0).sum()), int((d > 1).sum())


@pytest.fixture(scope="session")
def golden_manifest():
    import json

    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)
