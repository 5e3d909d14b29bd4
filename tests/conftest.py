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


def load_flippy_levels(name):
    """the stored levels of a Flippy container (common/formatflippy.nim:77-99): list of (h, w, 4) uint8 STRAIGHT-alpha arrays.
    Raw-snappy blocks decoded here in Python, independently of the C parsers under test."""
    import struct

    import numpy as np

    d = open(os.path.join(GOLDEN, name), "rb").read()
    assert d[:4] == b"flip" and struct.unpack("<I", d[4:8])[0] == 1

    def snappy(b):
        n = shift = p = 0
        while True:
            c = b[p]; p += 1; n |= (c & 0x7F) << shift; shift += 7
            if c < 0x80:
                break
        out = bytearray()
        while p < len(b):
            t = b[p]; p += 1
            if t & 3 == 0:
                ln = t >> 2
                if ln >= 60:
                    nb = ln - 59; ln = int.from_bytes(b[p:p + nb], "little"); p += nb
                ln += 1; out += b[p:p + ln]; p += ln
                continue
            if t & 3 == 1:
                ln = ((t >> 2) & 7) + 4; off = ((t >> 5) << 8) | b[p]; p += 1
            elif t & 3 == 2:
                ln = (t >> 2) + 1; off = b[p] | (b[p + 1] << 8); p += 2
            else:
                ln = (t >> 2) + 1; off = int.from_bytes(b[p:p + 4], "little"); p += 4
            for _ in range(ln):
                out.append(out[-off])
        assert len(out) == n
        return bytes(out)

    p, levels = 8, []
    while p < len(d):
        assert d[p:p + 4] == b"mip!"
        w, h, z = struct.unpack("<III", d[p + 4:p + 16])
        levels.append(np.frombuffer(snappy(d[p + 16:p + 16 + z]), np.uint8).reshape(h, w, 4).copy())
        p += 16 + z
    return levels


def diff_stats(a, b):
    import numpy as np

    d = np.abs(a.astype(int) - b.astype(int)).max(axis=2)
    return int(d.max()), int((d > 0).sum()), int((d > 1).sum())


@pytest.fixture(scope="session")
def golden_manifest():
    import json

    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)
