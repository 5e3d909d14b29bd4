#!/usr/bin/env python3
"""Regenerates tests/golden/ (run in the build container only; needs /root/reference + SwiftShader).

  ref_<name>.png   byte-for-byte image content of the reference's own test fixtures
                   (tests/expected/*.png in the reference): data the reference's tests hold.
  ss_<name>.png    the reference's GLSL shaders run on SwiftShader (oracle/ref_swiftshader.py) over the
                   BackendContext call stream the scene decomposes into -- the stream recorded from the HIP
                   library's OWN front-end (fdh_record_begin / fdh_record_json on a FDH_CREATE_RECORD_ONLY
                   context), checked here to equal the oracle's: the goldens pin the product's decomposition,
                   not only the oracle's.
  manifest.json    sizes, provenance and the measured oracle-vs-golden agreement at generation time.
"""
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_scenes as RS  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import ref_swiftshader as R  # noqa: E402
from figdraw_amd.context import HipContext  # noqa: E402
from test_frontend_calls import _same  # noqa: E402


def hip_calls(scene, w, h, atlas_size=1024, images=None):
    """the call stream of the HIP library's front-end for `scene` (no GPU: a record-only context)"""
    ctx = HipContext(atlas_size=atlas_size, record_only=True)
    for k in sorted(images or {}):
        ctx.put_image(k, images[k])
    ctx.record_begin()
    ctx.render_frame(scene, w, h)
    calls = ctx.record_calls()
    ctx.close()
    return calls

GOLD = os.path.join(ROOT, "tests", "golden")
REF_EXPECTED = "/root/reference/tests/expected"


def stats(a, b):
    d = np.abs(a.astype(int) - b.astype(int)).max(axis=2)
    return {"max": int(d.max()), "n_gt0": int((d > 0).sum()), "n_gt1": int((d > 1).sum())}


def big_blur_goldens():
    """blur.frag (glsl/blur.frag:11-32, two passes with RGBA8 between them: glcontext.nim:1743-1786) over HOSTILE content at a size the
    product blurs on its matrix-pipe kernels without being forced to (>= 384 K pixels): opaque white noise and a 0 / 255 checkerboard,
    both rebuilt from their recipe by the tests (ref_scenes.hostile_blur_source) -- only the blurred frames are stored."""
    out = {}
    w, h = RS.HOSTILE_BLUR_SIZE
    for kind in RS.HOSTILE_BLUR_KINDS:
        src = RS.hostile_blur_source(kind)
        for radius in RS.HOSTILE_BLUR_RADII:
            gl = R.RefGL(w, h)
            ss = gl.blur_only(src, radius)
            gl.close()
            Image.fromarray(ss).save(os.path.join(GOLD, f"ss_blur_big_{kind}_r{radius:g}.png"), optimize=True)
            out[f"blur_big_{kind}_r{radius:g}"] = {"width": w, "height": h, "oracle_vs_swiftshader": stats(O.blur_image(src, radius), ss)}
            print("blur_big", kind, radius, out[f"blur_big_{kind}_r{radius:g}"])
    return out


def main():
    assert R.available(), "needs /root/reference and SwiftShader"
    os.makedirs(GOLD, exist_ok=True)
    manifest = {}
    scenes = {k: v[:3] for k, v in RS.REFERENCE_PNG_SCENES.items()}
    scenes.update(RS.SWIFTSHADER_SCENES)
    scenes.update({k: v[:3] for k, v in RS.OUTLIER_SCENES.items()})
    for name, (fn, w, h) in scenes.items():
        o = O.Oracle(threads=8)
        o.record_begin()
        o.render_frame(fn(float(w), float(h)), w, h)
        calls = hip_calls(fn(float(w), float(h)), w, h)
        _same(o.record_calls(), calls)
        img = o.read_pixels()
        ss = R.replay(calls, w, h)
        Image.fromarray(ss).save(os.path.join(GOLD, f"ss_{name}.png"), optimize=True)
        entry = {"width": w, "height": h, "n_calls": len(calls), "calls_from": "libfigdraw_hip front-end (== oracle's)",
                 "oracle_vs_swiftshader": stats(img, ss)}
        if name in RS.REFERENCE_PNG_SCENES:
            png = RS.REFERENCE_PNG_SCENES[name][3]
            exp = np.array(Image.open(os.path.join(REF_EXPECTED, png)).convert("RGBA"))
            Image.fromarray(exp).save(os.path.join(GOLD, f"ref_{png}"), optimize=True)
            entry["reference_png"] = png
            entry["oracle_vs_reference_png"] = stats(img, exp)
            entry["swiftshader_vs_reference_png"] = stats(ss, exp)
        manifest[name] = entry
        print(name, entry)
    # the reference's Flippy fixture and the PNG its own test expects for it (tests/trender_image.nim:13-38)
    import shutil

    shutil.copyfile("/root/reference/data/img1.flippy", os.path.join(GOLD, "img1.flippy"))
    exp = np.array(Image.open(os.path.join(REF_EXPECTED, "render_image.png")).convert("RGBA"))
    Image.fromarray(exp).save(os.path.join(GOLD, "ref_render_image.png"), optimize=True)
    o = O.Oracle(atlas_size=2048, threads=8)
    o.put_flippy(RS.FLIPPY_IMAGE_KEY, open(os.path.join(GOLD, "img1.flippy"), "rb").read())
    o.render_frame(RS.image_flippy(), 800, 600)
    manifest["image_flippy"] = {"width": 800, "height": 600, "oracle_vs_reference_png": stats(o.read_pixels(), exp)}
    print("image_flippy", manifest["image_flippy"])
    # atlas scenes: images uploaded in sorted-key order into a 256^2 atlas (see ref_scenes.ATLAS_GOLDEN_SIZE)
    from figdraw_amd.scenes import load_glyph_fixture

    all_images = load_glyph_fixture(os.path.join(GOLD, "glyphs_ubuntu20.npz"))
    for name, (fn, w, h) in RS.ATLAS_SCENES.items():
        sc = fn(float(w), float(h), all_images)
        images = RS.used_images(sc, all_images)
        o = O.Oracle(atlas_size=RS.ATLAS_GOLDEN_SIZE, threads=8)
        for k in sorted(images):
            o.put_image(k, images[k])
        o.record_begin()
        o.render_frame(sc, w, h)
        calls = hip_calls(sc, w, h, atlas_size=RS.ATLAS_GOLDEN_SIZE, images=images)
        _same(o.record_calls(), calls)
        img = o.read_pixels()
        ss = R.replay(calls, w, h, atlas_size=RS.ATLAS_GOLDEN_SIZE, images=images)
        Image.fromarray(ss).save(os.path.join(GOLD, f"ss_{name}.png"), optimize=True)
        manifest[name] = {"width": w, "height": h, "n_calls": len(calls), "atlas_size": RS.ATLAS_GOLDEN_SIZE,
                          "n_images": len(images), "calls_from": "libfigdraw_hip front-end (== oracle's)", "oracle_vs_swiftshader": stats(img, ss)}
        print(name, manifest[name])
    # blur-only vectors: random RGBA8 through blur.frag H+V at several radii
    rng = np.random.default_rng(7)
    src = rng.integers(0, 256, size=(64, 96, 4), dtype=np.uint8)
    Image.fromarray(src).save(os.path.join(GOLD, "blur_src.png"))
    for radius in (0.4, 1.0, 5.0, 18.0, 64.0, 100.0):
        gl = R.RefGL(96, 64)
        out = gl.blur_only(src, radius)
        gl.close()
        Image.fromarray(out).save(os.path.join(GOLD, f"ss_blur_r{radius:g}.png"))
        manifest[f"blur_r{radius:g}"] = {"oracle_vs_swiftshader": stats(O.blur_image(src, radius), out)}
        print("blur", radius, manifest[f"blur_r{radius:g}"])
    manifest.update(big_blur_goldens())
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1:] == ["big_blur"]:  # only the hostile-content blur frames (the manifest keeps its other entries)
        with open(os.path.join(GOLD, "manifest.json")) as f:
            m = json.load(f)
        m.update(big_blur_goldens())
        with open(os.path.join(GOLD, "manifest.json"), "w") as f:
            json.dump(m, f, indent=1, sort_keys=True)
    else:
        main()
