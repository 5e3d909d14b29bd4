"""CPU suite, part 1: the oracle against everything that pins it.

  * the reference's own golden PNGs (tests/expected/*.png, committed as tests/golden/ref_*.png)
  * the reference's GLSL run on SwiftShader (tests/golden/ss_*.png, made by tools/make_goldens.py)
  * known answers restated from tests/ttransform.nim and figbackend.nim
Tolerance: 1 LSB per channel on every pixel (the north-star bar is 2).
"""
import numpy as np
import pytest

import ref_scenes as RS
from conftest import diff_stats, load_png
from figdraw_amd.scene import (Fig, FigFlags, FigKind, FillGradientAxis, Renders, fill, linear, rect, rgba)
from oracle import oracle as O


def _render(fn, w, h, threads=4):
    o = O.Oracle(threads=threads)
    o.render_frame(fn(float(w), float(h)), w, h)
    return o.read_pixels()


@pytest.mark.parametrize("name", ["rgb_boxes_sdf", "linear_gradient", "layers_clip"])
def test_oracle_matches_reference_pngs(name):
    fn, w, h, png = RS.REFERENCE_PNG_SCENES[name]
    img = _render(fn, w, h)
    mx, n0, n1 = diff_stats(img, load_png("ref_" + png))
    assert mx <= 1, (name, mx, n0, n1)


def test_rect_mask_variant_within_reference_threshold():
    # trender_layers_clip.nim:322-325 accepts the rect-mask variant against the clip PNG at diff <= 1 %:
    # the two differ only on the clipped edge columns (alpha vs alpha^3).
    fn, w, h, png = RS.REFERENCE_PNG_SCENES["layers_rect_mask"]
    img = _render(fn, w, h)
    exp = load_png("ref_" + png)
    d = np.abs(img.astype(int) - exp.astype(int)).max(axis=2)
    assert (d > 1).sum() < 0.001 * w * h
    # the reference's own point checks (trender_layers_clip.nim:272-289)
    for (x, y, c) in [(int(24 + 24 + 312 * 0.9), int(37.5 + 144 + 32), (43, 159, 234)), (720, int(37.5 + 144 + 32), (255, 255, 255)),
                      (int(400 + 24 + 312 * 0.2), int(37.5 + 144 + 32), (43, 159, 234)), (int(24 + 120), int(37.5 + 240 + 32), (208, 208, 208)),
                      (int(400 + 120), int(37.5 + 240 + 32), (208, 208, 208))]:
        assert np.abs(img[y, x, :3].astype(int) - np.array(c)).max() <= 12


@pytest.mark.parametrize("name", sorted(list(RS.SWIFTSHADER_SCENES) + list(RS.REFERENCE_PNG_SCENES)))
def test_oracle_matches_reference_shaders_on_swiftshader(name):
    if name in RS.SWIFTSHADER_SCENES:
        fn, w, h = RS.SWIFTSHADER_SCENES[name]
    else:
        fn, w, h, _ = RS.REFERENCE_PNG_SCENES[name]
    img = _render(fn, w, h)
    mx, n0, n1 = diff_stats(img, load_png(f"ss_{name}.png"))
    assert mx <= 1, (name, mx, n0, n1)


def test_reference_point_checks():
    # tfigrender_oneframe_screenshot.nim:89-92
    img = _render(RS.oneframe, 240, 160)
    assert img.shape == (160, 240, 4)
    assert np.abs(img[12, 12, :3].astype(int) - 255).max() <= 12
    assert np.abs(img[48, 64, :3].astype(int) - np.array([220, 40, 40])).max() <= 12
    # trender_linear_gradient.nim:123-138
    img = _render(RS.linear_gradient, 800, 600)
    for (x, y, c) in [(120, 140, (220, 40, 40)), (300, 140, (40, 200, 90)), (480, 140, (50, 90, 225)),
                      (190, 270, (240, 210, 40)), (190, 430, (110, 60, 210))]:
        assert np.abs(img[y, x, :3].astype(int) - np.array(c)).max() <= 40
    assert int(img[252, 365, 0]) > int(img[252, 365, 2]) + 40 and int(img[252, 555, 2]) > int(img[252, 555, 0]) + 40
    assert int(img[400, 602, 0]) > int(img[400, 602, 2]) + 20 and int(img[400, 768, 2]) > int(img[400, 768, 0]) + 20
    # trender_layers_clip.nim:350-353
    img = _render(RS.rect_mask_mixed_batch, 480, 180)
    for (x, y, c) in [(74, 88, (230, 70, 52)), (160, 88, (255, 255, 255)), (204, 88, (56, 168, 88)), (336, 88, (54, 118, 230))]:
        assert np.abs(img[y, x, :3].astype(int) - np.array(c)).max() <= 12


@pytest.mark.parametrize("radius", [0.4, 1.0, 5.0, 18.0, 64.0, 100.0])
def test_blur_matches_reference_blur_frag(radius):
    src = load_png("blur_src.png")
    out = O.blur_image(src, radius)
    mx, n0, n1 = diff_stats(out, load_png(f"ss_blur_r{radius:g}.png"))
    assert mx <= 1


def _record(renders, w=64, h=64):
    o = O.Oracle()
    o.record_begin()
    o.render_frame(renders, w, h)
    return o.record_calls()


def _xf_point(calls, upto, x, y):
    """Apply the recorded transform stack state at call index `upto` to (x, y)."""
    import math

    m = np.eye(3)
    stack = []
    for c in calls[:upto]:
        if c[0] == "save_transform":
            stack.append(m.copy())
        elif c[0] == "restore_transform":
            m = stack.pop()
        elif c[0] == "translate":
            m = m @ np.array([[1, 0, c[1]], [0, 1, c[2]], [0, 0, 1]])
        elif c[0] == "scale":
            m = m @ np.diag([c[1], c[2], 1])
        elif c[0] == "rotate":
            m = m @ np.array([[math.cos(c[1]), -math.sin(c[1]), 0], [math.sin(c[1]), math.cos(c[1]), 0], [0, 0, 1]])
        elif c[0] == "apply_transform":
            a = np.array(c[1]).reshape(4, 4).T
            m = m @ np.array([[a[0, 0], a[0, 1], a[0, 3]], [a[1, 0], a[1, 1], a[1, 3]], [0, 0, 1]])
    p = m @ np.array([x, y, 1.0])
    return p[0], p[1]


def test_known_answers_from_ttransform():
    """tests/ttransform.nim:146-267 restated with rectangle children (nkDrawable is a 'next' row)."""
    K = FigKind
    # elliptical radii passthrough (:147-166)
    r = Renders()
    r.addRoot(0, Fig(kind=K.nkRectangle, screenBox=rect(5, 7, 40, 20), fill=rgba(255, 0, 0, 255),
                     flags=FigFlags.NfEllipticalCorners, corners=[12, 10, 8, 6], cornerRadiiY=[4, 5, 6, 7]))
    draws = [c for c in _record(r) if c[0] == "draw_rounded_rect_sdf"]
    assert len(draws) == 1 and draws[0][3] == [12, 10, 8, 6] and draws[0][4] == [4, 5, 6, 7]
    # circular promotion (:168-185)
    r = Renders()
    r.addRoot(0, Fig(kind=K.nkRectangle, screenBox=rect(5, 7, 40, 20), fill=rgba(255, 0, 0, 255), corners=[12, 10, 8, 6]))
    draws = [c for c in _record(r) if c[0] == "draw_rounded_rect_sdf"]
    assert draws[0][3] == draws[0][4]
    # backdrop blur radii (:187-205)
    r = Renders()
    r.addRoot(0, Fig(kind=K.nkBackdropBlur, flags=FigFlags.NfEllipticalCorners, screenBox=rect(5, 7, 40, 20),
                     corners=[12, 10, 8, 6], cornerRadiiY=[4, 5, 6, 7], blur=10.0))
    b = [c for c in _record(r) if c[0] == "draw_backdrop_blur"]
    assert len(b) == 1 and b[0][2] == [12, 10, 8, 6] and b[0][3] == [4, 5, 6, 7]
    # translation (:207-234): rect at (2,2) under translation (5,-4) -> (7,-2)
    r = Renders()
    t = r.addRoot(0, Fig(kind=K.nkTransform, translation=(5.0, -4.0)))
    r.addChild(0, t, Fig(kind=K.nkRectangle, screenBox=rect(2, 2, 1, 1), fill=rgba(255, 0, 0, 255)))
    calls = _record(r)
    i = next(i for i, c in enumerate(calls) if c[0] == "draw_rounded_rect_sdf")
    x, y = _xf_point(calls, i, *calls[i][1][:2])
    assert abs(x - 7.0) < 1e-4 and abs(y + 2.0) < 1e-4
    # matrix (:236-267): translate(10,20) then scale(2,3): (2,2) -> (14,26)
    r = Renders()
    t = r.addRoot(0, Fig(kind=K.nkTransform, translation=(10.0, 20.0), useMatrix=True,
                         matrix=[2, 0, 0, 0, 0, 3, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]))
    r.addChild(0, t, Fig(kind=K.nkRectangle, screenBox=rect(2, 2, 1, 1), fill=rgba(255, 0, 0, 255)))
    calls = _record(r)
    i = next(i for i, c in enumerate(calls) if c[0] == "draw_rounded_rect_sdf")
    x, y = _xf_point(calls, i, *calls[i][1][:2])
    assert abs(x - 14.0) < 1e-4 and abs(y - 26.0) < 1e-4


def test_decomposition_order_and_skip_rules():
    """figrender.nim:1756-1839 stage order; :659-663,721-724,837,855 skip rules."""
    sc = RS.rgb_boxes_sdf()
    modes = [c[5] for c in _record(sc, 800, 600) if c[0] == "draw_rounded_rect_sdf"]
    assert modes == [3, 3, 12, 7, 3, 3, 9, 9]  # SURVEY.md App. A draw sequence
    r = Renders()
    from figdraw_amd.scene import RenderShadow, RenderStroke, ShadowStyle
    r.addRoot(0, Fig(kind=FigKind.nkRectangle, screenBox=rect(1, 1, 10, 10), fill=rgba(0, 0, 0, 0),
                     stroke=RenderStroke(weight=0.0, fill=fill(rgba(0, 0, 0, 255))),
                     shadows=[RenderShadow(style=ShadowStyle.DropShadow, blur=0, spread=0, fill=fill(rgba(0, 0, 0, 255))),
                              RenderShadow(style=ShadowStyle.DropShadow, blur=3, spread=0, fill=fill(rgba(0, 0, 0, 0))),
                              RenderShadow(style=ShadowStyle.InnerShadow, blur=0, spread=0, fill=fill(rgba(0, 0, 0, 255)))]))
    assert [c for c in _record(r) if c[0] == "draw_rounded_rect_sdf"] == []
    # disabled subtree
    r = Renders()
    p = r.addRoot(0, Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, 8, 8), fill=rgba(1, 2, 3, 255), flags=FigFlags.NfDisableRender))
    r.addChild(0, p, Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, 4, 4), fill=rgba(1, 2, 3, 255)))
    assert [c for c in _record(r) if c[0].startswith("draw")] == []


def test_gradient_colors_and_radii_packing():
    """figbackend.nim:161-183 corner order BL,BR,TR,TL; glcontext.nim:745-817 + SURVEY.md §8c elliptical vector."""
    g = O.gradient_colors(linear(rgba(0, 0, 0, 255), rgba(255, 255, 255, 255), axis=FillGradientAxis.fgaX))
    assert g == [(0, 0, 0, 255), (255, 255, 255, 255), (255, 255, 255, 255), (0, 0, 0, 255)]
    g = O.gradient_colors(linear(rgba(0, 0, 0, 255), rgba(255, 255, 255, 255), axis=FillGradientAxis.fgaY))
    assert g == [(255, 255, 255, 255), (255, 255, 255, 255), (0, 0, 0, 255), (0, 0, 0, 255)]
    g = O.gradient_colors(linear(rgba(0, 0, 0, 255), rgba(255, 255, 255, 255), axis=FillGradientAxis.fgaDiagTLBR))
    assert g == [(128, 128, 128, 255), (255, 255, 255, 255), (128, 128, 128, 255), (0, 0, 0, 255)]
    g = O.gradient_colors(linear(rgba(0, 0, 0, 255), rgba(100, 100, 100, 255), rgba(255, 255, 255, 255), axis=FillGradientAxis.fgaDiagBLTR, midPos=64))
    assert g[0] == (0, 0, 0, 255) and g[2] == (255, 255, 255, 255) and g[1] == g[3]
    # circular: clampRadius = round(max(1, min(r, min(hx,hy)))); order TR,BR,TL,BL
    r4, e = O.rounded_radii_vec([10, 20, 30, 40], [10, 20, 30, 40], 110, 70)
    assert (r4, e) == ([20, 40, 10, 30], False)
    r4, e = O.rounded_radii_vec([0.4, 200, 0, -3], [0.4, 200, 0, -3], 50, 30)
    assert (r4, e) == ([30, 0, 1, 0], False)
    # elliptical: x=30,10,8,40 / y=30,20,8,80 on half extents (100.25, 60.125)
    r4, e = O.rounded_radii_vec([30, 10, 8, 40], [30, 20, 8, 80], 100.25, 60.125)
    assert e and r4 == [5579160.0, 16737890.0, -31.0, -9.0]


@pytest.mark.parametrize("name", sorted(RS.ATLAS_SCENES))
def test_oracle_atlas_sampling_matches_reference_shaders(name):
    """Mode 0 (1:1 glyphs, magnified / minified / flipped images through the mip chain), MSDF / MTSDF / annular MSDF.
    Tolerance 2 LSB: SwiftShader quantises texture coordinates to 16 normalised bits (ref_scenes.ATLAS_GOLDEN_SIZE)."""
    import os

    from conftest import GOLDEN
    from figdraw_amd.scenes import load_glyph_fixture

    fn, w, h = RS.ATLAS_SCENES[name]
    all_images = load_glyph_fixture(os.path.join(GOLDEN, "glyphs_ubuntu20.npz"))
    sc = fn(float(w), float(h), all_images)
    o = O.Oracle(atlas_size=RS.ATLAS_GOLDEN_SIZE, threads=4)
    for k, img in RS.used_images(sc, all_images).items():
        o.put_image(k, img)
    o.render_frame(sc, w, h)
    mx, n0, n1 = diff_stats(o.read_pixels(), load_png(f"ss_{name}.png"))
    assert mx <= 2 and n1 <= 0.001 * w * h, (name, mx, n0, n1)


def test_atlas_packer_known_answers():
    """findEmptyRect (glcontext.nim:541-579): skyline with margin 4; entries are packed pixel rects."""
    o = O.Oracle(atlas_size=256)
    a = o.put_image(1, np.zeros((10, 20, 4), np.uint8))   # first image lands at (margin, margin)
    b = o.put_image(2, np.zeros((10, 20, 4), np.uint8))   # next free column run: x = 20 + 2*4 + 4
    c = o.put_image(3, np.zeros((30, 200, 4), np.uint8))  # does not fit beside them: goes where the skyline is lowest
    assert a == (4, 4, 20, 10)
    assert b == (32, 4, 20, 10)
    assert c[2:] == (200, 30) and c[0] >= 4 and c[1] >= 4
    with pytest.raises(RuntimeError):
        o.put_image(4, np.zeros((300, 300, 4), np.uint8))
