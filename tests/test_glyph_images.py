"""Glyph images on their way into the atlas: the LCD filter the reference applies to a rasterised glyph
(common/textrasters/pixie_raster.nim:12-43, FreeType's default 5-tap filter) -- an integer algorithm spelled out in the reference
itself, so it is pinned: known answers on the oracle's restatement (CPU), the HIP kernel against the oracle through sampling (GPU)."""
import numpy as np
import pytest

from oracle import oracle as O


def _numpy_lcd(img):
    """pixie_raster.nim:12-43 once more, vectorised: an independent restatement for the oracle to be checked against"""
    h, w, _ = img.shape
    wt = np.array([8, 77, 86, 77, 8], dtype=np.int64)
    xs = np.clip(np.arange(w)[:, None] + np.arange(5)[None, :] - 2, 0, w - 1)  # (w, 5)
    taps = img.astype(np.int64)[:, xs, :]                                      # (h, w, 5, 4)
    return (((taps * wt[None, None, :, None]).sum(axis=2) + 128) >> 8).astype(np.uint8)


def test_lcd_filter_known_answers():
    # impulse response: a single 255 texel spreads into 8, 77, 86, 77, 8 (the weights sum to 256)
    img = np.zeros((3, 11, 4), np.uint8)
    img[1, 5] = 255
    out = O.lcd_filter(img)
    assert out[1, 3:8, 0].tolist() == [8, 77, 86, 77, 8] and out[1, 3:8, 3].tolist() == [8, 77, 86, 77, 8]
    assert out[0].max() == 0 and out[2].max() == 0  # horizontal only
    # a constant image is a fixed point; columns are clamped to the image, so the borders keep their full weight
    img = np.full((2, 7, 4), 200, np.uint8)
    assert (O.lcd_filter(img) == 200).all()
    img = np.zeros((1, 6, 4), np.uint8)
    img[0, 0] = 255  # the first column is read three times through the clamp: 8 + 77 + 86 = 171 of 256
    out = O.lcd_filter(img)[0, :, 1].tolist()
    assert out == [(255 * 171 + 128) >> 8, (255 * 85 + 128) >> 8, (255 * 8 + 128) >> 8, 0, 0, 0]
    # 1 x 1 and 1-wide images (every tap clamps onto the same column)
    assert O.lcd_filter(np.full((4, 1, 4), 37, np.uint8)).ravel().tolist() == [37] * 16


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_lcd_filter_matches_an_independent_restatement(seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(int(rng.integers(1, 40)), int(rng.integers(1, 70)), 4), dtype=np.uint8)
    assert np.array_equal(O.lcd_filter(img), _numpy_lcd(img))


@pytest.mark.gpu
@pytest.mark.parametrize("lcd", [False, True])
def test_device_glyph_upload_matches_oracle(lcd):
    """fdh_put_glyph_image (LCD filter + mip chain as kernels) against the oracle's host-side chain: glyphs drawn 1:1, magnified
    and minified (the minified ones sample the device-built mip levels)."""
    import os

    from conftest import GOLDEN, diff_stats
    from figdraw_amd.context import HipContext
    from figdraw_amd.scene import Fig, FigKind, RenderList, Renders, fill, rect, rgba
    from figdraw_amd.scenes import load_glyph_fixture

    imgs = load_glyph_fixture(os.path.join(GOLDEN, "glyphs_ubuntu20.npz"))
    keys = sorted(k for k in imgs if 1000 <= k < 1100)[:24]
    w, h = 640, 400
    ctx, orc = HipContext(atlas_size=512, device=0), O.Oracle(atlas_size=512, threads=8)
    for k in keys:
        assert ctx.put_glyph_image(k, imgs[k], lcd_filter=lcd) == orc.put_glyph_image(k, imgs[k], lcd_filter=lcd)
    lst = RenderList()
    lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, w, h), fill=rgba(20, 24, 40, 255)))
    for i, k in enumerate(keys):
        gh, gw = imgs[k].shape[:2]
        for row, scale in enumerate((1.0, 2.5, 0.45)):
            f = Fig(kind=FigKind.nkImage, screenBox=rect(12 + 26 * i, 20 + 110 * row + (i % 3) * 0.5, gw * scale, gh * scale), fill=rgba(255, 255, 255, 255))
            f.image_id = k
            lst.addRoot(f)
    sc = Renders()
    sc.setLayer(0, lst)
    ctx.render_frame(sc, w, h)
    orc.render_frame(sc, w, h)
    mx, n0, n1 = diff_stats(ctx.read_pixels(), orc.read_pixels())
    assert mx <= 1 and n0 <= 0.005 * w * h, (lcd, mx, n0, n1)
    if lcd:  # and the filter did something: the unfiltered upload renders differently
        ref = HipContext(atlas_size=512, device=0)
        for k in keys:
            ref.put_glyph_image(k, imgs[k], lcd_filter=False)
        ref.render_frame(sc, w, h)
        assert (ref.read_pixels() != ctx.read_pixels()).any()
        ref.close()
    ctx.close()


# ---------------------------------------------------------------------- glyph outlines -> coverage
NAN = float("nan")


def _poly(points):
    """closed polygon -> line segments in the outline format"""
    n = len(points)
    return np.array([[points[i][0], points[i][1], NAN, NAN, points[(i + 1) % n][0], points[(i + 1) % n][1]] for i in range(n)], np.float32)


def test_rasteriser_known_answers():
    # an axis-aligned box at fractional coordinates: coverage = the exact overlap of every pixel with the box
    x0, y0, x1, y1 = 2.25, 1.5, 7.75, 5.25
    img = O.rasterize_outline(_poly([(x0, y0), (x1, y0), (x1, y1), (x0, y1)]), 10, 8)
    want = np.zeros((8, 10))
    for y in range(8):
        for x in range(10):
            want[y, x] = max(0.0, min(x + 1, x1) - max(x, x0)) * max(0.0, min(y + 1, y1) - max(y, y0))
    assert np.array_equal(img[..., 0], np.floor(want * 255.0 + 0.5).astype(np.uint8))
    assert (img[..., 0] == img[..., 3]).all() and (img[..., 1] == img[..., 2]).all()  # premultiplied white
    # winding direction does not matter (|sum|), and a hole (opposite winding inside) is empty
    cw = O.rasterize_outline(_poly([(x0, y0), (x0, y1), (x1, y1), (x1, y0)]), 10, 8)
    assert np.array_equal(cw, img)
    outer = [(1, 1), (9, 1), (9, 7), (1, 7)]
    hole = [(3, 3), (3, 5), (7, 5), (7, 3)]
    ring = O.rasterize_outline(np.concatenate([_poly(outer), _poly(hole)]), 10, 8)
    assert ring[4, 5, 0] == 0 and ring[2, 5, 0] == 255 and ring[4, 2, 0] == 255
    # a triangle: total coverage = its area (shoelace), up to the 8-bit rounding of each pixel
    tri = [(1.3, 0.7), (8.6, 2.2), (3.1, 6.9)]
    area = 0.5 * abs(sum(tri[i][0] * tri[(i + 1) % 3][1] - tri[(i + 1) % 3][0] * tri[i][1] for i in range(3)))
    cov = O.rasterize_outline(_poly(tri), 10, 8)[..., 0].astype(float).sum() / 255.0
    assert abs(cov - area) < 0.12
    # a quadratic segment: the area under a parabola arch; a shape partly left of / above the image is clipped, not wrapped
    arch = np.array([[1, 6, 5, -2, 9, 6], [9, 6, NAN, NAN, 1, 6]], np.float32)
    cov = O.rasterize_outline(arch, 10, 8)[..., 0].astype(float).sum() / 255.0
    assert abs(cov - (2.0 / 3.0) * 8 * 4) < 0.2  # (curves are flattened to 0.025 px chord error) chord (1,6)-(9,6), apex height 4: area = 2/3 * base * height
    clip = O.rasterize_outline(_poly([(-3, -2), (4.5, -2), (4.5, 3.5), (-3, 3.5)]), 10, 8)
    assert clip[0, 0, 0] == 255 and clip[2, 4, 0] == 128 and clip[3, 0, 0] == 128 and clip[3, 4, 0] == 64 and clip[3, 5, 0] == 0 and clip[4].max() == 0


def test_rasteriser_on_the_font_outlines():
    """every ASCII glyph of data/Ubuntu.ttf at 20 px (tests/golden/outlines_ubuntu20.npz, tools/make_outline_fixture.py): the
    coverage total equals the outline's enclosed area, and agrees with the FreeType rasters of the same glyphs in the atlas
    fixture in total ink within a quarter (FreeType hints: edges move by fractions of a pixel, thin strokes gain weight) -- a sanity
    check that the outlines, their scale and their winding are the font's."""
    import os

    from conftest import GOLDEN

    z = np.load(os.path.join(GOLDEN, "outlines_ubuntu20.npz"))
    ft = np.load(os.path.join(GOLDEN, "glyphs_ubuntu20.npz"))
    for code in range(33, 127):
        segs, (w, h) = z[f"segs_{code}"], z[f"size_{code}"]
        img = O.rasterize_outline(segs, int(w), int(h))[..., 0].astype(float) / 255.0
        # signed area by Green's theorem over the flattened outline (lines exactly, curves via their control polygons' formula)
        area = 0.0
        for x0, y0, cx, cy, x1, y1 in segs.astype(float):
            if np.isnan(cx):
                area += 0.5 * (x0 * y1 - x1 * y0)
            else:  # quadratic: chord term + 1/3 of the control triangle on each side
                area += 0.5 * (x0 * y1 - x1 * y0) + (1.0 / 3.0) * ((x0 - cx) * (y1 - cy) - (x1 - cx) * (y0 - cy)) * -1.0
        assert abs(img.sum() - abs(area)) < 0.02 * abs(area) + 0.3, (chr(code), img.sum(), area)
        ink_ft = ft[f"cov_{code}"][..., 0].astype(float).sum() / 255.0
        assert abs(img.sum() - ink_ft) < 0.25 * ink_ft + 1.5, (chr(code), img.sum(), ink_ft)  # FreeType's hinting thickens thin strokes at 20 px


@pytest.mark.gpu
@pytest.mark.parametrize("lcd", [False, True])
def test_device_rasteriser_matches_oracle(lcd):
    """fdh_put_glyph_outline (k_rasterize_lines, LCD filter, mip chain on the device) against the oracle: the same outlines into
    both atlases, glyphs drawn 1:1 at integer positions on black -- the frame then IS the atlas content, so equality of the
    frames is equality of the texels."""
    import os

    from conftest import GOLDEN
    from figdraw_amd.context import HipContext
    from figdraw_amd.scene import Fig, FigKind, RenderList, Renders, rect, rgba

    z = np.load(os.path.join(GOLDEN, "outlines_ubuntu20.npz"))
    w, h = 1024, 128
    ctx, orc = HipContext(atlas_size=1024, device=0), O.Oracle(atlas_size=1024, threads=8)
    lst = RenderList()
    lst.addRoot(Fig(kind=FigKind.nkRectangle, screenBox=rect(0, 0, w, h), fill=rgba(0, 0, 0, 255)))
    x = 2
    for code in range(33, 127):
        segs, (gw, gh) = z[f"segs_{code}"], z[f"size_{code}"]
        assert ctx.put_glyph_outline(3000 + code, segs, int(gw), int(gh), lcd_filter=lcd) == orc.put_glyph_outline(3000 + code, segs, int(gw), int(gh), lcd_filter=lcd)
        f = Fig(kind=FigKind.nkImage, screenBox=rect(x % 1000, 4 + 40 * (x // 1000), int(gw), int(gh)), fill=rgba(255, 255, 255, 255))
        f.image_id = 3000 + code
        lst.addRoot(f)
        x += int(gw) + 3
    sc = Renders()
    sc.setLayer(0, lst)
    ctx.render_frame(sc, w, h)
    orc.render_frame(sc, w, h)
    got, want = ctx.read_pixels(), orc.read_pixels()
    assert got[..., :3].max() > 200  # something was drawn
    assert np.array_equal(got, want)
    ctx.close()
