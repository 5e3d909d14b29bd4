#!/usr/bin/env python3
"""Builds tests/golden/glyphs_ubuntu20.npz (container only: reads the reference's data/Ubuntu.ttf).

The reference rasterises glyphs with pixie (third-party, not under /root/reference) and builds MSDF textures with
sdfy (third-party, only used by one example), so no reference test pins glyph TEXEL values: parity
for the atlas path is defined on SAMPLING -- reference shaders / oracle / HIP all sample these same bytes.

  cov_<c>   (h, w, 4) uint8  premultiplied white coverage glyph, 20 px Ubuntu, ASCII 33..126
            (what pixie hands to putImage: white * coverage in all four channels)
  msdf_<c>  (32, 32, 4) uint8 multi-channel distance field with pxRange 4: the true signed distance in the
            median channel plus a deterministic per-channel perturbation so median3() is exercised; alpha
            carries the plain distance (MTSDF layout)
"""
import os
import sys

import numpy as np
from PIL import Image, ImageDraw, ImageFont
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FONT = "/root/reference/data/Ubuntu.ttf"


def coverage(font, ch):
    x0, y0, x1, y1 = font.getbbox(ch)
    w, h = max(1, x1 - x0), max(1, y1 - y0)
    im = Image.new("L", (w, h), 0)
    ImageDraw.Draw(im).text((-x0, -y0), ch, font=font, fill=255)
    a = np.array(im, dtype=np.uint8)
    return np.stack([a, a, a, a], axis=2)  # premultiplied white


def msdf(ch, size=32, px_range=4.0, ss=8):
    font = ImageFont.truetype(FONT, int(size * 0.72 * ss))
    x0, y0, x1, y1 = font.getbbox(ch)
    W = size * ss
    im = Image.new("L", (W, W), 0)
    ox = (W - (x1 - x0)) // 2 - x0
    oy = (W - (y1 - y0)) // 2 - y0
    ImageDraw.Draw(im).text((ox, oy), ch, font=font, fill=255)
    inside = np.array(im) >= 128
    d_out = ndimage.distance_transform_edt(~inside)
    d_in = ndimage.distance_transform_edt(inside)
    sd = (d_in - d_out) / ss  # signed distance in output texels, positive inside
    sd = sd.reshape(size, ss, size, ss).mean(axis=(1, 3))
    base = np.clip(0.5 + sd / px_range, 0.0, 1.0)
    yy, xx = np.mgrid[0:size, 0:size]
    wob = 0.06 * np.sin(0.9 * xx + 0.3) * np.cos(0.7 * yy + 0.5)
    r = np.clip(base + np.abs(wob), 0, 1)
    g = base
    b = np.clip(base - np.abs(wob), 0, 1)
    out = np.stack([r, g, b, base], axis=2)
    return np.floor(out * 255.0 + 0.5).astype(np.uint8)


def main():
    font = ImageFont.truetype(FONT, 20)
    arrays = {}
    for code in range(33, 127):
        ch = chr(code)
        arrays[f"cov_{code}"] = coverage(font, ch)
        arrays[f"msdf_{code}"] = msdf(ch)
    # the reference's own test image for nkImage (tests/trender_image.nim draws data/img1.png): premultiplied like pixie's loadImage
    img = np.array(Image.open("/root/reference/data/img1.png").convert("RGBA")).astype(np.float32)
    img[..., :3] = np.floor(img[..., :3] * img[..., 3:4] / 255.0 + 0.5)
    arrays["img1_premul"] = img.astype(np.uint8)
    out = os.path.join(ROOT, "tests", "golden", "glyphs_ubuntu20.npz")
    np.savez_compressed(out, **arrays)
    print("wrote", out, os.path.getsize(out), "bytes;", len(arrays), "arrays")


if __name__ == "__main__":
    main()
