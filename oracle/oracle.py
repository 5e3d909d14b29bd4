"""TEST INFRASTRUCTURE: ctypes binding of oracle/libfigdraw_oracle.so (the C restatement).

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import numpy as np

from figdraw_amd import scene as S

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libfigdraw_oracle.so")
_lib = None

_F4 = C.c_float * 4
_F2 = C.c_float * 2
_COL4 = S.CColor * 4


def build(force: bool = False) -> str:
    """gcc the oracle (seconds)."""
    src = os.path.join(_HERE, "figdraw_oracle.c")
    hdr = os.path.join(_HERE, "figdraw_oracle.h")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.fo_create.restype = C.c_void_p
        L.fo_create.argtypes = [C.c_int, C.c_float]
        L.fo_destroy.argtypes = [C.c_void_p]
        L.fo_set_threads.argtypes = [C.c_int]
        L.fo_set_ui_scale.argtypes = [C.c_void_p, C.c_float]
        L.fo_begin_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _F4]
        L.fo_end_frame.argtypes = [C.c_void_p]
        L.fo_save_transform.argtypes = [C.c_void_p]
        L.fo_restore_transform.argtypes = [C.c_void_p]
        L.fo_translate.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.fo_rotate.argtypes = [C.c_void_p, C.c_float]
        L.fo_scale.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.fo_apply_transform.argtypes = [C.c_void_p, C.c_float * 16]
        L.fo_set_aa_factor.argtypes = [C.c_void_p, C.c_float]
        L.fo_draw_rounded_rect_sdf.argtypes = [C.c_void_p, _F4, _COL4, _F4, _F4, C.c_int, C.c_float, C.c_float, _F2,
                                               C.c_int, S.CColor, S.CColor, C.c_float]
        L.fo_draw_rounded_rect_fill.argtypes = [C.c_void_p, _F4, C.POINTER(S.CFill), _F4, _F4, C.c_int, C.c_float,
                                                C.c_float, _F2]
        L.fo_draw_image.argtypes = [C.c_void_p, C.c_int64, _F2, _COL4, _F2, C.c_int]
        L.fo_draw_image_adj.argtypes = [C.c_void_p, C.c_int64, _F2, S.CColor, _F2]
        L.fo_draw_msdf.argtypes = [C.c_void_p, C.c_int64, _F2, S.CColor, _F2, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
        L.fo_draw_quadratic_bezier_sdf.argtypes = [C.c_void_p, _F4, C.POINTER(S.CFill), _F2, _F2, _F2, C.c_float, C.c_int]
        L.fo_draw_filled_quad.argtypes = [C.c_void_p, C.c_float * 8, _COL4]
        L.fo_draw_rect.argtypes = [C.c_void_p, _F4, S.CColor]
        L.fo_draw_backdrop_blur.argtypes = [C.c_void_p, _F4, _F4, _F4, C.c_float]
        L.fo_begin_mask.argtypes = [C.c_void_p, _F4, _F4, _F4]
        L.fo_end_mask.argtypes = [C.c_void_p]
        L.fo_pop_mask.argtypes = [C.c_void_p]
        L.fo_begin_rect_mask.argtypes = [C.c_void_p, _F4, _F4, _F4]
        L.fo_pop_rect_mask.argtypes = [C.c_void_p]
        L.fo_put_glyph_outline.restype = C.c_int
        L.fo_put_glyph_outline.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_uint, C.c_int * 4]
        L.fo_put_glyph_image.restype = C.c_int
        L.fo_put_glyph_image.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_int * 4]
        L.fo_lcd_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.fo_put_image.restype = C.c_int
        L.fo_put_image.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int * 4]
        L.fo_put_flippy.restype = C.c_int
        L.fo_put_flippy.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_size_t, C.c_int * 4]
        L.fo_set_text_subpixel.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.fo_set_text_subpixel_glyph_variants.argtypes = [C.c_void_p, C.c_int]
        L.fo_set_text_subpixel_shift.argtypes = [C.c_void_p, C.c_float]
        L.fo_read_pixels.restype = C.c_int
        L.fo_read_pixels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.fo_read_mask.restype = C.c_int
        L.fo_read_mask.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.fo_render_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, _F4]
        L.fo_record_begin.argtypes = [C.c_void_p]
        L.fo_record_json.restype = C.c_char_p
        L.fo_record_json.argtypes = [C.c_void_p]
        L.fo_rounded_radii_vec.argtypes = [_F4, _F4, C.c_float, C.c_float, _F4, C.POINTER(C.c_int)]
        L.fo_gradient_colors.argtypes = [C.POINTER(S.CFill), _COL4]
        L.fo_blur_image.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float]
        L.fo_sizeof_fig.restype = C.c_int
        L.fo_sizeof_glyph.restype = C.c_int
        assert L.fo_sizeof_fig() == C.sizeof(S.CFig), (L.fo_sizeof_fig(), C.sizeof(S.CFig))
        assert L.fo_sizeof_glyph() == C.sizeof(S.CGlyph)
        L.fo_sizeof_draw_op.restype = C.c_int
        assert L.fo_sizeof_draw_op() == C.sizeof(S.CDrawOp)
        L.fo_sizeof_text_rect.restype = C.c_int
        assert L.fo_sizeof_text_rect() == C.sizeof(S.CTextRect)
        _lib = L
    return _lib


def _cols(colors):
    return _COL4(*[S.CColor(*[int(v) for v in c]) for c in colors])


class Oracle:
    def __init__(self, atlas_size: int = 1024, pixel_scale: float = 1.0, threads: int = 1):
        self.L = lib()
        self.h = self.L.fo_create(atlas_size, pixel_scale)
        self.L.fo_set_threads(threads)
        self.W = self.H = 0

    def close(self):
        if self.h:
            self.L.fo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # ---- BackendContext-level calls (same names as ref_swiftshader.RefGL / figdraw_amd.HipContext)
    def begin_frame(self, w, h, clear=True, color=(1.0, 1.0, 1.0, 1.0)):
        self.W, self.H = int(w), int(h)
        self.L.fo_begin_frame(self.h, int(w), int(h), int(bool(clear)), _F4(*color))

    def end_frame(self):
        self.L.fo_end_frame(self.h)

    def save_transform(self):
        self.L.fo_save_transform(self.h)

    def restore_transform(self):
        self.L.fo_restore_transform(self.h)

    def translate(self, x, y):
        self.L.fo_translate(self.h, x, y)

    def rotate(self, a):
        self.L.fo_rotate(self.h, a)

    def scale(self, sx, sy=None):
        self.L.fo_scale(self.h, sx, sx if sy is None else sy)

    def apply_transform(self, m16):
        self.L.fo_apply_transform(self.h, (C.c_float * 16)(*m16))

    def set_aa_factor(self, aa):
        self.L.fo_set_aa_factor(self.h, aa)

    def draw_rounded_rect_sdf(self, rect, colors, radii_x, radii_y, mode, factor=4.0, spread=0.0, shape=(0.0, 0.0),
                              fill_mode=0, mid=(0, 0, 0, 0), stop=(0, 0, 0, 0), mid_pos=0.5):
        self.L.fo_draw_rounded_rect_sdf(self.h, _F4(*rect), _cols(colors), _F4(*radii_x), _F4(*radii_y), int(mode),
                                        factor, spread, _F2(*shape), int(fill_mode), S.CColor(*mid), S.CColor(*stop), mid_pos)

    def draw_rounded_rect_fill(self, rect, fill: S.Fill, radii_x, radii_y, mode, factor=4.0, spread=0.0, shape=(0.0, 0.0)):
        cf = S.cfill(fill)
        self.L.fo_draw_rounded_rect_fill(self.h, _F4(*rect), C.byref(cf), _F4(*radii_x), _F4(*radii_y), int(mode), factor,
                                         spread, _F2(*shape))

    def draw_image(self, key, pos, colors, size=(0.0, 0.0), flip_y=False):
        self.L.fo_draw_image(self.h, int(key), _F2(*pos), _cols(colors), _F2(*size), int(bool(flip_y)))

    def draw_image_adj(self, key, pos, color, size):
        self.L.fo_draw_image_adj(self.h, int(key), _F2(*pos), S.CColor(*color), _F2(*size))

    def draw_msdf(self, key, pos, color, size, px_range, sd_threshold=0.5, stroke_weight=0.0, mtsdf=False, flip_y=False):
        self.L.fo_draw_msdf(self.h, int(key), _F2(*pos), S.CColor(*color), _F2(*size), px_range, sd_threshold,
                            stroke_weight, int(bool(mtsdf)), int(bool(flip_y)))

    def draw_quadratic_bezier_sdf(self, rect, fill, p0, p1, p2, stroke_weight, cap):
        cf = S.cfill(S.fill_from_json(fill))
        self.L.fo_draw_quadratic_bezier_sdf(self.h, _F4(*rect), C.byref(cf), _F2(*p0), _F2(*p1), _F2(*p2), stroke_weight, int(cap))

    def draw_filled_quad(self, verts, colors):
        self.L.fo_draw_filled_quad(self.h, (C.c_float * 8)(*verts), _cols(colors))

    def draw_rect(self, rect, color):
        self.L.fo_draw_rect(self.h, _F4(*rect), S.CColor(*color))

    def draw_backdrop_blur(self, rect, radii_x, radii_y, blur_radius):
        self.L.fo_draw_backdrop_blur(self.h, _F4(*rect), _F4(*radii_x), _F4(*radii_y), blur_radius)

    def begin_mask(self, rect, radii_x, radii_y):
        self.L.fo_begin_mask(self.h, _F4(*rect), _F4(*radii_x), _F4(*radii_y))

    def end_mask(self):
        self.L.fo_end_mask(self.h)

    def pop_mask(self):
        self.L.fo_pop_mask(self.h)

    def begin_rect_mask(self, rect, radii_x, radii_y):
        self.L.fo_begin_rect_mask(self.h, _F4(*rect), _F4(*radii_x), _F4(*radii_y))

    def pop_rect_mask(self):
        self.L.fo_pop_rect_mask(self.h)

    def put_image(self, key, rgba: np.ndarray):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        out = (C.c_int * 4)()
        rc = self.L.fo_put_image(self.h, int(key), rgba.shape[1], rgba.shape[0], rgba.ctypes.data, out)
        if rc != 0:
            raise RuntimeError("oracle atlas full")
        return tuple(out)

    def put_glyph_outline(self, key, segs: np.ndarray, w: int, h: int, lcd_filter: bool = False):
        segs = np.ascontiguousarray(segs, dtype=np.float32).reshape(-1, 6)
        out = (C.c_int * 4)()
        rc = self.L.fo_put_glyph_outline(self.h, int(key), int(w), int(h), segs.ctypes.data, len(segs), 1 if lcd_filter else 0, out)
        if rc != 0:
            raise RuntimeError("oracle atlas full")
        return tuple(out)

    def put_glyph_image(self, key, rgba: np.ndarray, lcd_filter: bool = False):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        out = (C.c_int * 4)()
        rc = self.L.fo_put_glyph_image(self.h, int(key), rgba.shape[1], rgba.shape[0], rgba.ctypes.data, 1 if lcd_filter else 0, out)
        if rc != 0:
            raise RuntimeError("oracle atlas full")
        return tuple(out)

    def set_text_subpixel(self, enabled: bool, shift: float = 0.0, glyph_variants: bool = False):
        self.L.fo_set_text_subpixel(self.h, int(bool(enabled)), float(shift))
        self.L.fo_set_text_subpixel_glyph_variants(self.h, int(bool(glyph_variants)))

    def put_flippy(self, key, file_bytes: bytes):
        out = (C.c_int * 4)()
        if self.L.fo_put_flippy(self.h, int(key), file_bytes, len(file_bytes), out) != 0:
            raise RuntimeError("bad flippy / atlas full")
        return tuple(out)

    def read_pixels(self, x=0, y=0, w=0, h=0) -> np.ndarray:
        if w <= 0 or h <= 0:
            x, y, w, h = 0, 0, self.W, self.H
        out = np.zeros((h, w, 4), dtype=np.uint8)
        rc = self.L.fo_read_pixels(self.h, x, y, w, h, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("read_pixels out of range")
        return out

    def read_mask(self, level) -> np.ndarray:
        out = np.zeros((self.H, self.W), dtype=np.uint8)
        if self.L.fo_read_mask(self.h, level, out.ctypes.data) != 0:
            raise RuntimeError("no such mask level")
        return out

    # ---- L2
    def render_frame(self, renders: S.Renders, w, h, clear=True, color=(1.0, 1.0, 1.0, 1.0), ui_scale=1.0):
        cs = renders.to_c()
        self.L.fo_set_ui_scale(self.h, ui_scale)
        self.W, self.H = int(w * ui_scale), int(h * ui_scale)
        self.L.fo_render_frame(self.h, cs.byref(), float(w), float(h), int(bool(clear)), _F4(*color))

    def set_text_subpixel_shift(self, shift: float):
        self.L.fo_set_text_subpixel_shift(self.h, float(shift))

    def record_begin(self):
        self.L.fo_record_begin(self.h)

    def record_calls(self):
        return json.loads(self.L.fo_record_json(self.h).decode())

    def replay(self, calls):
        """Replay a recorded call stream (same format ref_swiftshader.replay takes)."""
        for call in calls:
            name, args = call[0], call[1:]
            if name == "begin_frame":
                self.begin_frame(self.W, self.H, *args)
            else:
                getattr(self, name)(*args)


def rounded_radii_vec(rx, ry, hx, hy):
    out = _F4()
    e = C.c_int()
    lib().fo_rounded_radii_vec(_F4(*rx), _F4(*ry), hx, hy, out, C.byref(e))
    return list(out), bool(e.value)


def gradient_colors(fill: S.Fill):
    out = _COL4()
    cf = S.cfill(fill)
    lib().fo_gradient_colors(C.byref(cf), out)
    return [(c.r, c.g, c.b, c.a) for c in out]


def blur_image(rgba: np.ndarray, radius: float) -> np.ndarray:
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    out = np.zeros_like(rgba)
    lib().fo_blur_image(rgba.shape[1], rgba.shape[0], rgba.ctypes.data, out.ctypes.data, radius)
    return out


def texcoord_model(model: int) -> None:
    """TEST-ONLY, process-wide: 1 = sample the atlas on SwiftShader's 16-bit normalised coordinate grid (the goldens' sampler),
    0 = float32 coordinates (the default; what the HIP path is compared with)."""
    lib().fo_debug_texcoord_model.argtypes = [C.c_int]
    lib().fo_debug_texcoord_model.restype = None
    lib().fo_debug_texcoord_model(int(model))


def minify_by2(rgba: np.ndarray) -> np.ndarray:
    """pixie Image.minifyBy2 (the mip step of textures.nim:106-119) on an (h, w, 4) uint8 premultiplied image"""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = rgba.shape[:2]
    out = np.zeros(((h + 1) // 2, (w + 1) // 2, 4), dtype=np.uint8)
    lib().fo_minify_by2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib().fo_minify_by2.restype = None
    lib().fo_minify_by2(rgba.ctypes.data, w, h, out.ctypes.data)
    return out


def lcd_filter(rgba: np.ndarray) -> np.ndarray:
    """applyLcdFilter (common/textrasters/pixie_raster.nim:12-43) on an (h, w, 4) uint8 image"""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    out = np.zeros_like(rgba)
    lib().fo_lcd_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib().fo_lcd_filter(rgba.ctypes.data, out.ctypes.data, rgba.shape[1], rgba.shape[0])
    return out


def rasterize_outline(segs: np.ndarray, w: int, h: int) -> np.ndarray:
    """glyph outline (n x 6: x0, y0, cx, cy, x1, y1; cx = NaN for a line; pixel units, y down) -> (h, w, 4) premultiplied white coverage"""
    segs = np.ascontiguousarray(segs, dtype=np.float32).reshape(-1, 6)
    out = np.zeros((h, w, 4), np.uint8)
    L = lib()
    L.fo_rasterize_outline.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.fo_rasterize_outline.restype = C.c_int
    L.fo_rasterize_outline(segs.ctypes.data, len(segs), w, h, out.ctypes.data)
    return out
