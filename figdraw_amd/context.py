"""HipContext: the BackendContext-shaped Python binding over libfigdraw_hip.so.

Method names follow the reference's `BackendContext` (src/figdraw/figbackend.nim:245-705) in
snake_case; `render_frame` is `renderFrame` (figrender.nim:1960-1995).  There is no fallback: if
the shared library is missing or no gfx950 GPU is usable, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from . import scene as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfigdraw_hip.so")
_lib = None

_F4 = C.c_float * 4
_F2 = C.c_float * 2
_COL4 = S.CColor * 4


class FigdrawHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class FrameStats(C.Structure):
    _fields_ = [("n_draws", C.c_int32), ("n_phases", C.c_int32), ("n_blurs", C.c_int32), ("n_bins", C.c_int32),
                ("ms_total", C.c_float), ("ms_bin", C.c_float), ("ms_composite", C.c_float),
                ("ms_composite_main", C.c_float), ("ms_blur_h", C.c_float), ("ms_blur_v", C.c_float),
                ("bytes_algorithmic", C.c_int64), ("bytes_composite_main", C.c_int64), ("bytes_blur", C.c_int64),
                ("fragments", C.c_int64), ("ms_host_record", C.c_float), ("ms_host_upload", C.c_float),
                ("ms_host_launch", C.c_float), ("clear_folded", C.c_float),
                ("ms_blur_big_h", C.c_float), ("ms_blur_big_v", C.c_float), ("bytes_blur_big_h", C.c_int64), ("bytes_blur_big_v", C.c_int64),
                ("fragments_main_by_mode", C.c_int64 * 4), ("fragments_main_elliptical", C.c_int64), ("fragments_main_other", C.c_int64),
                ("flops_composite_main", C.c_int64), ("ms_blur_fused", C.c_float), ("deep_bins", C.c_float),
                ("bytes_blur_fused", C.c_int64), ("bytes_frame_implementation", C.c_int64)]


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 the C-ABI library in-tree (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    args = ["make", "-C", csrc, "-s", "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("FIGDRAW_HIP_LIB", LIB_PATH)  # override: instrumented builds (csrc/Makefile `stats`)
    if not os.path.exists(path):
        raise FigdrawHipError(-2, f"{path} is missing: run __graft_entry__.build() (there is no CPU fallback)")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.fdh_last_error.restype = C.c_char_p
    L.fdh_version.restype = C.c_char_p
    L.fdh_create.argtypes = [C.POINTER(vp), C.c_int, C.c_float, C.c_int, C.c_uint32]
    L.fdh_destroy.argtypes = [vp]
    L.fdh_set_stream.argtypes = [vp, vp]
    L.fdh_begin_frame.argtypes = [vp, C.c_int, C.c_int, C.c_int, _F4]
    L.fdh_end_frame.argtypes = [vp]
    L.fdh_save_transform.argtypes = [vp]
    L.fdh_restore_transform.argtypes = [vp]
    L.fdh_translate.argtypes = [vp, C.c_float, C.c_float]
    L.fdh_rotate.argtypes = [vp, C.c_float]
    L.fdh_scale.argtypes = [vp, C.c_float, C.c_float]
    L.fdh_apply_transform.argtypes = [vp, C.c_float * 16]
    L.fdh_transform_mirrors_y.argtypes = [vp, C.POINTER(C.c_int)]
    L.fdh_set_aa_factor.argtypes = [vp, C.c_float]
    L.fdh_get_aa_factor.argtypes = [vp, C.POINTER(C.c_float)]
    L.fdh_get_pixel_scale.argtypes = [vp, C.POINTER(C.c_float)]
    L.fdh_draw_rounded_rect_sdf.argtypes = [vp, _F4, _COL4, _F4, _F4, C.c_int, C.c_float, C.c_float, _F2, C.c_int,
                                            S.CColor, S.CColor, C.c_float]
    L.fdh_draw_rounded_rect_fill.argtypes = [vp, _F4, C.POINTER(S.CFill), _F4, _F4, C.c_int, C.c_float, C.c_float, _F2]
    L.fdh_draw_image.argtypes = [vp, C.c_int64, _F2, _COL4, _F2, C.c_int]
    L.fdh_draw_msdf.argtypes = [vp, C.c_int64, _F2, S.CColor, _F2, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
    L.fdh_draw_image_adj.argtypes = [vp, C.c_int64, _F2, S.CColor, _F2]
    L.fdh_set_text_lcd_filtering.argtypes = [vp, C.c_int]
    L.fdh_get_text_lcd_filtering.argtypes = [vp, C.POINTER(C.c_int)]
    L.fdh_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.fdh_draw_quadratic_bezier_sdf.argtypes = [vp, _F4, C.POINTER(S.CFill), _F2, _F2, _F2, C.c_float, C.c_int]
    L.fdh_draw_filled_quad.argtypes = [vp, C.c_float * 8, _COL4]
    L.fdh_draw_rect.argtypes = [vp, _F4, S.CColor]
    L.fdh_draw_backdrop_blur.argtypes = [vp, _F4, _F4, _F4, C.c_float]
    L.fdh_begin_mask.argtypes = [vp, _F4, _F4, _F4]
    L.fdh_end_mask.argtypes = [vp]
    L.fdh_pop_mask.argtypes = [vp]
    L.fdh_begin_rect_mask.argtypes = [vp, _F4, _F4, _F4]
    L.fdh_pop_rect_mask.argtypes = [vp]
    L.fdh_set_text_subpixel_positioning.argtypes = [vp, C.c_int]
    L.fdh_set_text_subpixel_shift.argtypes = [vp, C.c_float]
    L.fdh_set_text_subpixel_glyph_variants.argtypes = [vp, C.c_int]
    L.fdh_put_image.argtypes = [vp, C.c_int64, C.c_int, C.c_int, vp, C.c_int * 4]
    L.fdh_update_image.argtypes = [vp, C.c_int64, C.c_int, C.c_int, vp]
    L.fdh_put_image_mips.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int * 4]
    L.fdh_put_flippy.argtypes = [vp, C.c_int64, C.c_char_p, C.c_size_t, C.c_int * 4]
    L.fdh_remove_image.argtypes = [vp, C.c_int64]
    L.fdh_has_image.argtypes = [vp, C.c_int64, C.POINTER(C.c_int)]
    L.fdh_reset_atlas.argtypes = [vp, C.c_int]
    L.fdh_atlas_size.argtypes = [vp, C.POINTER(C.c_int)]
    L.fdh_atlas_packed_area.argtypes = [vp, C.POINTER(C.c_int64)]
    L.fdh_put_glyph_outline.argtypes = [vp, C.c_int64, C.c_int, C.c_int, vp, C.c_int, C.c_uint32, C.c_int * 4]
    L.fdh_put_glyph_image.argtypes = [vp, C.c_int64, C.c_int, C.c_int, vp, C.c_uint32, C.c_int * 4]
    L.fdh_read_pixels.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.fdh_debug_read_surface.argtypes = [vp, C.c_int, vp]
    L.fdh_scene_retain.argtypes = [vp, vp, C.c_float, C.c_float, C.c_int, _F4]
    L.fdh_scene_update_nodes.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.fdh_scene_replace_root.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp]
    L.fdh_scene_insert_root.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp]
    L.fdh_scene_render.argtypes = [vp]
    L.fdh_scene_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.fdh_debug_record_digest.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.fdh_debug_verify_upload.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.fdh_debug_bin_digest.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.fdh_debug_staging_store_bytes.argtypes = [C.c_int, C.POINTER(C.c_int64)]
    L.fdh_last_upload_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
    L.fdh_record_begin.argtypes = [vp]
    L.fdh_record_json.argtypes = [vp]
    L.fdh_record_json.restype = C.c_char_p
    L.fdh_frame_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.fdh_sync.argtypes = [vp]
    L.fdh_flush.argtypes = [vp]
    L.fdh_set_ui_scale.argtypes = [vp, C.c_float]
    L.fdh_render_frame.argtypes = [vp, vp, C.c_float, C.c_float, C.c_int, _F4]
    L.fdh_set_stripe.argtypes = [vp, C.c_int, C.c_int]
    L.fdh_set_blur_route.argtypes = [vp, C.c_int]
    L.fdh_set_cull.argtypes = [vp, C.c_int]
    L.fdh_culled_draws.argtypes = [vp, C.POINTER(C.c_int64)]
    L.fdh_set_walk_threads.argtypes = [vp, C.c_int]
    L.fdh_debug_host_times.argtypes = [vp, C.POINTER(C.c_int64)]
    L.fdh_walk_stats.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.fdh_stripe_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.fdh_comm_unique_id.argtypes = [C.c_char_p]
    L.fdh_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.fdh_comm_destroy.argtypes = [vp]
    L.fdh_comm_share.argtypes = [vp, vp]
    L.fdh_gather_stripes.argtypes = [vp, C.c_int, vp]
    L.fdh_gather_frames.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.fdh_replay.argtypes = [vp, C.c_int]
    L.fdh_replay_async.argtypes = [vp, C.c_int]
    L.fdh_replay_timed.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.fdh_profile.argtypes = [vp, C.c_int]
    L.fdh_get_frame_stats.argtypes = [vp, C.POINTER(FrameStats)]
    assert L.fdh_sizeof_fig() == C.sizeof(S.CFig), (L.fdh_sizeof_fig(), C.sizeof(S.CFig))
    assert L.fdh_sizeof_glyph() == C.sizeof(S.CGlyph)
    assert L.fdh_sizeof_draw_op() == C.sizeof(S.CDrawOp)
    assert L.fdh_sizeof_text_rect() == C.sizeof(S.CTextRect)
    _lib = L
    return L


def _cols(colors):
    return _COL4(*[S.CColor(*[int(v) for v in c]) for c in colors])


class HipContext:
    """One GPU, one HIP stream, one RGBA8 surface (newContext, glcontext.nim:255-261)."""

    RECORD_ONLY = 1  # FDH_CREATE_RECORD_ONLY
    SYNC_SUBMIT = 2  # FDH_CREATE_SYNC_SUBMIT

    def __init__(self, atlas_size: int = 1024, pixel_scale: float = 1.0, device: int = 0, record_only: bool = False,
                 sync_submit: bool = False):
        """record_only: a call recorder -- the scene front-end and the atlas packer run, nothing is drawn and no GPU is
        needed (the RecordingBackend of the reference's tests/ttransform.nim); see record_begin / record_calls.
        sync_submit: end_frame prepares, uploads and launches on the calling thread instead of the context's submit thread."""
        self.L = load()
        h = C.c_void_p()
        rc = self.L.fdh_create(C.byref(h), atlas_size, pixel_scale, device,
                               (self.RECORD_ONLY if record_only else 0) | (self.SYNC_SUBMIT if sync_submit else 0))
        if rc != 0:
            raise FigdrawHipError(rc, self.L.fdh_last_error().decode())
        self.h = h
        self.W = self.H = 0
        self.device = device

    def _ck(self, rc):
        if rc != 0:
            raise FigdrawHipError(rc, self.L.fdh_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.fdh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- BackendContext surface
    def begin_frame(self, w, h, clear=True, color=(1.0, 1.0, 1.0, 1.0)):
        self.W, self.H = int(w), int(h)
        self._ck(self.L.fdh_begin_frame(self.h, int(w), int(h), int(bool(clear)), _F4(*color)))

    def end_frame(self):
        self._ck(self.L.fdh_end_frame(self.h))

    def save_transform(self):
        self._ck(self.L.fdh_save_transform(self.h))

    def restore_transform(self):
        self._ck(self.L.fdh_restore_transform(self.h))

    def translate(self, x, y):
        self._ck(self.L.fdh_translate(self.h, x, y))

    def rotate(self, a):
        self._ck(self.L.fdh_rotate(self.h, a))

    def scale(self, sx, sy=None):
        self._ck(self.L.fdh_scale(self.h, sx, sx if sy is None else sy))

    def apply_transform(self, m16):
        self._ck(self.L.fdh_apply_transform(self.h, (C.c_float * 16)(*m16)))

    def transform_mirrors_y(self) -> bool:
        out = C.c_int()
        self._ck(self.L.fdh_transform_mirrors_y(self.h, C.byref(out)))
        return bool(out.value)

    def set_aa_factor(self, aa):
        self._ck(self.L.fdh_set_aa_factor(self.h, aa))

    def sdf_aa_factor(self) -> float:
        out = C.c_float()
        self._ck(self.L.fdh_get_aa_factor(self.h, C.byref(out)))
        return out.value

    def draw_rounded_rect_sdf(self, rect, colors, radii_x, radii_y, mode, factor=4.0, spread=0.0, shape=(0.0, 0.0),
                              fill_mode=0, mid=(0, 0, 0, 0), stop=(0, 0, 0, 0), mid_pos=0.5):
        self._ck(self.L.fdh_draw_rounded_rect_sdf(self.h, _F4(*rect), _cols(colors), _F4(*radii_x), _F4(*radii_y), int(mode),
                                                  factor, spread, _F2(*shape), int(fill_mode), S.CColor(*mid),
                                                  S.CColor(*stop), mid_pos))

    def draw_rounded_rect_fill(self, rect, fill: S.Fill, radii_x, radii_y, mode, factor=4.0, spread=0.0, shape=(0.0, 0.0)):
        cf = S.cfill(fill)
        self._ck(self.L.fdh_draw_rounded_rect_fill(self.h, _F4(*rect), C.byref(cf), _F4(*radii_x), _F4(*radii_y), int(mode),
                                                   factor, spread, _F2(*shape)))

    def draw_image(self, key, pos, colors, size=(0.0, 0.0), flip_y=False):
        self._ck(self.L.fdh_draw_image(self.h, int(key), _F2(*pos), _cols(colors), _F2(*size), int(bool(flip_y))))

    def draw_image_adj(self, key, pos, color, size):
        """drawImageAdj (glcontext.nim:1369-1381): the image with its uv rect pulled in by two texels on every side"""
        self._ck(self.L.fdh_draw_image_adj(self.h, int(key), _F2(*pos), S.CColor(*color), _F2(*size)))

    def set_text_lcd_filtering(self, enabled: bool):
        self._ck(self.L.fdh_set_text_lcd_filtering(self.h, int(bool(enabled))))

    def text_lcd_filtering(self) -> bool:
        out = C.c_int()
        self._ck(self.L.fdh_get_text_lcd_filtering(self.h, C.byref(out)))
        return bool(out.value)

    def comm_info(self):
        """(rank, world) of the context's communicator; (0, 1) without one"""
        r, w = C.c_int(), C.c_int()
        self._ck(self.L.fdh_comm_info(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def draw_msdf(self, key, pos, color, size, px_range, sd_threshold=0.5, stroke_weight=0.0, mtsdf=False, flip_y=False):
        self._ck(self.L.fdh_draw_msdf(self.h, int(key), _F2(*pos), S.CColor(*color), _F2(*size), px_range, sd_threshold,
                                      stroke_weight, int(bool(mtsdf)), int(bool(flip_y))))

    def draw_quadratic_bezier_sdf(self, rect, fill, p0, p1, p2, stroke_weight, cap):
        cf = S.cfill(S.fill_from_json(fill))
        self._ck(self.L.fdh_draw_quadratic_bezier_sdf(self.h, _F4(*rect), C.byref(cf), _F2(*p0), _F2(*p1), _F2(*p2),
                                                      stroke_weight, int(cap)))

    def draw_filled_quad(self, verts, colors):
        self._ck(self.L.fdh_draw_filled_quad(self.h, (C.c_float * 8)(*verts), _cols(colors)))

    def draw_rect(self, rect, color):
        self._ck(self.L.fdh_draw_rect(self.h, _F4(*rect), S.CColor(*color)))

    def draw_backdrop_blur(self, rect, radii_x, radii_y, blur_radius):
        self._ck(self.L.fdh_draw_backdrop_blur(self.h, _F4(*rect), _F4(*radii_x), _F4(*radii_y), blur_radius))

    def begin_mask(self, rect, radii_x, radii_y):
        self._ck(self.L.fdh_begin_mask(self.h, _F4(*rect), _F4(*radii_x), _F4(*radii_y)))

    def end_mask(self):
        self._ck(self.L.fdh_end_mask(self.h))

    def pop_mask(self):
        self._ck(self.L.fdh_pop_mask(self.h))

    def begin_rect_mask(self, rect, radii_x, radii_y):
        self._ck(self.L.fdh_begin_rect_mask(self.h, _F4(*rect), _F4(*radii_x), _F4(*radii_y)))

    def pop_rect_mask(self):
        self._ck(self.L.fdh_pop_rect_mask(self.h))

    def set_text_subpixel(self, enabled: bool, shift: float = 0.0, glyph_variants: bool = False):
        self._ck(self.L.fdh_set_text_subpixel_positioning(self.h, int(bool(enabled))))
        self._ck(self.L.fdh_set_text_subpixel_glyph_variants(self.h, int(bool(glyph_variants))))
        self._ck(self.L.fdh_set_text_subpixel_shift(self.h, shift))

    def set_text_subpixel_shift(self, shift: float):
        self._ck(self.L.fdh_set_text_subpixel_shift(self.h, float(shift)))

    # ---- atlas
    def put_image(self, key, rgba: np.ndarray):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        out = (C.c_int * 4)()
        self._ck(self.L.fdh_put_image(self.h, int(key), rgba.shape[1], rgba.shape[0], rgba.ctypes.data, out))
        return tuple(out)

    def put_glyph_outline(self, key, segs: np.ndarray, w: int, h: int, lcd_filter: bool = False):
        """rasterise a glyph outline (n x 6: x0, y0, cx, cy, x1, y1; cx = NaN for a line) on the device into the atlas"""
        segs = np.ascontiguousarray(segs, dtype=np.float32).reshape(-1, 6)
        out = (C.c_int * 4)()
        self._ck(self.L.fdh_put_glyph_outline(self.h, int(key), int(w), int(h), segs.ctypes.data, len(segs), 1 if lcd_filter else 0, out))
        return tuple(out)

    def put_glyph_image(self, key, rgba: np.ndarray, lcd_filter=False):
        """a rasterised glyph, processed on the device on its way into the atlas (LCD filter, mip chain): pixie_raster.nim:12-95"""
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        out = (C.c_int * 4)()
        flag = 2 if lcd_filter == "context" else (1 if lcd_filter else 0)  # "context": follow set_text_lcd_filtering (FDH_GLYPH_LCD_CONTEXT)
        self._ck(self.L.fdh_put_glyph_image(self.h, int(key), rgba.shape[1], rgba.shape[0], rgba.ctypes.data, flag, out))
        return tuple(out)

    def put_image_mips(self, key, mips):
        """Upload an explicit mip chain (premultiplied RGBA8 arrays, level 0 first), as putFlippy does."""
        mips = [np.ascontiguousarray(m, dtype=np.uint8) for m in mips]
        n = len(mips)
        ws = (C.c_int * n)(*[m.shape[1] for m in mips])
        hs = (C.c_int * n)(*[m.shape[0] for m in mips])
        ptrs = (C.c_void_p * n)(*[m.ctypes.data for m in mips])
        out = (C.c_int * 4)()
        self._ck(self.L.fdh_put_image_mips(self.h, int(key), n, ws, hs, ptrs, out))
        return tuple(out)

    def put_flippy(self, key, file_bytes: bytes):
        out = (C.c_int * 4)()
        self._ck(self.L.fdh_put_flippy(self.h, int(key), file_bytes, len(file_bytes), out))
        return tuple(out)

    def update_image(self, key, rgba: np.ndarray):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        self._ck(self.L.fdh_update_image(self.h, int(key), rgba.shape[1], rgba.shape[0], rgba.ctypes.data))

    def has_image(self, key) -> bool:
        out = C.c_int()
        self._ck(self.L.fdh_has_image(self.h, int(key), C.byref(out)))
        return bool(out.value)

    def remove_image(self, key):
        self._ck(self.L.fdh_remove_image(self.h, int(key)))

    def reset_atlas(self, minimum_size=0):
        self._ck(self.L.fdh_reset_atlas(self.h, int(minimum_size)))

    def atlas_size(self) -> int:
        out = C.c_int()
        self._ck(self.L.fdh_atlas_size(self.h, C.byref(out)))
        return out.value

    # ---- readback
    def read_pixels(self, x=0, y=0, w=0, h=0) -> np.ndarray:
        if w <= 0 or h <= 0:
            x, y, w, h = 0, 0, self.W, self.H
        out = np.zeros((h, w, 4), dtype=np.uint8)
        self._ck(self.L.fdh_read_pixels(self.h, x, y, w, h, out.ctypes.data))
        return out

    def record_begin(self):
        """start recording the backend-level calls this context receives (tests/ttransform.nim's RecordingBackend)"""
        self._ck(self.L.fdh_record_begin(self.h))

    def record_calls(self):
        import json

        return json.loads(self.L.fdh_record_json(self.h).decode())

    def debug_read_surface(self, which: int) -> np.ndarray:
        """0: frame, 1: horizontal blur pass output, 2: blurred snapshot (diagnostic)"""
        out = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        self._ck(self.L.fdh_debug_read_surface(self.h, which, out.ctypes.data))
        return out

    def frame_device_ptr(self):
        p, w, h, pitch = C.c_void_p(), C.c_int(), C.c_int(), C.c_int64()
        self._ck(self.L.fdh_frame_device_ptr(self.h, C.byref(p), C.byref(w), C.byref(h), C.byref(pitch)))
        return p.value, w.value, h.value, pitch.value

    def sync(self):
        self._ck(self.L.fdh_sync(self.h))

    def flush(self):
        """every submitted frame is enqueued on the stream (no wait for the GPU)"""
        self._ck(self.L.fdh_flush(self.h))

    def set_stream(self, stream_ptr: Optional[int]):
        self._ck(self.L.fdh_set_stream(self.h, C.c_void_p(stream_ptr or 0)))

    # ---- whole scenes
    def render_frame(self, renders: S.Renders, w, h, clear=True, color=(1.0, 1.0, 1.0, 1.0), ui_scale=1.0):
        cs = renders.to_c()
        self._ck(self.L.fdh_set_ui_scale(self.h, ui_scale))
        self.W, self.H = int(w * ui_scale), int(h * ui_scale)
        self._ck(self.L.fdh_render_frame(self.h, cs.byref(), float(w), float(h), int(bool(clear)), _F4(*color)))

    # ---- retained scenes (renderfragments.nim:426-544: the tree lives in the context, edits re-decompose only what they touch)
    @staticmethod
    def _marshal_nodes(figs):
        """a node list -> (CScene keeping the arrays alive, pointer to its FdhFig array, count, pointer to the side-array scene)"""
        tmp = S.Renders()
        lst = S.RenderList()
        lst.nodes = list(figs)
        tmp.setLayer(0, lst)
        cs = tmp.to_c()
        return cs, C.cast(cs.struct.layers[0].nodes, C.c_void_p), len(lst.nodes), cs.byref()

    def scene_retain(self, renders: S.Renders, w, h, clear=True, color=(1.0, 1.0, 1.0, 1.0), ui_scale=1.0):
        cs = renders.to_c()
        self._ck(self.L.fdh_set_ui_scale(self.h, ui_scale))
        self.W, self.H = int(w * ui_scale), int(h * ui_scale)
        self._ck(self.L.fdh_scene_retain(self.h, cs.byref(), float(w), float(h), int(bool(clear)), _F4(*color)))

    def scene_update_nodes(self, layer: int, first: int, figs):
        """overwrite nodes [first, first + len(figs)) of layer number `layer` (same tree shape, new properties)"""
        cs, ptr, n, side = self._marshal_nodes(figs)
        self._ck(self.L.fdh_scene_update_nodes(self.h, int(layer), int(first), n, ptr, side))

    def scene_replace_root(self, layer: int, slot: int, figs):
        """replace the subtree under root slot `slot`; figs[0] is the new root (parent -1), later parents are relative to figs;
        an empty list removes the root"""
        cs, ptr, n, side = self._marshal_nodes(figs)
        self._ck(self.L.fdh_scene_replace_root(self.h, int(layer), int(slot), ptr if n else None, n, side))

    def scene_insert_root(self, layer: int, slot: int, figs):
        cs, ptr, n, side = self._marshal_nodes(figs)
        self._ck(self.L.fdh_scene_insert_root(self.h, int(layer), int(slot), ptr if n else None, n, side))

    def scene_render(self):
        self._ck(self.L.fdh_scene_render(self.h))

    def scene_stats(self):
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.L.fdh_scene_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_upload_bytes(self) -> int:
        out = C.c_int64()
        self._ck(self.L.fdh_last_upload_bytes(self.h, C.byref(out)))
        return out.value

    def verify_upload(self):
        """fault hunting: the device's copy of the last frame's records against the host lanes (fdh_debug_verify_upload)"""
        out = (C.c_uint32 * 24)()
        self._ck(self.L.fdh_debug_verify_upload(self.h, out))
        return list(out)

    def bin_digest(self):
        """fault hunting: hash / totals of the bin kernel's output for the last frame (fdh_debug_bin_digest)"""
        out = (C.c_uint64 * 8)()
        self._ck(self.L.fdh_debug_bin_digest(self.h, out))
        return list(out)[:5]

    def record_digest(self) -> int:
        out = C.c_uint64()
        self._ck(self.L.fdh_debug_record_digest(self.h, C.byref(out)))
        return out.value

    def replay_calls(self, calls):
        """Replay a recorded BackendContext call stream (same format as the test harnesses use)."""
        for call in calls:
            name, args = call[0], call[1:]
            if name == "begin_frame":
                self.begin_frame(self.W, self.H, *args)
            else:
                getattr(self, name)(*args)

    # ---- multi-GPU / measurement
    def set_stripe(self, y0: int, y1: int):
        self._ck(self.L.fdh_set_stripe(self.h, int(y0), int(y1)))

    def set_cull(self, mode: int):
        """0: record every draw; 1 (default): drop draws / clipped subtrees no produced pixel lies under, except while the call
        recorder runs; 2: also then.  Same pixels either way."""
        self._ck(self.L.fdh_set_cull(self.h, int(mode)))

    def set_walk_threads(self, n: int):
        """pool threads the scene front-end decomposes large sibling groups on, beside the calling thread (0: serial; < 0: default)"""
        self._ck(self.L.fdh_set_walk_threads(self.h, int(n)))

    def host_times(self):
        """ns the calling thread spent in the last frame: begin_frame, upload wait, walk, end, prepare, publish, drain, groups, pool, merge"""
        out = (C.c_int64 * 12)()
        self._ck(self.L.fdh_debug_host_times(self.h, out))
        names = ["begin_frame", "wait_upload", "walk", "end", "prepare", "publish", "drain", "groups", "pool_run", "merge"]
        return {k: out[i] for i, k in enumerate(names)}

    def walk_stats(self):
        """(threads in force, sibling groups of the last frame that went to the pool)"""
        t, g = C.c_int(), C.c_int64()
        self._ck(self.L.fdh_walk_stats(self.h, C.byref(t), C.byref(g)))
        return t.value, g.value

    def culled_draws(self) -> int:
        out = C.c_int64()
        self._ck(self.L.fdh_culled_draws(self.h, C.byref(out)))
        return out.value

    def set_blur_route(self, route: int):
        """blur routes: 1 the one-kernel routes (k_blur_fx / k_blur_small), 0 two passes per node, -1 the library's default = the one-kernel routes (same pixels either way)"""
        self._ck(self.L.fdh_set_blur_route(self.h, int(route)))

    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId through the library (rank 0 makes it; the host carries the 128 bytes to the other ranks)"""
        L = load()
        buf = C.create_string_buffer(128)
        rc = L.fdh_comm_unique_id(buf)
        if rc != 0:
            raise FigdrawHipError(rc, L.fdh_last_error().decode())
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        self._ck(self.L.fdh_comm_init(self.h, unique_id, int(rank), int(world)))

    def comm_share(self, owner: "HipContext"):
        """use `owner`'s communicator (several contexts of one process share one)"""
        self._ck(self.L.fdh_comm_share(self.h, owner.h))

    def comm_destroy(self):
        self._ck(self.L.fdh_comm_destroy(self.h))

    def gather_stripes(self, dst_rank: int = 0, dst_ptr: Optional[int] = None):
        """row-stripe mode: this rank's rows (fdh_stripe_rows) go to dst_rank's image (device pointer; None: its own surface)"""
        self._ck(self.L.fdh_gather_stripes(self.h, int(dst_rank), C.c_void_p(dst_ptr or 0)))

    def gather_frames(self, dst_rank: int = 0, dst_ptrs=None):
        """frame-parallel mode: every rank's whole frame goes to dst_rank's dst_ptrs[r] (device pointers)"""
        arr = (C.c_void_p * len(dst_ptrs))(*dst_ptrs) if dst_ptrs else None
        self._ck(self.L.fdh_gather_frames(self.h, int(dst_rank), arr))

    def replay(self, times: int = 1):
        self._ck(self.L.fdh_replay(self.h, int(times)))

    def replay_async(self, times: int = 1):
        self._ck(self.L.fdh_replay_async(self.h, int(times)))

    def replay_timed(self, times: int):
        """Per-frame stream times (ms) of `times` back-to-back frames."""
        out = (C.c_float * int(times))()
        self._ck(self.L.fdh_replay_timed(self.h, int(times), out))
        return np.array(out, dtype=np.float32)

    def profile(self, times: int = 1):
        self._ck(self.L.fdh_profile(self.h, int(times)))

    def frame_stats(self) -> FrameStats:
        st = FrameStats()
        self._ck(self.L.fdh_get_frame_stats(self.h, C.byref(st)))
        return st
