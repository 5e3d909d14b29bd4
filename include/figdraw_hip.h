/* figdraw_hip.h -- C ABI of libfigdraw_hip.so: an MI355X (gfx950) offscreen
 * rasteriser for figdraw's per-pixel SDF path ("node list -> RGBA8 framebuffer").
 *
 * The drop-in seam in the reference is the Nim `BackendContext` plug-in object
 * (src/figdraw/figbackend.nim:185-190 type, :245-705 methods; implemented today by
 * `OpenGlContext`, src/figdraw/opengl/glcontext.nim:46).  Every entry point below
 * names the reference method it replaces; INTEGRATION.md shows the Nim shim
 * (`HipContext = ref object of BackendContext`) that binds them with `importc`.
 *
 * Conventions
 *   - plain C, opaque handle, no HIP/torch types in any signature
 *   - every call returns 0 on success or a negative FdhStatus; fdh_last_error()
 *     returns a human-readable message for the calling thread's last failure
 *   - one handle = one GPU + one HIP stream; a handle is single-threaded, exactly
 *     like a BackendContext (all reference call sites are on the render thread:
 *     figrender.nim `{.forbids: [AppMainThreadEff].}`)
 *   - coordinates are the same pre-transform pixel units the reference hands to its
 *     backend (already multiplied by uiScale by the front-end); radii arrays are
 *     in DirectionCorners order TL,TR,BL,BR (figbasics.nim:25-29); colour arrays
 *     are in vertex order BL,BR,TR,TL (figbackend.nim:162)
 *   - there is NO CPU fallback: if no gfx950 device is usable the create call fails
 */
#ifndef FIGDRAW_HIP_H
#define FIGDRAW_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define FDH_API __attribute__((visibility("default")))
#else
#define FDH_API
#endif

typedef struct FdhContext FdhContext;

typedef enum {
  FDH_OK = 0,
  FDH_ERR_INVALID = -1,    /* bad argument / call out of order (the reference asserts: glcontext.nim:1888-1889,1984-1986) */
  FDH_ERR_NO_DEVICE = -2,  /* no usable gfx950 device: the library never falls back to the CPU */
  FDH_ERR_HIP = -3,        /* a HIP runtime call failed */
  FDH_ERR_ATLAS_FULL = -4, /* atlas cannot hold the image even after growing */
  FDH_ERR_UNSUPPORTED = -5 /* the reference's "Backend ... unavailable" ValueError (figbackend.nim:468-705 base methods) */
} FdhStatus;

/* ------------------------------------------------------------------ scene model
 * POD mirror of Fig / RenderList / Renders (src/figdraw/fignodes.nim:44-92,
 * figbasics.nim:31-113, common/filltypes.nim:11-42).  The Nim `Fig` is a variant
 * object and not C-stable, so the shim copies nodes into these. */
typedef struct { uint8_t r, g, b, a; } FdhColor;

enum { FDH_FILL_COLOR = 0, FDH_FILL_LINEAR2 = 1, FDH_FILL_LINEAR3 = 2 };                  /* FillKind filltypes.nim:18-21 */
enum { FDH_AXIS_X = 0, FDH_AXIS_Y = 1, FDH_AXIS_DIAG_TLBR = 2, FDH_AXIS_DIAG_BLTR = 3 }; /* FillGradientAxis :12-16 */

typedef struct {
  int32_t kind;
  int32_t axis;
  FdhColor start; /* flColor: the colour */
  FdhColor mid;
  FdhColor stop;
  uint8_t mid_pos; /* 0..255 (Linear3.midPos) */
  uint8_t _pad[3];
} FdhFill;

enum { FDH_SHADOW_NONE = 0, FDH_SHADOW_DROP = 1, FDH_SHADOW_INNER = 2 }; /* ShadowStyle figbasics.nim:60-64 */
typedef struct { int32_t style; FdhFill fill; float blur, spread, x, y; } FdhShadow; /* RenderShadow :78-84 */
typedef struct { float weight; FdhFill fill; int32_t cap, join; } FdhStroke;         /* RenderStroke :86-90 */

enum { /* FigKind figbasics.nim:37-48 */
  FDH_NK_FRAME = 0, FDH_NK_TEXT = 1, FDH_NK_RECTANGLE = 2, FDH_NK_DRAWABLE = 3, FDH_NK_SCROLLBAR = 4,
  FDH_NK_IMAGE = 5, FDH_NK_MSDF_IMAGE = 6, FDH_NK_MTSDF_IMAGE = 7, FDH_NK_BACKDROP_BLUR = 8, FDH_NK_TRANSFORM = 9
};
enum { /* FigFlags figbasics.nim:50-58, bit = enum ordinal */
  FDH_NF_CLIP_CONTENT = 1 << 0, FDH_NF_DISABLE_RENDER = 1 << 1, FDH_NF_ROOT_WINDOW = 1 << 2, FDH_NF_INACTIVE = 1 << 3,
  FDH_NF_SELECT_TEXT = 1 << 4, FDH_NF_INVERT_Y = 1 << 5, FDH_NF_RECT_MASK_CONTENT = 1 << 6, FDH_NF_ELLIPTICAL_CORNERS = 1 << 7
};
enum { /* SdfMode figbackend.nim:36-52 */
  FDH_SDF_ATLAS = 0, FDH_SDF_CLIP_AA = 3, FDH_SDF_DROP_SHADOW = 7, FDH_SDF_DROP_SHADOW_AA = 8, FDH_SDF_INSET_SHADOW = 9,
  FDH_SDF_INSET_SHADOW_ANNULAR = 10, FDH_SDF_ANNULAR = 11, FDH_SDF_ANNULAR_AA = 12, FDH_SDF_MSDF = 13, FDH_SDF_MTSDF = 14,
  FDH_SDF_MSDF_ANNULAR = 15, FDH_SDF_MTSDF_ANNULAR = 16, FDH_SDF_BACKDROP_BLUR = 17, FDH_SDF_BEZIER_STROKE_AA = 18,
  FDH_SDF_BEZIER_STROKE_BUTT_AA = 19, FDH_SDF_BEZIER_STROKE_SQUARE_AA = 20
};

typedef struct {
  int32_t kind;
  uint32_t flags;
  int32_t parent; /* FigIdx, -1 = root */
  int32_t child_count;
  int32_t zlevel;
  float box[4];   /* screenBox x,y,w,h in UI units */
  float rotation; /* degrees about the box centre */
  FdhFill fill;
  uint16_t corners[4];        /* TL,TR,BL,BR */
  uint16_t corner_radii_y[4]; /* when NfEllipticalCorners */
  FdhShadow shadows[4];       /* nkRectangle (ShadowCount = 4, figbasics.nim:12) */
  FdhStroke stroke;           /* nkRectangle */
  int64_t image_id;           /* nkImage / nkMsdfImage / nkMtsdfImage (ImageId = distinct Hash) */
  FdhFill image_fill;
  float px_range, sd_threshold, stroke_weight; /* MsdfImageStyle figbasics.nim:96-105 */
  float blur;                 /* BackdropBlurStyle */
  float translation[2];       /* TransformStyle */
  float matrix[16];           /* column-major (vmath Mat4 memory order) */
  int32_t use_matrix;
  int32_t glyph_first, glyph_count; /* nkText: range into FdhScene.glyphs */
  FdhStroke draw_stroke;            /* nkDrawable: drawStroke (fignodes.nim:78-82) */
  uint16_t draw_steps;              /*             drawSteps: default step count for curve ops, 0 = adaptive */
  uint16_t _pad0;
  float draw_aa;                    /*             drawAa: SDF AA factor override, <= 0 = keep the backend's */
  int32_t op_first, op_count;       /*             drawOps: range into FdhScene.ops */
  int32_t text_rect_first, text_rect_count; /* nkText: selection / decoration rectangles, range into FdhScene.text_rects */
} FdhFig;

/* DrawableOp (fignodes.nim:13-42).  `v` holds, by kind: line a.xy b.xy | circle center.xy radius | rectangle
 * box x,y,w,h | arc center.xy radius startAngle sweepAngle | ellipse center.xy radii.xy.  Bezier control points
 * are pairs of floats in FdhScene.controls[ctrl_first .. ctrl_first + ctrl_count). */
enum { FDH_DK_LINE = 0, FDH_DK_CIRCLE = 1, FDH_DK_RECTANGLE = 2, FDH_DK_BEZIER = 3, FDH_DK_ARC = 4, FDH_DK_ELLIPSE = 5 };
enum { FDH_CAP_AUTO = 0, FDH_CAP_ROUND = 1, FDH_CAP_BUTT = 2, FDH_CAP_SQUARE = 3 };    /* StrokeCap figbasics.nim:66-70 */
enum { FDH_JOIN_AUTO = 0, FDH_JOIN_ROUND = 1, FDH_JOIN_BEVEL = 2, FDH_JOIN_MITER = 3 }; /* StrokeJoin :72-76 */
typedef struct {
  int32_t kind;
  uint16_t steps;      /* dkBezier.steps / dkArc.arcSteps */
  uint16_t corners[4]; /* dkRectangle.corners TL,TR,BL,BR */
  uint16_t _pad;
  float v[6];
  int32_t ctrl_first, ctrl_count;
} FdhDrawOp;

/* One positioned glyph quad: what figrender.nim:456-496 hands to drawImage after typesetting
 * (typesetting / glyph rasterisation are CPU pre-processing done by the caller). */
typedef struct {
  int64_t image_id;
  float x, y;         /* local top-left, UI-scaled (glyphLocalPos + imageOffset) */
  FdhColor colors[4]; /* BL,BR,TR,TL */
  float subpixel_shift; /* >= 0: the shift to use (setTextSubpixelShift).  < 0: derive it from x the way renderText does when
                         * sub-pixel positioning is on: x snaps to floor(x), the fraction becomes the shift -- or, with glyph
                         * variants enabled, selects FdhScene.glyph_variant_ids[glyph][step] (figrender.nim:464-476) */
} FdhGlyph;

/* The rectangles renderText draws before the glyphs (figrender.nim:435-452, 355-415), in the text node's local
 * UI units; typesetting (selectionRectsFor, glyphRect) is the caller's.  kind 0 = selection rectangle: drawn with the
 * NODE's fill, only when NfSelectText is set and that fill is not fully transparent, width forced to >= 1, skipped when
 * h <= 0.  kind 1 = underline / strikethrough: drawn with `fill`, skipped when w <= 0 or h <= 0. */
enum { FDH_TEXT_RECT_SELECTION = 0, FDH_TEXT_RECT_DECORATION = 1 };
typedef struct { float x, y, w, h; FdhFill fill; int32_t kind; } FdhTextRect;
#define FDH_GLYPH_VARIANT_STEPS 10 /* glyphVariantSubpixelSteps, common/fontglyphs.nim:43 */

typedef struct { int32_t zlevel; int32_t n_nodes; int32_t n_roots; int32_t _pad; const FdhFig* nodes; const int32_t* root_ids; } FdhLayer;
typedef struct {
  const FdhLayer* layers;
  const FdhGlyph* glyphs;
  int32_t n_layers;
  int32_t n_glyphs;
  const FdhDrawOp* ops;  /* nkDrawable ops of all nodes */
  const float* controls; /* x,y pairs of all bezier ops */
  int32_t n_ops;
  int32_t n_controls;    /* number of POINTS */
  const FdhTextRect* text_rects;
  int32_t n_text_rects;
  int32_t _pad;
  const int64_t* glyph_variant_ids; /* optional: [n_glyphs][FDH_GLYPH_VARIANT_STEPS] atlas keys of the sub-pixel variants */
} FdhScene;

/* ------------------------------------------------------------------ lifetime */
/* newContext(atlasSize, ..., pixelScale): glcontext.nim:255-261.  device = HIP ordinal. */
/* fdh_create flags.  FDH_CREATE_RECORD_ONLY: a context that records the calls it receives (fdh_record_begin / fdh_record_json)
 * and runs the scene front-end and the atlas packer, but draws nothing and touches no device: the counterpart of the
 * RecordingBackend in the reference's tests/ttransform.nim.  Every entry point that needs pixels (read_pixels, replay, ...)
 * fails with FDH_ERR_NO_DEVICE on such a context. */
enum { FDH_CREATE_RECORD_ONLY = 1,
       /* fdh_end_frame prepares, uploads and launches the frame on the calling thread.  Default (flag clear): the recorded frame
        * is handed to the context's own submit thread and fdh_end_frame returns at once, so the caller records frame n + 1
        * (the reference's render loop: examples/windy_non_clip_benchmark.nim:113-147) while frame n is being submitted.  Results
        * are identical either way; every entry point that needs the submitted frame waits for the submit thread first. */
       FDH_CREATE_SYNC_SUBMIT = 2 };
FDH_API int fdh_create(FdhContext** out, int atlas_size, float pixel_scale, int device, uint32_t flags);
FDH_API int fdh_destroy(FdhContext*);
FDH_API const char* fdh_last_error(void);
/* Use a caller-owned hipStream_t (passed as void*) instead of the context's own stream; NULL restores it. */
FDH_API int fdh_set_stream(FdhContext*, void* hip_stream);

/* ------------------------------------------------------------------ BackendContext methods on the path */
FDH_API int fdh_begin_frame(FdhContext*, int width, int height, int clear_main, const float clear_rgba[4]); /* beginFrame figbackend.nim:628-631, glcontext.nim:2080-2092 */
FDH_API int fdh_end_frame(FdhContext*);      /* endFrame :633-634, glcontext.nim:1982-1989: submits the frame to the GPU */
FDH_API int fdh_save_transform(FdhContext*);                   /* :654 */
FDH_API int fdh_restore_transform(FdhContext*);                /* :657 */
FDH_API int fdh_translate(FdhContext*, float x, float y);      /* :636 */
FDH_API int fdh_rotate(FdhContext*, float radians);            /* :639 */
FDH_API int fdh_scale(FdhContext*, float sx, float sy);        /* :642-646 */
FDH_API int fdh_apply_transform(FdhContext*, const float m16[16]); /* :648 */
FDH_API int fdh_transform_mirrors_y(FdhContext*, int* out);    /* :659, glcontext.nim:2019-2024 */
FDH_API int fdh_set_aa_factor(FdhContext*, float aa);          /* setSdfAaFactor :277-279 */
FDH_API int fdh_get_aa_factor(FdhContext*, float* out);        /* sdfAaFactor :274 */
FDH_API int fdh_get_pixel_scale(FdhContext*, float* out);      /* pixelScale :271 */

/* drawRoundedRectSdf(rect, colors, radii, mode, factor, spread, shapeSize) figbackend.nim:522-532;
 * fill_mode/mid/stop/mid_pos carry the in-shader 3-stop gradient of glcontext.nim:1591-1607 (0 = vertex colours). */
FDH_API int fdh_draw_rounded_rect_sdf(FdhContext*, const float rect[4], const FdhColor colors[4], const float radii_x[4],
                                      const float radii_y[4], int mode, float factor, float spread, const float shape_size[2],
                                      int fill_mode, FdhColor mid, FdhColor stop, float mid_pos);
/* drawRoundedRectSdf(rect, fill: BackendFill, ...) figbackend.nim:534-552, glcontext.nim:1581-1617 */
FDH_API int fdh_draw_rounded_rect_fill(FdhContext*, const float rect[4], const FdhFill* fill, const float radii_x[4],
                                       const float radii_y[4], int mode, float factor, float spread, const float shape_size[2]);
/* drawImage(path, pos, colors, size, flipY) figbackend.nim:468-475, glcontext.nim:1350-1367 */
FDH_API int fdh_draw_image(FdhContext*, int64_t key, const float pos[2], const FdhColor colors[4], const float size[2], int flip_y);
/* drawImageAdj(path, pos, color, size) figbackend.nim:499-502, glcontext.nim:1369-1381: the image with its uv rect pulled in by
 * two texels on every side */
FDH_API int fdh_draw_image_adj(FdhContext*, int64_t key, const float pos[2], FdhColor color, const float size[2]);
/* drawMsdfImage / drawMtsdfImage figbackend.nim:575-601, glcontext.nim:1097-1155 */
FDH_API int fdh_draw_msdf(FdhContext*, int64_t key, const float pos[2], FdhColor color, const float size[2], float px_range,
                          float sd_threshold, float stroke_weight, int mtsdf, int flip_y);
/* drawBackdropBlur(rect, radii, blurRadius) figbackend.nim:603-606, glcontext.nim:1788-1841 */
FDH_API int fdh_draw_backdrop_blur(FdhContext*, const float rect[4], const float radii_x[4], const float radii_y[4], float blur_radius);
FDH_API int fdh_begin_mask(FdhContext*, const float rect[4], const float radii_x[4], const float radii_y[4]); /* beginMask :608-611, glcontext.nim:1886-1914 */
FDH_API int fdh_end_mask(FdhContext*);                                                                         /* endMask :613, glcontext.nim:1916-1925 */
FDH_API int fdh_pop_mask(FdhContext*);                                                                         /* popMask :616, glcontext.nim:1927-1930 */
FDH_API int fdh_begin_rect_mask(FdhContext*, const float rect[4], const float radii_x[4], const float radii_y[4]); /* :619-623, glcontext.nim:1932-1943 */
FDH_API int fdh_pop_rect_mask(FdhContext*);                                                                     /* :625, glcontext.nim:1945-1949 */
/* drawQuadraticBezierSdf(rect, fill, p0, p1, p2, strokeWeight, cap) figbackend.nim:512-520, glcontext.nim:1619-1741;
 * p0..p2 are relative to the rect centre. */
FDH_API int fdh_draw_quadratic_bezier_sdf(FdhContext*, const float rect[4], const FdhFill* fill, const float p0[2],
                                          const float p1[2], const float p2[2], float stroke_weight, int cap);
/* drawFilledQuad(verts, colors) figbackend.nim:507, glcontext.nim:963-982; verts = x0,y0 .. x3,y3 */
FDH_API int fdh_draw_filled_quad(FdhContext*, const float verts[8], const FdhColor colors[4]);
/* drawRect(rect, color) figbackend.nim:504, glcontext.nim:1410-1426 */
FDH_API int fdh_draw_rect(FdhContext*, const float rect[4], FdhColor color);

/* text flags: figbackend.nim:663-686 */
FDH_API int fdh_set_text_subpixel_positioning(FdhContext*, int enabled);
FDH_API int fdh_set_text_subpixel_shift(FdhContext*, float shift);
FDH_API int fdh_set_text_subpixel_glyph_variants(FdhContext*, int enabled); /* textSubpixelGlyphVariantsEnabled (figbackend.nim) */
/* setTextLcdFilteringEnabled / textLcdFilteringEnabled (figbackend.nim:663-667, glcontext.nim:2059-2063): the flag the renderer
 * reads before it rasterises glyphs (figrender.nim:420); glyph uploads flagged FDH_GLYPH_LCD_CONTEXT follow it. */
FDH_API int fdh_set_text_lcd_filtering(FdhContext*, int enabled);
FDH_API int fdh_get_text_lcd_filtering(FdhContext*, int* out);

/* ------------------------------------------------------------------ atlas (hasImage/putImage/updateImage/removeImage/... figbackend.nim:281-294,400-432) */
/* putImage: skyline packer with margin 4 (glcontext.nim:541-586); out_rect = packed pixel rect x,y,w,h.
 * The shim derives the UV entry as rect / atlasSize (glcontext.nim:584).  The atlas doubles when full (:536-539),
 * which invalidates every earlier entry exactly as in the reference (resetImageAtlas :634-641). */
FDH_API int fdh_put_image(FdhContext*, int64_t key, int width, int height, const uint8_t* rgba8, int out_rect[4]);
/* A freshly rasterised glyph (what renderPixieGlyph hands to loadGlyphImage, common/textrasters/pixie_raster.nim:45-95), processed
 * on the device: with FDH_GLYPH_LCD_FILTER FreeType's default 5-tap LCD filter is applied first -- applyLcdFilter :12-43:
 * weights 8, 77, 86, 77, 8 over x-2 .. x+2, columns clamped to the image, per channel (sum + 128) >> 8 -- then the image and
 * its minifyBy2 chain go into the atlas like fdh_put_image's.  Integer arithmetic, bit-exact with the reference's. */
enum { FDH_GLYPH_LCD_FILTER = 1,
       FDH_GLYPH_LCD_CONTEXT = 2 /* filter iff fdh_set_text_lcd_filtering is on: what renderText's generateGlyph call does with
                                    ctx.textLcdFilteringEnabled() (figrender.nim:420) */ };
FDH_API int fdh_put_glyph_image(FdhContext*, int64_t key, int width, int height, const uint8_t* rgba8, uint32_t flags, int out_rect[4]);
/* A glyph OUTLINE rasterised on the device into the atlas -- generateGlyph's job (common/fontglyphs.nim:61-106; the reference calls
 * pixie's fillText for it, common/textrasters/pixie_raster.nim:83-87).  segs: n x 6 floats {x0, y0, cx, cy, x1, y1} in pixel units of
 * the width x height glyph image, y down: quadratic Bezier segments with control point (cx, cy), or straight lines when cx is
 * NaN; contours closed, non-zero winding.  Sub-pixel variants are the same outline shifted by variant / steps in x
 * (pixie_raster.nim:69-72).  Coverage = exact-area scanline accumulation (curves flattened to <= 0.025 px chord error), stored
 * premultiplied white like pixie's white paint; flags as for fdh_put_glyph_image.  pixie's own texels are third-party and
 * unpinned: the oracle restates the same published algorithm and the two agree bit for bit. */
FDH_API int fdh_put_glyph_outline(FdhContext*, int64_t key, int width, int height, const float* segs, int n_segs, uint32_t flags, int out_rect[4]);
/* putFlippy (glcontext.nim:610-620): `bytes` is a whole .flippy file (common/formatflippy.nim:77-149: "flip", version 1, then per
 * mip "mip!", w, h, zlen, raw-snappy straight RGBA8); every stored level is uploaded as is at (x >> l, y >> l). */
FDH_API int fdh_put_image_mips(FdhContext*, int64_t key, int n_levels, const int* widths, const int* heights,
                               const uint8_t* const* premul_rgba8, int out_rect[4]); /* an already decoded Flippy (flippy.mipmaps) */
FDH_API int fdh_put_flippy(FdhContext*, int64_t key, const uint8_t* bytes, size_t n_bytes, int out_rect[4]);
FDH_API int fdh_update_image(FdhContext*, int64_t key, int width, int height, const uint8_t* rgba8); /* glcontext.nim:591-604 */
FDH_API int fdh_remove_image(FdhContext*, int64_t key);
FDH_API int fdh_has_image(FdhContext*, int64_t key, int* out);
FDH_API int fdh_reset_atlas(FdhContext*, int minimum_size); /* resetImageAtlas / clearImageAtlas */
FDH_API int fdh_atlas_size(FdhContext*, int* out);
FDH_API int fdh_atlas_packed_area(FdhContext*, int64_t* out); /* atlasPackedArea glcontext.nim:2052-2054 */

/* ------------------------------------------------------------------ readback */
/* readPixels(frame, readFront) figbackend.nim:660-661, glcontext.nim:2094-2135: RGBA8, top-down rows.
 * (x, y, w, h) is in top-down pixel coordinates; w <= 0 or h <= 0 reads the whole frame. Blocks until the frame is done. */
FDH_API int fdh_read_pixels(FdhContext*, int x, int y, int w, int h, uint8_t* out_rgba8);
/* Device pointer (hipDeviceptr_t as void*) + pitch of the RGBA8 surface that holds the last submitted frame, for zero-copy
 * consumers (torch.from_blob, a gather of one's own).  Ask again after every frame: a context owns two surfaces of the frame's
 * size and a frame may end in either (a full-frame blur node rendered as one out-of-place kernel flips them). */
FDH_API int fdh_frame_device_ptr(FdhContext*, void** out_ptr, int* out_width, int* out_height, int64_t* out_pitch_bytes);
FDH_API int fdh_sync(FdhContext*);
/* Returns when every frame submitted so far has been ENQUEUED on the context's stream (it does not wait for the GPU): call it
 * before ordering other work after the frame on that stream (fdh_set_stream with a caller-owned stream, a collective). */
FDH_API int fdh_flush(FdhContext*);

/* ------------------------------------------------------------------ whole-scene entry (renderFrame figrender.nim:1960-1995) */
FDH_API int fdh_set_ui_scale(FdhContext*, float s); /* common/shared.nim:69-98 */
FDH_API int fdh_render_frame(FdhContext*, const FdhScene*, float frame_w, float frame_h, int clear_main, const float clear_rgba[4]);

/* ------------------------------------------------------------------ retained scenes (renderfragments.nim:426-544, common/transfer.nim:44-191)
 * The reference lets an application keep a base `Renders` and insert / append / replace fragments of it between frames
 * (insertChildren :426, addChildren :457, insertRoot, updateFragment :523) and ships such updates across threads
 * (transfer.nim); its renderer re-walks the whole tree every frame.  Here the tree lives in the context:
 *   fdh_scene_retain        copy `scene` into the context (nodes, roots, glyph / op / control / text-rect arrays), render it
 *   fdh_scene_update_nodes  overwrite nodes [first, first + count) of a layer -- same tree shape, new properties (the
 *                           animation case; = updateFragment with a fragment of unchanged shape)
 *   fdh_scene_replace_root  replace the whole subtree under root slot `slot` by `subtree` (n nodes, node 0 its root with
 *                           parent -1, later nodes' parents relative to the subtree); n = 0 removes the root
 *   fdh_scene_insert_root   insert a new root subtree in front of root slot `slot` (slot = number of roots: append)
 *   fdh_scene_render        render the retained scene: only roots an update touched (and roots holding a blur node) are
 *                           decomposed again; the draw records of every other root are spliced back from a per-root cache
 * `side` carries the glyph / drawable-op / control-point / text-rect arrays the NEW nodes index (NULL when they use none);
 * the ranges are re-based into the context's own arrays.  Results are identical, record for record, to fdh_render_frame of
 * the edited tree (tests/test_retained_scene.py). */
FDH_API int fdh_scene_retain(FdhContext*, const FdhScene* scene, float frame_w, float frame_h, int clear_main, const float clear_rgba[4]);
FDH_API int fdh_scene_update_nodes(FdhContext*, int layer, int first, int count, const FdhFig* nodes, const FdhScene* side);
FDH_API int fdh_scene_replace_root(FdhContext*, int layer, int slot, const FdhFig* subtree, int n, const FdhScene* side);
FDH_API int fdh_scene_insert_root(FdhContext*, int layer, int slot, const FdhFig* subtree, int n, const FdhScene* side);
FDH_API int fdh_scene_render(FdhContext*);
/* roots decomposed / roots reused from the cache by the last fdh_scene_render */
FDH_API int fdh_scene_stats(FdhContext*, int64_t* roots_walked, int64_t* roots_reused);
/* bytes of the frame block (records, bounds, bin boxes, blur weight tables) the last frame submission sent to the device: the
 * whole block for a new layout, only the 256-byte chunks that differ from what the device already holds otherwise */
FDH_API int fdh_last_upload_bytes(FdhContext*, int64_t* out);
/* Diagnostic: FNV-1a digest of the draw records, bounds, quad extensions and phase table of the last frame (also on
 * FDH_CREATE_RECORD_ONLY contexts): two frames with equal digests hand the kernels identical input. */
FDH_API int fdh_debug_record_digest(FdhContext*, uint64_t* out);
/* Fault hunting: the frame block on the device (what the upload kernel gathered for the frame last submitted) read back and compared
 * with the host-side records it was gathered from.  out[0..2] = differing bytes in records / bin records / quad extensions, out[3] =
 * bytes compared, out[4..9] = first difference (array, byte offset, lane, device dword, host dword, zero dwords in that run),
 * out[10] = pieces, out[11] = records, out[12..16] = the block behind them (chunk boxes, phase table, blur tables: differing bytes,
 * first offset, device dword, host dword, bytes compared), out[17..20] = derived bin boxes (differing, first index, device, host).
 * No counterpart in the reference. */
FDH_API int fdh_debug_verify_upload(FdhContext*, uint32_t out[24]);
/* Fault hunting: what the bin kernel left on the device for the frame last submitted.  out[0] = a hash of every (phase, bin) count and
 * the list entries it covers, out[1] = sum of the counts, out[2] = bins with count 0, out[3] = entries whose first word is 0,
 * out[4] = bins whose count exceeds the list stride.  No counterpart in the reference. */
FDH_API int fdh_debug_bin_digest(FdhContext*, uint64_t out[8]);
/* Bytes of released staging blocks (device memory the host writes through the PCIe BAR) the library holds for `device` -- its store is keyed
 * by device ordinal: a block allocated on one GPU never reaches a context on another.  0 for an ordinal no context has used.  No counterpart
 * in the reference. */
FDH_API int fdh_debug_staging_store_bytes(int device, int64_t* out);

/* ------------------------------------------------------------------ multi-GPU / measurement hooks (no reference counterpart) */
/* Restrict rasterisation to rows [y0, y1) of the frame (row-stripe sharding); blur halos are rendered redundantly so
 * no exchange is needed.  Rows outside the stripe hold nothing a caller may use: a frame with a full-frame blur node flips between
 * the context's two surfaces, so they show whatever an earlier frame left there.  y1 <= y0 disables it.  fdh_gather_stripes sends
 * the rows fdh_stripe_rows gives this rank and refuses a context whose stripe is another one. */
FDH_API int fdh_set_stripe(FdhContext*, int y0, int y1);
/* Culling.  A draw whose pixel bounds reach no pixel the frame will produce -- off the frame, or, under fdh_set_stripe with
 * fdh_render_frame / fdh_scene_render, off the stripe's rows widened by the reach of the scene's blur nodes -- is not recorded, and
 * the scene front-end does not walk the subtree of a node that clips its content (NfClipContent / NfRectMaskContent) to a mask
 * lying outside: the mask is 0 on every produced pixel.  The reference hands such draws to GL, which clips them
 * (examples/windy_non_clip_benchmark.nim submits 180 rows to a window that shows 34); the pixels are the same bit for bit
 * (tests/test_hip_parity.py, tests/test_culling.py).  mode 0: off; 1 (default): on, except while the call recorder runs (recorded
 * streams stay call for call the reference's); 2: on even then.  fdh_culled_draws: draws dropped from the last frame. */
FDH_API int fdh_set_cull(FdhContext*, int mode);
/* The scene front-end (fdh_render_frame) decomposes large sibling groups of the tree -- the roots of a layer, the children of a
 * node: 48 siblings or more, no backdrop-blur node below them -- on `n` threads of a process-wide pool beside the calling thread:
 * every thread records into its own arrays, the upload gathers them in painter's order.  n = 0: the calling thread alone (the
 * reference's model: one render thread, figrender.nim:1960-2002); n < 0 (default): FDH_WALK_THREADS from the environment, or
 * min(7, cores / 8 + 1); the pool's threads keep to the hardware threads that share a last-level cache with the thread that first
 * used the pool (FDH_WALK_AFFINITY=0 lifts that).  The records are the same whatever n is (fdh_debug_record_digest;
 * tests/test_parallel_walk.py).
 * fdh_walk_stats: the thread count in force and how many sibling groups of the last frame went to the pool. */
FDH_API int fdh_set_walk_threads(FdhContext*, int n);
/* Diagnostic: where the calling thread spent the last frame, nanoseconds: [0] begin_frame, [1] of it waiting for the previous use
 * of the frame's record arrays to be uploaded, [2] the calls / the tree walk, [3] end_frame before submission, [4] preparing the
 * submission, [5] of it copying records into the staging mirrors the GPU reads (device memory written through the PCIe BAR, or pinned host memory), [6] waiting for the submit thread, [7] sibling groups on the walk
 * pool (part of [2]), [8] of it inside the pool, [9] of it merging; [10], [11] reserved. */
FDH_API int fdh_debug_host_times(FdhContext*, int64_t out_ns[12]);
FDH_API int fdh_walk_stats(FdhContext*, int* threads, int64_t* parallel_groups);
FDH_API int fdh_culled_draws(FdhContext*, int64_t* out);
/* ---- the gather over RCCL / xGMI (one process per GPU).  Nothing is exchanged while a frame renders; the one
 * collective of the path is the gather of the finished RGBA8 rows or frames to one rank: grouped ncclSend / ncclRecv on the
 * context's stream, queued behind the frame's kernels (no host synchronisation; fdh_sync on the destination waits for it).
 * librccl is loaded on first use (dlopen; FDH_RCCL_LIB overrides the name), never linked.
 *   fdh_stripe_rows      the partition rule: rows [y0, y1) of `rank` when `height` rows are cut into `world` contiguous stripes
 *                        of whole 8-row strips -- what each rank passes to fdh_set_stripe
 *   fdh_comm_unique_id   ncclGetUniqueId: rank 0 makes the 128-byte id, the host carries it to the other ranks (its own channel)
 *   fdh_comm_init        ncclCommInitRank on the context's device; fdh_comm_destroy (also done by fdh_destroy)
 *   fdh_comm_share       a second context of the same process (frames in flight) shares `owner`'s communicator instead of
 *                        creating one; the communicator goes when the last context holding it is destroyed (any order).  A lock
 *                        in the shared object keeps two contexts' gathers from interleaving; EVERY RANK MUST ISSUE THE GATHERS OF A
 *                        SHARED COMMUNICATOR IN THE SAME ORDER (context by context, frame by frame), or the send / recv pairs
 *                        of different ranks mismatch and the collective hangs
 *   fdh_gather_stripes   row-stripe mode: rank r sends rows fdh_stripe_rows(H, world, r) of its surface to dst_rank, which
 *                        receives them into the same rows of dst_image (device, W x H RGBA8; NULL: its own surface, whose own
 *                        rows are already in place)
 *   fdh_gather_frames    frame-parallel mode: every rank sends its whole frame, dst_rank receives rank r's into dst_images[r]
 * Without fdh_comm_init a context is rank 0 of 1 and the gathers reduce to the destination's device copy. */
#define FDH_COMM_ID_BYTES 128
FDH_API int fdh_stripe_rows(int height, int world, int rank, int* y0, int* y1);
FDH_API int fdh_comm_unique_id(uint8_t out[FDH_COMM_ID_BYTES]);
FDH_API int fdh_comm_init(FdhContext*, const uint8_t id[FDH_COMM_ID_BYTES], int rank, int world);
FDH_API int fdh_comm_share(FdhContext*, FdhContext* owner);
/* this context's rank and the communicator's size as fdh_comm_init / fdh_comm_share left them (0 of 1 without a communicator) */
FDH_API int fdh_comm_info(FdhContext*, int* rank, int* world);
FDH_API int fdh_comm_destroy(FdhContext*);
FDH_API int fdh_gather_stripes(FdhContext*, int dst_rank, void* dst_image);
FDH_API int fdh_gather_frames(FdhContext*, int dst_rank, void* const* dst_images);
/* Which routes the blur nodes of a frame take: 1 = the one-kernel routes (a node covering the whole frame: both passes as one
 * out-of-place kernel, k_blur_fx, half the bytes; a small node: k_blur_small), 0 = the horizontal and the vertical pass as two
 * kernels each, -1 (default) = the library's choice: the one-kernel routes (until round 4 it chose per frame, two-pass routes beside
 * other contexts' frames; they no longer measure faster there).  The pixels are the same bit for bit; FDH_BLUR_FUSED=0|1 in the
 * environment sets the default. */
FDH_API int fdh_set_blur_route(FdhContext*, int route);
/* Re-run the GPU work of the last submitted frame `times` times from the draw records already resident in HBM
 * (the host-side decomposition and the upload are not repeated). */
FDH_API int fdh_replay(FdhContext*, int times);
/* enqueue `times` frames and return without waiting (fdh_sync waits): lets several contexts overlap on one GPU */
FDH_API int fdh_replay_async(FdhContext*, int times);
/* the same with a hipEvent between consecutive frames: ms_out[i] = stream time of frame i (min / p50 / p95 reporting) */
FDH_API int fdh_replay_timed(FdhContext*, int times, float* ms_out);
typedef struct {
  int32_t n_draws, n_phases, n_blurs, n_bins;
  float ms_total;          /* fdh_replay: hipEvent time per frame over the whole batch (no per-kernel events in it) */
  float ms_bin;            /* fdh_profile: binning kernel, per frame */
  float ms_composite;      /* fdh_profile: all composite launches of a frame */
  float ms_composite_main; /* fdh_profile: the phase-0 composite launch (the dominant kernel) */
  float ms_blur_h;         /* fdh_profile: all horizontal blur launches of a frame */
  float ms_blur_v;         /* fdh_profile: all vertical blur launches of a frame */
  int64_t bytes_algorithmic; /* SURVEY.md 8(d) B_frame for the last frame */
  int64_t bytes_composite_main; /* algorithmic bytes of the phase-0 composite launch: surface store (+load) + records */
  int64_t bytes_blur;        /* algorithmic bytes of all blur launches: H read + H write + V read + V write */
  int64_t fragments;         /* sum of covered fragments over all draws (GL-equivalent work unit) */
  /* host-side cost of the last begin_frame .. end_frame: recording the calls (tree walk for fdh_render_frame), building
   * the upload, and issuing the copies + kernel launches; all asynchronous to the GPU */
  float ms_host_record, ms_host_upload, ms_host_launch;
  float clear_folded;      /* 1: the frame's first draw -- one colour at full coverage over the whole cleared frame -- was folded into the clear colour */
  /* the frame's LARGEST blur node on its own (the bench frame: the full-frame node, i.e. the HBM-bound launches):
   * fdh_profile times of its horizontal and vertical pass, and the algorithmic bytes each must move -- H: region + halo rows
   * read and written; V: the same rows read, the region written, plus the region read where the fused composite has to
   * blend (not when the surface is known opaque: the pass then only stores) */
  float ms_blur_big_h, ms_blur_big_v;
  int64_t bytes_blur_big_h, bytes_blur_big_v;
  /* the phase-0 composite launch in SURVEY.md 8(d)'s work units: covered fragments (sum of quad areas) by SdfMode 3 (ClipAA) / 7
   * (DropShadow) / 9 (InsetShadow) / 12 (AnnularAA), those of draws with elliptical corners, all other modes; and the algorithmic
   * flops they amount to: 25 / 36 / 71 / 28 per fragment (other modes priced as ClipAA), + 30 elliptical, + 16 blend + re-quantise */
  int64_t fragments_main_by_mode[4], fragments_main_elliptical, fragments_main_other;
  int64_t flops_composite_main;
  /* a full-frame blur node whose two passes ran as ONE out-of-place kernel (k_blur_fx): fdh_profile time, and the bytes THAT
   * kernel must move (the region read once + written once; bytes_blur / bytes_algorithmic keep SURVEY.md 8(d)'s two-pass
   * formula so that frames stay comparable); bytes_frame_implementation: bytes_algorithmic with such nodes priced at what
   * this implementation moves */
  /* deep_bins: how many of the frame's longest-listed bins the LAST full-frame compositor launch shaded with four waves per
   * strip (k_composite_tiles' quarter strips, round 6; 0: none -- no list reached the threshold, or the order is not known yet) */
  float ms_blur_fused, deep_bins;
  int64_t bytes_blur_fused, bytes_frame_implementation;
} FdhFrameStats;
/* Run `times` more frames and fill the per-kernel averages.  Each launch is stamped with its own start / end events
 * (hipExtLaunchKernelGGL): kernel execution time as rocprofv3 --kernel-trace reports it, no launch gaps in it. */
FDH_API int fdh_profile(FdhContext*, int times);
FDH_API int fdh_get_frame_stats(FdhContext*, FdhFrameStats* out);
FDH_API int fdh_sizeof_fig(void);
FDH_API int fdh_sizeof_glyph(void);
FDH_API int fdh_sizeof_draw_op(void);
FDH_API int fdh_sizeof_text_rect(void);
/* Call recorder -- the library-side counterpart of the RecordingBackend the reference's front-end tests use
 * (tests/ttransform.nim:19-144, tests/trenderfragments.nim): between fdh_record_begin and fdh_record_json every
 * BackendContext-level call this context receives -- from the caller or from its own scene front-end (fdh_render_frame) -- is
 * appended to a JSON array of ["name", args...] entries (same names and argument order as the entry points above; colours as
 * [r,g,b,a]).  fdh_record_json ends the recording and returns the text (valid until the next fdh_record_begin). */
FDH_API int fdh_record_begin(FdhContext*);
FDH_API const char* fdh_record_json(FdhContext*);

/* Diagnostic: copy one of the context's working surfaces to the host (W x H RGBA8, tightly packed) after waiting for its
 * stream.  which = 0: the frame (= fdh_read_pixels), 1: the horizontal blur pass's output (the reference's intermediate blur
 * texture, glcontext.nim:1743-1786) as the last blur node left it, 2: the blurred snapshot of the last unfused blur node.
 * tools/race_contexts.py uses it to tell which pass a wrong pixel came from. */
FDH_API int fdh_debug_read_surface(FdhContext*, int which, uint8_t* out_rgba8);
/* Diagnostic, host-only (no device, no context): the pixel rectangle [x0,x1) x [y0,y1) the submission path marks as the
 * draw's saturated core (coverage exactly 1, or a provable no-op for strokes / inner shadows) under the identity
 * transform; all zeros when there is none.  Arguments as fdh_draw_rounded_rect_sdf. */
FDH_API int fdh_saturated_core(const float rect[4], const float radii_x[4], const float radii_y[4], int mode, float factor,
                               float spread, const float shape[2], float aa, int out_px[4]);
/* Diagnostic, host-only: the tap table of a backdrop blur of `blur_radius` (the merged FIR of blur.frag:11-32, glcontext.nim
 * :1743-1786: dense weights over offsets -reach..+reach into `dense`, capacity >= 133) and the weight fragments the
 * matrix-pipe pass (vertical != 0: vertical pass) multiplies with: k-steps x 2 x 64 lanes x 8 binary16 values into
 * `frag_bits` (capacity >= 11 * 2 * 64 * 8) -- the first of the two halves holds the taps at scale 2^10, one binary16 each, rounded
 * from the centre tap outwards with the rounding error carried to the next tap; the second half (rounds 2 - 4: the low part of a
 * 22-bit weight) is zero.  Returns the tap reach in *reach and the number of k-steps in *k_steps. */
FDH_API int fdh_blur_weight_fragments(float blur_radius, int vertical, float* dense, uint16_t* frag_bits, int* reach, int* k_steps);
FDH_API const char* fdh_version(void);

#ifdef __cplusplus
}
#endif
#endif
