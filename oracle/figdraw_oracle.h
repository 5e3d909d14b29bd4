/* figdraw_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, float32, single-source CPU restatement of figdraw's
 * "node list -> RGBA8 framebuffer" path (reference: elcritch/figdraw v0.35.1):
 *   L2  figrender.nim   node -> backend-call decomposition
 *   L4  opengl/glcontext.nim   backend call -> quad attributes, masks, blur, atlas
 *   L5  opengl/glsl/{atlas,atlas_rect_mask,mask,blur}.frag + fixed-function blend
 * Every function in figdraw_oracle.c cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / the timed CPU baseline.  The product
 * (libfigdraw_hip.so) never links, loads or calls it.
 *
 * Pinning: checked against the reference's own golden
 * PNGs (tests/expected/render_{rgb_boxes_sdf,linear_gradient,layers_clip}.png)
 * and against the reference's GLSL run on SwiftShader (oracle/ref_swiftshader.py).
 */
#ifndef FIGDRAW_ORACLE_H
#define FIGDRAW_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scene model: POD mirror of fignodes.nim:51-92 / figbasics.nim:31-113 / filltypes.nim:11-42 */
typedef struct { uint8_t r, g, b, a; } FoColor;

enum { FO_FILL_COLOR = 0, FO_FILL_LINEAR2 = 1, FO_FILL_LINEAR3 = 2 };          /* FillKind */
enum { FO_AXIS_X = 0, FO_AXIS_Y = 1, FO_AXIS_DIAG_TLBR = 2, FO_AXIS_DIAG_BLTR = 3 }; /* FillGradientAxis */

typedef struct {
  int32_t kind;
  int32_t axis;
  FoColor start; /* flColor: the colour */
  FoColor mid;
  FoColor stop;
  uint8_t mid_pos; /* 0..255 */
  uint8_t _pad[3];
} FoFill; /* 24 B */

enum { FO_SHADOW_NONE = 0, FO_SHADOW_DROP = 1, FO_SHADOW_INNER = 2 };            /* ShadowStyle */
typedef struct { int32_t style; FoFill fill; float blur, spread, x, y; } FoShadow; /* 44 B */
typedef struct { float weight; FoFill fill; int32_t cap, join; } FoStroke;         /* 36 B */

/* FigKind ordinals (figbasics.nim:37-48) */
enum {
  FO_NK_FRAME = 0, FO_NK_TEXT = 1, FO_NK_RECTANGLE = 2, FO_NK_DRAWABLE = 3, FO_NK_SCROLLBAR = 4,
  FO_NK_IMAGE = 5, FO_NK_MSDF_IMAGE = 6, FO_NK_MTSDF_IMAGE = 7, FO_NK_BACKDROP_BLUR = 8, FO_NK_TRANSFORM = 9
};
/* FigFlags ordinals (figbasics.nim:50-58) as bit positions */
enum {
  FO_NF_CLIP_CONTENT = 1 << 0, FO_NF_DISABLE_RENDER = 1 << 1, FO_NF_ROOT_WINDOW = 1 << 2, FO_NF_INACTIVE = 1 << 3,
  FO_NF_SELECT_TEXT = 1 << 4, FO_NF_INVERT_Y = 1 << 5, FO_NF_RECT_MASK_CONTENT = 1 << 6, FO_NF_ELLIPTICAL_CORNERS = 1 << 7
};

typedef struct {
  int32_t kind;
  uint32_t flags;
  int32_t parent; /* FigIdx, -1 = root */
  int32_t child_count;
  int32_t zlevel;
  float box[4]; /* screenBox x,y,w,h (UI units) */
  float rotation; /* degrees */
  FoFill fill;
  uint16_t corners[4];        /* TL,TR,BL,BR (DirectionCorners order) */
  uint16_t corner_radii_y[4]; /* used when NfEllipticalCorners */
  FoShadow shadows[4];        /* nkRectangle */
  FoStroke stroke;            /* nkRectangle */
  int64_t image_id;           /* nkImage / nkMsdfImage / nkMtsdfImage */
  FoFill image_fill;
  float px_range, sd_threshold, stroke_weight; /* msdf */
  float blur;                 /* nkBackdropBlur */
  float translation[2];       /* nkTransform */
  float matrix[16];           /* column-major, vmath Mat4 memory order */
  int32_t use_matrix;
  int32_t glyph_first, glyph_count; /* nkText: range into FoScene.glyphs */
  FoStroke draw_stroke;             /* nkDrawable (fignodes.nim:78-82) */
  uint16_t draw_steps;
  uint16_t _pad0;
  float draw_aa;
  int32_t op_first, op_count;       /* range into FoScene.ops */
  int32_t text_rect_first, text_rect_count; /* nkText: range into FoScene.text_rects */
} FoFig;

/* DrawableOp (fignodes.nim:13-42); v by kind: line a.xy b.xy | circle c.xy r | rectangle x,y,w,h | arc c.xy r start sweep | ellipse c.xy radii.xy */
enum { FO_DK_LINE = 0, FO_DK_CIRCLE = 1, FO_DK_RECTANGLE = 2, FO_DK_BEZIER = 3, FO_DK_ARC = 4, FO_DK_ELLIPSE = 5 };
enum { FO_CAP_AUTO = 0, FO_CAP_ROUND = 1, FO_CAP_BUTT = 2, FO_CAP_SQUARE = 3 };
enum { FO_JOIN_AUTO = 0, FO_JOIN_ROUND = 1, FO_JOIN_BEVEL = 2, FO_JOIN_MITER = 3 };
typedef struct {
  int32_t kind;
  uint16_t steps;
  uint16_t corners[4];
  uint16_t _pad;
  float v[6];
  int32_t ctrl_first, ctrl_count;
} FoDrawOp;

/* A pre-shaped glyph quad: typesetting and glyph rasterisation are third-party
 * (pixie) CPU pre-processing in the reference; the path only consumes
 * (atlas key, local top-left position, vertex colours) -- figrender.nim:456-496. */
typedef struct {
  int64_t image_id;
  float x, y;        /* glyphLocalPos(...) + imageOffset, UI-scaled, local to the text box */
  FoColor colors[4]; /* BL,BR,TR,TL */
  float subpixel_shift;
} FoGlyph;

/* selection (kind 0, node fill) and decoration (kind 1, own fill) rectangles of renderText, figrender.nim:355-452 */
typedef struct { float x, y, w, h; FoFill fill; int32_t kind; } FoTextRect;
#define FO_GLYPH_VARIANT_STEPS 10 /* common/fontglyphs.nim:43 */

typedef struct { int32_t zlevel; int32_t n_nodes; int32_t n_roots; int32_t _pad; const FoFig* nodes; const int32_t* root_ids; } FoLayer;
typedef struct {
  const FoLayer* layers;
  const FoGlyph* glyphs;
  int32_t n_layers;
  int32_t n_glyphs;
  const FoDrawOp* ops;
  const float* controls;
  int32_t n_ops;
  int32_t n_controls;
  const FoTextRect* text_rects;
  int32_t n_text_rects;
  int32_t _pad;
  const int64_t* glyph_variant_ids; /* optional [n_glyphs][FO_GLYPH_VARIANT_STEPS] */
} FoScene;

/* ---- the CPU backend (restates glcontext.nim) */
typedef struct FoCtx FoCtx;

FoCtx* fo_create(int atlas_size, float pixel_scale);
void fo_destroy(FoCtx*);
void fo_set_threads(int n); /* OpenMP threads used by the rasteriser (cpu_baseline leg) */
void fo_set_ui_scale(FoCtx*, float s); /* common/shared.nim:57-98 */

/* BackendContext methods on the path (figbackend.nim:245-705) */
void fo_begin_frame(FoCtx*, int w, int h, int clear, const float rgba[4]);
void fo_end_frame(FoCtx*);
void fo_save_transform(FoCtx*);
void fo_restore_transform(FoCtx*);
void fo_translate(FoCtx*, float x, float y);
void fo_rotate(FoCtx*, float radians);
void fo_scale(FoCtx*, float sx, float sy);
void fo_apply_transform(FoCtx*, const float m16[16]);
void fo_set_aa_factor(FoCtx*, float aa);
void fo_draw_rounded_rect_sdf(FoCtx*, const float rect[4], const FoColor colors[4], const float radii_x[4],
                              const float radii_y[4], int mode, float factor, float spread, const float shape[2],
                              int fill_mode, FoColor mid, FoColor stop, float mid_pos);
void fo_draw_rounded_rect_fill(FoCtx*, const float rect[4], const FoFill* fill, const float radii_x[4],
                               const float radii_y[4], int mode, float factor, float spread, const float shape[2]);
void fo_draw_image(FoCtx*, int64_t key, const float pos[2], const FoColor colors[4], const float size[2], int flip_y);
void fo_draw_image_adj(FoCtx*, int64_t key, const float pos[2], FoColor color, const float size[2]); /* glcontext.nim:1369-1381 */
void fo_draw_msdf(FoCtx*, int64_t key, const float pos[2], FoColor color, const float size[2], float px_range,
                  float sd_threshold, float stroke_weight, int mtsdf, int flip_y);
void fo_draw_quadratic_bezier_sdf(FoCtx*, const float rect[4], const FoFill* fill, const float p0[2], const float p1[2],
                                  const float p2[2], float stroke_weight, int cap);
void fo_draw_filled_quad(FoCtx*, const float verts[8], const FoColor colors[4]);
void fo_draw_rect(FoCtx*, const float rect[4], FoColor color);
void fo_draw_backdrop_blur(FoCtx*, const float rect[4], const float radii_x[4], const float radii_y[4], float blur_radius);
void fo_begin_mask(FoCtx*, const float rect[4], const float radii_x[4], const float radii_y[4]);
void fo_end_mask(FoCtx*);
void fo_pop_mask(FoCtx*);
void fo_begin_rect_mask(FoCtx*, const float rect[4], const float radii_x[4], const float radii_y[4]);
void fo_pop_rect_mask(FoCtx*);
/* pixie Image.minifyBy2 on RGBA8 texels (premultiplied): dst is ((w + 1) / 2) x ((h + 1) / 2).  textures.nim:106-119's mip step;
 * arithmetic pinned by the reference's data/img1.flippy (figdraw_oracle.c header). */
void fo_minify_by2(const uint8_t* src, int w, int h, uint8_t* dst);
int fo_put_image(FoCtx*, int64_t key, int w, int h, const uint8_t* rgba, int out_rect[4]);
/* glyph outlines -> premultiplied white coverage (exact-area accumulation; pixie's texels themselves are unpinned) */
int fo_flatten_outline(const float* segs, int n, float* lines, int cap);
void fo_rasterize_lines(const float* lines, int n, int w, int h, uint8_t* out_rgba);
int fo_rasterize_outline(const float* segs, int n, int w, int h, uint8_t* out_rgba);
int fo_put_glyph_outline(FoCtx*, int64_t key, int w, int h, const float* segs, int n, unsigned flags, int out_rect[4]);
/* applyLcdFilter (common/textrasters/pixie_raster.nim:12-43) and the glyph upload with it (flags bit 0) */
void fo_lcd_filter(const uint8_t* src, uint8_t* dst, int w, int h);
int fo_put_glyph_image(FoCtx*, int64_t key, int w, int h, const uint8_t* rgba, unsigned flags, int out_rect[4]);
int fo_put_flippy(FoCtx*, int64_t key, const uint8_t* file_bytes, size_t n, int out_rect[4]); /* putFlippy glcontext.nim:610-620 */
void fo_set_text_subpixel(FoCtx*, int enabled, float shift);
void fo_set_text_subpixel_glyph_variants(FoCtx*, int enabled);
void fo_set_text_subpixel_shift(FoCtx*, float shift);
/* readPixels: top-down RGBA8, (x,y,w,h) in top-down pixel coordinates; w<=0 -> whole frame */
int fo_read_pixels(FoCtx*, int x, int y, int w, int h, uint8_t* out);
int fo_read_mask(FoCtx*, int level, uint8_t* out_r8);

/* L2: figrender.nim renderFrame / renderRoot / render */
void fo_render_frame(FoCtx*, const FoScene* scene, float frame_w, float frame_h, int clear, const float rgba[4]);

/* Recording mode (the reference's own RecordingBackend idea, tests/ttransform.nim:7-125):
 * when enabled, every backend call is appended to a JSON array text which the
 * tests replay on SwiftShader (ref_swiftshader.replay) or inspect directly. */
void fo_record_begin(FoCtx*);
const char* fo_record_json(FoCtx*); /* valid until the next fo_record_begin / fo_destroy */

/* helpers exposed for known-answer tests */
void fo_rounded_radii_vec(const float rx[4], const float ry[4], float hx, float hy, float out4[4], int* elliptical);
void fo_gradient_colors(const FoFill* fill, FoColor out[4]);
void fo_blur_image(int w, int h, const uint8_t* src, uint8_t* dst, float radius); /* H then V, RGBA8 between */
void fo_debug_texcoord_model(int model); /* TEST-ONLY: 1 = atlas coordinates on SwiftShader's 16-bit normalised grid (process-wide) */
int fo_sizeof_fig(void);
int fo_sizeof_glyph(void);
int fo_sizeof_draw_op(void);
int fo_sizeof_text_rect(void);

#ifdef __cplusplus
}
#endif
#endif
