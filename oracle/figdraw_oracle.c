/* figdraw_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see figdraw_oracle.h).
 *
 * CPU restatement (plain C, float32, IEEE, no FMA contraction) of figdraw's
 * node-list -> RGBA8 path.  Citations are relative to /root/reference/.
 *
 * Structure mirrors the GL implementation on purpose (one full-frame pass per
 * draw, real R8 mask planes per nesting level, real full-frame blur textures,
 * RGBA8 store after every draw) -- it is NOT structured like the HIP product
 * (tile-binned, masks analytic, blur restricted to footprints), which is what
 * makes HIP-vs-oracle parity meaningful.
 *
 * Third-party arithmetic that is not under /root/reference and is restated from its published definition.  Both items
 * below turned out to be pinned by data the reference ships; what remains "parity unpinned" is said where it is used (the
 * straight -> premultiplied rounding of translucent Flippy texels, pixie's glyph rasteriser texels):
 *   - vmath (any version, figdraw.nimble:20): Mat4 column-major, translate/scale/ortho/inverse; rotateZ(a) maps
 *     (x, y) to (cos x + sin y, -sin x + cos y) -- this one IS pinned, by tests/expected/render_line_rect.png.
 *   - pixie >= 5.0.1 Image.minifyBy2 used for atlas mip levels (opengl/textures.nim:106-119).  pixie is not in
 *     /root/reference; its arithmetic is PINNED by the reference's own data/img1.flippy, whose eight stored levels
 *     (100, 50, 25, 13, 7, 4, 2, 1 px) are pngToFlippy's minifyBy2 chain of the opaque level 0 (formatflippy.nim:101-112):
 *     2x2 box sum DIV 4 (no rounding); an odd width or height rounds the result size UP, and the extra column / row is
 *     mix(a, b, 0.5) * 0.5 of the last source column / row -- mix = (a*127 + b*128) div 255, * 0.5 = (v*128) div 255 --
 *     the extra corner the last source texel * 0.25 = (v*64) div 255 (premultiplied: the edge's coverage is halved).
 *     tests/test_oracle.py::test_minify_by2_reproduces_the_flippy_levels.
 */
#include "figdraw_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ small helpers */
typedef struct { float x, y; } v2;
typedef struct { float x, y, z, w; } v4;

static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline float minf(float a, float b) { return a < b ? a : b; }
static inline float maxf(float a, float b) { return a > b ? a : b; }
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; } /* GLSL mix */
/* Nim math.round: half away from zero */
static inline float nim_round(float x) { return x >= 0.0f ? floorf(x + 0.5f) : -floorf(-x + 0.5f); }
/* GL unorm8 store: round to nearest */
static inline uint8_t to_unorm8(float x) { return (uint8_t)floorf(clampf(x, 0.0f, 1.0f) * 255.0f + 0.5f); }
static inline float from_unorm8(uint8_t v) { return (float)v / 255.0f; }

/* 2D affine part of the vmath Mat4 transform stack: [a c tx; b d ty] acting on column vectors */
typedef struct { float a, b, c, d, tx, ty; } Aff;
static Aff aff_identity(void) { Aff m = {1, 0, 0, 1, 0, 0}; return m; }
static Aff aff_mul(Aff m, Aff n) { /* m * n */
  Aff r;
  r.a = m.a * n.a + m.c * n.b;
  r.b = m.b * n.a + m.d * n.b;
  r.c = m.a * n.c + m.c * n.d;
  r.d = m.b * n.c + m.d * n.d;
  r.tx = m.a * n.tx + m.c * n.ty + m.tx;
  r.ty = m.b * n.tx + m.d * n.ty + m.ty;
  return r;
}
static v2 aff_apply(Aff m, float x, float y) { v2 r = {m.a * x + m.c * y + m.tx, m.b * x + m.d * y + m.ty}; return r; }
static Aff aff_inverse(Aff m) {
  float det = m.a * m.d - m.b * m.c;
  float id = 1.0f / det;
  Aff r;
  r.a = m.d * id;
  r.b = -m.b * id;
  r.c = -m.c * id;
  r.d = m.a * id;
  r.tx = -(r.a * m.tx + r.c * m.ty);
  r.ty = -(r.b * m.tx + r.d * m.ty);
  return r;
}

/* ------------------------------------------------------------------ L5: glsl/atlas.frag restated */
/* atlas.frag:51-69 */
static float sd_rounded_box(float px, float py, float bx, float by, v4 r) {
  float rr;
  if (px > 0.0f) rr = (py > 0.0f) ? r.x : r.y;
  else rr = (py > 0.0f) ? r.z : r.w;
  float qx = fabsf(px) - bx + rr, qy = fabsf(py) - by + rr;
  float mx = maxf(qx, 0.0f), my = maxf(qy, 0.0f);
  return minf(maxf(qx, qy), 0.0f) + sqrtf(mx * mx + my * my) - rr;
}
/* atlas.frag:71-79 */
static float sd_ellipse(float px, float py, float rx, float ry) {
  float sx = maxf(rx, 0.000001f), sy = maxf(ry, 0.000001f);
  float ax = px / sx, ay = py / sy;
  float k0 = sqrtf(ax * ax + ay * ay);
  if (k0 <= 0.000001f) return -minf(sx, sy);
  float bx = px / (sx * sx), by = py / (sy * sy);
  float k1 = sqrtf(bx * bx + by * by);
  return k0 * (k0 - 1.0f) / maxf(k1, 0.000001f);
}
/* atlas.frag:81-86 */
static float select_corner_radius(v4 r, float px, float py) {
  if (px > 0.0f) return (py > 0.0f) ? r.x : r.y;
  return (py > 0.0f) ? r.z : r.w;
}
/* atlas.frag:88-115 */
static float sd_elliptical_rounded_box(float px, float py, float bx, float by, v4 packed) {
  float sel = select_corner_radius(packed, px, py);
  if (sel < 0.0f) {
    float r = -sel - 1.0f;
    v4 rr = {r, r, r, r};
    return sd_rounded_box(px, py, bx, by, rr);
  }
  float pv = floorf(sel + 0.5f);
  float rx = (pv - 4096.0f * floorf(pv / 4096.0f)) * bx / 4095.0f; /* mod(pv, 4096) */
  float ry = floorf(pv / 4096.0f) * by / 4095.0f;
  if (rx <= 0.0f || ry <= 0.0f) {
    float qx = fabsf(px) - bx, qy = fabsf(py) - by;
    float mx = maxf(qx, 0.0f), my = maxf(qy, 0.0f);
    return minf(maxf(qx, qy), 0.0f) + sqrtf(mx * mx + my * my);
  }
  if (rx == ry) {
    v4 rr = {rx, rx, rx, rx};
    return sd_rounded_box(px, py, bx, by, rr);
  }
  float qx = fabsf(px) - bx + rx, qy = fabsf(py) - by + ry;
  if (qx > 0.0f && qy > 0.0f) return sd_ellipse(qx, qy, rx, ry);
  return maxf(qx - rx, qy - ry);
}
/* atlas.frag:211-216 */
static float shadow_profile(float sd, float blur_radius) {
  float sigma = maxf(0.5f * blur_radius, 0.5f);
  float z = sd / sigma;
  return expf(-0.5f * z * z);
}
/* atlas.frag:218-231 */
static float linear3_t(int fill_mode, float u, float v) {
  switch (fill_mode) {
    case 1: return u;
    case 2: return v;
    case 3: return 0.5f * (u + v);
    case 4: return 0.5f * (u + (1.0f - v));
    default: return 0.0f;
  }
}
/* atlas.frag:121-160 */
static float sd_bezier(float px, float py, v2 A, v2 B, v2 C) {
  float ax = B.x - A.x, ay = B.y - A.y;
  float bx = A.x - 2.0f * B.x + C.x, by = A.y - 2.0f * B.y + C.y;
  float bb = bx * bx + by * by;
  if (bb <= 0.000001f) {
    float bax = C.x - A.x, bay = C.y - A.y;
    float h = clampf(((px - A.x) * bax + (py - A.y) * bay) / maxf(bax * bax + bay * bay, 0.000001f), 0.0f, 1.0f);
    float dx = px - (A.x + bax * h), dy = py - (A.y + bay * h);
    return sqrtf(dx * dx + dy * dy);
  }
  float cx = ax * 2.0f, cy = ay * 2.0f;
  float dx = A.x - px, dy = A.y - py;
  float kk = 1.0f / bb;
  float kx = kk * (ax * bx + ay * by);
  float ky = kk * (2.0f * (ax * ax + ay * ay) + (dx * bx + dy * by)) / 3.0f;
  float kz = kk * (dx * ax + dy * ay);
  float p = ky - kx * kx;
  float p3 = p * p * p;
  float q = kx * (2.0f * kx * kx - 3.0f * ky) + kz;
  float h = q * q + 4.0f * p3;
  float res;
  if (h >= 0.0f) {
    h = sqrtf(h);
    float x0 = (h - q) / 2.0f, x1 = (-h - q) / 2.0f;
    float r0 = (x0 > 0.0f ? 1.0f : (x0 < 0.0f ? -1.0f : 0.0f)) * powf(fabsf(x0), 1.0f / 3.0f);
    float r1 = (x1 > 0.0f ? 1.0f : (x1 < 0.0f ? -1.0f : 0.0f)) * powf(fabsf(x1), 1.0f / 3.0f);
    float t = clampf(r0 + r1 - kx, 0.0f, 1.0f);
    float ex = dx + (cx + bx * t) * t, ey = dy + (cy + by * t) * t;
    res = ex * ex + ey * ey;
  } else {
    float z = sqrtf(-p);
    float v = acosf(clampf(q / (p * z * 2.0f), -1.0f, 1.0f)) / 3.0f;
    float m = cosf(v);
    float n = sinf(v) * 1.732050808f;
    float t1 = clampf((m + m) * z - kx, 0.0f, 1.0f);
    float t2 = clampf((-n - m) * z - kx, 0.0f, 1.0f);
    float e1x = dx + (cx + bx * t1) * t1, e1y = dy + (cy + by * t1) * t1;
    float e2x = dx + (cx + bx * t2) * t2, e2y = dy + (cy + by * t2) * t2;
    res = minf(e1x * e1x + e1y * e1y, e2x * e2x + e2y * e2y);
  }
  return sqrtf(res);
}
static v2 safe_normalize(float x, float y, v2 fallback) { /* atlas.frag:174-177 */
  float len = sqrtf(x * x + y * y);
  v2 r;
  if (len <= 0.000001f) return fallback;
  r.x = x / len; r.y = y / len;
  return r;
}
/* atlas.frag:179-209 */
static float bezier_stroke_sd(float dist, float px, float py, v2 A, v2 B, v2 C, float half_w, int mode) {
  if (mode == 18) return dist - half_w;
  v2 one = {1.0f, 0.0f};
  v2 fallback = safe_normalize(C.x - A.x, C.y - A.y, one);
  v2 st = safe_normalize(B.x - A.x, B.y - A.y, fallback);
  v2 et = safe_normalize(C.x - B.x, C.y - B.y, fallback);
  float start_proj = (px - A.x) * st.x + (py - A.y) * st.y;
  float end_proj = (px - C.x) * et.x + (py - C.y) * et.y;
  float trim = mode == 20 ? half_w : 0.0f;
  float tube = dist;
  if (mode == 20) {
    if (start_proj < 0.0f) tube = minf(tube, fabsf((px - A.x) * st.y - (py - A.y) * st.x));
    if (end_proj > 0.0f) tube = minf(tube, fabsf((px - C.x) * et.y - (py - C.y) * et.x));
  }
  float cap = maxf(-start_proj - trim, end_proj - trim);
  return maxf(tube - half_w, cap);
}
static float median3(float a, float b, float c) { return maxf(minf(a, b), minf(maxf(a, b), c)); } /* atlas.frag:41-43 */

/* ------------------------------------------------------------------ L4 state */
#define FO_MAX_MASKS 96 /* mask planes are allocated per level on demand; the reference has no limit (glcontext.nim:1886-1914) */
#define FO_MAX_MATS 256
#define FO_MAX_MIPS 14

typedef struct { int64_t key; float x, y, w, h; /* UV units */ int used; } AtlasEntry;

typedef struct {
  int kind; /* 1 = fast (analytic), 2 = mask texture */
  v4 params, radii, matx, maty;
} RectMask;

typedef struct {
  char* buf;
  size_t len, cap;
  int on, first;
} Recorder;

struct FoCtx {
  int W, H;
  uint8_t* fb;                    /* RGBA8 top-down */
  uint8_t* mask[FO_MAX_MASKS];    /* R8 planes, index 0 unused ("white", never sampled) */
  uint8_t *backdrop, *backdrop_tmp;
  int mask_write, mask_begun, frame_begun;
  RectMask rect_masks[FO_MAX_MASKS];
  int n_rect_masks;
  Aff mat, mats[FO_MAX_MATS];
  int n_mats;
  float aa, pixel_scale, ui_scale;
  int subpixel_enabled;
  int subpixel_variants;
  float subpixel_shift;
  /* atlas */
  int atlas_size, atlas_margin, n_mips;
  uint8_t* atlas[FO_MAX_MIPS];
  uint16_t* heights;
  AtlasEntry* entries;
  int n_entries, cap_entries;
  Recorder rec;
};

static int g_threads = 1;
void fo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
void fo_set_ui_scale(FoCtx* c, float s) { c->ui_scale = s; }
int fo_sizeof_fig(void) { return (int)sizeof(FoFig); }
int fo_sizeof_glyph(void) { return (int)sizeof(FoGlyph); }
int fo_sizeof_draw_op(void) { return (int)sizeof(FoDrawOp); }
int fo_sizeof_text_rect(void) { return (int)sizeof(FoTextRect); }

/* ------------------------------------------------------------------ recorder */
static void rec_printf(FoCtx* c, const char* fmt, ...) {
  Recorder* r = &c->rec;
  if (!r->on) return;
  for (;;) {
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(r->buf + r->len, r->cap - r->len, fmt, ap);
    va_end(ap);
    if (n >= 0 && (size_t)n < r->cap - r->len) { r->len += (size_t)n; return; }
    r->cap = r->cap ? r->cap * 2 : 4096;
    r->buf = (char*)realloc(r->buf, r->cap);
  }
}
static void rec_open(FoCtx* c, const char* name) {
  if (!c->rec.on) return;
  rec_printf(c, "%s[\"%s\"", c->rec.first ? "" : ",\n", name);
  c->rec.first = 0;
}
static void rec_f(FoCtx* c, double v) { rec_printf(c, ",%.9g", v); }
static void rec_i(FoCtx* c, long long v) { rec_printf(c, ",%lld", v); }
static void rec_fv(FoCtx* c, const float* v, int n) {
  rec_printf(c, ",[");
  for (int i = 0; i < n; i++) rec_printf(c, "%s%.9g", i ? "," : "", (double)v[i]);
  rec_printf(c, "]");
}
static void rec_col(FoCtx* c, FoColor k) { rec_printf(c, ",[%d,%d,%d,%d]", k.r, k.g, k.b, k.a); }
static void rec_cols(FoCtx* c, const FoColor k[4]) {
  rec_printf(c, ",[");
  for (int i = 0; i < 4; i++) rec_printf(c, "%s[%d,%d,%d,%d]", i ? "," : "", k[i].r, k[i].g, k[i].b, k[i].a);
  rec_printf(c, "]");
}
static void rec_close(FoCtx* c) { rec_printf(c, "]"); }
void fo_record_begin(FoCtx* c) {
  c->rec.on = 1;
  c->rec.first = 1;
  c->rec.len = 0;
  rec_printf(c, "[");
}
const char* fo_record_json(FoCtx* c) {
  if (!c->rec.on) return "[]";
  rec_printf(c, "\n]");
  c->rec.on = 0;
  return c->rec.buf;
}

/* ------------------------------------------------------------------ context */
FoCtx* fo_create(int atlas_size, float pixel_scale) {
  FoCtx* c = (FoCtx*)calloc(1, sizeof(FoCtx));
  c->atlas_size = atlas_size > 0 ? atlas_size : 1024; /* newContext defaults glcontext.nim:255-261 */
  c->atlas_margin = 4;
  c->pixel_scale = pixel_scale;
  c->ui_scale = 1.0f;
  c->aa = 1.2f; /* DefaultSdfAaFactor figbackend.nim:34 */
  c->mat = aff_identity();
  c->heights = (uint16_t*)calloc((size_t)c->atlas_size, sizeof(uint16_t));
  int s = c->atlas_size, l = 0;
  while (s >= 1 && l < FO_MAX_MIPS) {
    c->atlas[l] = (uint8_t*)calloc((size_t)s * s * 4, 1);
    l++;
    if (s == 1) break;
    s /= 2;
  }
  c->n_mips = l;
  return c;
}
static void free_frame(FoCtx* c) {
  free(c->fb);
  free(c->backdrop);
  free(c->backdrop_tmp);
  c->fb = c->backdrop = c->backdrop_tmp = NULL;
  for (int i = 0; i < FO_MAX_MASKS; i++) { free(c->mask[i]); c->mask[i] = NULL; }
}
void fo_destroy(FoCtx* c) {
  if (!c) return;
  free_frame(c);
  for (int i = 0; i < FO_MAX_MIPS; i++) free(c->atlas[i]);
  free(c->heights);
  free(c->entries);
  free(c->rec.buf);
  free(c);
}

/* transforms: glcontext.nim:1991-2017 */
void fo_save_transform(FoCtx* c) { rec_open(c, "save_transform"); rec_close(c); if (c->n_mats < FO_MAX_MATS) c->mats[c->n_mats++] = c->mat; }
void fo_restore_transform(FoCtx* c) { rec_open(c, "restore_transform"); rec_close(c); if (c->n_mats > 0) c->mat = c->mats[--c->n_mats]; }
void fo_translate(FoCtx* c, float x, float y) {
  rec_open(c, "translate"); rec_f(c, x); rec_f(c, y); rec_close(c);
  Aff t = {1, 0, 0, 1, x, y};
  c->mat = aff_mul(c->mat, t);
}
void fo_rotate(FoCtx* c, float a) {
  rec_open(c, "rotate"); rec_f(c, a); rec_close(c);
  float cs = cosf(a), sn = sinf(a);
  /* vmath rotateZ: m[0,1] = -sin, m[1,0] = sin, i.e. column 0 = (cos, -sin), column 1 = (sin, cos): x' = cos x + sin y,
   * y' = -sin x + cos y.  Pinned by the reference golden tests/expected/render_line_rect.png (a line drawn as a rotated box). */
  Aff r = {cs, -sn, sn, cs, 0, 0};
  c->mat = aff_mul(c->mat, r);
}
void fo_scale(FoCtx* c, float sx, float sy) {
  rec_open(c, "scale"); rec_f(c, sx); rec_f(c, sy); rec_close(c);
  Aff s = {sx, 0, 0, sy, 0, 0};
  c->mat = aff_mul(c->mat, s);
}
void fo_apply_transform(FoCtx* c, const float m[16]) {
  rec_open(c, "apply_transform"); rec_fv(c, m, 16); rec_close(c);
  /* column-major Mat4; `ctx.mat * vec3(x, y, 0)` (glcontext.nim:905-906) only uses the 2D affine part */
  Aff n = {m[0], m[1], m[4], m[5], m[12], m[13]};
  c->mat = aff_mul(c->mat, n);
}
void fo_set_aa_factor(FoCtx* c, float aa) { rec_open(c, "set_aa_factor"); rec_f(c, aa); rec_close(c); c->aa = aa; }
void fo_set_text_subpixel(FoCtx* c, int enabled, float shift) { c->subpixel_enabled = enabled; c->subpixel_shift = shift; }
void fo_set_text_subpixel_glyph_variants(FoCtx* c, int enabled) { c->subpixel_variants = enabled; }
/* setTextSubpixelShift figbackend.nim:663-686: what renderText calls before every glyph (figrender.nim:476) */
void fo_set_text_subpixel_shift(FoCtx* c, float shift) { rec_open(c, "set_text_subpixel_shift"); rec_f(c, shift); rec_close(c); c->subpixel_shift = shift; }

/* beginFrame: glcontext.nim:2080-2092, 1951-1980 */
void fo_begin_frame(FoCtx* c, int w, int h, int clear, const float rgba[4]) {
  rec_open(c, "begin_frame"); rec_i(c, clear); rec_fv(c, rgba, 4); rec_close(c);
  if (w != c->W || h != c->H || !c->fb) {
    free_frame(c);
    c->W = w;
    c->H = h;
    c->fb = (uint8_t*)calloc((size_t)w * h * 4, 1);
    c->backdrop = (uint8_t*)calloc((size_t)w * h * 4, 1);
    c->backdrop_tmp = (uint8_t*)calloc((size_t)w * h * 4, 1);
  }
  if (clear) {
    uint8_t k[4] = {to_unorm8(rgba[0]), to_unorm8(rgba[1]), to_unorm8(rgba[2]), to_unorm8(rgba[3])};
    for (size_t i = 0; i < (size_t)w * h; i++) memcpy(c->fb + i * 4, k, 4);
  }
  c->frame_begun = 1;
  c->n_rect_masks = 0;
  c->mask_write = 0;
  c->mask_begun = 0;
}
void fo_end_frame(FoCtx* c) { rec_open(c, "end_frame"); rec_close(c); c->frame_begun = 0; }

/* ------------------------------------------------------------------ radii packing: glcontext.nim:745-817 */
static float clamp_radius(float r, float m) { return r <= 0.0f ? 0.0f : nim_round(maxf(1.0f, minf(r, m))); }
void fo_rounded_radii_vec(const float rx[4], const float ry[4], float hx, float hy, float out[4], int* elliptical) {
  enum { TL = 0, TR = 1, BL = 2, BR = 3 };
  int circular = 1;
  for (int i = 0; i < 4; i++) if (rx[i] != ry[i]) circular = 0;
  if (circular) {
    float m = minf(hx, hy);
    out[0] = clamp_radius(rx[TR], m);
    out[1] = clamp_radius(rx[BR], m);
    out[2] = clamp_radius(rx[TL], m);
    out[3] = clamp_radius(rx[BL], m);
    *elliptical = 0;
    return;
  }
  float cm = minf(hx, hy);
  static const int order[4] = {TR, BR, TL, BL};
  for (int k = 0; k < 4; k++) {
    int i = order[k];
    float cx = clamp_radius(rx[i], hx), cy = clamp_radius(ry[i], hy);
    float v;
    if (rx[i] == ry[i]) v = -(clamp_radius(rx[i], cm) + 1.0f);
    else if (cx == cy) v = -(cx + 1.0f);
    else {
      float qx = nim_round(clampf(cx / maxf(hx, 0.000001f), 0.0f, 1.0f) * 4095.0f);
      float qy = nim_round(clampf(cy / maxf(hy, 0.000001f), 0.0f, 1.0f) * 4095.0f);
      v = qx + qy * 4096.0f;
    }
    out[k] = v;
  }
  *elliptical = 1;
}

/* ------------------------------------------------------------------ fills: figbackend.nim:96-183 */
static FoColor lerp_color(FoColor a, FoColor b, float t) { /* figbackend.nim:129-136 */
  float ct = clampf(t, 0.0f, 1.0f), it = 1.0f - ct;
  FoColor r;
  r.r = (uint8_t)nim_round((float)a.r * it + (float)b.r * ct);
  r.g = (uint8_t)nim_round((float)a.g * it + (float)b.g * ct);
  r.b = (uint8_t)nim_round((float)a.b * it + (float)b.b * ct);
  r.a = (uint8_t)nim_round((float)a.a * it + (float)b.a * ct);
  return r;
}
static float fill_mid_pos01(const FoFill* f) { return clampf((float)f->mid_pos / 255.0f, 0.01f, 0.99f); } /* figbackend.nim:125 */
static FoColor sample_color(const FoFill* f, float t) { /* figbackend.nim:138-153 */
  if (f->kind == FO_FILL_COLOR) return f->start;
  if (f->kind == FO_FILL_LINEAR2) return lerp_color(f->start, f->stop, t);
  float ct = clampf(t, 0.0f, 1.0f), mid = fill_mid_pos01(f);
  if (ct <= mid) return lerp_color(f->start, f->mid, ct / mid);
  return lerp_color(f->mid, f->stop, (ct - mid) / (1.0f - mid));
}
void fo_gradient_colors(const FoFill* f, FoColor out[4]) { /* figbackend.nim:161-183; order BL,BR,TR,TL */
  int axis = f->kind == FO_FILL_COLOR ? FO_AXIS_X : f->axis;
  static const float T[4][4] = {{0, 1, 1, 0}, {1, 1, 0, 0}, {0.5f, 1, 0.5f, 0}, {0, 0.5f, 1, 0.5f}};
  for (int i = 0; i < 4; i++) out[i] = sample_color(f, T[axis][i]);
}
static uint8_t fill_alpha_max(const FoFill* f) { /* figrender.nim:587-594 */
  if (f->kind == FO_FILL_COLOR) return f->start.a;
  if (f->kind == FO_FILL_LINEAR2) return f->start.a > f->stop.a ? f->start.a : f->stop.a;
  uint8_t m = f->start.a > f->mid.a ? f->start.a : f->mid.a;
  return m > f->stop.a ? m : f->stop.a;
}

/* ------------------------------------------------------------------ texture sampling */
/* GL_LINEAR, clamp-to-edge RGBA8 fetch at texel-space coordinate (x,y) = s*W-0.5 */
static void bilinear_rgba8_clamp(const uint8_t* tex, int W, int H, float x, float y, float out[4]) {
  float fx = floorf(x), fy = floorf(y);
  float ax = x - fx, ay = y - fy;
  int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  if (x0 < 0) x0 = 0; if (x0 > W - 1) x0 = W - 1;
  if (x1 < 0) x1 = 0; if (x1 > W - 1) x1 = W - 1;
  if (y0 < 0) y0 = 0; if (y0 > H - 1) y0 = H - 1;
  if (y1 < 0) y1 = 0; if (y1 > H - 1) y1 = H - 1;
  const uint8_t *p00 = tex + ((size_t)y0 * W + x0) * 4, *p10 = tex + ((size_t)y0 * W + x1) * 4;
  const uint8_t *p01 = tex + ((size_t)y1 * W + x0) * 4, *p11 = tex + ((size_t)y1 * W + x1) * 4;
  for (int k = 0; k < 4; k++) {
    float top = from_unorm8(p00[k]) * (1.0f - ax) + from_unorm8(p10[k]) * ax;
    float bot = from_unorm8(p01[k]) * (1.0f - ax) + from_unorm8(p11[k]) * ax;
    out[k] = top * (1.0f - ay) + bot * ay;
  }
}
/* TEST-ONLY sampler model (fo_debug_texcoord_model).  The goldens tests/golden/ss_*.png come from SwiftShader, whose GLES sampler
 * takes texture coordinates as 16-bit NORMALISED fixed point: on the 256^2 golden atlas a bilinear fraction has 8 bits.  Model 0
 * (the default, and what every parity test compares the HIP path with) filters at float32 coordinates -- GL leaves filter
 * precision to the implementation.  Model 1 truncates the coordinate to that grid first (coordinate -> 16 bits, half a texel
 * subtracted in the same units, texel = high half of coordinate * size, fraction = low half) and filters in float as before:
 * fitted to the goldens, it takes the MSDF scene's 2-LSB pixels from 13 to 1 (tests/test_oracle.py), which is the evidence that
 * those differences are the sampler's grid and not the restatement's arithmetic. */
static int g_tc_model = 0;
void fo_debug_texcoord_model(int model) { g_tc_model = model; }
static void bilinear_rgba8_repeat_grid16(const uint8_t* tex, int S, float u, float v, float out[4]) {
  float c[2] = {u, v}, fr[2];
  int i0[2], i1[2];
  for (int k = 0; k < 2; k++) {
    float w = c[k] - floorf(c[k]);
    int q = ((int)(w * 65536.0f) - 0x8000 / S) & 0xFFFF;
    unsigned prod = (unsigned)q * (unsigned)S;
    i0[k] = (int)(prod >> 16) % S;
    i1[k] = (i0[k] + 1) % S;
    fr[k] = (float)(prod & 0xFFFFu) / 65536.0f;
  }
  const uint8_t *p00 = tex + ((size_t)i0[1] * S + i0[0]) * 4, *p10 = tex + ((size_t)i0[1] * S + i1[0]) * 4;
  const uint8_t *p01 = tex + ((size_t)i1[1] * S + i0[0]) * 4, *p11 = tex + ((size_t)i1[1] * S + i1[0]) * 4;
  for (int k = 0; k < 4; k++) {
    float top = from_unorm8(p00[k]) * (1.0f - fr[0]) + from_unorm8(p10[k]) * fr[0];
    float bot = from_unorm8(p01[k]) * (1.0f - fr[0]) + from_unorm8(p11[k]) * fr[0];
    out[k] = top * (1.0f - fr[1]) + bot * fr[1];
  }
}
/* GL_LINEAR, GL_REPEAT (atlas default wrap, glcontext.nim:157-169) */
static void bilinear_rgba8_repeat(const uint8_t* tex, int S, float x, float y, float out[4]) {
  if (g_tc_model > 0) { bilinear_rgba8_repeat_grid16(tex, S, (x + 0.5f) / (float)S, (y + 0.5f) / (float)S, out); return; }
  float fx = floorf(x), fy = floorf(y);
  float ax = x - fx, ay = y - fy;
  int x0 = ((int)fx % S + S) % S, y0 = ((int)fy % S + S) % S;
  int x1 = (x0 + 1) % S, y1 = (y0 + 1) % S;
  const uint8_t *p00 = tex + ((size_t)y0 * S + x0) * 4, *p10 = tex + ((size_t)y0 * S + x1) * 4;
  const uint8_t *p01 = tex + ((size_t)y1 * S + x0) * 4, *p11 = tex + ((size_t)y1 * S + x1) * 4;
  for (int k = 0; k < 4; k++) {
    float top = from_unorm8(p00[k]) * (1.0f - ax) + from_unorm8(p10[k]) * ax;
    float bot = from_unorm8(p01[k]) * (1.0f - ax) + from_unorm8(p11[k]) * ax;
    out[k] = top * (1.0f - ay) + bot * ay;
  }
}
/* texture(atlasTex, uv): LINEAR_MIPMAP_LINEAR min / LINEAR mag; rho from the quad's
 * (per-triangle constant) uv derivatives, in level-0 texels per pixel. */
static void sample_atlas(const FoCtx* c, float u, float v, float rho, int lod0_only, float out[4]) {
  int S = c->atlas_size;
  float lambda = (lod0_only || rho <= 0.0f) ? 0.0f : log2f(rho);
  if (lambda <= 0.0f || c->n_mips < 2) {
    bilinear_rgba8_repeat(c->atlas[0], S, u * (float)S - 0.5f, v * (float)S - 0.5f, out);
    return;
  }
  float maxl = (float)(c->n_mips - 1);
  if (lambda > maxl) lambda = maxl;
  int l0 = (int)floorf(lambda);
  int l1 = l0 + 1 > c->n_mips - 1 ? c->n_mips - 1 : l0 + 1;
  float f = lambda - (float)l0;
  float a[4], b[4];
  int S0 = S >> l0, S1 = S >> l1;
  bilinear_rgba8_repeat(c->atlas[l0], S0, u * (float)S0 - 0.5f, v * (float)S0 - 0.5f, a);
  bilinear_rgba8_repeat(c->atlas[l1], S1, u * (float)S1 - 0.5f, v * (float)S1 - 0.5f, b);
  for (int k = 0; k < 4; k++) out[k] = a[k] * (1.0f - f) + b[k] * f;
}

/* ------------------------------------------------------------------ one quad through the pipeline */
typedef struct {
  v2 pos[4], uv[4];  /* vertex order BL,BR,TR,TL (glcontext.nim:1498-1509) */
  v4 col[4];         /* unorm8 -> float */
  v4 mid, stop;
  v4 params, radii;
  int mode_word;
  float factor0, factor1;
  float subpixel_shift;
} Quad;

typedef struct { float u, v; v4 col; float rho; float fw_u, fw_v; } Frag;

/* atlas_rect_mask.frag:222-237 */
static float rect_mask_alpha(const RectMask* rm, float aa, float px, float py) {
  if (rm->params.z < 0.0f || rm->params.w < 0.0f) return 1.0f;
  float lx = (rm->matx.x * px + rm->matx.y * py) + rm->matx.z;
  float ly = (rm->maty.x * px + rm->maty.y * py) + rm->maty.z;
  float qx = lx - rm->params.x, qy = ly - rm->params.y;
  float dist = rm->maty.w > 0.5f ? sd_elliptical_rounded_box(qx, -qy, rm->params.z, rm->params.w, rm->radii)
                                 : sd_rounded_box(qx, -qy, rm->params.z, rm->params.w, rm->radii);
  return 1.0f - clampf(aa * dist + 0.5f, 0.0f, 1.0f);
}

/* mask.frag:186-234 -> scalar alpha (fragColor = vec4(alpha)) */
static float shade_mask(const FoCtx* c, const Quad* q, const Frag* f) {
  int word = q->mode_word;
  int fill_mode = word / 256;
  int mode = word - fill_mode * 256;
  int ellip = mode >= 128;
  if (ellip) mode -= 128;
  float alpha;
  if (mode == 0) {
    float t[4];
    sample_atlas(c, f->u, f->v, f->rho, 0, t);
    alpha = t[3] * f->col.w;
  } else {
    float px = (f->u - 0.5f) * 2.0f * q->params.x, py = (f->v - 0.5f) * 2.0f * q->params.y;
    float dist = ellip ? sd_elliptical_rounded_box(px, -py, q->params.z, q->params.w, q->radii)
                       : sd_rounded_box(px, -py, q->params.z, q->params.w, q->radii);
    if (mode == 12) {
      float hw = maxf(q->factor0, 0.0f) * 0.5f;
      dist = fabsf(dist + hw) - hw;
    }
    float cl = clampf(c->aa * dist + 0.5f, 0.0f, 1.0f);
    alpha = (1.0f - cl) * f->col.w;
  }
  return alpha;
}

/* atlas.frag:252-399 -> fragColor before the mask multiply.  (x,y) = integer pixel (for mode 17) */
static v4 shade_main(const FoCtx* c, const Quad* q, const Frag* f, int x, int y) {
  int word = q->mode_word;
  int fill_mode = word / 256;
  int mode = word - fill_mode * 256;
  int ellip = mode >= 128;
  if (ellip) mode -= 128;
  float qhx = q->params.x, qhy = q->params.y;
  int inset = mode == 9;
  float shx = inset ? qhx : q->params.z, shy = inset ? qhy : q->params.w;
  float px = (f->u - 0.5f) * 2.0f * qhx, py = (f->v - 0.5f) * 2.0f * qhy;
  int bezier = mode >= 18 && mode <= 20; /* isBezierStrokeMode atlas.frag:162-168 */
  v2 bA = {q->params.z, q->params.w}, bB = {q->radii.x, q->radii.y}, bC = {q->radii.z, q->radii.w};
  float dist = bezier ? sd_bezier(px, py, bA, bB, bC)
                      : (ellip ? sd_elliptical_rounded_box(px, -py, shx, shy, q->radii) : sd_rounded_box(px, -py, shx, shy, q->radii));
  float sdf_factor = q->factor0;
  float sdf_spread = fill_mode == 0 ? q->factor1 : 0.0f;
  /* evalFillColor atlas.frag:233-250 */
  v4 fc = f->col;
  if (fill_mode != 0) {
    float t = clampf(linear3_t(fill_mode, f->u, f->v), 0.0f, 1.0f);
    float mid = clampf(q->factor1, 0.01f, 0.99f);
    if (t <= mid) {
      float k = t / mid;
      fc.x = mixf(f->col.x, q->mid.x, k); fc.y = mixf(f->col.y, q->mid.y, k);
      fc.z = mixf(f->col.z, q->mid.z, k); fc.w = mixf(f->col.w, q->mid.w, k);
    } else {
      float k = (t - mid) / (1.0f - mid);
      fc.x = mixf(q->mid.x, q->stop.x, k); fc.y = mixf(q->mid.y, q->stop.y, k);
      fc.z = mixf(q->mid.z, q->stop.z, k); fc.w = mixf(q->mid.w, q->stop.w, k);
    }
  }
  float alpha = 0.0f;
  v4 out;
  if (mode == 0) { /* atlas.frag:284-295 */
    float u = f->u;
    if (c->subpixel_enabled) u -= q->subpixel_shift * (1.0f / maxf((float)c->atlas_size, 1.0f));
    float t[4];
    sample_atlas(c, u, f->v, f->rho, 0, t);
    out.x = t[0] * f->col.x; out.y = t[1] * f->col.y; out.z = t[2] * f->col.z; out.w = t[3] * f->col.w;
    return out;
  }
  if (mode >= 13 && mode <= 16) { /* atlas.frag:296-318 */
    float t[4];
    sample_atlas(c, f->u, f->v, 0.0f, 1, t); /* textureLod(atlasTex, uv, 0.0) */
    int is_mtsdf = (mode == 14 || mode == 16), is_stroke = (mode == 15 || mode == 16);
    float sd = is_mtsdf ? t[3] : median3(t[0], t[1], t[2]);
    /* msdfScreenPxRange atlas.frag:45-49 */
    float unit = q->factor0 / (float)c->atlas_size;
    float spr = maxf(0.5f * (unit * (1.0f / f->fw_u) + unit * (1.0f / f->fw_v)), 1.0f);
    float spd = spr * (sd - q->factor1);
    if (is_stroke) {
      float hw = maxf(q->params.y, 0.0f) * 0.5f;
      alpha = clampf(hw - fabsf(spd) + 0.5f, 0.0f, 1.0f);
    } else {
      alpha = clampf(spd + 0.5f, 0.0f, 1.0f);
    }
    out.x = fc.x; out.y = fc.y; out.z = fc.z; out.w = fc.w * alpha;
    return out;
  }
  switch (mode) {
    case 18: case 19: case 20: { /* atlas.frag:321-336 */
      float sd = bezier_stroke_sd(dist, px, py, bA, bB, bC, maxf(sdf_factor, 0.0f) * 0.5f, mode);
      alpha = 1.0f - clampf(c->aa * sd + 0.5f, 0.0f, 1.0f);
      break;
    }
    case 11: { float h = sdf_factor * 0.5f; float sd = fabsf(dist + h) - h; alpha = sd < 0.0f ? 1.0f : 0.0f; break; }
    case 12: { float h = sdf_factor * 0.5f; float sd = fabsf(dist + h) - h; alpha = 1.0f - clampf(c->aa * sd + 0.5f, 0.0f, 1.0f); break; }
    case 7: { float sd = dist - sdf_spread; float a = shadow_profile(sd, sdf_factor); alpha = sd > 0.0f ? minf(a, 1.0f) : 1.0f; break; }
    case 8: {
      float inside = 1.0f - clampf(c->aa * dist + 0.5f, 0.0f, 1.0f);
      float sd = dist - sdf_spread; float a = shadow_profile(sd, sdf_factor);
      alpha = sd >= 0.0f ? minf(a, 1.0f) : inside; break;
    }
    case 9: { /* atlas.frag:364-380 */
      float qx = px, qy = -py;
      float sx = qx - q->params.z, sy = qy - (-q->params.w);
      float clip_d = ellip ? sd_elliptical_rounded_box(qx, qy, qhx, qhy, q->radii) : sd_rounded_box(qx, qy, qhx, qhy, q->radii);
      float clip_a = 1.0f - clampf(c->aa * clip_d + 0.5f, 0.0f, 1.0f);
      float sh_d = ellip ? sd_elliptical_rounded_box(sx, sy, qhx, qhy, q->radii) : sd_rounded_box(sx, sy, qhx, qhy, q->radii);
      float sd = sh_d + sdf_spread; float a = shadow_profile(sd, sdf_factor);
      float inset_a = sd < 0.0f ? minf(a, 1.0f) : 1.0f;
      alpha = clip_a * inset_a; break;
    }
    case 17: { /* atlas.frag:381-388: blurred backdrop at the fragment's own pixel */
      alpha = 1.0f - clampf(c->aa * dist + 0.5f, 0.0f, 1.0f);
      const uint8_t* b = c->backdrop + ((size_t)y * c->W + x) * 4;
      out.x = from_unorm8(b[0]); out.y = from_unorm8(b[1]); out.z = from_unorm8(b[2]); out.w = from_unorm8(b[3]) * alpha;
      return out;
    }
    default: alpha = 1.0f - clampf(c->aa * dist + 0.5f, 0.0f, 1.0f); break; /* ClipAA & others atlas.frag:389-393 */
  }
  out.x = fc.x; out.y = fc.y; out.z = fc.z; out.w = fc.w * alpha;
  return out;
}

/* edge function helpers for the two triangles (3,0,1) and (2,3,1): glcontext.nim:418-429 */
static inline double edge_fn(v2 a, v2 b, double px, double py) { return ((double)b.x - a.x) * (py - a.y) - ((double)b.y - a.y) * (px - a.x); }
/* top-left rule in the final image orientation (y down): an edge a->b of a triangle whose
 * interior is on the positive side of edge_fn owns the pixels lying exactly on it iff it is a
 * top edge (horizontal, interior below) or a left edge (interior to the right). */
static int edge_owns(v2 a, v2 b, v2 opp) {
  if (a.y == b.y) return opp.y > a.y;
  /* x of the edge at the opposite vertex's y */
  double t = ((double)opp.y - a.y) / ((double)b.y - a.y);
  double ex = a.x + t * ((double)b.x - a.x);
  return opp.x > ex;
}

static void draw_quad(FoCtx* c, const Quad* q) {
  int W = c->W, H = c->H;
  float minx = q->pos[0].x, maxx = minx, miny = q->pos[0].y, maxy = miny;
  for (int i = 1; i < 4; i++) {
    minx = minf(minx, q->pos[i].x); maxx = maxf(maxx, q->pos[i].x);
    miny = minf(miny, q->pos[i].y); maxy = maxf(maxy, q->pos[i].y);
  }
  int x0 = (int)floorf(minx), x1 = (int)ceilf(maxx), y0 = (int)floorf(miny), y1 = (int)ceilf(maxy);
  if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (x1 > W) x1 = W; if (y1 > H) y1 = H;
  if (x0 >= x1 || y0 >= y1) return;
  static const int TRI[2][3] = {{3, 0, 1}, {2, 3, 1}};
  /* per-triangle setup */
  double area[2];
  int own[2][3];
  float rho[2], fwu[2], fwv[2];
  for (int t = 0; t < 2; t++) {
    v2 a = q->pos[TRI[t][0]], b = q->pos[TRI[t][1]], d = q->pos[TRI[t][2]];
    area[t] = edge_fn(a, b, d.x, d.y);
    /* edge k is opposite vertex k: e0 = (b->d), e1 = (d->a), e2 = (a->b) */
    own[t][0] = edge_owns(b, d, a);
    own[t][1] = edge_owns(d, a, b);
    own[t][2] = edge_owns(a, b, d);
    /* uv derivatives of this triangle's affine map (for LOD / fwidth) */
    v2 ua = q->uv[TRI[t][0]], ub = q->uv[TRI[t][1]], ud = q->uv[TRI[t][2]];
    double e1x = (double)b.x - a.x, e1y = (double)b.y - a.y, e2x = (double)d.x - a.x, e2y = (double)d.y - a.y;
    double det = e1x * e2y - e1y * e2x;
    if (det != 0.0) {
      double du1 = (double)ub.x - ua.x, du2 = (double)ud.x - ua.x, dv1 = (double)ub.y - ua.y, dv2 = (double)ud.y - ua.y;
      double dudx = (du1 * e2y - du2 * e1y) / det, dudy = (du2 * e1x - du1 * e2x) / det;
      double dvdx = (dv1 * e2y - dv2 * e1y) / det, dvdy = (dv2 * e1x - dv1 * e2x) / det;
      double S = (double)c->atlas_size;
      double rx = sqrt(dudx * dudx + dvdx * dvdx) * S, ry = sqrt(dudy * dudy + dvdy * dvdy) * S;
      rho[t] = (float)(rx > ry ? rx : ry);
      fwu[t] = (float)(fabs(dudx) + fabs(dudy));
      fwv[t] = (float)(fabs(dvdx) + fabs(dvdy));
    } else {
      rho[t] = 0.0f; fwu[t] = fwv[t] = 1.0f;
    }
  }
  const RectMask* fast = NULL;
  if (!c->mask_begun)
    for (int i = c->n_rect_masks - 1; i >= 0; i--)
      if (c->rect_masks[i].kind == 1) { fast = &c->rect_masks[i]; break; }
  int mask_read = c->mask_begun ? c->mask_write - 1 : c->mask_write; /* flush(maskTextureRead) glcontext.nim:643,1891,1920 */
  const uint8_t* mask_tex = mask_read != 0 ? c->mask[mask_read] : NULL;
  uint8_t* mask_dst = c->mask_begun ? c->mask[c->mask_write] : NULL;

#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
  for (int y = y0; y < y1; y++) {
    for (int x = x0; x < x1; x++) {
      double cx = x + 0.5, cy = y + 0.5;
      int hit = -1;
      double w0 = 0, w1 = 0, w2 = 0;
      for (int t = 0; t < 2 && hit < 0; t++) {
        if (area[t] == 0.0) continue;
        v2 a = q->pos[TRI[t][0]], b = q->pos[TRI[t][1]], d = q->pos[TRI[t][2]];
        double sgn = area[t] > 0 ? 1.0 : -1.0;
        double e0 = sgn * edge_fn(b, d, cx, cy), e1 = sgn * edge_fn(d, a, cx, cy), e2 = sgn * edge_fn(a, b, cx, cy);
        if (e0 < 0 || e1 < 0 || e2 < 0) continue;
        if ((e0 == 0 && !own[t][0]) || (e1 == 0 && !own[t][1]) || (e2 == 0 && !own[t][2])) continue;
        hit = t; w0 = e0; w1 = e1; w2 = e2;
      }
      if (hit < 0) continue;
      double inv = 1.0 / (w0 + w1 + w2);
      float l0 = (float)(w0 * inv), l1 = (float)(w1 * inv), l2 = (float)(w2 * inv);
      const int* T = TRI[hit];
      Frag f;
      f.u = l0 * q->uv[T[0]].x + l1 * q->uv[T[1]].x + l2 * q->uv[T[2]].x;
      f.v = l0 * q->uv[T[0]].y + l1 * q->uv[T[1]].y + l2 * q->uv[T[2]].y;
      f.col.x = l0 * q->col[T[0]].x + l1 * q->col[T[1]].x + l2 * q->col[T[2]].x;
      f.col.y = l0 * q->col[T[0]].y + l1 * q->col[T[1]].y + l2 * q->col[T[2]].y;
      f.col.z = l0 * q->col[T[0]].z + l1 * q->col[T[1]].z + l2 * q->col[T[2]].z;
      f.col.w = l0 * q->col[T[0]].w + l1 * q->col[T[1]].w + l2 * q->col[T[2]].w;
      f.rho = rho[hit]; f.fw_u = fwu[hit]; f.fw_v = fwv[hit];
      size_t pi = (size_t)y * W + x;
      if (mask_dst) {
        /* mask.frag + blend into the R8 plane: r = a*a + dst*(1-a)  (utils/glutils.nim:150-154) */
        float a = shade_mask(c, q, &f);
        if (mask_tex) a *= from_unorm8(mask_tex[pi]);
        float d = from_unorm8(mask_dst[pi]);
        mask_dst[pi] = to_unorm8(a * a + d * (1.0f - a));
      } else {
        v4 s = shade_main(c, q, &f, x, y);
        if (mask_tex) s.w *= from_unorm8(mask_tex[pi]); /* atlas.frag:401-404 */
        if (fast) s.w *= rect_mask_alpha(fast, c->aa, (float)cx, (float)cy); /* atlas_rect_mask.frag:425 */
        uint8_t* d = c->fb + pi * 4;
        float dr = from_unorm8(d[0]), dg = from_unorm8(d[1]), db = from_unorm8(d[2]), da = from_unorm8(d[3]);
        float sa = s.w, ia = 1.0f - sa;
        d[0] = to_unorm8(s.x * sa + dr * ia);
        d[1] = to_unorm8(s.y * sa + dg * ia);
        d[2] = to_unorm8(s.z * sa + db * ia);
        d[3] = to_unorm8(sa + da * ia);
      }
    }
  }
}

static v4 col_to_v4(FoColor k) { v4 r = {from_unorm8(k.r), from_unorm8(k.g), from_unorm8(k.b), from_unorm8(k.a)}; return r; }
static v2 ceil_xf(const FoCtx* c, float x, float y) { v2 p = aff_apply(c->mat, x, y); p.x = ceilf(p.x); p.y = ceilf(p.y); return p; }
static float active_subpixel_shift(const FoCtx* c) { /* glcontext.nim:819-822 */
  if (!c->subpixel_enabled) return 0.0f;
  return maxf(0.0f, minf(c->subpixel_shift, 0.999f));
}

/* drawRoundedRectSdfOpenGl: glcontext.nim:1449-1559 */
void fo_draw_rounded_rect_sdf(FoCtx* c, const float rect[4], const FoColor colors[4], const float radii_x[4],
                              const float radii_y[4], int mode, float factor, float spread, const float shape[2],
                              int fill_mode, FoColor mid, FoColor stop, float mid_pos) {
  rec_open(c, "draw_rounded_rect_sdf");
  rec_fv(c, rect, 4); rec_cols(c, colors); rec_fv(c, radii_x, 4); rec_fv(c, radii_y, 4); rec_i(c, mode);
  rec_f(c, factor); rec_f(c, spread); rec_fv(c, shape, 2); rec_i(c, fill_mode); rec_col(c, mid); rec_col(c, stop); rec_f(c, mid_pos);
  rec_close(c);
  float x = rect[0], y = rect[1], w = rect[2], h = rect[3];
  if (w <= 0 || h <= 0) return;
  Quad q;
  memset(&q, 0, sizeof q);
  float qhx = w * 0.5f, qhy = h * 0.5f;
  int inset = mode == 9;
  float rsx = (shape[0] > 0.0f && shape[1] > 0.0f) ? shape[0] : w;
  float rsy = (shape[0] > 0.0f && shape[1] > 0.0f) ? shape[1] : h;
  float shx = inset ? qhx : rsx * 0.5f, shy = inset ? qhy : rsy * 0.5f;
  q.params.x = qhx; q.params.y = qhy;
  if (inset) { q.params.z = shape[0]; q.params.w = shape[1]; } else { q.params.z = shx; q.params.w = shy; }
  float r4[4];
  int ellip;
  fo_rounded_radii_vec(radii_x, radii_y, shx, shy, r4, &ellip);
  q.radii.x = r4[0]; q.radii.y = r4[1]; q.radii.z = r4[2]; q.radii.w = r4[3];
  q.pos[0] = ceil_xf(c, x, y + h); q.pos[1] = ceil_xf(c, x + w, y + h);
  q.pos[2] = ceil_xf(c, x + w, y); q.pos[3] = ceil_xf(c, x, y);
  q.uv[0].x = 0; q.uv[0].y = 1; q.uv[1].x = 1; q.uv[1].y = 1; q.uv[2].x = 1; q.uv[2].y = 0; q.uv[3].x = 0; q.uv[3].y = 0;
  for (int i = 0; i < 4; i++) q.col[i] = col_to_v4(colors[i]);
  q.mid = col_to_v4(mid); q.stop = col_to_v4(stop);
  q.factor0 = factor;
  q.factor1 = fill_mode == 0 ? spread : clampf(mid_pos, 0.01f, 0.99f);
  q.mode_word = mode + (ellip ? 128 : 0) + fill_mode * 256; /* encodeSdfMode glcontext.nim:1002-1008 */
  q.subpixel_shift = active_subpixel_shift(c);
  draw_quad(c, &q);
}

/* drawRoundedRectSdf(fill: BackendFill): glcontext.nim:1581-1617 */
void fo_draw_rounded_rect_fill(FoCtx* c, const float rect[4], const FoFill* fill, const float radii_x[4],
                               const float radii_y[4], int mode, float factor, float spread, const float shape[2]) {
  FoColor zero = {0, 0, 0, 0};
  if (fill->kind == FO_FILL_LINEAR3 && (mode == 3 || mode == 11 || mode == 12)) {
    FoColor cols[4] = {fill->start, fill->start, fill->start, fill->start};
    fo_draw_rounded_rect_sdf(c, rect, cols, radii_x, radii_y, mode, factor, spread, shape, 1 + fill->axis, fill->mid,
                             fill->stop, fill_mid_pos01(fill));
  } else {
    FoColor cols[4];
    fo_gradient_colors(fill, cols);
    fo_draw_rounded_rect_sdf(c, rect, cols, radii_x, radii_y, mode, factor, spread, shape, 0, zero, zero, 0.5f);
  }
}

/* ------------------------------------------------------------------ atlas: glcontext.nim:541-586 */
static AtlasEntry* find_entry(FoCtx* c, int64_t key) {
  for (int i = 0; i < c->n_entries; i++) if (c->entries[i].used && c->entries[i].key == key) return &c->entries[i];
  return NULL;
}
static int place_rect(FoCtx* c, int64_t key, int w, int h, int* ox, int* oy) { /* findEmptyRect + entry bookkeeping */
  int S = c->atlas_size, M = c->atlas_margin;
  int iw = w + M * 2, ih = h + M * 2;
  int lowest = S, at = 0;
  for (int i = 0; i < S; i++) {
    int v = c->heights[i];
    if (v < lowest) {
      int fit = 1;
      for (int j = 0; j <= iw; j++) {
        if (i + j >= S) { fit = 0; break; }
        if ((int)c->heights[i + j] > v) { fit = 0; break; }
      }
      if (fit) { lowest = v; at = i; }
    }
  }
  if (lowest + ih > S) return -1; /* reference grows the atlas (glcontext.nim:536-539,564-567); callers size it up front */
  for (int j = at; j < at + iw; j++) c->heights[j] = (uint16_t)(lowest + ih + M * 2);
  int rx = at + M, ry = lowest + M;
  AtlasEntry* e = find_entry(c, key);
  if (!e) {
    if (c->n_entries == c->cap_entries) {
      c->cap_entries = c->cap_entries ? c->cap_entries * 2 : 64;
      c->entries = (AtlasEntry*)realloc(c->entries, (size_t)c->cap_entries * sizeof(AtlasEntry));
    }
    e = &c->entries[c->n_entries++];
  }
  e->key = key; e->used = 1;
  e->x = (float)rx / (float)S; e->y = (float)ry / (float)S; e->w = (float)w / (float)S; e->h = (float)h / (float)S;
  *ox = rx; *oy = ry;
  return 0;
}
/* ------------------------------------------------------------------ glyph outlines -> coverage
 * The reference rasterises glyphs with pixie (`image.fillText`, common/textrasters/pixie_raster.nim:83-87): third-party code that is
 * not under /root/reference, so the TEXELS it produces are unpinned.  What is restated here is the published
 * exact-area scanline accumulation every modern font rasteriser uses (font-rs / stb_truetype v2 / pixie's own fill): each
 * outline segment adds its signed area to the cells of the pixel rows it crosses, a running sum along the row turns areas
 * into coverage, non-zero winding via |sum| clamped to 1.  Output = what pixie hands to putImage for white paint:
 * premultiplied white, coverage in all four channels.
 *   segs: n x 6 floats {x0, y0, cx, cy, x1, y1} in pixel units of the w x h image, y down; a quadratic Bezier with control
 *   point (cx, cy), or a straight line when cx is NaN.  Contours must be closed (each ends where it started). */
static int flatten_count(const float* q) { /* segments for a quadratic so that the chord error stays under 0.025 px */
  float ddx = q[0] - 2.0f * q[2] + q[4], ddy = q[1] - 2.0f * q[3] + q[5];
  float dev = sqrtf(ddx * ddx + ddy * ddy);
  int n = (int)ceilf(sqrtf(dev * 10.0f)); /* error of n chords = dev / (4 n^2) <= 0.025 px */
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}
int fo_flatten_outline(const float* segs, int n, float* lines /* 4 floats each */, int cap) {
  int m = 0;
  for (int i = 0; i < n; i++) {
    const float* q = segs + 6 * i;
    if (q[2] != q[2]) { /* NaN: a line */
      if (m < cap) { lines[4 * m] = q[0]; lines[4 * m + 1] = q[1]; lines[4 * m + 2] = q[4]; lines[4 * m + 3] = q[5]; }
      m++;
      continue;
    }
    int k = flatten_count(q);
    float px = q[0], py = q[1];
    for (int j = 1; j <= k; j++) {
      float t = (float)j / (float)k, u = 1.0f - t;
      float x = j == k ? q[4] : (u * u) * q[0] + (2.0f * u * t) * q[2] + (t * t) * q[4];
      float y = j == k ? q[5] : (u * u) * q[1] + (2.0f * u * t) * q[3] + (t * t) * q[5];
      if (m < cap) { lines[4 * m] = px; lines[4 * m + 1] = py; lines[4 * m + 2] = x; lines[4 * m + 3] = y; }
      m++;
      px = x; py = y;
    }
  }
  return m;
}
/* one pixel row of the accumulation: the part of line (x0,y0)-(x1,y1) inside rows [y, y + 1) adds to acc[0 .. w] */
static void raster_row_line(float* acc, int w, int y, float x0, float y0, float x1, float y1) {
  if (y0 == y1) return;
  float dir = 1.0f;
  if (y0 > y1) { float t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; dir = -1.0f; }
  float ya = y0 > (float)y ? y0 : (float)y, yb = y1 < (float)(y + 1) ? y1 : (float)(y + 1);
  if (!(yb > ya)) return;
  float dxdy = (x1 - x0) / (y1 - y0);
  float xa = x0 + (ya - y0) * dxdy, xb = x0 + (yb - y0) * dxdy;
  float d = (yb - ya) * dir;
  float xl = xa < xb ? xa : xb, xr = xa < xb ? xb : xa;
  /* cells left of the image take everything at cell 0 (full coverage to their right), cells right of it nothing */
  if (xl < 0.0f) xl = 0.0f;
  if (xr < 0.0f) xr = 0.0f;
  if (xl > (float)w) xl = (float)w;
  if (xr > (float)w) xr = (float)w;
  float x0floor = floorf(xl);
  int x0i = (int)x0floor;
  float x1ceil = ceilf(xr);
  int x1i = (int)x1ceil;
  if (x1i <= x0i + 1) {
    float xmf = 0.5f * (xl + xr) - x0floor;
    acc[x0i] += d - d * xmf;
    if (x0i + 1 <= w) acc[x0i + 1] += d * xmf;
  } else {
    float s = 1.0f / (xr - xl);
    float x0f = xl - x0floor;
    float a0 = 0.5f * s * (1.0f - x0f) * (1.0f - x0f);
    float x1f = xr - x1ceil + 1.0f;
    float am = 0.5f * s * x1f * x1f;
    acc[x0i] += d * a0;
    if (x1i == x0i + 2) {
      acc[x0i + 1] += d * (1.0f - a0 - am);
    } else {
      float a1 = s * (1.5f - x0f);
      acc[x0i + 1] += d * (a1 - a0);
      for (int xi = x0i + 2; xi < x1i - 1; xi++) acc[xi] += d * s;
      float a2 = a1 + (float)(x1i - x0i - 3) * s;
      acc[x1i - 1] += d * (1.0f - a2 - am);
    }
    if (x1i <= w) acc[x1i] += d * am;
  }
}
void fo_rasterize_lines(const float* lines, int n, int w, int h, uint8_t* out_rgba) {
  float* acc = (float*)malloc((size_t)(w + 2) * sizeof(float));
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w + 2; x++) acc[x] = 0.0f;
    for (int i = 0; i < n; i++) raster_row_line(acc, w, y, lines[4 * i], lines[4 * i + 1], lines[4 * i + 2], lines[4 * i + 3]);
    float sum = 0.0f;
    for (int x = 0; x < w; x++) {
      sum += acc[x];
      float c = fabsf(sum);
      if (c > 1.0f) c = 1.0f;
      uint8_t v = (uint8_t)(c * 255.0f + 0.5f);
      uint8_t* p = out_rgba + ((size_t)y * w + x) * 4;
      p[0] = p[1] = p[2] = p[3] = v;
    }
  }
  free(acc);
}
int fo_rasterize_outline(const float* segs, int n, int w, int h, uint8_t* out_rgba) {
  int m = fo_flatten_outline(segs, n, NULL, 0);
  float* lines = (float*)malloc((size_t)(m > 0 ? m : 1) * 4 * sizeof(float));
  fo_flatten_outline(segs, n, lines, m);
  fo_rasterize_lines(lines, m, w, h, out_rgba);
  free(lines);
  return m;
}
int fo_put_glyph_image(FoCtx* c, int64_t key, int w, int h, const uint8_t* rgba, unsigned flags, int out_rect[4]);
/* generateGlyph's job (common/fontglyphs.nim:61-106) with the rasteriser above in pixie's place: outline -> coverage -> (LCD) -> atlas */
int fo_put_glyph_outline(FoCtx* c, int64_t key, int w, int h, const float* segs, int n, unsigned flags, int out_rect[4]) {
  uint8_t* img = (uint8_t*)malloc((size_t)w * h * 4);
  fo_rasterize_outline(segs, n, w, h, img);
  int rc = fo_put_glyph_image(c, key, w, h, img, flags, out_rect);
  free(img);
  return rc;
}

/* applyLcdFilter: common/textrasters/pixie_raster.nim:12-43 (FreeType's default 5-tap LCD filter, horizontal, columns clamped) */
void fo_lcd_filter(const uint8_t* src, uint8_t* dst, int w, int h) {
  static const int wt[5] = {8, 77, 86, 77, 8};
  int maxx = w - 1;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int sum[4] = {0, 0, 0, 0};
      for (int i = 0; i < 5; i++) {
        int sx = x + i - 2;
        sx = sx < 0 ? 0 : (sx > maxx ? maxx : sx);
        const uint8_t* p = src + ((size_t)y * w + sx) * 4;
        for (int k = 0; k < 4; k++) sum[k] += (int)p[k] * wt[i];
      }
      for (int k = 0; k < 4; k++) dst[((size_t)y * w + x) * 4 + k] = (uint8_t)((sum[k] + 128) >> 8);
    }
}
/* renderPixieGlyph's tail (pixie_raster.nim:83-91): optional LCD filter, then loadGlyphImage -> putImage */
int fo_put_glyph_image(FoCtx* c, int64_t key, int w, int h, const uint8_t* rgba, unsigned flags, int out_rect[4]) {
  if (!(flags & 1u)) return fo_put_image(c, key, w, h, rgba, out_rect);
  uint8_t* f = (uint8_t*)malloc((size_t)w * h * 4);
  fo_lcd_filter(rgba, f, w, h);
  int rc = fo_put_image(c, key, w, h, f, out_rect);
  free(f);
  return rc;
}
/* pixie Image.minifyBy2 (see the header of this file for how its arithmetic is pinned): dst is ((w + 1) / 2) x ((h + 1) / 2) */
void fo_minify_by2(const uint8_t* src, int w, int h, uint8_t* dst) {
  const int ew = w / 2, eh = h / 2, nw = (w + 1) / 2, nh = (h + 1) / 2;
  for (int y = 0; y < nh; y++)
    for (int x = 0; x < nw; x++)
      for (int k = 0; k < 4; k++) {
        unsigned v;
        if (x < ew && y < eh) {
          v = (src[((size_t)(2 * y) * w + 2 * x) * 4 + k] + src[((size_t)(2 * y) * w + 2 * x + 1) * 4 + k] +
               src[((size_t)(2 * y + 1) * w + 2 * x + 1) * 4 + k] + src[((size_t)(2 * y + 1) * w + 2 * x) * 4 + k]) / 4;
        } else if (y < eh) { /* odd width: the last source column, two rows mixed, half coverage */
          unsigned a = src[((size_t)(2 * y) * w + (w - 1)) * 4 + k], b = src[((size_t)(2 * y + 1) * w + (w - 1)) * 4 + k];
          v = (((a * 127u + b * 128u) / 255u) * 128u) / 255u;
        } else if (x < ew) { /* odd height: the last source row */
          unsigned a = src[((size_t)(h - 1) * w + 2 * x) * 4 + k], b = src[((size_t)(h - 1) * w + 2 * x + 1) * 4 + k];
          v = (((a * 127u + b * 128u) / 255u) * 128u) / 255u;
        } else { /* both odd: the last texel, quarter coverage */
          v = (src[((size_t)(h - 1) * w + (w - 1)) * 4 + k] * 64u) / 255u;
        }
        dst[((size_t)y * nw + x) * 4 + k] = (uint8_t)v;
      }
}

int fo_put_image(FoCtx* c, int64_t key, int w, int h, const uint8_t* rgba, int out_rect[4]) {
  int S = c->atlas_size, rx, ry;
  if (place_rect(c, key, w, h, &rx, &ry) != 0) return -1;
  if (out_rect) { out_rect[0] = rx; out_rect[1] = ry; out_rect[2] = w; out_rect[3] = h; }
  /* updateSubImage with the minifyBy2 mip chain: textures.nim:106-119 */
  int cw = w, ch = h, lx = rx, ly = ry, level = 0;
  uint8_t* cur = (uint8_t*)malloc((size_t)w * h * 4);
  memcpy(cur, rgba, (size_t)w * h * 4);
  while (cw > 1 && ch > 1 && level < c->n_mips) {
    int LS = S >> level;
    for (int yy = 0; yy < ch; yy++)
      for (int xx = 0; xx < cw; xx++) {
        int tx = lx + xx, ty = ly + yy;
        if (tx >= 0 && ty >= 0 && tx < LS && ty < LS) memcpy(c->atlas[level] + ((size_t)ty * LS + tx) * 4, cur + ((size_t)yy * cw + xx) * 4, 4);
      }
    int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
    uint8_t* nxt = (uint8_t*)malloc((size_t)nw * nh * 4);
    fo_minify_by2(cur, cw, ch, nxt);
    free(cur);
    cur = nxt; cw = nw; ch = nh; lx /= 2; ly /= 2; level++;
  }
  free(cur);
  return 0;
}

/* Flippy container (common/formatflippy.nim:77-149): "flip", u32 version 1, then per mip "mip!", u32 w, u32 h, u32 zlen and a
 * snappy block (third-party supersnappy; the raw-snappy format is public: varint length, then literal / copy tags) of straight
 * RGBA8; on load every texel is converted to pixie's premultiplied ColorRGBX.  putFlippy (glcontext.nim:610-620) uploads mip l
 * at (x >> l, y >> l).  The premultiply rounding is pixie's (floor(c*a/255)); parity unpinned for 0 < a < 255. */
static int snappy_uncompress(const uint8_t* in, size_t n, uint8_t** out, size_t* out_n) {
  size_t i = 0, len = 0;
  int shift = 0;
  for (;;) {
    if (i >= n) return -1;
    uint8_t c = in[i++];
    len |= (size_t)(c & 0x7f) << shift;
    if (c < 0x80) break;
    shift += 7;
    if (shift > 35) return -1;
  }
  uint8_t* o = (uint8_t*)malloc(len ? len : 1);
  size_t w = 0;
  while (i < n) {
    uint8_t tag = in[i++];
    int t = tag & 3;
    if (t == 0) {
      size_t l = tag >> 2;
      if (l < 60) l += 1;
      else {
        int nb = (int)l - 59;
        if (i + nb > n) { free(o); return -1; }
        l = 0;
        for (int k = 0; k < nb; k++) l |= (size_t)in[i + k] << (8 * k);
        l += 1;
        i += nb;
      }
      if (i + l > n || w + l > len) { free(o); return -1; }
      memcpy(o + w, in + i, l);
      w += l; i += l;
    } else {
      size_t l, off;
      if (t == 1) { if (i + 1 > n) { free(o); return -1; } l = ((tag >> 2) & 7) + 4; off = ((size_t)(tag >> 5) << 8) | in[i]; i += 1; }
      else if (t == 2) { if (i + 2 > n) { free(o); return -1; } l = (tag >> 2) + 1; off = in[i] | ((size_t)in[i + 1] << 8); i += 2; }
      else { if (i + 4 > n) { free(o); return -1; } l = (tag >> 2) + 1; off = in[i] | ((size_t)in[i + 1] << 8) | ((size_t)in[i + 2] << 16) | ((size_t)in[i + 3] << 24); i += 4; }
      if (off == 0 || off > w || w + l > len) { free(o); return -1; }
      for (size_t k = 0; k < l; k++) { o[w] = o[w - off]; w++; }
    }
  }
  if (w != len) { free(o); return -1; }
  *out = o; *out_n = len;
  return 0;
}
static uint32_t rd_u32(const uint8_t* p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static int place_rect(FoCtx* c, int64_t key, int w, int h, int* ox, int* oy);
int fo_put_flippy(FoCtx* c, int64_t key, const uint8_t* data, size_t n, int out_rect[4]) {
  if (n < 8 || memcmp(data, "flip", 4) != 0 || rd_u32(data + 4) != 1) return -2;
  size_t i = 8;
  int level = 0, rx = 0, ry = 0, S = c->atlas_size;
  while (i < n) {
    if (i + 16 > n || memcmp(data + i, "mip!", 4) != 0) return -2;
    int w = (int)rd_u32(data + i + 4), h = (int)rd_u32(data + i + 8);
    size_t z = rd_u32(data + i + 12);
    i += 16;
    if (i + z > n) return -2;
    uint8_t* px; size_t pn;
    if (snappy_uncompress(data + i, z, &px, &pn) != 0 || pn != (size_t)w * h * 4) return -2;
    i += z;
    for (size_t k = 0; k < (size_t)w * h; k++) { /* ColorRGBA -> ColorRGBX */
      unsigned a = px[4 * k + 3];
      px[4 * k + 0] = (uint8_t)((px[4 * k + 0] * a) / 255);
      px[4 * k + 1] = (uint8_t)((px[4 * k + 1] * a) / 255);
      px[4 * k + 2] = (uint8_t)((px[4 * k + 2] * a) / 255);
    }
    if (level == 0) {
      if (place_rect(c, key, w, h, &rx, &ry) != 0) { free(px); return -1; }
      if (out_rect) { out_rect[0] = rx; out_rect[1] = ry; out_rect[2] = w; out_rect[3] = h; }
    }
    if (level < c->n_mips) {
      int LS = S >> level, lx = rx >> level, ly = ry >> level;
      for (int yy = 0; yy < h; yy++)
        for (int xx = 0; xx < w; xx++) {
          int tx = lx + xx, ty = ly + yy;
          if (tx >= 0 && ty >= 0 && tx < LS && ty < LS) memcpy(c->atlas[level] + ((size_t)ty * LS + tx) * 4, px + ((size_t)yy * w + xx) * 4, 4);
        }
    }
    free(px);
    level++;
  }
  return level > 0 ? 0 : -2;
}

static void draw_uv_quad(FoCtx* c, float ax, float ay, float tx, float ty, v2 uv_at, v2 uv_to, const FoColor colors[4],
                         int mode_word, v4 params, float f0, float f1) {
  Quad q;
  memset(&q, 0, sizeof q);
  q.pos[0] = ceil_xf(c, ax, ty); q.pos[1] = ceil_xf(c, tx, ty); q.pos[2] = ceil_xf(c, tx, ay); q.pos[3] = ceil_xf(c, ax, ay);
  q.uv[0].x = uv_at.x; q.uv[0].y = uv_to.y; q.uv[1].x = uv_to.x; q.uv[1].y = uv_to.y;
  q.uv[2].x = uv_to.x; q.uv[2].y = uv_at.y; q.uv[3].x = uv_at.x; q.uv[3].y = uv_at.y;
  for (int i = 0; i < 4; i++) q.col[i] = col_to_v4(colors[i]);
  q.params = params;
  q.mode_word = mode_word;
  q.factor0 = f0; q.factor1 = f1;
  q.subpixel_shift = active_subpixel_shift(c);
  draw_quad(c, &q);
}

/* drawImage(imageId, pos, colors, size, flipY): glcontext.nim:1350-1367, drawUvRect :1236-1302 */
void fo_draw_image(FoCtx* c, int64_t key, const float pos[2], const FoColor colors[4], const float size[2], int flip_y) {
  rec_open(c, "draw_image"); rec_i(c, key); rec_fv(c, pos, 2); rec_cols(c, colors); rec_fv(c, size, 2); rec_i(c, flip_y); rec_close(c);
  AtlasEntry* e = find_entry(c, key);
  if (!e) return; /* "missing image in context": warn + no-op glcontext.nim:1310-1315 */
  float S = (float)c->atlas_size;
  float dw = (size[0] > 0.0f && size[1] > 0.0f) ? size[0] : e->w * S;
  float dh = (size[0] > 0.0f && size[1] > 0.0f) ? size[1] : e->h * S;
  v2 at, to;
  if (flip_y) { at.x = e->x; at.y = e->y + e->h; to.x = e->x + e->w; to.y = e->y; }
  else { at.x = e->x; at.y = e->y; to.x = e->x + e->w; to.y = e->y + e->h; }
  v4 z = {0, 0, 0, 0};
  draw_uv_quad(c, pos[0], pos[1], pos[0] + dw, pos[1] + dh, at, to, colors, 0, z, 0.0f, 0.0f);
}

/* drawImageAdj(imageId, pos, color, size): glcontext.nim:1369-1381 -- the image's uv rect pulled in by two texels on every side */
void fo_draw_image_adj(FoCtx* c, int64_t key, const float pos[2], FoColor color, const float size[2]) {
  rec_open(c, "draw_image_adj"); rec_i(c, key); rec_fv(c, pos, 2); rec_col(c, color); rec_fv(c, size, 2); rec_close(c);
  AtlasEntry* e = find_entry(c, key);
  if (!e) return;
  float adj = 2.0f / (float)c->atlas_size;
  v2 at = {e->x + adj, e->y + adj}, to = {e->x + e->w - adj, e->y + e->h - adj};
  v4 z = {0, 0, 0, 0};
  FoColor cols[4] = {color, color, color, color};
  draw_uv_quad(c, pos[0], pos[1], pos[0] + size[0], pos[1] + size[1], at, to, cols, 0, z, 0.0f, 0.0f);
}

/* drawMsdfImage / drawMtsdfImage: glcontext.nim:1097-1155, drawUvRectAtlasSdf :1022-1093 */
void fo_draw_msdf(FoCtx* c, int64_t key, const float pos[2], FoColor color, const float size[2], float px_range,
                  float sd_threshold, float stroke_weight, int mtsdf, int flip_y) {
  rec_open(c, "draw_msdf"); rec_i(c, key); rec_fv(c, pos, 2); rec_col(c, color); rec_fv(c, size, 2); rec_f(c, px_range);
  rec_f(c, sd_threshold); rec_f(c, stroke_weight); rec_i(c, mtsdf); rec_i(c, flip_y); rec_close(c);
  AtlasEntry* e = find_entry(c, key);
  if (!e) return;
  v2 at, to;
  if (flip_y) { at.x = e->x; at.y = e->y + e->h; to.x = e->x + e->w; to.y = e->y; }
  else { at.x = e->x; at.y = e->y; to.x = e->x + e->w; to.y = e->y + e->h; }
  float sw = maxf(0.0f, stroke_weight);
  v4 params = {(float)c->atlas_size, sw, 0, 0};
  int mode = mtsdf ? (sw > 0.0f ? 16 : 14) : (sw > 0.0f ? 15 : 13);
  FoColor cols[4] = {color, color, color, color};
  draw_uv_quad(c, pos[0], pos[1], pos[0] + size[0], pos[1] + size[1], at, to, cols, mode, params, px_range, sd_threshold);
}

/* drawQuadraticBezierSdf: glcontext.nim:1619-1741 */
void fo_draw_quadratic_bezier_sdf(FoCtx* c, const float rect[4], const FoFill* fill, const float p0[2], const float p1[2],
                                  const float p2[2], float stroke_weight, int cap) {
  rec_open(c, "draw_quadratic_bezier_sdf");
  rec_fv(c, rect, 4);
  rec_printf(c, ",{\"kind\":%d,\"axis\":%d,\"start\":[%d,%d,%d,%d],\"mid\":[%d,%d,%d,%d],\"stop\":[%d,%d,%d,%d],\"mid_pos\":%d}", fill->kind,
             fill->axis, fill->start.r, fill->start.g, fill->start.b, fill->start.a, fill->mid.r, fill->mid.g, fill->mid.b, fill->mid.a,
             fill->stop.r, fill->stop.g, fill->stop.b, fill->stop.a, fill->mid_pos);
  rec_fv(c, p0, 2); rec_fv(c, p1, 2); rec_fv(c, p2, 2); rec_f(c, stroke_weight); rec_i(c, cap);
  rec_close(c);
  if (rect[2] <= 0.0f || rect[3] <= 0.0f || stroke_weight <= 0.0f) return;
  Quad q;
  memset(&q, 0, sizeof q);
  float x = rect[0], y = rect[1], w = rect[2], h = rect[3];
  q.params.x = w * 0.5f; q.params.y = h * 0.5f; q.params.z = p0[0]; q.params.w = p0[1];
  q.radii.x = p1[0]; q.radii.y = p1[1]; q.radii.z = p2[0]; q.radii.w = p2[1];
  q.pos[0] = ceil_xf(c, x, y + h); q.pos[1] = ceil_xf(c, x + w, y + h);
  q.pos[2] = ceil_xf(c, x + w, y); q.pos[3] = ceil_xf(c, x, y);
  q.uv[0].x = 0; q.uv[0].y = 1; q.uv[1].x = 1; q.uv[1].y = 1; q.uv[2].x = 1; q.uv[2].y = 0; q.uv[3].x = 0; q.uv[3].y = 0;
  int fill_mode = 0;
  FoColor cols[4];
  if (fill->kind == FO_FILL_LINEAR3) {
    fill_mode = 1 + fill->axis;
    cols[0] = cols[1] = cols[2] = cols[3] = fill->start;
    q.mid = col_to_v4(fill->mid); q.stop = col_to_v4(fill->stop);
  } else {
    fo_gradient_colors(fill, cols);
  }
  for (int i = 0; i < 4; i++) q.col[i] = col_to_v4(cols[i]);
  q.factor0 = stroke_weight;
  q.factor1 = fill_mode == 0 ? 0.0f : clampf(fill_mid_pos01(fill), 0.01f, 0.99f);
  int mode = cap == FO_CAP_BUTT ? 19 : (cap == FO_CAP_SQUARE ? 20 : 18); /* bezierStrokeSdfMode figbackend.nim:54-58 */
  q.mode_word = mode + fill_mode * 256;
  q.subpixel_shift = active_subpixel_shift(c);
  draw_quad(c, &q);
}

/* the 4x4 white "rect" atlas image both drawRect and drawFilledQuad sample (glcontext.nim:966-970,1411-1415) */
#define FO_RECT_IMAGE_KEY 0x7265637452454354LL
static AtlasEntry* rect_entry(FoCtx* c) {
  AtlasEntry* e = find_entry(c, FO_RECT_IMAGE_KEY);
  if (!e) {
    uint8_t white[4 * 4 * 4];
    memset(white, 255, sizeof white);
    fo_put_image(c, FO_RECT_IMAGE_KEY, 4, 4, white, NULL);
    e = find_entry(c, FO_RECT_IMAGE_KEY);
  }
  return e;
}
/* drawFilledQuad: glcontext.nim:963-982 + drawQuad :908-961 */
void fo_draw_filled_quad(FoCtx* c, const float verts[8], const FoColor colors[4]) {
  rec_open(c, "draw_filled_quad"); rec_fv(c, verts, 8); rec_cols(c, colors); rec_close(c);
  AtlasEntry* e = rect_entry(c);
  if (!e) return;
  Quad q;
  memset(&q, 0, sizeof q);
  float u = e->x + e->w / 2.0f, v = e->y + e->h / 2.0f;
  for (int i = 0; i < 4; i++) {
    q.pos[i] = ceil_xf(c, verts[2 * i], verts[2 * i + 1]);
    q.uv[i].x = u; q.uv[i].y = v;
    q.col[i] = col_to_v4(colors[i]);
  }
  q.mode_word = 0;
  q.subpixel_shift = active_subpixel_shift(c);
  draw_quad(c, &q);
}
/* drawRect: glcontext.nim:1410-1426 */
void fo_draw_rect(FoCtx* c, const float rect[4], FoColor color) {
  rec_open(c, "draw_rect"); rec_fv(c, rect, 4); rec_col(c, color); rec_close(c);
  AtlasEntry* e = rect_entry(c);
  if (!e) return;
  v2 uvc = {e->x + e->w / 2.0f, e->y + e->h / 2.0f};
  FoColor cols[4] = {color, color, color, color};
  v4 z = {0, 0, 0, 0};
  draw_uv_quad(c, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], uvc, uvc, cols, 0, z, 0.0f, 0.0f);
}

/* ------------------------------------------------------------------ masks: glcontext.nim:1873-1949 */
void fo_begin_mask(FoCtx* c, const float rect[4], const float rx[4], const float ry[4]) {
  rec_open(c, "begin_mask"); rec_fv(c, rect, 4); rec_fv(c, rx, 4); rec_fv(c, ry, 4); rec_close(c);
  int was = c->rec.on;
  c->rec.on = 0;
  c->mask_begun = 1;
  c->mask_write++;
  if (c->mask_write >= FO_MAX_MASKS) { fprintf(stderr, "figdraw_oracle: mask stack overflow\n"); abort(); }
  if (!c->mask[c->mask_write]) c->mask[c->mask_write] = (uint8_t*)malloc((size_t)c->W * c->H);
  memset(c->mask[c->mask_write], 0, (size_t)c->W * c->H);
  FoColor red = {255, 0, 0, 255}, zero = {0, 0, 0, 0};
  FoColor cols[4] = {red, red, red, red};
  float shape[2] = {0, 0};
  fo_draw_rounded_rect_sdf(c, rect, cols, rx, ry, 3, 4.0f, 0.0f, shape, 0, zero, zero, 0.5f);
  c->rec.on = was;
}
void fo_end_mask(FoCtx* c) { rec_open(c, "end_mask"); rec_close(c); c->mask_begun = 0; }
void fo_pop_mask(FoCtx* c) { rec_open(c, "pop_mask"); rec_close(c); c->mask_write--; }

/* makeRectMask glcontext.nim:831-850, beginRectMask :1932-1943 */
void fo_begin_rect_mask(FoCtx* c, const float rect[4], const float rx[4], const float ry[4]) {
  rec_open(c, "begin_rect_mask"); rec_fv(c, rect, 4); rec_fv(c, rx, 4); rec_fv(c, ry, 4); rec_close(c);
  int was = c->rec.on;
  c->rec.on = 0;
  RectMask* rm = &c->rect_masks[c->n_rect_masks];
  if (c->n_rect_masks == 0 && rect[2] > 0.0f && rect[3] > 0.0f) {
    float hx = rect[2] * 0.5f, hy = rect[3] * 0.5f;
    float r4[4];
    int ellip;
    fo_rounded_radii_vec(rx, ry, hx, hy, r4, &ellip);
    Aff inv = aff_inverse(c->mat);
    rm->kind = 1;
    rm->params.x = rect[0] + hx; rm->params.y = rect[1] + hy; rm->params.z = hx; rm->params.w = hy;
    rm->radii.x = r4[0]; rm->radii.y = r4[1]; rm->radii.z = r4[2]; rm->radii.w = r4[3];
    rm->matx.x = inv.a; rm->matx.y = inv.c; rm->matx.z = inv.tx; rm->matx.w = 1.0f;
    rm->maty.x = inv.b; rm->maty.y = inv.d; rm->maty.z = inv.ty; rm->maty.w = ellip ? 1.0f : 0.0f;
  } else {
    fo_begin_mask(c, rect, rx, ry);
    fo_end_mask(c);
    rm->kind = 2;
  }
  c->n_rect_masks++;
  c->rec.on = was;
}
void fo_pop_rect_mask(FoCtx* c) {
  rec_open(c, "pop_rect_mask"); rec_close(c);
  int was = c->rec.on;
  c->rec.on = 0;
  if (c->n_rect_masks > 0) {
    RectMask rm = c->rect_masks[--c->n_rect_masks];
    if (rm.kind == 2) fo_pop_mask(c);
  }
  c->rec.on = was;
}

/* ------------------------------------------------------------------ blur: blur.frag:11-32, glcontext.nim:1743-1786 */
static void blur_pass(int W, int H, const uint8_t* src, uint8_t* dst, float blur_radius, int vertical) {
  float radius = clampf(blur_radius, 0.0f, 64.0f);
  float sigma = maxf(0.5f * radius, 0.5f);
  float step_px = maxf(radius / 8.0f, 1.0f);
  float tsx = vertical ? 0.0f : 1.0f / (float)W, tsy = vertical ? 1.0f / (float)H : 0.0f;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      /* uv of the full-screen triangle at the fragment centre */
      float u = ((float)x + 0.5f) / (float)W, v = ((float)y + 0.5f) / (float)H;
      float acc[4] = {0, 0, 0, 0}, wsum = 0.0f;
      if (radius <= 0.5f) {
        memcpy(dst + ((size_t)y * W + x) * 4, src + ((size_t)y * W + x) * 4, 4);
        continue;
      }
      for (int i = -8; i <= 8; i++) {
        float xx = (float)i * step_px;
        float w = expf(-0.5f * (xx * xx) / (sigma * sigma));
        float su = u + tsx * xx, sv = v + tsy * xx;
        float t[4];
        bilinear_rgba8_clamp(src, W, H, su * (float)W - 0.5f, sv * (float)H - 0.5f, t);
        for (int k = 0; k < 4; k++) acc[k] += t[k] * w;
        wsum += w;
      }
      float d = maxf(wsum, 1e-5f);
      uint8_t* o = dst + ((size_t)y * W + x) * 4;
      for (int k = 0; k < 4; k++) o[k] = to_unorm8(acc[k] / d);
    }
}
void fo_blur_image(int w, int h, const uint8_t* src, uint8_t* dst, float radius) {
  if (radius <= 0.5f) { memcpy(dst, src, (size_t)w * h * 4); return; }
  uint8_t* tmp = (uint8_t*)malloc((size_t)w * h * 4);
  blur_pass(w, h, src, tmp, radius, 0);
  blur_pass(w, h, tmp, dst, radius, 1);
  free(tmp);
}

/* drawBackdropBlur: glcontext.nim:1788-1841 */
void fo_draw_backdrop_blur(FoCtx* c, const float rect[4], const float rx[4], const float ry[4], float blur_radius) {
  rec_open(c, "draw_backdrop_blur"); rec_fv(c, rect, 4); rec_fv(c, rx, 4); rec_fv(c, ry, 4); rec_f(c, blur_radius); rec_close(c);
  if (blur_radius <= 0.0f || rect[2] <= 0.0f || rect[3] <= 0.0f) return;
  int was = c->rec.on;
  c->rec.on = 0;
  memcpy(c->backdrop, c->fb, (size_t)c->W * c->H * 4); /* glCopyTexSubImage2D of the whole frame */
  if (blur_radius > 0.5f) {
    blur_pass(c->W, c->H, c->backdrop, c->backdrop_tmp, blur_radius, 0);
    blur_pass(c->W, c->H, c->backdrop_tmp, c->backdrop, blur_radius, 1);
  }
  FoColor white = {255, 255, 255, 255}, zero = {0, 0, 0, 0};
  FoColor cols[4] = {white, white, white, white};
  float shape[2] = {0, 0};
  fo_draw_rounded_rect_sdf(c, rect, cols, rx, ry, 17, blur_radius, 0.0f, shape, 0, zero, zero, 0.5f);
  c->rec.on = was;
}

/* readPixels + flipVertical: glcontext.nim:2094-2135 (top-down result) */
int fo_read_pixels(FoCtx* c, int x, int y, int w, int h, uint8_t* out) {
  if (!c->fb) return -1;
  if (w <= 0 || h <= 0) { x = 0; y = 0; w = c->W; h = c->H; }
  if (x < 0 || y < 0 || x + w > c->W || y + h > c->H) return -1;
  for (int r = 0; r < h; r++) memcpy(out + (size_t)r * w * 4, c->fb + ((size_t)(y + r) * c->W + x) * 4, (size_t)w * 4);
  return 0;
}
int fo_read_mask(FoCtx* c, int level, uint8_t* out) {
  if (level <= 0 || level >= FO_MAX_MASKS || !c->mask[level]) return -1;
  memcpy(out, c->mask[level], (size_t)c->W * c->H);
  return 0;
}

/* ================================================================== L2: figrender.nim */
static float scaled(const FoCtx* c, float v) { return v * c->ui_scale; } /* common/shared.nim:94-95 */

static void node_radii(const FoCtx* c, const FoFig* n, float rx[4], float ry[4]) { /* resolvedCorners+scaledCorners figrender.nim:549-571 */
  for (int i = 0; i < 4; i++) {
    rx[i] = scaled(c, (float)n->corners[i]);
    ry[i] = (n->flags & FO_NF_ELLIPTICAL_CORNERS) ? scaled(c, (float)n->corner_radii_y[i]) : rx[i];
  }
}
static void node_box(const FoCtx* c, const FoFig* n, float b[4]) { for (int i = 0; i < 4; i++) b[i] = n->box[i] * c->ui_scale; }

/* renderDropShadows figrender.nim:654-689 */
static void render_drop_shadows(FoCtx* c, const FoFig* n) {
  for (int s = 0; s < 4; s++) {
    const FoShadow* sh = &n->shadows[s];
    if (sh->style != FO_SHADOW_DROP) continue;
    if (sh->blur <= 0.0f && sh->spread <= 0.0f) continue;
    if (fill_alpha_max(&sh->fill) == 0) continue;
    float box[4];
    node_box(c, n, box);
    float sx = scaled(c, sh->x), sy = scaled(c, sh->y), sb = scaled(c, sh->blur), ss = scaled(c, sh->spread);
    float blur_pad = nim_round(1.5f * sb);
    float pad = maxf(nim_round(ss) + blur_pad, 0.0f);
    float srx = box[0] + sx, sry = box[1] + sy, srw = box[2], srh = box[3];
    float quad[4] = {srx - pad, sry - pad, srw + 2.0f * pad, srh + 2.0f * pad};
    float shape[2] = {srw, srh};
    float rx[4], ry[4];
    node_radii(c, n, rx, ry);
    fo_draw_rounded_rect_fill(c, quad, &sh->fill, rx, ry, 7, sb, ss, shape);
  }
}
/* renderInnerShadows figrender.nim:716-744 */
static void render_inner_shadows(FoCtx* c, const FoFig* n) {
  for (int s = 0; s < 4; s++) {
    const FoShadow* sh = &n->shadows[s];
    if (sh->style != FO_SHADOW_INNER) continue;
    if (sh->blur <= 0.0f && sh->spread <= 0.0f) continue;
    if (fill_alpha_max(&sh->fill) == 0) continue;
    float box[4];
    node_box(c, n, box);
    float off[2] = {scaled(c, sh->x), scaled(c, sh->y)};
    float rx[4], ry[4];
    node_radii(c, n, rx, ry);
    fo_draw_rounded_rect_fill(c, box, &sh->fill, rx, ry, 9, scaled(c, sh->blur), scaled(c, sh->spread), off);
  }
}
/* renderRoundedShapeScaledCorners figrender.nim:806-873 */
static void render_rounded_shape(FoCtx* c, const float box_unscaled[4], const FoFill* fill, const FoStroke* stroke,
                                 const float rx[4], const float ry[4]) {
  float box[4];
  for (int i = 0; i < 4; i++) box[i] = box_unscaled[i] * c->ui_scale;
  float shape[2] = {0, 0};
  int has_gradient = (fill->kind == FO_FILL_LINEAR2 || fill->kind == FO_FILL_LINEAR3) && fill_alpha_max(fill) > 0;
  if (has_gradient) {
    fo_draw_rounded_rect_fill(c, box, fill, rx, ry, 3, 4.0f, 0.0f, shape);
  } else if (fill_alpha_max(fill) > 0) {
    /* fillCenterColor -> Color -> rgba(): an RGBA8 round trip, identity for flColor */
    FoFill solid = *fill;
    solid.kind = FO_FILL_COLOR;
    solid.start = sample_color(fill, 0.5f);
    fo_draw_rounded_rect_fill(c, box, &solid, rx, ry, 3, 4.0f, 0.0f, shape);
  }
  if (stroke && fill_alpha_max(&stroke->fill) > 0 && stroke->weight > 0.0f)
    fo_draw_rounded_rect_fill(c, box, &stroke->fill, rx, ry, 12, scaled(c, stroke->weight), 0.0f, shape);
}


/* ------------------------------------------------------------------ L2: nkDrawable (figrender.nim:910-1667) */
typedef struct { v2 p0, p1, p2; } QSpan;
static v2 v2mk(float x, float y) { v2 r = {x, y}; return r; }
static v2 v2add(v2 a, v2 b) { return v2mk(a.x + b.x, a.y + b.y); }
static v2 v2sub(v2 a, v2 b) { return v2mk(a.x - b.x, a.y - b.y); }
static v2 v2mul(v2 a, float s) { return v2mk(a.x * s, a.y * s); }
static float v2len(v2 v) { return sqrtf(v.x * v.x + v.y * v.y); }                  /* vectorLength :910-911 */
static v2 normalized_or(v2 v, v2 fb) { float l = v2len(v); return l <= 0.000001f ? fb : v2mk(v.x / l, v.y / l); } /* :913-918 */
static v2 normal_left(v2 d) { return v2mk(-d.y, d.x); }
static float cross2f(v2 a, v2 b) { return a.x * b.y - a.y * b.x; }
static float descaled(const FoCtx* c, float v) { return v / c->ui_scale; }
static uint16_t radius_corner(float r) { /* :797-802 */
  if (r <= 0.0f) return 0;
  if (r >= 65535.0f) return 65535;
  return (uint16_t)nim_round(r);
}

typedef struct { FoCtx* c; const FoScene* sc; } DrawEnv;

static void shape_u16(FoCtx* c, const float box[4], const FoFill* fill, const FoStroke* stroke, const uint16_t corners[4]) {
  float rx[4];
  for (int i = 0; i < 4; i++) rx[i] = scaled(c, (float)corners[i]);
  render_rounded_shape(c, box, fill, stroke, rx, rx);
}
static void draw_stroke_cap(FoCtx* c, v2 center, float radius, const FoFill* fill) { /* renderDrawableStrokeCap :993-1005 */
  if (radius <= 0.0f || fill_alpha_max(fill) == 0) return;
  float d = radius * 2.0f;
  float box[4] = {center.x - radius, center.y - radius, d, d};
  uint16_t rc = radius_corner(radius);
  uint16_t corners[4] = {rc, rc, rc, rc};
  shape_u16(c, box, fill, NULL, corners);
}
static void draw_line(FoCtx* c, v2 origin, v2 pa, v2 pb, const FoStroke* stroke) { /* renderDrawableLine :943-991 */
  float weight = maxf(0.0f, stroke->weight);
  if (weight <= 0.0f || fill_alpha_max(&stroke->fill) == 0) return;
  v2 a = v2add(origin, pa), b = v2add(origin, pb), delta = v2sub(b, a);
  float length = v2len(delta);
  if (length <= 0.0f) return;
  int cap = stroke->cap == FO_CAP_AUTO ? FO_CAP_BUTT : stroke->cap; /* resolveLineCap */
  float cap_radius = weight * 0.5f;
  v2 dir = v2mk(delta.x / length, delta.y / length);
  v2 da = a, db = b;
  float dl = length;
  if (cap == FO_CAP_SQUARE) { da = v2sub(a, v2mul(dir, cap_radius)); db = v2add(b, v2mul(dir, cap_radius)); dl = length + weight; }
  v2 center = v2mk((da.x + db.x) / 2.0f, (da.y + db.y) / 2.0f);
  float box[4] = {center.x - dl / 2.0f, center.y - weight / 2.0f, dl, weight};
  float sb[4] = {box[0] * c->ui_scale, box[1] * c->ui_scale, box[2] * c->ui_scale, box[3] * c->ui_scale};
  float pvx = sb[0] + sb[2] / 2.0f, pvy = sb[1] + sb[3] / 2.0f;
  float angle = atan2f(delta.y, delta.x);
  fo_save_transform(c);
  fo_translate(c, pvx, pvy);
  fo_rotate(c, angle);
  fo_translate(c, -pvx, -pvy);
  uint16_t zero[4] = {0, 0, 0, 0};
  shape_u16(c, box, &stroke->fill, NULL, zero);
  fo_restore_transform(c);
  if (cap == FO_CAP_ROUND) { draw_stroke_cap(c, a, cap_radius, &stroke->fill); draw_stroke_cap(c, b, cap_radius, &stroke->fill); }
}
static void draw_endpoint_cap(FoCtx* c, v2 origin, v2 point, v2 tangent, float radius, const FoStroke* stroke, int cap, int is_start) { /* :1007-1040 */
  if (radius <= 0.0f || fill_alpha_max(&stroke->fill) == 0) return;
  if (cap == FO_CAP_ROUND) { draw_stroke_cap(c, v2add(origin, point), radius, &stroke->fill); return; }
  if (cap == FO_CAP_SQUARE) {
    v2 dir = normalized_or(tangent, v2mk(1.0f, 0.0f));
    v2 a = is_start ? v2sub(point, v2mul(dir, radius)) : point;
    v2 b = is_start ? point : v2add(point, v2mul(dir, radius));
    FoStroke s2 = *stroke;
    s2.cap = FO_CAP_BUTT;
    draw_line(c, origin, a, b, &s2);
  }
}
static void draw_filled_quad_l2(FoCtx* c, const v2 verts[4], const FoFill* fill) { /* renderDrawableFilledQuad :1050-1058 */
  if (fill_alpha_max(fill) == 0) return;
  FoColor k = sample_color(fill, 0.5f);
  FoColor cols[4] = {k, k, k, k};
  float vv[8];
  for (int i = 0; i < 4; i++) { vv[2 * i] = verts[i].x * c->ui_scale; vv[2 * i + 1] = verts[i].y * c->ui_scale; }
  fo_draw_filled_quad(c, vv, cols);
}
static void draw_stroke_join(FoCtx* c, v2 origin, v2 point, v2 in_t, v2 out_t, float radius, const FoFill* fill, int join) { /* :1060-1109 */
  if (radius <= 0.0f || fill_alpha_max(fill) == 0) return;
  if (join == FO_JOIN_ROUND) { draw_stroke_cap(c, v2add(origin, point), radius, fill); return; }
  if (join != FO_JOIN_BEVEL && join != FO_JOIN_MITER) return;
  v2 incoming = normalized_or(in_t, v2mk(1.0f, 0.0f));
  v2 outgoing = normalized_or(out_t, incoming);
  float turn = cross2f(incoming, outgoing);
  if (fabsf(turn) <= 0.0001f) return;
  float side = turn > 0.0f ? -1.0f : 1.0f;
  v2 in_outer = v2add(point, v2mul(normal_left(incoming), radius * side));
  v2 out_outer = v2add(point, v2mul(normal_left(outgoing), radius * side));
  if (join == FO_JOIN_MITER) {
    float denom = cross2f(incoming, outgoing); /* lineIntersection :1042-1048 */
    if (fabsf(denom) > 0.000001f) {
      float t = cross2f(v2sub(out_outer, in_outer), outgoing) / denom;
      v2 miter = v2add(in_outer, v2mul(incoming, t));
      if (v2len(v2sub(miter, point)) <= radius * 4.0f) {
        v2 q[4] = {v2add(origin, point), v2add(origin, in_outer), v2add(origin, miter), v2add(origin, out_outer)};
        draw_filled_quad_l2(c, q, fill);
        return;
      }
    }
  }
  v2 q[4] = {v2add(origin, point), v2add(origin, in_outer), v2add(origin, out_outer), v2add(origin, out_outer)};
  draw_filled_quad_l2(c, q, fill);
}
static v2 bezier_point(const float* ctrl, int n, float t) { /* bezierPoint :1139-1153 */
  v2 work[32];
  if (n <= 0) return v2mk(0, 0);
  if (n > 32) n = 32;
  for (int i = 0; i < n; i++) work[i] = v2mk(ctrl[2 * i], ctrl[2 * i + 1]);
  for (int count = n; count > 1; count--)
    for (int i = 0; i < count - 1; i++) work[i] = v2add(v2mul(work[i], 1.0f - t), v2mul(work[i + 1], t));
  return work[0];
}
static v2 quadratic_point(v2 p0, v2 p1, v2 p2, float t) { /* :1155-1157 */
  float it = 1.0f - t;
  return v2add(v2add(v2mul(p0, it * it), v2mul(p1, 2.0f * it * t)), v2mul(p2, t * t));
}
static void quadratic_bounds(v2 p0, v2 p1, v2 p2, float pad, float out[4]) { /* :1177-1202 */
  v2 mn = v2mk(minf(p0.x, p2.x), minf(p0.y, p2.y)), mx = v2mk(maxf(p0.x, p2.x), maxf(p0.y, p2.y));
  float dx = p0.x - 2.0f * p1.x + p2.x;
  if (fabsf(dx) > 0.000001f) {
    float t = (p0.x - p1.x) / dx;
    if (t > 0.0f && t < 1.0f) { v2 q = quadratic_point(p0, p1, p2, t); mn.x = minf(mn.x, q.x); mn.y = minf(mn.y, q.y); mx.x = maxf(mx.x, q.x); mx.y = maxf(mx.y, q.y); }
  }
  float dy = p0.y - 2.0f * p1.y + p2.y;
  if (fabsf(dy) > 0.000001f) {
    float t = (p0.y - p1.y) / dy;
    if (t > 0.0f && t < 1.0f) { v2 q = quadratic_point(p0, p1, p2, t); mn.x = minf(mn.x, q.x); mn.y = minf(mn.y, q.y); mx.x = maxf(mx.x, q.x); mx.y = maxf(mx.y, q.y); }
  }
  out[0] = mn.x - pad; out[1] = mn.y - pad; out[2] = mx.x - mn.x + pad * 2.0f; out[3] = mx.y - mn.y + pad * 2.0f;
}
static v2 span_start_tangent(const QSpan* s) { return normalized_or(v2sub(s->p1, s->p0), normalized_or(v2sub(s->p2, s->p0), v2mk(1.0f, 0.0f))); }
static v2 span_end_tangent(const QSpan* s) { return normalized_or(v2sub(s->p2, s->p1), normalized_or(v2sub(s->p2, s->p0), v2mk(1.0f, 0.0f))); }
static QSpan bezier_span(const float* ctrl, int n, float t0, float t2) { /* bezierQuadraticSpan :1238-1247 */
  float tm = (t0 + t2) * 0.5f;
  QSpan s;
  s.p0 = bezier_point(ctrl, n, t0);
  v2 pm = bezier_point(ctrl, n, tm);
  s.p2 = bezier_point(ctrl, n, t2);
  s.p1 = v2sub(v2mul(pm, 2.0f), v2mul(v2add(s.p0, s.p2), 0.5f));
  return s;
}
#define FO_MAX_ADAPTIVE_STEPS 192 /* max(DefaultDrawableBezierSteps*4, 64) :1171 */
#define FO_MAX_CURVE_DEPTH 8
static void adaptive_spans(const FoCtx* c, const float* ctrl, int n, float t0, float t2, int depth, QSpan* spans, int* ns) { /* :1267-1282 */
  QSpan s = bezier_span(ctrl, n, t0, t2);
  float err = 0.0f;
  const float lt[2] = {0.25f, 0.75f};
  for (int i = 0; i < 2; i++) { /* quadraticApproxErrorPx :1256-1265 */
    float t = t0 + (t2 - t0) * lt[i];
    v2 actual = bezier_point(ctrl, n, t), approx = quadratic_point(s.p0, s.p1, s.p2, lt[i]);
    err = maxf(err, v2len(v2mul(v2sub(actual, approx), c->ui_scale)));
  }
  if (err <= 0.5f || depth >= FO_MAX_CURVE_DEPTH || *ns >= FO_MAX_ADAPTIVE_STEPS - 1) { spans[(*ns)++] = s; return; }
  float tm = (t0 + t2) * 0.5f;
  adaptive_spans(c, ctrl, n, t0, tm, depth + 1, spans, ns);
  adaptive_spans(c, ctrl, n, tm, t2, depth + 1, spans, ns);
}
static void draw_quadratic_sdf(FoCtx* c, v2 origin, v2 p0, v2 p1, v2 p2, const FoStroke* stroke, int cap) { /* renderDrawableQuadraticBezierSdf :1335-1376 */
  int rcap = cap == FO_CAP_AUTO ? (stroke->cap == FO_CAP_AUTO ? FO_CAP_ROUND : stroke->cap) : cap;
  if (fabsf(cross2f(v2sub(p1, p0), v2sub(p2, p1))) <= 0.0001f) { /* isFlatQuadratic :1165-1166 */
    FoStroke s2 = *stroke;
    s2.cap = rcap;
    draw_line(c, origin, p0, p2, &s2);
    return;
  }
  float sw = maxf(0.0f, stroke->weight);
  float padding = sw * 0.5f + descaled(c, 2.0f); /* DrawableSdfPaddingPx */
  v2 a = v2add(origin, p0), b = v2add(origin, p1), cc = v2add(origin, p2);
  float box[4];
  quadratic_bounds(a, b, cc, padding, box);
  if (box[2] <= 0.0f || box[3] <= 0.0f) return;
  v2 center = v2mk(box[0] + box[2] * 0.5f, box[1] + box[3] * 0.5f);
  float us = c->ui_scale;
  float la[2] = {(a.x - center.x) * us, (a.y - center.y) * us}, lb[2] = {(b.x - center.x) * us, (b.y - center.y) * us};
  float lc[2] = {(cc.x - center.x) * us, (cc.y - center.y) * us};
  float sbox[4] = {box[0] * us, box[1] * us, box[2] * us, box[3] * us};
  fo_draw_quadratic_bezier_sdf(c, sbox, &stroke->fill, la, lb, lc, sw * us, rcap);
}
static void draw_spans(FoCtx* c, v2 origin, const QSpan* spans, int n, const FoStroke* stroke) { /* :1412-1455, :1569-1604 */
  int cap = stroke->cap == FO_CAP_AUTO ? FO_CAP_ROUND : stroke->cap;
  int join = stroke->join == FO_JOIN_AUTO ? FO_JOIN_ROUND : stroke->join;
  int simple = cap == FO_CAP_ROUND && join == FO_JOIN_ROUND;
  int span_cap = simple ? FO_CAP_ROUND : FO_CAP_BUTT;
  float cap_radius = maxf(0.0f, stroke->weight) / 2.0f;
  for (int i = 0; i < n; i++) {
    draw_quadratic_sdf(c, origin, spans[i].p0, spans[i].p1, spans[i].p2, stroke, span_cap);
    if (!simple) {
      if (i == 0) draw_endpoint_cap(c, origin, spans[i].p0, span_start_tangent(&spans[i]), cap_radius, stroke, cap, 1);
      else draw_stroke_join(c, origin, spans[i].p0, span_end_tangent(&spans[i - 1]), span_start_tangent(&spans[i]), cap_radius, &stroke->fill, join);
      if (i == n - 1) draw_endpoint_cap(c, origin, spans[i].p2, span_end_tangent(&spans[i]), cap_radius, stroke, cap, 0);
    }
  }
}
static int explicit_steps(uint16_t steps, uint16_t node_steps) { /* :1204-1210 */
  if (steps != 0) return steps > 1 ? steps : 1;
  if (node_steps != 0) return node_steps > 1 ? node_steps : 1;
  return 0;
}
static void seg_points(const FoCtx* c, const float* ctrl, int n, float t0, float t2, int depth, v2* pts, int* np) { /* :1297-1313 */
  v2 p0 = bezier_point(ctrl, n, t0), p2 = bezier_point(ctrl, n, t2);
  float tm = (t0 + t2) * 0.5f;
  v2 pm = bezier_point(ctrl, n, tm);
  v2 P = v2mul(pm, c->ui_scale), A = v2mul(p0, c->ui_scale), B = v2mul(p2, c->ui_scale), ab = v2sub(B, A); /* distanceToLinePx */
  float denom = ab.x * ab.x + ab.y * ab.y, err;
  if (denom <= 0.000001f) err = v2len(v2sub(P, A));
  else {
    float h = clampf(((P.x - A.x) * ab.x + (P.y - A.y) * ab.y) / denom, 0.0f, 1.0f);
    err = v2len(v2sub(P, v2add(A, v2mul(ab, h))));
  }
  if (err <= 0.5f || depth >= FO_MAX_CURVE_DEPTH || *np >= FO_MAX_ADAPTIVE_STEPS) { pts[(*np)++] = p2; return; }
  seg_points(c, ctrl, n, t0, tm, depth + 1, pts, np);
  seg_points(c, ctrl, n, tm, t2, depth + 1, pts, np);
}
static void draw_bezier_segments(FoCtx* c, v2 origin, const float* ctrl, int n, uint16_t steps, const FoStroke* stroke, uint16_t node_steps) { /* :1378-1410 */
  if (n < 2 || stroke->weight <= 0.0f || fill_alpha_max(&stroke->fill) == 0) return;
  static v2 pts[FO_MAX_ADAPTIVE_STEPS + 70000];
  int np = 0, fixed = explicit_steps(steps, node_steps);
  pts[np++] = bezier_point(ctrl, n, 0.0f);
  if (fixed > 0) for (int s = 1; s <= fixed; s++) pts[np++] = bezier_point(ctrl, n, (float)s / (float)fixed);
  else seg_points(c, ctrl, n, 0.0f, 1.0f, 0, pts, &np);
  if (np < 2) return;
  int cap = stroke->cap == FO_CAP_AUTO ? FO_CAP_ROUND : stroke->cap;
  int join = stroke->join == FO_JOIN_AUTO ? FO_JOIN_ROUND : stroke->join;
  float cap_radius = maxf(0.0f, stroke->weight) / 2.0f;
  FoStroke seg = *stroke;
  seg.cap = FO_CAP_BUTT;
  v2 prev = pts[0], prev_t = v2mk(1.0f, 0.0f);
  for (int s = 1; s < np; s++) {
    v2 cur = pts[s], tan = v2sub(cur, prev);
    draw_line(c, origin, prev, cur, &seg);
    if (s == 1) draw_endpoint_cap(c, origin, prev, tan, cap_radius, stroke, cap, 1);
    else draw_stroke_join(c, origin, prev, prev_t, tan, cap_radius, &stroke->fill, join);
    if (s == np - 1) draw_endpoint_cap(c, origin, cur, tan, cap_radius, stroke, cap, 0);
    prev = cur;
    prev_t = tan;
  }
}
static void render_drawable_ops(FoCtx* c, const FoScene* sc, const FoFig* n) { /* renderDrawableOps :1627-1645 */
  v2 origin = v2mk(n->box[0], n->box[1]);
  const FoStroke* stroke = &n->draw_stroke;
  for (int oi = n->op_first; oi < n->op_first + n->op_count && oi < sc->n_ops; oi++) {
    const FoDrawOp* op = &sc->ops[oi];
    switch (op->kind) {
      case FO_DK_LINE: draw_line(c, origin, v2mk(op->v[0], op->v[1]), v2mk(op->v[2], op->v[3]), stroke); break;
      case FO_DK_CIRCLE: { /* :1111-1125 */
        float r = maxf(0.0f, op->v[2]);
        if (r <= 0.0f) break;
        float box[4] = {origin.x + op->v[0] - r, origin.y + op->v[1] - r, r * 2.0f, r * 2.0f};
        uint16_t rc = radius_corner(r);
        uint16_t corners[4] = {rc, rc, rc, rc};
        shape_u16(c, box, &n->fill, stroke, corners);
        break;
      }
      case FO_DK_RECTANGLE: { /* :1127-1131 */
        float box[4] = {origin.x + op->v[0], origin.y + op->v[1], op->v[2], op->v[3]};
        shape_u16(c, box, &n->fill, stroke, op->corners);
        break;
      }
      case FO_DK_ELLIPSE: { /* :1606-1625 */
        float rx0 = maxf(0.0f, op->v[2]), ry0 = maxf(0.0f, op->v[3]);
        if (rx0 <= 0.0f || ry0 <= 0.0f) break;
        float box[4] = {origin.x + op->v[0] - rx0, origin.y + op->v[1] - ry0, rx0 * 2.0f, ry0 * 2.0f};
        float rx[4], ry[4];
        for (int i = 0; i < 4; i++) { rx[i] = scaled(c, rx0); ry[i] = scaled(c, ry0); }
        render_rounded_shape(c, box, &n->fill, stroke, rx, ry);
        break;
      }
      case FO_DK_BEZIER: { /* renderDrawableBezier :1457-1486 */
        int nc = op->ctrl_count;
        const float* ctrl = sc->controls + 2 * op->ctrl_first;
        if (nc < 2 || stroke->weight <= 0.0f || fill_alpha_max(&stroke->fill) == 0) break;
        if (nc == 3) {
          draw_quadratic_sdf(c, origin, v2mk(ctrl[0], ctrl[1]), v2mk(ctrl[2], ctrl[3]), v2mk(ctrl[4], ctrl[5]), stroke,
                             stroke->cap == FO_CAP_AUTO ? FO_CAP_ROUND : stroke->cap);
        } else if (nc > 3) {
          static QSpan spans[70000];
          int ns = 0, fixed = explicit_steps(op->steps, n->draw_steps);
          if (fixed > 0) for (int st = 0; st < fixed; st++) spans[ns++] = bezier_span(ctrl, nc, (float)st / (float)fixed, (float)(st + 1) / (float)fixed);
          else adaptive_spans(c, ctrl, nc, 0.0f, 1.0f, 0, spans, &ns);
          draw_spans(c, origin, spans, ns, stroke);
        } else {
          draw_bezier_segments(c, origin, ctrl, nc, op->steps, stroke, n->draw_steps);
        }
        break;
      }
      case FO_DK_ARC: { /* renderDrawableArc :1606-1625, arcQuadraticSpan :1531-1546, adaptiveArcStepCount :1315-1328 */
        float radius = maxf(0.0f, op->v[2]), start = op->v[3], sweep = op->v[4];
        if (radius <= 0.0f || sweep == 0.0f || stroke->weight <= 0.0f || fill_alpha_max(&stroke->fill) == 0) break;
        int steps = explicit_steps(op->steps, n->draw_steps);
        if (steps <= 0) {
          float rpx = maxf(0.0f, scaled(c, radius)), asw = fabsf(sweep);
          if (rpx <= 0.0f || asw <= 0.0f) steps = 1;
          else {
            float cl = clampf(1.0f - 0.5f / rpx, -1.0f, 1.0f);
            float max_angle = maxf(0.01f, 2.0f * acosf(cl));
            int k = (int)ceilf(asw / max_angle);
            steps = k < 1 ? 1 : (k > FO_MAX_ADAPTIVE_STEPS ? FO_MAX_ADAPTIVE_STEPS : k);
          }
        }
        static QSpan spans[70000];
        v2 cen = v2mk(op->v[0], op->v[1]);
        for (int st = 0; st < steps; st++) {
          float t0 = (float)st / (float)steps, t2 = (float)(st + 1) / (float)steps, tm = (t0 + t2) * 0.5f;
          float a0 = start + sweep * t0, a2 = start + sweep * t2, am = start + sweep * tm;
          QSpan sp;
          sp.p0 = v2add(cen, v2mk(cosf(a0) * radius, sinf(a0) * radius));
          v2 pm = v2add(cen, v2mk(cosf(am) * radius, sinf(am) * radius));
          sp.p2 = v2add(cen, v2mk(cosf(a2) * radius, sinf(a2) * radius));
          sp.p1 = v2sub(v2mul(pm, 2.0f), v2mul(v2add(sp.p0, sp.p2), 0.5f));
          spans[st] = sp;
        }
        draw_spans(c, origin, spans, steps, stroke);
        break;
      }
      default: break;
    }
  }
}
static void render_drawable(FoCtx* c, const FoScene* sc, const FoFig* n) { /* renderDrawable :1647-1667 */
  if (n->draw_aa <= 0.0f || c->aa == n->draw_aa) { render_drawable_ops(c, sc, n); return; }
  float old = c->aa;
  fo_set_aa_factor(c, n->draw_aa);
  render_drawable_ops(c, sc, n);
  fo_set_aa_factor(c, old);
}

static void render_node(FoCtx* c, const FoScene* sc, const FoLayer* L, int idx);

static void draw_text_rect(FoCtx* c, float x, float y, float w, float h, const FoFill* fill) { /* figrender.nim:355-369, 444-452 */
  const float rect[4] = {scaled(c, x), scaled(c, y), scaled(c, w), scaled(c, h)};
  const float zero[4] = {0, 0, 0, 0}, shape[2] = {0, 0};
  fo_draw_rounded_rect_fill(c, rect, fill, zero, zero, 3 /* sdfModeClipAA */, 4.0f, 0.0f, shape);
}
static void render_text(FoCtx* c, const FoScene* sc, const FoFig* n) { /* renderText figrender.nim:417-497 */
  fo_save_transform(c);
  fo_translate(c, scaled(c, n->box[0]), scaled(c, n->box[1]));
  if (n->flags & FO_NF_INVERT_Y) {
    fo_translate(c, 0.0f, scaled(c, n->box[3]));
    fo_scale(c, 1.0f, -1.0f);
  }
  /* selection rectangles first (:435-452), then underline / strikethrough (:371-415), then the glyphs */
  for (int pass = 0; pass < 2; pass++) {
    for (int k = n->text_rect_first; k < n->text_rect_first + n->text_rect_count && k < sc->n_text_rects; k++) {
      const FoTextRect* tr = &sc->text_rects[k];
      if (tr->kind != pass) continue;
      if (pass == 0) {
        if (!(n->flags & FO_NF_SELECT_TEXT) || fill_alpha_max(&n->fill) == 0 || !(tr->h > 0.0f)) continue;
        draw_text_rect(c, tr->x, tr->y, maxf(tr->w, 1.0f), tr->h, &n->fill);
      } else {
        if (tr->w <= 0.0f || tr->h <= 0.0f) continue;
        draw_text_rect(c, tr->x, tr->y, tr->w, tr->h, &tr->fill);
      }
    }
  }
  for (int g = n->glyph_first; g < n->glyph_first + n->glyph_count && g < sc->n_glyphs; g++) {
    const FoGlyph* gl = &sc->glyphs[g];
    float pos[2] = {gl->x, gl->y}, size[2] = {0, 0};
    int64_t key = gl->image_id;
    float shift = gl->subpixel_shift;
    if (shift < 0.0f) { /* figrender.nim:464-471 */
      shift = 0.0f;
      if (c->subpixel_enabled) {
        const float snapped = floorf(pos[0]);
        const float frac = maxf(0.0f, minf(pos[0] - snapped, 0.999f));
        pos[0] = snapped;
        if (c->subpixel_variants && sc->glyph_variant_ids) { /* toGlyphVariantSubpixelStep common/fontglyphs.nim:50-52 */
          int step = (int)(frac * (float)FO_GLYPH_VARIANT_STEPS);
          if (step > FO_GLYPH_VARIANT_STEPS - 1) step = FO_GLYPH_VARIANT_STEPS - 1;
          key = sc->glyph_variant_ids[(size_t)g * FO_GLYPH_VARIANT_STEPS + step];
        } else {
          shift = frac;
        }
      }
    }
    fo_set_text_subpixel_shift(c, shift);
    fo_draw_image(c, key, pos, gl->colors, size, 0);
  }
  fo_set_text_subpixel_shift(c, 0.0f);
  fo_restore_transform(c);
}

/* render figrender.nim:1756-1839 (stage order fixed by renderStages :501-547) */
static void render_node(FoCtx* c, const FoScene* sc, const FoLayer* L, int idx) {
  const FoFig* n = &L->nodes[idx];
  if (n->flags & FO_NF_DISABLE_RENDER) return;
  float box[4];
  node_box(c, n, box);
  float rx[4], ry[4];
  node_radii(c, n, rx, ry);
  int did_rot = 0, did_xf = 0, did_clip = 0, did_rmask = 0;
  if (n->rotation != 0.0f) {
    did_rot = 1;
    fo_save_transform(c);
    float cx = box[0] + box[2] / 2.0f, cy = box[1] + box[3] / 2.0f;
    fo_translate(c, cx, cy);
    fo_rotate(c, n->rotation / 180.0f * 3.14159265358979323846f); /* node.rotation / 180 * PI in float32 */
    fo_translate(c, -cx, -cy);
  }
  if (n->kind == FO_NK_TRANSFORM) {
    did_xf = 1;
    fo_save_transform(c);
    if (n->translation[0] != 0.0f || n->translation[1] != 0.0f) fo_translate(c, scaled(c, n->translation[0]), scaled(c, n->translation[1]));
    if (n->use_matrix) fo_apply_transform(c, n->matrix);
  }
  if (n->kind == FO_NK_RECTANGLE) render_drop_shadows(c, n);
  if (n->flags & FO_NF_CLIP_CONTENT) {
    did_clip = 1;
    fo_begin_mask(c, box, rx, ry);
    fo_end_mask(c);
  }
  if (n->flags & FO_NF_RECT_MASK_CONTENT) {
    did_rmask = 1;
    fo_begin_rect_mask(c, box, rx, ry);
  }
  switch (n->kind) {
    case FO_NK_TEXT: render_text(c, sc, n); break;
    case FO_NK_DRAWABLE: render_drawable(c, sc, n); break;
    case FO_NK_RECTANGLE: render_rounded_shape(c, n->box, &n->fill, &n->stroke, rx, ry); break; /* renderBoxes :1669-1671 */
    case FO_NK_IMAGE: { /* renderImage :1673-1684 */
      if (n->image_id == 0) break;
      FoColor k = sample_color(&n->image_fill, 0.5f);
      FoColor cols[4] = {k, k, k, k};
      float pos[2] = {box[0], box[1]}, size[2] = {box[2], box[3]};
      fo_draw_image(c, n->image_id, pos, cols, size, (n->flags & FO_NF_INVERT_Y) != 0);
      break;
    }
    case FO_NK_MSDF_IMAGE:
    case FO_NK_MTSDF_IMAGE: { /* renderMsdfImage / renderMtsdfImage :1686-1732 */
      if (n->image_id == 0) break;
      float pr = n->px_range > 0.0f ? n->px_range : 4.0f;
      float th = (n->sd_threshold > 0.0f && n->sd_threshold < 1.0f) ? n->sd_threshold : 0.5f;
      float sw = scaled(c, maxf(0.0f, n->stroke_weight));
      float pos[2] = {box[0], box[1]}, size[2] = {box[2], box[3]};
      fo_draw_msdf(c, n->image_id, pos, sample_color(&n->image_fill, 0.5f), size, pr, th, sw, n->kind == FO_NK_MTSDF_IMAGE,
                   (n->flags & FO_NF_INVERT_Y) != 0);
      break;
    }
    case FO_NK_BACKDROP_BLUR: { /* renderBackdropBlur :1734-1754 */
      if (n->blur > 0.0f) fo_draw_backdrop_blur(c, box, rx, ry, scaled(c, n->blur));
      if (fill_alpha_max(&n->fill) != 0) {
        FoStroke none;
        memset(&none, 0, sizeof none);
        render_rounded_shape(c, n->box, &n->fill, &none, rx, ry);
      }
      break;
    }
    default: break; /* nkFrame / nkScrollBar / nkTransform draw nothing themselves */
  }
  if (n->kind == FO_NK_RECTANGLE) render_inner_shadows(c, n); /* hasActiveInnerShadow :778-789 folded into the loop's skip rules */
  /* children: childIndex fignodes.nim:165-177 */
  int cnt = 0;
  for (int i = idx + 1; i < L->n_nodes && cnt < n->child_count; i++)
    if (L->nodes[i].parent == idx) { cnt++; render_node(c, sc, L, i); }
  /* postRender: cleanups in reverse stage order */
  if (did_rmask) fo_pop_rect_mask(c);
  if (did_clip) fo_pop_mask(c);
  if (did_xf) fo_restore_transform(c);
  if (did_rot) fo_restore_transform(c);
}

/* renderFrame figrender.nim:1960-1995, renderRoot :1946-1955 */
void fo_render_frame(FoCtx* c, const FoScene* sc, float frame_w, float frame_h, int clear, const float rgba[4]) {
  float fw = frame_w * c->ui_scale, fh = frame_h * c->ui_scale;
  if (fw <= 0.0f || fh <= 0.0f) return;
  fo_begin_frame(c, (int)fw, (int)fh, clear, rgba);
  fo_save_transform(c);
  fo_scale(c, c->pixel_scale, c->pixel_scale);
  for (int l = 0; l < sc->n_layers; l++) {
    const FoLayer* L = &sc->layers[l];
    for (int r = 0; r < L->n_roots; r++) render_node(c, sc, L, L->root_ids[r]);
  }
  fo_restore_transform(c);
  fo_end_frame(c);
}
