// fdh_kernels.hip -- hand-written gfx950 (CDNA4) kernels for figdraw's per-pixel SDF path.
//
//   k_bin_draws        coarse binning: one wavefront per (phase, 64x64-pixel bin) builds the bin's draw list in painter's
//                      order; an entry carries the draw index, flags and two 16-bit strip masks (touched / inside the core)
//   k_composite_tiles  one single-wave workgroup per 32x8-pixel strip (four 8x8 tiles side by side, four pixels per lane)
//                      walks the bin's list in painter's order keeping RGBA in registers, re-quantising to RGBA8 after every
//                      draw like the GL framebuffer does (utils/glutils.nim:150-154 blend + RGBA8 target), and stores the
//                      strip once.  Four builds picked per phase: <4> SDF draws only, <0> + clip masks, <2> + the 4-wide atlas
//                      path, <3> + general quads (one pixel slot at a time)
//   k_blur_mx<NK,kV>   the separable backdrop blur of glsl/blur.frag for regions of 0.4 Mpx and more: the merged FIR as a banded
//                      Toeplitz product on the matrix pipe (v_mfma_f32_32x32x16_f16), texels staged by LDS-DMA into a per-wave ring
//   k_blur_h/k_blur_v  the same FIR with VALU FMAs for small regions and unaligned pitches, LDS line staging
//
// The per-pixel math restates src/figdraw/opengl/glsl/{atlas,atlas_rect_mask,mask,blur}.frag; every device function cites
// the lines it follows.  Only the blur FIR is a contraction and runs on MFMA; everything else is VALU + transcendental work
// (SURVEY.md 8d).  The translation unit is compiled WITHOUT packed-FP32 instructions (csrc/Makefile): DESIGN.md section 4.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>

#include "fdh_kernels.h"

#include <cstddef>
#include <cstdlib>

namespace fdh {

// Build switches.  What was tried and dropped is in DESIGN.md section 4 (register prefetch of the next record, 4-wave
// workgroups, several strips per wave, sequential per-pixel shading, ...).
#ifndef FDH_SIMPLE_EDGE
#define FDH_SIMPLE_EDGE 1  // hand-packed path for the commonest edge strips (-4 % VALU instructions)
#endif
#ifndef FDH_STATS
#define FDH_STATS 0  // `make stats`: per-strip draw classification counters (tools/strip_stats.py); never in the product build
#endif
// Two translation units from this one file (csrc/Makefile).  FDH_TU 0: everything except the compositor builds <0>, <2>
// and <4>.  FDH_TU 1 (fdh_composite_uniform.hip): those three and their launcher only, compiled with
// -structurizecfg-skip-uniform-regions.  hipcc structurizes EVERY region of a kernel's control flow, uniform branches
// included; in the draw loop that turns each wave-uniform branch into a predicate in an SGPR pair (s_cselect_b64 / s_and_b64
// / s_cbranch_vccnz where one s_cbranch_scc would do) and keeps the texels that merge at the loop latch out of the
// registers they came from (eight v_mov_b64 per draw).  With the switch, a region whose branches are all wave-uniform is
// left as the branches it is: phase 0 of the bench frame 38 -> 34 us, and with the registers that freed, six waves per SIMD.
// The switch is NOT safe for code that nests uniform branches inside divergent ones: with it the slot path <3> and the blur
// passes come out wrong (measured: 23 of the 57 GPU parity tests fail), so it stays confined to kernels whose draw loop
// nest holds no divergent branch at all -- tools/lint_isa.py checks exactly that on the built code object after every
// build, and the note at shadow_profile() says how the source keeps it so.  Instrumented builds (FDH_STATS, FDH_TIMING,
// FDH_EDGE_CHECK, FDH_MX_CHECK: device-side counters) are single-unit builds (`make variant SINGLE=1`).
#ifndef FDH_TU
#define FDH_TU 0
#endif
#ifndef FDH_SPLIT_UNIFORM
#define FDH_SPLIT_UNIFORM 0  // the Makefile's product and variant builds set 1
#endif

// ------------------------------------------------------------------ small device helpers
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float clamp01(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

struct F4 { float x, y, z, w; };

__device__ __forceinline__ F4 unpack255(uint32_t c) {  // RGBA8 -> floats in 0..255
  F4 r;
  r.x = (float)(c & 255u);
  r.y = (float)((c >> 8) & 255u);
  r.z = (float)((c >> 16) & 255u);
  r.w = (float)(c >> 24);
  return r;
}
__device__ __forceinline__ uint32_t pack255(F4 f) {  // v_cvt_pk_u8_f32 x 4 (the values are integers 0..255 already)
  uint32_t o = __builtin_amdgcn_cvt_pk_u8_f32(f.x, 0, 0u);
  o = __builtin_amdgcn_cvt_pk_u8_f32(f.y, 1, o);
  o = __builtin_amdgcn_cvt_pk_u8_f32(f.z, 2, o);
  return __builtin_amdgcn_cvt_pk_u8_f32(f.w, 3, o);
}

// atlas.frag:51-69
__device__ __forceinline__ float sd_rounded_box(float px, float py, float bx, float by, float r0, float r1, float r2, float r3) {
  float rr = (px > 0.0f) ? ((py > 0.0f) ? r0 : r1) : ((py > 0.0f) ? r2 : r3);
  float qx = __builtin_fabsf(px) - bx + rr, qy = __builtin_fabsf(py) - by + rr;
  float mx = __builtin_fmaxf(qx, 0.0f), my = __builtin_fmaxf(qy, 0.0f);
  return __builtin_fminf(__builtin_fmaxf(qx, qy), 0.0f) + fsqrt(mx * mx + my * my) - rr;
}
// atlas.frag:71-79
__device__ __forceinline__ float sd_ellipse(float px, float py, float rx, float ry) {
  float sx = __builtin_fmaxf(rx, 0.000001f), sy = __builtin_fmaxf(ry, 0.000001f);
  float isx = frcp(sx), isy = frcp(sy);
  float ax = px * isx, ay = py * isy;
  float k0 = fsqrt(ax * ax + ay * ay);
  if (k0 <= 0.000001f) return -__builtin_fminf(sx, sy);
  float bx = ax * isx, by = ay * isy;
  float k1 = fsqrt(bx * bx + by * by);
  return k0 * (k0 - 1.0f) * frcp(__builtin_fmaxf(k1, 0.000001f));
}
// atlas.frag:88-115
__device__ __forceinline__ float sd_elliptical_rounded_box(float px, float py, float bx, float by, float r0, float r1, float r2, float r3) {
  float sel = (px > 0.0f) ? ((py > 0.0f) ? r0 : r1) : ((py > 0.0f) ? r2 : r3);
  if (sel < 0.0f) {
    float r = -sel - 1.0f;
    return sd_rounded_box(px, py, bx, by, r, r, r, r);
  }
  float pv = __builtin_floorf(sel + 0.5f);
  float hi = __builtin_floorf(pv * (1.0f / 4096.0f));
  float rx = (pv - 4096.0f * hi) * bx * (1.0f / 4095.0f);
  float ry = hi * by * (1.0f / 4095.0f);
  float ax = __builtin_fabsf(px), ay = __builtin_fabsf(py);
  if (rx <= 0.0f || ry <= 0.0f) {
    float qx = ax - bx, qy = ay - by;
    float mx = __builtin_fmaxf(qx, 0.0f), my = __builtin_fmaxf(qy, 0.0f);
    return __builtin_fminf(__builtin_fmaxf(qx, qy), 0.0f) + fsqrt(mx * mx + my * my);
  }
  if (rx == ry) return sd_rounded_box(px, py, bx, by, rx, rx, rx, rx);
  float qx = ax - bx + rx, qy = ay - by + ry;
  if (qx > 0.0f && qy > 0.0f) return sd_ellipse(qx, qy, rx, ry);
  return __builtin_fmaxf(qx - rx, qy - ry);
}
__device__ __forceinline__ float shape_dist(bool ellip, float px, float py, float bx, float by, float r0, float r1, float r2, float r3) {
  return ellip ? sd_elliptical_rounded_box(px, py, bx, by, r0, r1, r2, r3) : sd_rounded_box(px, py, bx, by, r0, r1, r2, r3);
}
// atlas.frag:211-216 -- exp(-0.5 z^2) as exp2
// NOTE on selects below: an expensive expression (v_exp / v_sqrt / v_rcp inside) is always evaluated in a statement of its
// own and then SELECTED, never written inside the arm of a ternary: there the compiler keeps a divergent branch around it,
// and ONE divergent branch anywhere in the compositor's draw loop makes LLVM structurize the whole loop nest -- every
// wave-uniform branch in it becomes a predicate in an SGPR pair (s_cselect_b64 / s_and_b64 / s_cbranch_vccnz instead of
// s_cbranch_scc) and the texels that merge at the loop latch can no longer share registers with the ones they replace
// (eight v_mov_b64 per draw).
__device__ __forceinline__ float shadow_profile(float sd, float blur_radius) {
  float sigma = __builtin_fmaxf(0.5f * blur_radius, 0.5f);
  float z = sd * frcp(sigma);
  return fexp2(-0.72134752044f * z * z);
}
// sdBezier atlas.frag:121-160 (exact quadratic-Bezier distance: cubic solve).  Rare path: libm-quality functions.
__device__ __forceinline__ float sd_bezier(float px, float py, float Ax, float Ay, float Bx, float By, float Cx, float Cy) {
  const float ax = Bx - Ax, ay = By - Ay;
  const float bx = Ax - 2.0f * Bx + Cx, by = Ay - 2.0f * By + Cy;
  const float bb = bx * bx + by * by;
  if (bb <= 0.000001f) {
    const float bax = Cx - Ax, bay = Cy - Ay;
    const float h = clamp01(((px - Ax) * bax + (py - Ay) * bay) / __builtin_fmaxf(bax * bax + bay * bay, 0.000001f));
    const float dx = px - (Ax + bax * h), dy = py - (Ay + bay * h);
    return __builtin_sqrtf(dx * dx + dy * dy);
  }
  const float cx = ax * 2.0f, cy = ay * 2.0f;
  const float dx = Ax - px, dy = Ay - py;
  const float kk = 1.0f / bb;
  const float kx = kk * (ax * bx + ay * by);
  const float ky = kk * (2.0f * (ax * ax + ay * ay) + (dx * bx + dy * by)) / 3.0f;
  const float kz = kk * (dx * ax + dy * ay);
  const float p = ky - kx * kx;
  const float p3 = p * p * p;
  const float q = kx * (2.0f * kx * kx - 3.0f * ky) + kz;
  float h = q * q + 4.0f * p3;
  float res;
  if (h >= 0.0f) {
    h = __builtin_sqrtf(h);
    const float x0 = (h - q) / 2.0f, x1 = (-h - q) / 2.0f;
    const float r0 = __builtin_copysignf(powf(__builtin_fabsf(x0), 1.0f / 3.0f), x0) * (x0 == 0.0f ? 0.0f : 1.0f);
    const float r1 = __builtin_copysignf(powf(__builtin_fabsf(x1), 1.0f / 3.0f), x1) * (x1 == 0.0f ? 0.0f : 1.0f);
    const float t = clamp01(r0 + r1 - kx);
    const float ex = dx + (cx + bx * t) * t, ey = dy + (cy + by * t) * t;
    res = ex * ex + ey * ey;
  } else {
    const float z = __builtin_sqrtf(-p);
    const float v = acosf(__builtin_fminf(__builtin_fmaxf(q / (p * z * 2.0f), -1.0f), 1.0f)) / 3.0f;
    const float m = cosf(v);
    const float n = sinf(v) * 1.732050808f;
    const float t1 = clamp01((m + m) * z - kx);
    const float t2 = clamp01((-n - m) * z - kx);
    const float e1x = dx + (cx + bx * t1) * t1, e1y = dy + (cy + by * t1) * t1;
    const float e2x = dx + (cx + bx * t2) * t2, e2y = dy + (cy + by * t2) * t2;
    res = __builtin_fminf(e1x * e1x + e1y * e1y, e2x * e2x + e2y * e2y);
  }
  return __builtin_sqrtf(res);
}
__device__ __forceinline__ void safe_normalize(float x, float y, float fx, float fy, float& ox, float& oy) {  // atlas.frag:174-177
  const float len = __builtin_sqrtf(x * x + y * y);
  if (len <= 0.000001f) { ox = fx; oy = fy; } else { ox = x / len; oy = y / len; }
}
// bezierStrokeSd atlas.frag:179-209
__device__ __forceinline__ float bezier_stroke_sd(float dist, float px, float py, float Ax, float Ay, float Bx, float By, float Cx, float Cy,
                                                  float half_w, uint32_t mode) {
  if (mode == 18u) return dist - half_w;
  float fx, fy, sx, sy, ex, ey;
  safe_normalize(Cx - Ax, Cy - Ay, 1.0f, 0.0f, fx, fy);
  safe_normalize(Bx - Ax, By - Ay, fx, fy, sx, sy);
  safe_normalize(Cx - Bx, Cy - By, fx, fy, ex, ey);
  const float start_proj = (px - Ax) * sx + (py - Ay) * sy;
  const float end_proj = (px - Cx) * ex + (py - Cy) * ey;
  const float trim = mode == 20u ? half_w : 0.0f;
  float tube = dist;
  if (mode == 20u) {
    if (start_proj < 0.0f) tube = __builtin_fminf(tube, __builtin_fabsf((px - Ax) * sy - (py - Ay) * sx));
    if (end_proj > 0.0f) tube = __builtin_fminf(tube, __builtin_fabsf((px - Cx) * ey - (py - Cy) * ex));
  }
  const float cap = __builtin_fmaxf(-start_proj - trim, end_proj - trim);
  return __builtin_fmaxf(tube - half_w, cap);
}
// ---- the same two functions for N pixels at once, straight-line (round 4): what depends on the curve alone is computed once per
// draw, the cubic's two cases are both evaluated and selected, and the transcendentals are the hardware's (v_log / v_exp for the
// cube roots, v_sqrt, v_rcp) or short polynomials (acos: Abramowitz & Stegun 4.4.46, |error| <= 2e-8; sin / cos on [0, pi / 3]:
// Taylor to v^11, <= 4e-9) -- the distance is stationary in the root, so these errors enter squared.  libm's powf / acosf / cosf /
// sinf cost several hundred instructions per pixel and made a curve the most expensive thing the compositor could draw.
__device__ __forceinline__ float cbrt_signed(float x) {
  const float a = __builtin_fabsf(x);
  const float r = fexp2(__builtin_amdgcn_logf(a) * (1.0f / 3.0f));  // (a = 0: log2 = -inf, exp2 = 0)
  return __builtin_copysignf(r, x);
}
__device__ __forceinline__ float acos_poly(float x) {
  const float a = __builtin_fabsf(x);
  float p = -0.0012624911f;
  p = __builtin_fmaf(p, a, 0.0066700901f); p = __builtin_fmaf(p, a, -0.0170881256f); p = __builtin_fmaf(p, a, 0.0308918810f);
  p = __builtin_fmaf(p, a, -0.0501743046f); p = __builtin_fmaf(p, a, 0.0889789874f); p = __builtin_fmaf(p, a, -0.2145988016f);
  p = __builtin_fmaf(p, a, 1.5707963050f);
  const float f = fsqrt(__builtin_fmaxf(1.0f - a, 0.0f)) * p;
  return x < 0.0f ? 3.14159265358979f - f : f;
}
template <int N, bool kPerY>
__device__ __forceinline__ void sd_bezierN(const float* px, const float* pyv, float Ax, float Ay, float Bx, float By, float Cx, float Cy, float* out) {
#pragma clang fp contract(off)
  const float ax = Bx - Ax, ay = By - Ay;
  const float bx = Ax - 2.0f * Bx + Cx, by = Ay - 2.0f * By + Cy;
  const float bb = bx * bx + by * by;
  if (bb <= 0.000001f) {  // wave-uniform: a straight span
    const float bax = Cx - Ax, bay = Cy - Ay;
    const float il = frcp(__builtin_fmaxf(bax * bax + bay * bay, 0.000001f));
#pragma unroll
    for (int k = 0; k < N; k++) {
      const float py = pyv[kPerY ? k : 0];
      const float h = clamp01(((px[k] - Ax) * bax + (py - Ay) * bay) * il);
      const float dx = px[k] - (Ax + bax * h), dy = py - (Ay + bay * h);
      out[k] = fsqrt(dx * dx + dy * dy);
    }
    return;
  }
  // The one-root case forms (sqrt(h) - q) / 2 with sqrt(h) ~ |q| wherever 4 p^3 << q^2: what survives the cancellation is the
  // rounding of the operations that led there, and a pixel's distance can move by a tenth of a pixel when sqrt(h) moves by one ulp
  // (measured; the trigonometric case, the cube roots and the inputs are benign: 1e-6 relative moves the result by 1e-4 px).  So
  // up to the roots the arithmetic is the oracle's and libm's operation for operation: IEEE division and square root, no fused
  // multiply-adds (the pragma at the top of the function).
  const float cx = ax * 2.0f, cy = ay * 2.0f;
  const float kk = 1.0f / bb;
  const float kx = kk * (ax * bx + ay * by);
  const float aa2 = 2.0f * (ax * ax + ay * ay);
#pragma unroll
  for (int k = 0; k < N; k++) {
    const float dx = Ax - px[k], dy = Ay - pyv[kPerY ? k : 0];
    const float ky = kk * (aa2 + (dx * bx + dy * by)) / 3.0f;
    const float kz = kk * (dx * ax + dy * ay);
    const float p = ky - kx * kx;
    const float p3 = p * p * p;
    const float q = kx * (2.0f * kx * kx - 3.0f * ky) + kz;
    const float h = q * q + 4.0f * p3;
    // h >= 0: one real root
    const float hs = __builtin_sqrtf(__builtin_fmaxf(h, 0.0f));
    const float tA = clamp01(cbrt_signed((hs - q) * 0.5f) + cbrt_signed((-hs - q) * 0.5f) - kx);
    const float eax = dx + (cx + bx * tA) * tA, eay = dy + (cy + by * tA) * tA;
    const float resA = eax * eax + eay * eay;
    // h < 0 (then p < 0): three real roots, the two that can be nearest
    const float z = fsqrt(__builtin_fmaxf(-p, 0.0f));
    const float den = p * z * 2.0f;
    const float arg = __builtin_fminf(__builtin_fmaxf(q * frcp(den), -1.0f), 1.0f);
    const float v = acos_poly(den == 0.0f ? 0.0f : arg) * (1.0f / 3.0f);
    const float v2 = v * v;
    float cm = -1.0f / 3628800.0f, sn = -1.0f / 39916800.0f;
    cm = __builtin_fmaf(cm, v2, 1.0f / 40320.0f); cm = __builtin_fmaf(cm, v2, -1.0f / 720.0f); cm = __builtin_fmaf(cm, v2, 1.0f / 24.0f); cm = __builtin_fmaf(cm, v2, -0.5f); cm = __builtin_fmaf(cm, v2, 1.0f);
    sn = __builtin_fmaf(sn, v2, 1.0f / 362880.0f); sn = __builtin_fmaf(sn, v2, -1.0f / 5040.0f); sn = __builtin_fmaf(sn, v2, 1.0f / 120.0f); sn = __builtin_fmaf(sn, v2, -1.0f / 6.0f); sn = __builtin_fmaf(sn, v2, 1.0f);
    const float m = cm, n = sn * v * 1.732050808f;
    const float t1 = clamp01((m + m) * z - kx), t2 = clamp01((-n - m) * z - kx);
    const float e1x = dx + (cx + bx * t1) * t1, e1y = dy + (cy + by * t1) * t1;
    const float e2x = dx + (cx + bx * t2) * t2, e2y = dy + (cy + by * t2) * t2;
    const float resB = __builtin_fminf(e1x * e1x + e1y * e1y, e2x * e2x + e2y * e2y);
    out[k] = fsqrt(h >= 0.0f ? resA : resB);
  }
}
__device__ __forceinline__ float median3(float a, float b, float c) {  // atlas.frag:41-43
  return __builtin_fmaxf(__builtin_fminf(a, b), __builtin_fminf(__builtin_fmaxf(a, b), c));
}

// GL_LINEAR + GL_REPEAT fetch from one atlas level, texel-space coords (s*S - 0.5); returns 0..1 floats
// the GL_LINEAR weighting of four RGBA8 texels (row y0: q00, q01; row y1: q10, q11), 0..1 floats
__device__ __forceinline__ F4 bilinear_of(uint32_t q00, uint32_t q01, uint32_t q10, uint32_t q11, float ax, float ay) {
  const F4 a = unpack255(q00), b = unpack255(q01), c = unpack255(q10), d = unpack255(q11);
  const float k = 1.0f / 255.0f;
  F4 o;
  o.x = (mixf(a.x, b.x, ax) * (1.0f - ay) + mixf(c.x, d.x, ax) * ay) * k;
  o.y = (mixf(a.y, b.y, ax) * (1.0f - ay) + mixf(c.y, d.y, ax) * ay) * k;
  o.z = (mixf(a.z, b.z, ax) * (1.0f - ay) + mixf(c.z, d.z, ax) * ay) * k;
  o.w = (mixf(a.w, b.w, ax) * (1.0f - ay) + mixf(c.w, d.w, ax) * ay) * k;
  return o;
}
__device__ __forceinline__ F4 atlas_bilinear(const uint32_t* __restrict__ tex, int S, float x, float y) {
  float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
  float ax = x - fx, ay = y - fy;
  int m = S - 1;  // S is a power of two
  int x0 = (int)fx & m, y0 = (int)fy & m, x1 = (x0 + 1) & m, y1 = (y0 + 1) & m;
  return bilinear_of(tex[(size_t)y0 * S + x0], tex[(size_t)y0 * S + x1], tex[(size_t)y1 * S + x0], tex[(size_t)y1 * S + x1], ax, ay);
}
// texture(atlasTex, uv) with LINEAR_MIPMAP_LINEAR min / LINEAR mag (glcontext.nim:157-169); lod = log2(rho)
__device__ __forceinline__ F4 atlas_sample(const AtlasView& A, float u, float v, float lod) {
  int S = A.size;
  if (!(lod > 0.0f) || A.n_levels < 2) return atlas_bilinear(A.level[0], S, u * (float)S - 0.5f, v * (float)S - 0.5f);
  float maxl = (float)(A.n_levels - 1);
  lod = __builtin_fminf(lod, maxl);
  int l0 = (int)__builtin_floorf(lod);
  int l1 = l0 + 1 > A.n_levels - 1 ? A.n_levels - 1 : l0 + 1;
  float f = lod - (float)l0;
  int S0 = S >> l0, S1 = S >> l1;
  F4 a = atlas_bilinear(A.level[l0], S0, u * (float)S0 - 0.5f, v * (float)S0 - 0.5f);
  F4 b = atlas_bilinear(A.level[l1], S1, u * (float)S1 - 0.5f, v * (float)S1 - 0.5f);
  F4 o = {mixf(a.x, b.x, f), mixf(a.y, b.y, f), mixf(a.z, b.z, f), mixf(a.w, b.w, f)};
  return o;
}

// ---- where a quadratic bezier can be: split at t = 1/2 (de Casteljau) into two sub-curves, each inside the box aligned with its own
// chord that reaches min(0, b.f) .. max(|chord|, b.f) along it and 0 .. b.g / 2 across (b = its middle control point relative to its
// start).  The sagitta of a half is a quarter of the whole curve's: the boxes hug the curve where one box around the whole span would
// be as fat as the curve is bent.  Used by k_bin_draws, per strip.  (Four boxes took the bin kernel from 20 to 36 us on the
// 1500-curve frame for a quarter fewer strip-draws; queueing a strip's near pixels in LDS and running the cubic on the queue was
// no faster than four lock-step pixel slots: the spans are short and their reach wide, most pixels of a kept strip are near.)
struct CurveBox { float ax, ay, fx, fy, x_lo, x_hi, y_lo, y_hi; };
__device__ __forceinline__ CurveBox curve_box(float Ax, float Ay, float Bx, float By, float Cx, float Cy) {
  CurveBox b;
  b.ax = Ax; b.ay = Ay;
  float fx = Cx - Ax, fy = Cy - Ay;
  const float fl = fsqrt(fx * fx + fy * fy);
  const float il = frcp(__builtin_fmaxf(fl, 0.000001f));
  const bool tiny = fl <= 0.000001f;
  b.fx = tiny ? 1.0f : fx * il;
  b.fy = tiny ? 0.0f : fy * il;
  const float bf = (Bx - Ax) * b.fx + (By - Ay) * b.fy, bg = (By - Ay) * b.fx - (Bx - Ax) * b.fy, lac = (Cx - Ax) * b.fx + (Cy - Ay) * b.fy;
  b.x_lo = __builtin_fminf(0.0f, bf); b.x_hi = __builtin_fmaxf(lac, bf);
  b.y_lo = __builtin_fminf(0.0f, 0.5f * bg); b.y_hi = __builtin_fmaxf(0.0f, 0.5f * bg);
  return b;
}
__device__ __forceinline__ void curve_boxes2(float Ax, float Ay, float Bx, float By, float Cx, float Cy, CurveBox (&out)[2]) {
  const float abx = 0.5f * (Ax + Bx), aby = 0.5f * (Ay + By), bcx = 0.5f * (Bx + Cx), bcy = 0.5f * (By + Cy);
  const float mx = 0.5f * (abx + bcx), my = 0.5f * (aby + bcy);
  out[0] = curve_box(Ax, Ay, abx, aby, mx, my);  // halves: (A, AB, M) and (M, BC, C)
  out[1] = curve_box(mx, my, bcx, bcy, Cx, Cy);
}
// ------------------------------------------------------------------ binning

// Which of a bin's 16 strips (32x8 px; strip s = (jy*2 + jx)*4 + w sits at column jx, row jy*4 + w) a bin-relative
// pixel box [x0,x1) x [y0,y1) touches.  The compositor's per-strip culling is then one bit test on the list entry
// instead of a dependent bounding-box fetch.
// strip rows (8 bits, row r = jy * 4 + w) x the two strip columns -> strip bits: column 0 holds rows 0..3 in bits 0..3 and rows 4..7 in
// bits 8..11, column 1 the same four bits higher (closed form: the loop over the eight rows it replaces was a fifth of k_bin_draws<false>)
__device__ __forceinline__ uint32_t strips_of_rows(uint32_t rows, bool col0, bool col1) {
  const uint32_t c0 = (rows & 15u) | ((rows & 0xf0u) << 4);
  return (col0 ? c0 : 0u) | (col1 ? c0 << 4 : 0u);
}
__device__ __forceinline__ uint32_t strip_mask(int x0, int y0, int x1, int y1) {
  x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0;
  x1 = x1 > kBin ? kBin : x1; y1 = y1 > kBin ? kBin : y1;
  const int r0 = y0 >> 3, r1 = (y1 + 7) >> 3;                   // strip rows [r0, r1) of 8
  const uint32_t rows = ((1u << r1) - 1u) & ~((1u << r0) - 1u);  // 8 bits
  return strips_of_rows(rows, x0 < kTileW, x1 > kTileW);
}

// the strips of a bin that lie entirely inside a bin-relative pixel box (the draw's saturated core)
// (no branch: the compositor's direct launches make entries inside their draw loop, which must hold no divergent one -- tools/lint_isa.py)
__device__ __forceinline__ uint32_t strip_mask_inside(int x0, int y0, int x1, int y1) {
  int r0 = (y0 + 7) >> 3, r1 = y1 >> 3;  // strip rows [r0, r1) fully inside
  r0 = r0 < 0 ? 0 : r0; r0 = r0 > 8 ? 8 : r0; r1 = r1 > 8 ? 8 : r1;
  r1 = r1 < r0 ? r0 : r1;                // (no row: the mask below is empty)
  const uint32_t rows = ((1u << r1) - 1u) & ~((1u << r0) - 1u);
  const uint32_t m = strips_of_rows(rows, x0 <= 0 && x1 >= kTileW, x0 <= kTileW && x1 >= 2 * kTileW);
  return (x1 <= x0 || y1 <= y0) ? 0u : m;
}

// List entry flags (uint2.x high bits; the low 30 bits are the draw index)
// One WAVEFRONT per (phase, bin): ordered stream compaction of the phase's draws that touch the bin.  An entry is
// {draw index | flags, strips touched (16 bits) | strips inside the draw's saturated core (16 bits)}; strips where an
// annular stroke is provably invisible (its core) are dropped from the entry, and the entry with them if none is left.
//
// The scan reads the 4-byte "bin boxes" (inclusive bin-index bounds, u8 x 4, built on the host), four draws per lane
// and step as one 16-byte load, so a step tests 256 draws with ~30 instructions and no barrier; only steps with a hit
// touch the pixel bounds and the records.  (The first version -- one 256-thread workgroup per bin, one draw per thread,
// three barriers per step -- spent 26 us on the 10 001-draw glyph frame, all of it instruction issue.)
// Bin box = x0 | y0 << 8 | (127 - x1) << 16 | (127 - y1) << 24, 7-bit bin indices (in 128-px units when a frame has
// more than 128 bins along an axis; the exact test follows for the hits).  With U = (bx | by << 8 | (127 - bx) << 16 |
// (127 - by) << 24) | 0x80808080, the four byte-wise differences U - q keep their guard bit exactly when
// x0 <= bx, y0 <= by, bx <= x1, by <= y1: one subtract, one and, one compare per draw.
// The two ends of an entry's making that every draw goes through -- both translation units hold them: the compositor of a frame with at
// most 64 draws per phase makes its entries itself (round 6: "direct" launches, k_composite_tiles).
// (the 24-byte BinRec in ONE round trip -- a 16- and an 8-byte load issued together, pinned: read field by field the compiler sank
// each field's load behind the test before it, three to four dependent L2 latencies per batch of hits)
__device__ __forceinline__ bool bin_entry_head(const BinRec* __restrict__ binrec, int i, int x0, int y0, BinRec& r, uint32_t& word, uint32_t& strips) {
  {
    const uint2* __restrict__ src = reinterpret_cast<const uint2*>(binrec + i);
    uint2 q0 = src[0], q1 = src[1], q2 = src[2];
    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2));
    r.box.x0 = (int16_t)(q0.x & 0xffffu); r.box.y0 = (int16_t)(q0.x >> 16); r.box.x1 = (int16_t)(q0.y & 0xffffu); r.box.y1 = (int16_t)(q0.y >> 16);
    r.ix0 = (int16_t)(q1.x & 0xffffu); r.iy0 = (int16_t)(q1.x >> 16); r.ix1 = (int16_t)(q1.y & 0xffffu); r.iy1 = (int16_t)(q1.y >> 16);
    r.flags = q2.x; r.pad = q2.y;
  }
  const BBox b = r.box;
  strips = strip_mask(b.x0 - x0, b.y0 - y0, b.x1 - x0, b.y1 - y0);
  word = (uint32_t)i | (r.flags & ~LE_INDEX);
  return b.x0 < x0 + kBin && b.x1 > x0 && b.y0 < y0 + kBin && b.y1 > y0;  // exact test
}
// (written without branches, for the same reason: every step is computed and selected)
__device__ __forceinline__ void bin_entry_tail(const BinRec& r, int x0, int y0, bool& hit, uint32_t& strips) {
  const bool has_core = (r.flags & BR_HAS_CORE) != 0u, removed = (r.flags & BR_CORE_REMOVED) != 0u, exact = (r.flags & BR_BOX_EXACT) != 0u;
  const uint32_t core = strip_mask_inside(r.ix0 - x0, r.iy0 - y0, r.ix1 - x0, r.iy1 - y0) & strips;
  // alpha == 0 on the core (stroke interior) or too small to change an 8-bit channel (deep inside an inner shadow): those strips leave
  // the entry, and the entry goes with them if none is left; any other core is marked in the high half
  const uint32_t s_removed = strips & ~core, s_marked = strips | (core << 16);
  const uint32_t s1 = has_core ? (removed ? s_removed : s_marked) : strips;
  const bool gone = has_core && removed && s_removed == 0u;
  // edge strips wholly inside the quad's pixel bounds: state (0, 1) -- fdh_types.h, BR_BOX_EXACT
  const BBox b = r.box;
  const uint32_t inq = strip_mask_inside(b.x0 - x0, b.y0 - y0, b.x1 - x0, b.y1 - y0) & s1 & ~(s1 >> 16) & 0xffffu;
  const uint32_t s2 = (s1 & ~inq) | (inq << 16);
  strips = (has_core && !gone && exact) ? s2 : s1;
  hit = hit && !gone;
}
#if FDH_TU == 0
__device__ __forceinline__ bool binbox_hits(uint32_t q, uint32_t U) { return ((U - q) & 0x80808080u) == 0x80808080u; }
// (kRefine: the build for frames that hold bezier strokes or rotated quads -- their per-strip tests cost registers, 193 against 56,
// which a frame without them should not pay in occupancy: bench frame 5.4 us against 8.7)
template <bool kRefine>
__device__ __forceinline__ void bin_entry(const BinParams& P, int i, int x0, int y0, bool& hit, uint32_t& word, uint32_t& strips) {
  BinRec r;
  if (!bin_entry_head(P.binrec, i, x0, y0, r, word, strips)) { hit = false; return; }
  if (kRefine && (r.flags & BR_CURVE)) {
    // A bezier stroke: strips whose pixels are all farther from the chord-aligned box around the curve than sqrt 2 (half width +
    // 0.5 / aa) hold no coverage (see the 4-wide bezier path of k_composite_tiles, which applies the same bound per strip after
    // fetching the record; here the strip never sees the draw).  The strip's pixel-centre rectangle is mapped into the quad's local
    // frame (upright quad: x and y map separately), its corners into the chord frame, and their bounding box is held against the
    // curve's box.
    const uint4* __restrict__ d4 = reinterpret_cast<const uint4*>(P.draws + i);
    const uint4 q0 = d4[0], q1 = d4[1], q2 = d4[2], q3 = d4[3], q5 = d4[5];
    const float ox = __uint_as_float(q0.z), oy = __uint_as_float(q0.w), inv_w = __uint_as_float(q1.x), inv_h = __uint_as_float(q1.y);
    const float p0 = __uint_as_float(q1.z), p1 = __uint_as_float(q1.w), Ax = __uint_as_float(q2.x), Ay = __uint_as_float(q2.y), f0 = __uint_as_float(q2.z);
    const float Bx = __uint_as_float(q3.x), By = __uint_as_float(q3.y), Cx = __uint_as_float(q3.z), Cy = __uint_as_float(q3.w), aa = __uint_as_float(q5.z);
    CurveBox cb[2];
    curve_boxes2(Ax, Ay, Bx, By, Cx, Cy, cb);
    const float reach = 1.41422f * (__builtin_fmaxf(f0, 0.0f) * 0.5f + 0.5f / aa) + 0.05f;  // (+ slack for the kernels' own rounding of the coordinates)
    // Pixel centre -> local frame is separable and affine (upright quad): lx = sx px + tx, ly = sy py + ty.  A rectangle of pixel
    // centres with centre (mx, my) and half sizes (hx, hy) has, in box c's chord frame (a rotation by (fx, fy) about (ax, ay)), the
    // bounding box  centre (Xc, Yc) = R (l(mx, my) - a),  half sizes (|fx| hx' + |fy| hy', |fy| hx' + |fx| hy')  with hx' = |sx| hx,
    // hy' = |sy| hy -- the same box the four mapped corners span (round 4 mapped the corners: ~30 operations per strip and box
    // against 8 here, and 103 VGPRs).  It is near the curve's box iff both centre distances are under the summed half sizes + reach.
    // The WHOLE bin first: most bins inside a long stroke's quad are nowhere near the curve, and sixteen strip tests end there.
    const float sx = inv_w * (2.0f * p0), sy = inv_h * (2.0f * p1);
    const float tx = (-ox * inv_w - 0.5f) * (2.0f * p0), ty = (-oy * inv_h - 0.5f) * (2.0f * p1);
    const float asx = __builtin_fabsf(sx), asy = __builtin_fabsf(sy);
    float X0[2], Y0[2], dXc[2], dXr[2], dYc[2], dYr[2], ex[2], ey[2], bcx[2], bcy[2];
    bool bin_near = false;
    const float eps = 0.002f;  // (the centre / half-size form rounds differently from the corner form by a few ulps of ~1e3: keep, never drop)
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const float afx = __builtin_fabsf(cb[c].fx), afy = __builtin_fabsf(cb[c].fy);
      bcx[c] = 0.5f * (cb[c].x_lo + cb[c].x_hi); bcy[c] = 0.5f * (cb[c].y_lo + cb[c].y_hi);
      const float bhx = 0.5f * (cb[c].x_hi - cb[c].x_lo), bhy = 0.5f * (cb[c].y_hi - cb[c].y_lo);
      // strip (0, 0): pixel centres x0 + 0.5 .. x0 + 31.5, y0 + 0.5 .. y0 + 7.5
      const float l0x = sx * ((float)x0 + 16.0f) + tx - cb[c].ax, l0y = sy * ((float)y0 + 4.0f) + ty - cb[c].ay;
      X0[c] = l0x * cb[c].fx + l0y * cb[c].fy; Y0[c] = l0y * cb[c].fx - l0x * cb[c].fy;
      dXc[c] = 32.0f * sx * cb[c].fx; dXr[c] = 8.0f * sy * cb[c].fy; dYc[c] = -32.0f * sx * cb[c].fy; dYr[c] = 8.0f * sy * cb[c].fx;
      const float hx = 15.5f * asx, hy = 3.5f * asy;
      ex[c] = bhx + afx * hx + afy * hy + reach + eps; ey[c] = bhy + afy * hx + afx * hy + reach + eps;
      // the bin: centre = strip (0, 0)'s + half a strip column + 3.5 strip rows, half sizes 31.5 px
      const float Xb = X0[c] + 0.5f * dXc[c] + 3.5f * dXr[c], Yb = Y0[c] + 0.5f * dYc[c] + 3.5f * dYr[c];
      const float Hx = 31.5f * asx, Hy = 31.5f * asy;
      bin_near = bin_near || (__builtin_fabsf(Xb - bcx[c]) < bhx + afx * Hx + afy * Hy + reach + eps && __builtin_fabsf(Yb - bcy[c]) < bhy + afy * Hx + afx * Hy + reach + eps);
    }
    uint32_t keep = 0;
    if (bin_near) {
#pragma unroll 4
      for (int s = 0; s < 16; s++) {
        const float col = (float)((s >> 2) & 1), row = (float)((s >> 3) * 4 + (s & 3));
        bool near = false;
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const float Xc = X0[c] + col * dXc[c] + row * dXr[c], Yc = Y0[c] + col * dYc[c] + row * dYr[c];
          near = near || (__builtin_fabsf(Xc - bcx[c]) < ex[c] && __builtin_fabsf(Yc - bcy[c]) < ey[c]);
        }
        if (near) keep |= 1u << s;
      }
    }
    strips &= keep;
    hit = strips != 0u;
    return;
  }
  if (kRefine && (r.flags & BR_GENERAL)) {
    // A rotated quad: a strip all of whose pixel centres fail ONE of the quad's outer edges (bottom, left, right, top: edges 0 and 2
    // of triangle (TL, BL, BR), 1 and 2 of (TR, TL, BR)) holds no pixel of it.  The largest value an edge function takes on a
    // strip is at the corner its coefficients' signs pick; 32-bit arithmetic is exact here (F_EDGE32).
    // (everything the strip loops need is fetched up front, as 16-byte pieces: left to the compiler the loads stayed inside the
    // loops -- conditional code does not get its loads hoisted -- and sixteen dependent round trips made the bin kernel 4x longer)
    const uint4* __restrict__ q4 = reinterpret_cast<const uint4*>(P.exts + P.draws[i].ext);
    const uint4 w0 = q4[0], w1 = q4[2], w2 = q4[4], w3 = q4[5], wc = q4[9], wl0 = q4[10], wl1 = q4[11], wl2 = q4[12];  // e[0][0], e[0][2], e[1][1], e[1][2], core, lm
    const int ea[4] = {(int)w0.x, (int)w1.x, (int)w2.x, (int)w3.x}, eb[4] = {(int)w0.y, (int)w1.y, (int)w2.y, (int)w3.y}, ec[4] = {(int)w0.z, (int)w1.z, (int)w2.z, (int)w3.z};
    // the value an edge function takes at the strip's corner (sx, sy) = the bin's corner + 64 a per strip column + 16 b per strip row;
    // its largest / smallest value on the strip is that plus the spans its coefficients' signs pick (62 |a|, 14 |b|)
    int e00[4], hi[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      e00[k] = ea[k] * (2 * x0 + 1) + eb[k] * (2 * y0 + 1) + ec[k];
      hi[k] = (ea[k] > 0 ? 62 * ea[k] : 0) + (eb[k] > 0 ? 14 * eb[k] : 0);
      lo[k] = (ea[k] < 0 ? 62 * ea[k] : 0) + (eb[k] < 0 ? 14 * eb[k] : 0);
    }
    const float cxl = __uint_as_float(wc.x), cxr = __uint_as_float(wc.y), cyb = __uint_as_float(wc.z), cyt = __uint_as_float(wc.w);
    const float lm[12] = {__uint_as_float(wl0.x), __uint_as_float(wl0.y), __uint_as_float(wl0.z), __uint_as_float(wl0.w), __uint_as_float(wl1.x), __uint_as_float(wl1.y),
                          __uint_as_float(wl1.z), __uint_as_float(wl1.w), __uint_as_float(wl2.x), __uint_as_float(wl2.y), __uint_as_float(wl2.z), __uint_as_float(wl2.w)};
    const bool has_core = cxr > cxl;
    uint32_t keep = 0, core = 0;
    // The WHOLE bin first (its pixel centres span 126 half-pixel units each way): outside one edge -> the draw leaves the bin; inside
    // all four AND, under both triangles' maps, inside the core rectangle -> every strip is a core strip; inside all four without
    // a core -> every strip stays.  A large rotated panel covers most of its bins whole: for those the sixteen strip tests are skipped.
    bool bin_out = false, bin_in = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int bhi = (ea[k] > 0 ? 126 * ea[k] : 0) + (eb[k] > 0 ? 126 * eb[k] : 0), blo = (ea[k] < 0 ? 126 * ea[k] : 0) + (eb[k] < 0 ? 126 * eb[k] : 0);
      bin_out = bin_out || e00[k] + bhi < 0;
      bin_in = bin_in && e00[k] + blo > 0;
    }
    if (bin_out) { hit = false; return; }
    bool bin_core = bin_in && has_core;
    if (bin_core) {
      const float Xlo = (float)(2 * x0 + 1), Ylo = (float)(2 * y0 + 1);
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const float* m = lm + 6 * t;
        const float lx0 = m[0] * Xlo + m[1] * Ylo + m[2], ly0 = m[3] * Xlo + m[4] * Ylo + m[5];
        const float dxx = 126.0f * m[0], dxy = 126.0f * m[1], dyx = 126.0f * m[3], dyy = 126.0f * m[4];
        const float lxmin = lx0 + __builtin_fminf(dxx, 0.0f) + __builtin_fminf(dxy, 0.0f), lxmax = lx0 + __builtin_fmaxf(dxx, 0.0f) + __builtin_fmaxf(dxy, 0.0f);
        const float lymin = ly0 + __builtin_fminf(dyx, 0.0f) + __builtin_fminf(dyy, 0.0f), lymax = ly0 + __builtin_fmaxf(dyx, 0.0f) + __builtin_fmaxf(dyy, 0.0f);
        bin_core = bin_core && lxmin >= cxl && lxmax <= cxr && lymin >= cyb && lymax <= cyt;
      }
    }
    if (bin_core) { keep = 0xffffu; core = 0xffffu; }
    else if (bin_in && !has_core) { keep = 0xffffu; }
    else
#pragma unroll 1
    for (int s = 0; s < 16; s++) {
      const int col = (s >> 2) & 1, row = (s >> 3) * 4 + (s & 3);
      bool out = false, in = has_core;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int e = e00[k] + 64 * col * ea[k] + 16 * row * eb[k];
        out = out || e + hi[k] < 0;
        in = in && e + lo[k] > 0;
      }
      if (!out) keep |= 1u << s;
      // inside the quad: do the strip's corner pixels map into the local-frame core rectangle under both triangles' maps? (QuadExt::core)
      const float Xlo = (float)(2 * (x0 + col * kTileW) + 1), Ylo = (float)(2 * (y0 + row * kTileH) + 1);
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const float* m = lm + 6 * t;
        const float lx0 = m[0] * Xlo + m[1] * Ylo + m[2], ly0 = m[3] * Xlo + m[4] * Ylo + m[5];
        const float dxx = 62.0f * m[0], dxy = 14.0f * m[1], dyx = 62.0f * m[3], dyy = 14.0f * m[4];
        const float lxmin = lx0 + __builtin_fminf(dxx, 0.0f) + __builtin_fminf(dxy, 0.0f), lxmax = lx0 + __builtin_fmaxf(dxx, 0.0f) + __builtin_fmaxf(dxy, 0.0f);
        const float lymin = ly0 + __builtin_fminf(dyx, 0.0f) + __builtin_fminf(dyy, 0.0f), lymax = ly0 + __builtin_fmaxf(dyx, 0.0f) + __builtin_fmaxf(dyy, 0.0f);
        in = in && lxmin >= cxl && lxmax <= cxr && lymin >= cyb && lymax <= cyt;
      }
      if (in) core |= 1u << s;
    }
    keep &= strips;
    core &= keep;
    strips = keep;
    if (r.flags & BR_CORE_REMOVED) strips &= ~core;
    else strips |= core << 16;
    hit = (strips & 0xffffu) != 0u;
    return;
  }
  bin_entry_tail(r, x0, y0, hit, strips);
}
template <bool kRefine>
__global__ __launch_bounds__(64) void k_bin_draws(BinParams P) {
  const int nb = P.bins_x * P.bins_y;
  int phase, bin, bx, by;
  if (P.sub_n) {  // (scalar: the table is in the kernel arguments)
    int s_first = 0, s_x0 = 0, s_y0 = 0, s_nx = P.sub_nx[0];
    phase = 0;
#pragma unroll
    for (int k = 1; k < BinParams::kBinSubs; k++)
      if (k < P.sub_n && (int)blockIdx.x >= P.sub_first[k]) { phase = k; s_first = P.sub_first[k]; s_x0 = P.sub_x0[k]; s_y0 = P.sub_y0[k]; s_nx = P.sub_nx[k]; }
    const int local = (int)blockIdx.x - s_first, ly = local / s_nx;
    bx = s_x0 + local - ly * s_nx; by = s_y0 + ly;
    bin = by * P.bins_x + bx;
  } else {
    phase = blockIdx.x / nb; bin = blockIdx.x - phase * nb;
    by = bin / P.bins_x; bx = bin - by * P.bins_x;
  }
  const int x0 = bx * kBin, y0 = by * kBin;
  const int first = P.phase_first[phase], last = P.phase_first[phase + 1];
  uint2* out = P.lists + ((size_t)phase * nb + bin) * P.stride;
  const int lane = threadIdx.x;
  // "The upload in front of this launch has finished" for the host (Context::issue: what releases a staging set): this launch has
  // started, so everything before it on the stream is done.  One posted store to a word of pinned host memory.
  if (P.seq_out && blockIdx.x == 0 && lane == 0) __hip_atomic_store(P.seq_out, P.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // Warm-up for the compositor: a frame's records were just written by the upload kernel, i.e. they sit in ONE XCD's L2 or in
  // memory, and the compositor fetches them with scalar loads it waits for (a record round trip per edge draw: a fresh frame's
  // phase-0 launch ran 34 us against 31 for a replayed one whose records were L2-resident).  A record is one 128-byte line:
  // the waves of this launch that run on XCD x (workgroup b runs on XCD b % 8) touch every record once between them, so each
  // XCD's L2 holds the frame's records before the compositor starts.  The loaded dword is only kept alive (end of the kernel).
  uint32_t warm = 0;
  {
    const int per_xcd = ((int)gridDim.x + 7) >> 3, w = (int)blockIdx.x >> 3;
    const int lpw = (P.n_draws + per_xcd - 1) / per_xcd;  // records per wave
    const int rec = w * lpw + lane;
    if (lane < lpw && rec < P.n_draws) warm = *reinterpret_cast<const uint32_t*>(P.draws + rec);
  }
  constexpr uint32_t kQueue = 512;
  __shared__ int hits[kQueue];
  uint32_t queued = 0;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const uint4* __restrict__ boxes4 = reinterpret_cast<const uint4*>(P.binbox);  // padded to a multiple of 4 draws
  const int ngroups = (P.n_draws + 3) >> 2;
  uint32_t count = 0;
  auto flush = [&]() {
    __builtin_amdgcn_wave_barrier();
    for (uint32_t c = 0; c < queued; c += 64) {
      bool ok = c + lane < queued;
      uint32_t word = 0, strips = 0;
      if (ok) bin_entry<kRefine>(P, hits[c + lane], x0, y0, ok, word, strips);  // a stroke may drop out
      const unsigned long long mb = __ballot(ok);
      if (ok) out[count + __builtin_popcountll(mb & lt)] = make_uint2(word, strips);
      count += __builtin_popcountll(mb);
    }
    queued = 0;
    __builtin_amdgcn_wave_barrier();
  };
  const uint32_t cbx = (uint32_t)bx >> P.binbox_shift, cby = (uint32_t)by >> P.binbox_shift;
  const uint32_t U = (cbx | (cby << 8) | ((127u - cbx) << 16) | ((127u - cby) << 24)) | 0x80808080u;
  // Two levels: P.chunkbox[c] is the union box of draws [256 c, 256 c + 256) (byte-wise min of their bin boxes).  The wave
  // tests 64 chunks at once (one box per lane) and then walks only the chunks that can reach this bin -- draws arrive in
  // layout order, so a run of 256 glyphs touches a handful of bins and most (bin, chunk) pairs end at the ballot.
  // The boxes of up to four live chunks are fetched together: with two waves per SIMD nothing else hides the L2 latency
  // of a dependent load per step.
  constexpr int kAhead = 4;  // (eight, round 5: no change on the 8910-draw curve frame -- 22.4 us either way; its launch is paced by the hits' record fetches)
  const int c_last = (last - 1) >> 8;
  for (int c0 = first >> 8; c0 <= c_last && first < last; c0 += 64) {
    const int cl = c0 + lane;
    unsigned long long live = __ballot(cl <= c_last && binbox_hits(P.chunkbox[min(cl, c_last)], U));
    while (live) {
    int cs[kAhead];
    uint4 qc[kAhead];
#pragma unroll
    for (int a = 0; a < kAhead; a++) {
      cs[a] = -1;
      if (live) {
        cs[a] = c0 + __builtin_ctzll(live);
        live &= live - 1ull;
        const int g = cs[a] * 64 + lane;
        qc[a] = g < ngroups ? boxes4[g] : make_uint4(0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu);  // never hits
      }
    }
#pragma unroll
    for (int a = 0; a < kAhead; a++) {  // (body kept at one indent level)
    if (cs[a] < 0) break;
    const int base = cs[a] << 8;
    const uint4 q = qc[a];
    const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
    const int i4 = base + lane * 4;
    bool hit[4];
    unsigned long long m[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = i4 + j;
      hit[j] = binbox_hits(qq[j], U) && i >= first && i < last;
      m[j] = __ballot(hit[j]);
    }
    if ((m[0] | m[1] | m[2] | m[3]) == 0ull) continue;
    // The hits, in draw order (lane-major, then j), are appended to an LDS queue; the expensive part -- pixel bounds,
    // record fields, strip masks, a chain of dependent loads -- runs when the queue fills up (and once at the end)
    // with a hit per lane, not once per step under divergence.
    const uint32_t total = __builtin_popcountll(m[0]) + __builtin_popcountll(m[1]) + __builtin_popcountll(m[2]) + __builtin_popcountll(m[3]);
    uint32_t rank = queued + __builtin_popcountll(m[0] & lt) + __builtin_popcountll(m[1] & lt) + __builtin_popcountll(m[2] & lt) +
                    __builtin_popcountll(m[3] & lt);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (hit[j]) { hits[rank] = i4 + j; rank++; }
    }
    queued += total;
    if (queued > kQueue - 256) flush();  // the next step can add up to 256
    }
    }
  }
  flush();
  if (lane == 0) P.counts[(size_t)phase * nb + bin] = count;
  asm volatile("" : : "v"(warm));  // (the warm-up load must be issued: nothing reads its result)
}

#endif  // FDH_TU == 0

// Longest-processing-time-first order for the compositor: bins sorted by list length, descending (counting sort on
// min(count, 255); the order among equal keys is whatever the LDS atomics give -- bins are independent, only the
// schedule changes).  A strip's cost is roughly its list length, and a launch ends when its last wave does: started in
// frame order, a 60-entry strip picked up near the end ran on alone for a fifth of the kernel (58 -> 48 us).  The sort is
// done by ONE wavefront with 256 words of LDS, from inside k_composite_tiles (see there)
// (round 3: ONE CLASS of bins per wave -- the bins with index congruent to `cls` mod 8, written to the positions congruent to
// cls mod 8 of `order`.  The compositor hands position p to XCD p % 8 and k_bin_draws builds bin b's list on XCD b % 8: with the
// order sorted per class, a bin's list, its count and -- next frame -- its order entry are read on the XCD whose L2 they were
// written in, instead of being pulled across XCDs at the very start of every strip's life.)
// (round 6: the wave also leaves, in pinned host memory, how many bins of its class hold at least `deep_min` draws -- the exclusive
// prefix of that bucket of the descending histogram.  The host sizes the NEXT launches' deep-strip part from it: k_composite_deep.)
__device__ __forceinline__ void order_bins_wave(const uint32_t* __restrict__ counts, int* __restrict__ order, int nx, int nb, uint32_t* offs, int lane, int cls,
                                                int deep_min = 0, uint32_t* __restrict__ deep_out = nullptr) {
#pragma unroll
  for (int k = 0; k < 4; k++) offs[lane + 64 * k] = 0;
  __builtin_amdgcn_wave_barrier();
  constexpr int kU = 8;
  const int nc = (nb - cls + 7) >> 3;  // bins of this class: cls, cls + 8, ...
  for (int i0 = lane; i0 < nc; i0 += 64 * kU) {
    uint32_t c[kU];
#pragma unroll
    for (int k = 0; k < kU; k++) c[k] = i0 + 64 * k < nc ? counts[cls + 8 * (i0 + 64 * k)] : 0xffffffffu;
#pragma unroll
    for (int k = 0; k < kU; k++) if (c[k] != 0xffffffffu) atomicAdd(&offs[255u - min(c[k], 255u)], 1u);
  }
  __builtin_amdgcn_wave_barrier();
  {
    const uint32_t a = offs[4 * lane], b = offs[4 * lane + 1], c2 = offs[4 * lane + 2], d = offs[4 * lane + 3];
    uint32_t incl = a + b + c2 + d;
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
      const uint32_t up = __shfl_up(incl, sh, 64);
      if (lane >= sh) incl += up;
    }
    const uint32_t base = incl - (a + b + c2 + d);
    __builtin_amdgcn_wave_barrier();
    offs[4 * lane] = base; offs[4 * lane + 1] = base + a; offs[4 * lane + 2] = base + a + b; offs[4 * lane + 3] = base + a + b + c2;
  }
  __builtin_amdgcn_wave_barrier();
  // bucket k holds the bins with min(count, 255) = 255 - k: counts >= m are buckets 0 .. 255 - m, the start of bucket 256 - m
  if (deep_out != nullptr && lane == 0) {
    const int m = min(max(deep_min, 1), 255);
    __hip_atomic_store(deep_out + cls, offs[256 - m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __builtin_amdgcn_wave_barrier();
  for (int i0 = lane; i0 < nc; i0 += 64 * kU) {
    uint32_t c[kU];
#pragma unroll
    for (int k = 0; k < kU; k++) c[k] = i0 + 64 * k < nc ? counts[cls + 8 * (i0 + 64 * k)] : 0xffffffffu;
#pragma unroll
    for (int k = 0; k < kU; k++)
      if (c[k] != 0xffffffffu) {  // stored as (row << 16 | column) of the launch's bin grid: the reader is spared a division
        const int i = cls + 8 * (i0 + 64 * k), row = i / nx;
        order[8 * (int)atomicAdd(&offs[255u - min(c[k], 255u)], 1u) + cls] = (row << 16) | (i - row * nx);
      }
  }
}

// ------------------------------------------------------------------ compositing

struct Frag {
  float u, v;     // interpolated quad uv
  F4 col;         // interpolated vertex colour, 0..1
  float fw_u, fw_v, lod;
  bool covered;
};

// per-triangle affine interpolation of the four vertex colours on an axis-aligned quad:
// triangles (TL,BL,BR) and (TR,TL,BR) (glcontext.nim:418-429); s,t = quad-normalised x, y-down
__device__ __forceinline__ float tri_upper(float tl, float br, float tr, float s, float t) { return tl + (tr - tl) * s + (br - tr) * t; }
__device__ __forceinline__ float tri_lower(float tl, float bl, float br, float s, float t) { return tl + (bl - tl) * t + (br - bl) * s; }
__device__ __forceinline__ float tri_lerp(float tl, float bl, float br, float tr, float s, float t) {
  const float upper = tri_upper(tl, br, tr, s, t), lower = tri_lower(tl, bl, br, s, t);  // (both, then a select)
  return (s > t) ? upper : lower;
}

__device__ __forceinline__ Frag make_frag(const DrawRec& r, const QuadExt* __restrict__ exts, int px, int py) {
  Frag f;
  const uint32_t om = r.op_mode;
  const uint32_t mode = om & 255u;
  const bool atlas_mode = (mode == 0u) || (mode >= 13u && mode <= 16u);
  // vertex uvs: BL=(at.x,to.y) BR=(to.x,to.y) TR=(to.x,at.y) TL=(at.x,at.y); SDF quads use (0,0)-(1,1)
  const float uax = atlas_mode ? r.r[0] : 0.0f, uay = atlas_mode ? r.r[1] : 0.0f;
  const float utx = atlas_mode ? r.r[2] : 1.0f, uty = atlas_mode ? r.r[3] : 1.0f;
  f.covered = px >= r.bx0 && px < r.bx1 && py >= r.by0 && py < r.by1;
  if (!(om & F_GENERAL)) {
    float s = ((float)px + 0.5f - r.ox) * r.inv_w;
    float t = ((float)py + 0.5f - r.oy) * r.inv_h;
    f.u = uax + (utx - uax) * s;
    f.v = uay + (uty - uay) * t;
    if (om & F_SOLID) {
      F4 c = unpack255(r.col[0]);
      const float k = 1.0f / 255.0f;
      f.col = {c.x * k, c.y * k, c.z * k, c.w * k};
    } else {
      F4 bl = unpack255(r.col[0]), br = unpack255(r.col[1]), tr = unpack255(r.col[2]), tl = unpack255(r.col[3]);
      const float k = 1.0f / 255.0f;
      f.col.x = tri_lerp(tl.x, bl.x, br.x, tr.x, s, t) * k;
      f.col.y = tri_lerp(tl.y, bl.y, br.y, tr.y, s, t) * k;
      f.col.z = tri_lerp(tl.z, bl.z, br.z, tr.z, s, t) * k;
      f.col.w = tri_lerp(tl.w, bl.w, br.w, tr.w, s, t) * k;
    }
    f.fw_u = __builtin_fabsf((utx - uax) * r.inv_w);
    f.fw_v = __builtin_fabsf((uty - uay) * r.inv_h);
    f.lod = r.aux2;
    return f;
  }
  // general quad: exact integer edge functions in half-pixel units, top-left rule
  const QuadExt& q = exts[r.ext];
  const int X = 2 * px + 1, Y = 2 * py + 1;
  int hit = -1;
  long long e0 = 0, e1 = 0, e2 = 0;
#pragma unroll
  for (int t = 0; t < 2; t++) {
    if (hit >= 0 || q.inv_sum[t] == 0.0f) continue;
    long long a0 = (long long)q.e[t][0].a * X + (long long)q.e[t][0].b * Y + q.e[t][0].c;
    long long a1 = (long long)q.e[t][1].a * X + (long long)q.e[t][1].b * Y + q.e[t][1].c;
    long long a2 = (long long)q.e[t][2].a * X + (long long)q.e[t][2].b * Y + q.e[t][2].c;
    bool in = a0 >= 0 && a1 >= 0 && a2 >= 0;
    uint32_t own = q.own >> (t * 3);
    if (a0 == 0 && !(own & 1u)) in = false;
    if (a1 == 0 && !(own & 2u)) in = false;
    if (a2 == 0 && !(own & 4u)) in = false;
    if (in) { hit = t; e0 = a0; e1 = a1; e2 = a2; }
  }
  f.covered = f.covered && hit >= 0;
  const int t = hit < 0 ? 0 : hit;
  const float is = q.inv_sum[t];
  const float l0 = (float)e0 * is, l1 = (float)e1 * is, l2 = (float)e2 * is;
  // triangle 0 = vertices (3,0,1) = (TL,BL,BR); triangle 1 = (2,3,1) = (TR,TL,BR)
  const float u0 = t == 0 ? uax : utx, v0 = t == 0 ? uay : uay;   // TL | TR
  const float u1 = t == 0 ? uax : uax, v1 = t == 0 ? uty : uay;   // BL | TL
  const float u2 = utx, v2 = uty;                                 // BR | BR
  f.u = l0 * u0 + l1 * u1 + l2 * u2;
  f.v = l0 * v0 + l1 * v1 + l2 * v2;
  F4 c0 = unpack255(t == 0 ? r.col[3] : r.col[2]);
  F4 c1 = unpack255(t == 0 ? r.col[0] : r.col[3]);
  F4 c2 = unpack255(r.col[1]);
  const float k = 1.0f / 255.0f;
  f.col.x = (l0 * c0.x + l1 * c1.x + l2 * c2.x) * k;
  f.col.y = (l0 * c0.y + l1 * c1.y + l2 * c2.y) * k;
  f.col.z = (l0 * c0.z + l1 * c1.z + l2 * c2.z) * k;
  f.col.w = (l0 * c0.w + l1 * c1.w + l2 * c2.w) * k;
  f.fw_u = q.fw_u[t];
  f.fw_v = q.fw_v[t];
  f.lod = q.lod[t];
  return f;
}

// fixed-function blend SRC_ALPHA/ONE_MINUS_SRC_ALPHA (rgb), ONE/ONE_MINUS_SRC_ALPHA (alpha), then the RGBA8
// store (utils/glutils.nim:150-154).  F holds the framebuffer texel as 0..255 integers in floats.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void blend(F4& F, float r, float g, float b, float sa) {
  const float ia = 1.0f - sa, A = 255.0f * sa;
  F.x = __builtin_rintf(__builtin_fmaf(F.x, ia, r * A));
  F.y = __builtin_rintf(__builtin_fmaf(F.y, ia, g * A));
  F.z = __builtin_rintf(__builtin_fmaf(F.z, ia, b * A));
  F.w = __builtin_rintf(__builtin_fmaf(F.w, ia, A));
}
// the same blend with the source term (rgb * 255 sa, 255 sa) and 1 - sa already formed: four FMAs per pixel (written on
// float2 pairs for packed FMAs once; the library is built without packed-FP32 instructions, see csrc/Makefile)
__device__ __forceinline__ void blend_pre(F4& F, f2 c_rg, f2 c_ba, float ia) {
  f2 xy = {F.x, F.y}, zw = {F.z, F.w};
  const f2 ia2 = {ia, ia};
  xy = __builtin_elementwise_fma(xy, ia2, c_rg);
  zw = __builtin_elementwise_fma(zw, ia2, c_ba);
  F.x = __builtin_rintf(xy.x); F.y = __builtin_rintf(xy.y); F.z = __builtin_rintf(zw.x); F.w = __builtin_rintf(zw.y);
}

// a black source: the r, g, b terms of the source are +0
__device__ __forceinline__ void blend_black(F4& F, float A, float ia) {
  F.x = __builtin_rintf(F.x * ia); F.y = __builtin_rintf(F.y * ia); F.z = __builtin_rintf(F.z * ia); F.w = __builtin_rintf(__builtin_fmaf(F.w, ia, A));
}

// atlas_rect_mask.frag:222-237
__device__ __forceinline__ float rect_mask_alpha(const DrawRec& r, float cx, float cy) {
  float lx = (r.ox * cx + r.oy * cy) + r.inv_w;
  float ly = (r.inv_h * cx + r.f0 * cy) + r.f1;
  float qx = lx - r.p0, qy = ly - r.p1;
  float dist = shape_dist((r.op_mode & F_ELLIP) != 0u, qx, -qy, r.p2, r.p3, r.r[0], r.r[1], r.r[2], r.r[3]);
  return 1.0f - clamp01(r.aa * dist + 0.5f);
}

// Wave-uniform record fetch: 8 x 16-byte scalar loads issued together so one s_waitcnt covers them all.
__device__ __forceinline__ DrawRec load_rec(const DrawRec* __restrict__ p) {
  DrawRec r;
  const uint4* __restrict__ src = reinterpret_cast<const uint4*>(p);
  uint4* dst = reinterpret_cast<uint4*>(&r);
#pragma unroll
  for (int i = 0; i < 8; i++) dst[i] = src[i];
  return r;
}

// The same with every field pinned in SGPRs at this point: the compiler may not sink part of the fetch into the branches
// that use it (a second round trip to L2 per draw)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ DrawRec load_rec_whole(const DrawRec* __restrict__ p) {
  const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(p);
  u32x4 q[8];
#pragma unroll
  for (int i = 0; i < 8; i++) q[i] = src[i];
  // (128-bit operands: the values stay in the aligned SGPR tuples the loads wrote; pinned dword by dword the compiler
  // reshuffled them with ~20 s_mov_b32 per record)
  asm volatile("" : "+s"(q[0]), "+s"(q[1]), "+s"(q[2]), "+s"(q[3]), "+s"(q[4]), "+s"(q[5]), "+s"(q[6]), "+s"(q[7]));
  DrawRec r;  // (field by field: a cast of &r to a vector pointer parks the record in scratch)
  const auto f = [](uint32_t v) { return __uint_as_float(v); };
  r.op_mode = q[0].x; r.ext = q[0].y; r.ox = f(q[0].z); r.oy = f(q[0].w);
  r.inv_w = f(q[1].x); r.inv_h = f(q[1].y); r.p0 = f(q[1].z); r.p1 = f(q[1].w);
  r.p2 = f(q[2].x); r.p3 = f(q[2].y); r.f0 = f(q[2].z); r.f1 = f(q[2].w);
  r.r[0] = f(q[3].x); r.r[1] = f(q[3].y); r.r[2] = f(q[3].z); r.r[3] = f(q[3].w);
  r.col[0] = q[4].x; r.col[1] = q[4].y; r.col[2] = q[4].z; r.col[3] = q[4].w;
  r.mid = q[5].x; r.stop = q[5].y; r.aa = f(q[5].z); r.aux = f(q[5].w);
  r.aux2 = f(q[6].x);
  r.bx0 = (int16_t)(q[6].y & 0xffffu); r.by0 = (int16_t)(q[6].y >> 16); r.bx1 = (int16_t)(q[6].z & 0xffffu); r.by1 = (int16_t)(q[6].z >> 16);
  r.ix0 = (int16_t)(q[6].w & 0xffffu); r.iy0 = (int16_t)(q[6].w >> 16); r.ix1 = (int16_t)(q[7].x & 0xffffu); r.iy1 = (int16_t)(q[7].x >> 16);
  r.kx = f(q[7].y); r.ky = f(q[7].z); r._pad = q[7].w;
  return r;
}
static_assert(LE_PLAIN == 0x80000000u, "the compositor tests LE_PLAIN as the sign bit");
static_assert(offsetof(DrawRec, p2) == 32 && offsetof(DrawRec, col) == 64 && offsetof(DrawRec, aux2) == 96 && offsetof(DrawRec, bx0) == 100 && offsetof(DrawRec, ix0) == 108 && offsetof(DrawRec, aa) == 88 && offsetof(DrawRec, kx) == 116, "load_rec_whole follows DrawRec's layout");

// Local-frame coordinates of a lane's pixels on an axis-aligned SDF quad (atlas.frag:252-262: p = (uv - 0.5) * 2 * quadHalfExtents,
// uv = (pixel centre - quad origin) / quad extent).  Pixel 0 follows the shader's own operations ((c - o) * inv - 0.5, times 2 p);
// its neighbours are pixel 0 plus multiples of the per-pixel step kx = 2 p0 inv_w the host put into the record -- 6
// instructions for the four x, 3 for y, where the shader's formula per pixel costs 4 each.  Every path of every build uses these
// (the builds must agree to the bit: tests/test_hip_parity.py::test_every_kernel_build_gives_the_same_pixels).
__device__ __forceinline__ void local_x4(const DrawRec& r, float cx0, float (&lx)[4]) {
  lx[0] = __builtin_fmaf(cx0 - r.ox, r.inv_w, -0.5f) * (2.0f * r.p0);
  lx[1] = lx[0] + r.kx;
  lx[2] = __builtin_fmaf(2.0f, r.kx, lx[0]);
  lx[3] = __builtin_fmaf(3.0f, r.kx, lx[0]);
}
__device__ __forceinline__ float local_y_up(const DrawRec& r, float cy) {  // -ly: the shader flips y (p.y = -p.y, atlas.frag:262)
  return __builtin_fmaf(r.oy - cy, r.inv_h, 0.5f) * (2.0f * r.p1);
}
// 1 - clamp(aa d + 0.5, 0, 1) (atlas.frag:389-393) as ONE instruction: clamp(0.5 - aa d, 0, 1) = v_fma with the clamp modifier
__device__ __forceinline__ float cover_aa(float d, float aa) { return clamp01(__builtin_fmaf(-d, aa, 0.5f)); }

// ---- branch-free shape distance: no per-lane exec juggling (divergent control flow is paid in
// s_and_saveexec/s_or sequences on the CU's single scalar unit).  The two-sqrt ellipse evaluation is skipped with
// ONE wave-uniform branch when no lane of the wave sits in an elliptical corner region.
struct Corner { float rx, ry; bool same; };
__device__ __forceinline__ Corner pick_corner(bool ellip, float px, float py, float bx, float by, float r0, float r1, float r2, float r3) {
  const float sel = (px > 0.0f) ? ((py > 0.0f) ? r0 : r1) : ((py > 0.0f) ? r2 : r3);
  Corner c;
  if (!ellip) { c.rx = sel; c.ry = sel; c.same = true; return c; }  // wave-uniform
  // decodeEllipticalCornerRadii atlas.frag:88-94; negative = circular corner of radius -v-1 (:98-100)
  const float pv = __builtin_floorf(sel + 0.5f);
  const float hi = __builtin_floorf(pv * (1.0f / 4096.0f));
  float rx = (pv - 4096.0f * hi) * bx * (1.0f / 4095.0f);
  float ry = hi * by * (1.0f / 4095.0f);
  const bool circle = sel < 0.0f;
  const float rc = -sel - 1.0f;
  rx = circle ? rc : rx;
  ry = circle ? rc : ry;
  const bool zero = rx <= 0.0f || ry <= 0.0f;  // :102-105 -> plain box, i.e. the rounded-box formula with r = 0
  c.rx = zero ? 0.0f : rx;
  c.ry = zero ? 0.0f : ry;
  c.same = c.rx == c.ry;
  return c;
}
__device__ __forceinline__ float sd_ellipse_nb(float px, float py, float rx, float ry) {  // atlas.frag:71-79 without branches
  const float sx = __builtin_fmaxf(rx, 0.000001f), sy = __builtin_fmaxf(ry, 0.000001f);
  const float isx = frcp(sx), isy = frcp(sy);
  const float ax = px * isx, ay = py * isy;
  const float k0 = fsqrt(ax * ax + ay * ay);
  const float bx = ax * isx, by = ay * isy;
  const float k1 = fsqrt(bx * bx + by * by);
  const float d = k0 * (k0 - 1.0f) * frcp(__builtin_fmaxf(k1, 0.000001f));
  const float inner = -__builtin_fminf(sx, sy);  // (both arms are plain values: see the note on selects at shadow_profile)
  return k0 <= 0.000001f ? inner : d;
}
// distance of N pixels at once (sdRoundedBox :51-69 / sdEllipticalRoundedBox :96-115): of one row (kPerY = false: they share
// their local y, *pyv) or each with a local y of its own (rotated quads: pyv[k])
template <int N, bool kPerY>
__device__ __forceinline__ void shape_distNy(bool ellip, const float* px, const float* pyv, float bx, float by, float r0, float r1, float r2,
                                             float r3, float* out) {
  Corner c[N];
  float qx[N], qy[N];
  bool need = false, diag = false;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const float py = pyv[kPerY ? k : 0];
    c[k] = pick_corner(ellip, px[k], py, bx, by, r0, r1, r2, r3);
    qx[k] = __builtin_fabsf(px[k]) - bx + c[k].rx;
    qy[k] = __builtin_fabsf(py) - by + c[k].ry;
    const bool corner = qx[k] > 0.0f && qy[k] > 0.0f;
    diag = diag || corner;
    need = need || (!c[k].same && corner);
  }
  // length(max(q, 0)) only needs the (quarter-rate) sqrt where BOTH components are positive, i.e. in the corner
  // arcs; elsewhere it is max(qx, qy, 0).  One wave-uniform branch skips the sqrt for every strip without an arc.
  const bool any_diag = __any(diag);
  // (min(max(q.x, q.y), 0) + length(max(q, 0)) = (both components positive ? |q| : max(q.x, q.y)), exactly: see dist4 in
  // k_composite_tiles)
#pragma unroll
  for (int k = 0; k < N; k++) {
    float m = __builtin_fmaxf(qx[k], qy[k]);
    if (any_diag) {  // wave-uniform
      const float e = fsqrt(qx[k] * qx[k] + qy[k] * qy[k]);
      m = (qx[k] > 0.0f && qy[k] > 0.0f) ? e : m;
    }
    out[k] = m - c[k].rx;
  }
  if (!ellip) return;  // wave-uniform
  const bool any_ellipse = __any(need);
#pragma unroll
  for (int k = 0; k < N; k++) {
    const bool corner = qx[k] > 0.0f && qy[k] > 0.0f;
    float de = __builtin_fmaxf(qx[k] - c[k].rx, qy[k] - c[k].ry);
    if (any_ellipse) {  // wave-uniform
      const float e = sd_ellipse_nb(qx[k], qy[k], c[k].rx, c[k].ry);
      de = corner ? e : de;
    }
    out[k] = c[k].same ? out[k] : de;
  }
}
template <int N>
__device__ __forceinline__ void shape_distN(bool ellip, const float* px, float py, float bx, float by, float r0, float r1, float r2,
                                            float r3, float* out) {
  shape_distNy<N, false>(ellip, px, &py, bx, by, r0, r1, r2, r3, out);
}

// evalFillColor atlas.frag:233-250, select-based.  Everything is passed BY VALUE: `c ? a.x : b.x` on lvalues is an
// lvalue conditional (pointer select, then a load), which pins arrays of F4 in scratch.
__device__ __forceinline__ float selectf(bool c, float a, float b) { return c ? a : b; }
__device__ __forceinline__ float fill_t(uint32_t fill_mode, float u, float v) {  // the gradient parameter, clamped (atlas.frag:236-243)
  float t;
  switch (fill_mode) {  // wave-uniform
    case 1u: t = u; break;
    case 2u: t = v; break;
    case 3u: t = 0.5f * (u + v); break;
    default: t = 0.5f * (u + (1.0f - v)); break;
  }
  return clamp01(t);
}
__device__ __forceinline__ F4 eval_fill_nb(F4 col, F4 m, F4 s, uint32_t fill_mode, float mid, float u, float v) {
  const float t = fill_t(fill_mode, u, v);
  const bool lo = t <= mid;
  const float w = selectf(lo, t * frcp(mid), (t - mid) * frcp(1.0f - mid));
  F4 o;
  o.x = mixf(selectf(lo, col.x, m.x), selectf(lo, m.x, s.x), w);
  o.y = mixf(selectf(lo, col.y, m.y), selectf(lo, m.y, s.y), w);
  o.z = mixf(selectf(lo, col.z, m.z), selectf(lo, m.z, s.z), w);
  o.w = mixf(selectf(lo, col.w, m.w), selectf(lo, m.w, s.w), w);
  return o;
}

__device__ __forceinline__ F4 eval_fill_rec(const DrawRec& r, F4 col, uint32_t fill_mode, float u, float v) {
  if (fill_mode == 0u) return col;
  const float k = 1.0f / 255.0f;
  const F4 mc = unpack255(r.mid), sc = unpack255(r.stop);
  const F4 m01 = {mc.x * k, mc.y * k, mc.z * k, mc.w * k}, s01 = {sc.x * k, sc.y * k, sc.z * k, sc.w * k};
  return eval_fill_nb(col, m01, s01, fill_mode, __builtin_fminf(__builtin_fmaxf(r.f1, 0.01f), 0.99f), u, v);
}

// ---- generic one-pixel shading (atlas / MSDF sampling, rotated or skewed quads, rect-mask setup): the rare draws.
// Reads the record through the global pointer (dynamic field selection must not force a local copy into scratch).
struct Src { float r, g, b, a; bool covered; };
__device__ __forceinline__ Src shade_one(const DrawRec* __restrict__ rp, const QuadExt* __restrict__ exts, const AtlasView* __restrict__ atlas,
                                      const uint32_t* __restrict__ backdrop, size_t pix, bool in_frame, int px, int py, F4 F) {
  const DrawRec& r = *rp;
  const uint32_t om = r.op_mode;
  const uint32_t mode = om & 255u;
  const bool ellip = (om & F_ELLIP) != 0u;
  const uint32_t fill_mode = (om >> 9) & 7u;
  const Frag f = make_frag(r, exts, px, py);
  Src s;
  s.covered = f.covered;
  if (((om >> 12) & 15u) == OP_MASK_PUSH) {  // mask.frag:186-234: returns the shape alpha (x colour alpha) in .a
    const float lx = (f.u - 0.5f) * 2.0f * r.p0, ly = (f.v - 0.5f) * 2.0f * r.p1;
    const float dist = shape_dist(ellip, lx, -ly, r.p2, r.p3, r.r[0], r.r[1], r.r[2], r.r[3]);
    s.r = s.g = s.b = 0.0f;
    s.a = (1.0f - clamp01(r.aa * dist + 0.5f)) * f.col.w;
    return s;
  }
  if (mode == 0u) {  // atlas.frag:284-295
    float u = f.u;
    if (om & F_SUBPIXEL) u -= r.aux * frcp(__builtin_fmaxf((float)atlas->size, 1.0f));
    const F4 t = atlas_sample(*atlas, u, f.v, f.lod);
    s.r = t.x * f.col.x; s.g = t.y * f.col.y; s.b = t.z * f.col.z; s.a = t.w * f.col.w;
    return s;
  }
  if (mode >= 13u && mode <= 16u) {  // atlas.frag:296-318
    const F4 fc = eval_fill_rec(r, f.col, fill_mode, f.u, f.v);
    const F4 t = atlas_sample(*atlas, f.u, f.v, 0.0f);  // textureLod(atlasTex, uv, 0.0)
    const bool is_mtsdf = (mode == 14u || mode == 16u), is_stroke = (mode == 15u || mode == 16u);
    const float sd = is_mtsdf ? t.w : median3(t.x, t.y, t.z);
    const float unit = r.f0 * frcp(r.p0);  // pxRange / atlas size (atlas.frag:45-49)
    const float spr = __builtin_fmaxf(0.5f * (unit * frcp(f.fw_u) + unit * frcp(f.fw_v)), 1.0f);
    const float spd = spr * (sd - r.f1);
    const float alpha = is_stroke ? clamp01(__builtin_fmaxf(r.p1, 0.0f) * 0.5f - __builtin_fabsf(spd) + 0.5f) : clamp01(spd + 0.5f);
    s.r = fc.x; s.g = fc.y; s.b = fc.z; s.a = fc.w * alpha;
    return s;
  }
  const float qhx = r.p0, qhy = r.p1;
  const bool inset = mode == 9u;
  const float shx = inset ? qhx : r.p2, shy = inset ? qhy : r.p3;
  const float lx = (f.u - 0.5f) * 2.0f * qhx, ly = (f.v - 0.5f) * 2.0f * qhy;
  const bool bezier = mode >= 18u && mode <= 20u;  // isBezierStrokeMode atlas.frag:162-168 (p is NOT y-flipped here)
  const float dist = bezier ? sd_bezier(lx, ly, r.p2, r.p3, r.r[0], r.r[1], r.r[2], r.r[3])
                            : shape_dist(ellip, lx, -ly, shx, shy, r.r[0], r.r[1], r.r[2], r.r[3]);
  const float spread = fill_mode == 0u ? r.f1 : 0.0f;
  float alpha;
  switch (mode) {
    case 18u: case 19u: case 20u: {  // atlas.frag:321-336
      const float sd = bezier_stroke_sd(dist, lx, ly, r.p2, r.p3, r.r[0], r.r[1], r.r[2], r.r[3], __builtin_fmaxf(r.f0, 0.0f) * 0.5f, mode);
      alpha = 1.0f - clamp01(r.aa * sd + 0.5f);
      break;
    }
    case 11u: { float h = r.f0 * 0.5f; float sd = __builtin_fabsf(dist + h) - h; alpha = sd < 0.0f ? 1.0f : 0.0f; break; }
    case 12u: { float h = r.f0 * 0.5f; float sd = __builtin_fabsf(dist + h) - h; alpha = 1.0f - clamp01(r.aa * sd + 0.5f); break; }
    case 7u: { float sd = dist - spread; const float sp = __builtin_fminf(shadow_profile(sd, r.f0), 1.0f); alpha = sd > 0.0f ? sp : 1.0f; break; }
    case 8u: {
      float inside = 1.0f - clamp01(r.aa * dist + 0.5f);
      float sd = dist - spread;
      const float sp = __builtin_fminf(shadow_profile(sd, r.f0), 1.0f);
      alpha = sd >= 0.0f ? sp : inside;
      break;
    }
    case 9u: {  // atlas.frag:364-380
      float clip_a = 1.0f - clamp01(r.aa * dist + 0.5f);
      float shd = shape_dist(ellip, lx - r.p2, -ly + r.p3, qhx, qhy, r.r[0], r.r[1], r.r[2], r.r[3]);
      float sd = shd + spread;
      const float sp = __builtin_fminf(shadow_profile(sd, r.f0), 1.0f);
      float ia = sd < 0.0f ? sp : 1.0f;
      alpha = clip_a * ia;
      break;
    }
    default: alpha = 1.0f - clamp01(r.aa * dist + 0.5f); break;
  }
  if (mode == 17u) {  // atlas.frag:381-388
    F4 b = F;
    if (!(om & F_SELF_BACKDROP) && in_frame) b = unpack255(backdrop[pix]);
    const float k = 1.0f / 255.0f;
    s.r = b.x * k; s.g = b.y * k; s.b = b.z * k; s.a = b.w * k * alpha;
  } else {
    const F4 fc = eval_fill_rec(r, f.col, fill_mode, f.u, f.v);
    s.r = fc.x; s.g = fc.y; s.b = fc.z; s.a = fc.w * alpha;
  }
  return s;
}

// One wavefront = one 32x8 pixel tile: lane l owns the 4 horizontally adjacent pixels x = tx0 + 4*(l&7) .. +3 of
// row ty0 + (l>>3) -- i.e. four side-by-side 8x8 sub-tiles shaded in lock-step, so every record fetch, mode
// dispatch and loop step is paid once per 256 pixels, loads/stores of the surface are 16 B per lane and a wave
// reads or writes 8 full 128-byte lines.  A workgroup is ONE wavefront; the sixteen strips of a 64x64 bin are sixteen
// consecutive workgroups of an XCD.
//
// Axis-aligned SDF draws (fills, strokes, shadows, clip pushes, blur composites -- all but a handful of calls in
// real scenes) take the 4-wide straight-line path.  Everything else goes through shade_one() one pixel slot at a
// time; the per-lane state arrays are rotated between slots so they are only ever indexed statically.
#ifndef FDH_TIMING
#define FDH_TIMING 0  // `make variant SINGLE=1 DEFS="-DFDH_STATS=1 -DFDH_TIMING=1"`: per-wave phase times (shader cycles) instead of counts
#endif
#if FDH_TIMING
#define FDH_NOW() clock64()
#endif
#if FDH_STATS
__device__ unsigned long long g_wave_times[16 * 65536];  // FDH_TIMING: one row per wave (no atomics: they would serialise)
__device__ unsigned long long g_counters[128];
#if FDH_TIMING
#define FDH_COUNT(i) do { } while (0)
#else
#define FDH_COUNT(i) do { if (lane == 0) atomicAdd(&g_counters[(i)], 1ull); } while (0)
#endif
#else
#define FDH_COUNT(i) do { } while (0)
#endif
// The builds without the one-pixel-slot path (phases made only of axis-aligned SDF draws, clips, axis-aligned atlas quads:
// no rotated quads, no bezier strokes, no rect-mask setup) need no scratch: 80 - 96 VGPRs, five or six waves per SIMD.
#ifndef FDH_EDGE_CHECK
#define FDH_EDGE_CHECK 0  // experiment builds only: every packed edge strip is also shaded by the generic path and compared
#endif
#if FDH_EDGE_CHECK
__device__ unsigned int g_edge_bad_n;
__device__ unsigned int g_edge_bad[4096 * 8];
#endif
#ifndef FDH_FAST_WAVES
#define FDH_FAST_WAVES 5
#endif
#ifndef FDH_ATLAS_WAVES
#define FDH_ATLAS_WAVES 5  // waves per SIMD of the atlas build <2>
#endif
#ifndef FDH_ROT_ATLAS4
#define FDH_ROT_ATLAS4 1  // 0 (experiment builds): rotated atlas quads stay on the one-pixel-slot path
#endif
#ifndef FDH_BEZIER4
#define FDH_BEZIER4 1  // 0 (experiment builds): bezier strokes stay on the one-pixel-slot path with libm's functions
#endif
#ifndef FDH_SLOW_WAVES
#define FDH_SLOW_WAVES 4  // waves per SIMD of the build with every path <3>
#endif
#ifndef FDH_ROT_WAVES
#define FDH_ROT_WAVES 4  // waves per SIMD of the rotated-quad build <8>
#endif
#ifndef FDH_DEEP_PRIO
#define FDH_DEEP_PRIO 1  // a deep strip's waves raise their issue priority (s_setprio: blender 3, shaders 2)
#endif
#ifndef FDH_UNIFORM_WAVES
#define FDH_UNIFORM_WAVES 6  // waves per SIMD of the no-clip build <4>: 80 VGPRs, no spills
#endif
// the strip's texel window in LDS (builds with the atlas path): up to kWinCols x kWinRows texels, rows kWinStride dwords apart
// (a multiple of four, for the 16-byte stores, that is not a multiple of 32: rows start in different banks)
constexpr int kWinCols = 64, kWinRows = 12, kWinStride = 68;
// kPaths: bit 3 = the 4-wide path for rotated / skewed SDF quads (F_EDGE32) on top of the axis-aligned SDF paths, with clip masks:
// build <8>, for phases whose only draws off the fast paths are such quads (a rotated panel does not drag the slot path in);
// bit 0 = the one-pixel-slot path (rotated / skewed quads, bezier strokes, rect-mask setup, minified images),
// bit 1 = the 4-wide atlas path (axis-aligned glyphs, images at >= 1:1, MSDF).  0: SDF draws, clips and rect masks only.
// kFull: the launch that starts a frame -- every bin of the grid, from the clear colour (nothing is loaded), bins taken longest
// list first, with the sort for the next frame riding along.  A symbol of its own, so that the dominant launch of a frame is a
// row of its own in a rocprofv3 kernel summary (the later phases' launches cover a blur node's footprint and take microseconds).
// ---- Deep strips (round 6).  A wave walks its strip's list one draw after the other, and alone on a SIMD it gets through a draw's ~200
// dependent instructions no faster than with five neighbours: at 1920 x 1080 the bench tree's lists are four times as deep as at 4K, and
// the full-frame launch was the serial chain of its longest strips -- the 128 strips of the eight longest bins, shaded with NOTHING else on
// the chip, take the launch's whole 29 us (profiles/r06_1080p_critical_path.txt).  What is serial in a strip is only the BLEND -- a draw's
// source term (coverage from the distance field, colour from the fill) depends on nothing before it.  So the strips of the frame's deepest
// bins get a workgroup of four waves (k_composite_deep): waves 1..3 (kRole 2, "shaders") each take every third draw that needs per-pixel
// work, evaluate its source term with the code below and put it into a ring of slots in LDS; wave 0 (kRole 1, the "blender") walks the same
// list, blends the one-colour core strips itself and every other draw's source term out of the ring, in list order.  Same operations on the
// same values as one wave would do (the blender's arithmetic is the tail of edge_blend / shade, moved): bit-identical frames.
// Slot payload per lane: packed edge paths 4 floats (the four source alphas; the draw's colour rides in the slot's header), the generic path
// 16 (r, g, b, masked alpha of the four pixels).
constexpr int kDeepSlots = 6;                               // ring depth: source terms a strip's shaders may be ahead of its blender
constexpr int kDeepSlotFloats = 16 * 64;                    // 4 KB of payload per slot: [float index 0..15][lane]
constexpr int kDeepHdr = 8;                                 // dwords of header per slot: tag, colour words / uniform terms
[[maybe_unused]] constexpr int kDeepLdsDwords = kDeepSlots * (kDeepSlotFloats + kDeepHdr) + kDeepSlots + 2;  // + ready[] + consumed
constexpr uint32_t DT_NOP = 0, DT_PACKED = 1, DT_PACKED_BLACK = 2, DT_GENERIC = 3, DT_SELF17 = 4, DT_UNIFORM_PRE = 5;
struct DeepRing {
  float* data;         // [slot][16][64]
  uint32_t* hdr;       // [slot][kDeepHdr]
  uint32_t* ready;     // [slot]: rank + 1 of the source term the slot holds
  uint32_t* consumed;  // ranks the blender is done with
  __device__ __forceinline__ explicit DeepRing(uint32_t* lds)
      : data(reinterpret_cast<float*>(lds)), hdr(lds + kDeepSlots * kDeepSlotFloats), ready(lds + kDeepSlots * (kDeepSlotFloats + kDeepHdr)),
        consumed(lds + kDeepSlots * (kDeepSlotFloats + kDeepHdr) + kDeepSlots) {}
};
// (a wait that can never be satisfied must not hang the device: after ~2^20 polls a wave goes on -- wrong pixels, which the tests see)
__device__ __forceinline__ void deep_wait_ge(const uint32_t* p, const uint32_t want) {
  for (int spins = 0; spins < (1 << 20); spins++) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
    if ((int32_t)(v - want) >= 0) return;
    __builtin_amdgcn_s_sleep(1);
  }
}

// How many draws of its list a strip has to SHADE -- survivors of the strip test and of the occlusion cut that are not one-colour core strips
// (those are a uniform blend: cheap) --, counted the way the draw loop walks the list.  A strip of a deep bin goes to k_composite_deep when
// this reaches P.deep_strip_min, and to its usual wave otherwise: both kernels ask this function, so they agree.
__device__ __forceinline__ uint32_t strip_shade_count(const CompositeParams& P, const int bin, const int sbit, const int lane) {
  const uint32_t cnt = P.counts[bin];
  const uint2* __restrict__ list = P.lists + (size_t)bin * P.stride;
  uint32_t n = 0;
  for (uint32_t base = 0; base < cnt; base += 64) {
    const uint32_t i = base + lane;
    const uint2 e = list[min(i, cnt - 1u)];
    const uint32_t ey = i < cnt ? e.y : 0u;
    const uint32_t st = (ey >> sbit) & 0x10001u;
    unsigned long long m = __ballot(st != 0u);
    const unsigned long long m_opaque = __ballot(st == 0x10001u && (e.x & LE_OPAQUE) != 0u);
    if (m_opaque != 0) m &= ~((1ull << (63 - __builtin_clzll(m_opaque))) - 1ull);
    m &= ~__ballot(st == 0x10001u && (int32_t)e.x < 0);
    n += (uint32_t)__builtin_popcountll(m);
  }
  return n;
}

template <int kPaths, bool kFull, int kRole>
__device__ __forceinline__ void composite_strip(const CompositeParams& P, const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts,
                                                uint32_t* composite_lds, const int bin, const int sidx, const int sbit, const int tx0, const int ty0,
                                                const int lane, const int shader_id);

template <int kPaths, bool kFull>
__global__ __launch_bounds__(64, (kPaths & 1) ? FDH_SLOW_WAVES : kPaths == 4 ? FDH_UNIFORM_WAVES : (kPaths & 2) ? FDH_ATLAS_WAVES : (kPaths & 8) ? FDH_ROT_WAVES : FDH_FAST_WAVES) void k_composite_tiles(
    // the sixteen dwords a wave needs before anything else, as leading scalar arguments: with kernel-argument preloading
    // (-amdgpu-kernarg-preload-count, csrc/Makefile) they arrive in SGPRs with the wave instead of through a first s_load
    const int* __restrict__ a_order, int* __restrict__ a_order_next, const uint32_t* __restrict__ a_counts, const uint2* __restrict__ a_lists,
    int a_bin_x0, int a_bin_y0, int a_bin_nx, int a_bin_ny, int a_bins_x, int a_stride, int a_row_lo, int a_row_hi,
    const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts, CompositeParams P) {
  P.order = kFull ? a_order : nullptr; P.order_next = kFull ? a_order_next : nullptr; P.counts = a_counts; P.lists = a_lists;
  if (kFull) P.load_fb = 0;
  P.bin_x0 = a_bin_x0; P.bin_y0 = a_bin_y0; P.bin_nx = a_bin_nx; P.bin_ny = a_bin_ny;
  P.bins_x = a_bins_x; P.stride = a_stride; P.row_lo = a_row_lo; P.row_hi = a_row_hi;
  // clip stack: 4 pixels' q8 mask values packed per lane and level.  Dynamic LDS: 4 KB when the phase has clip operations,
  // 1 KB (what the bin-ordering wavefront needs) when it has none -- a phase's waves then fit beside the 18-KB rings of
  // another frame's blur pass on the same CU (20 x 4 KB + 8 x 18 KB do not)
  extern __shared__ uint32_t composite_lds[];
  uint32_t (*mask_stack)[kMaskDepth][64] = reinterpret_cast<uint32_t (*)[kMaskDepth][64]>(composite_lds);
  // XCD-aware mapping: the dispatcher places workgroup b on XCD b % 8.  XCD x takes the bins x, x+8, x+16, ... of this
  // launch (row-major), all 16 strips of a bin back to back: a bin's draw list and records stay in ONE L2, and every
  // XCD gets an even sample of the frame -- contiguous bands per XCD left the XCDs holding the busy rows 3x the work
  // of the ones holding the emptier top and bottom of the frame.
  constexpr int kStripsPerBin = kWgsPerBin * kWavesPerWg;  // 16
  // (a workgroup is ONE wavefront: nothing is shared between strips, and the dispatcher refills wave slots one at a time)
  // (with a sort riding along, the first eight workgroups are its -- one per XCD, each sorts the bins of its class --, dispatched
  // first so they are done long before the launch ends; the numbering of the rest shifts by eight and keeps its XCD phase)
  const int blk = P.order_next ? (int)blockIdx.x - 8 : (int)blockIdx.x;
  const int q = blk >> 3, xcd = blk & 7;
  if (blk < 0) {
    // One extra wavefront per full-frame launch sorts THIS frame's bin counts for the NEXT frame's launch (any
    // permutation is a correct schedule, and list lengths barely change from frame to frame).  As a kernel of its own the
    // sort was a ~6 us serial step of every frame; here it runs beside 32 000 compositing waves.
    order_bins_wave(P.counts, P.order_next, P.bin_nx, P.bin_nx * P.bin_ny, &mask_stack[0][0][0], threadIdx.x, (int)blockIdx.x, P.deep_min, P.deep_out);
    return;
  }
  int bin_local = xcd + 8 * (q / kStripsPerBin);
  const int sidx = q % kStripsPerBin;
  if (bin_local >= P.bin_nx * P.bin_ny) return;
  const int j = sidx >> 2, wave = sidx & 3, lane = threadIdx.x & 63;
  const int sbit = j * 4 + wave;  // this strip's bit in the list entries' strip masks
  int bly, blx;
  if (P.order) {  // longest lists first (order_bins_wave): entries are row << 16 | column
    const int rc = P.order[bin_local];
    bly = rc >> 16; blx = rc & 0xffff;
  } else {
    bly = bin_local / P.bin_nx; blx = bin_local - bly * P.bin_nx;
  }
  const int bin_x = P.bin_x0 + blx, bin_y = P.bin_y0 + bly;
  const int bin = bin_y * P.bins_x + bin_x;
  const int tx0 = bin_x * kBin + (j & 1) * kWgW;
  const int ty0 = bin_y * kBin + (j >> 1) * kWgH + wave * kTileH;
  // (four plain scalar compare-and-branch pairs: written with ||, each pair became two s_cselect_b64 masks, an s_and_b64 and a vcc branch)
  // (the empty asm statements keep the compiler from folding the four exits back into that form)
  if (tx0 >= P.W) return;
  asm volatile("");
  if (ty0 >= P.H) return;
  asm volatile("");
  if (ty0 + kTileH <= P.row_lo) return;
  asm volatile("");
  if (ty0 >= P.row_hi) return;
  composite_strip<kPaths, kFull, 0>(P, draws, exts, composite_lds, bin, sidx, sbit, tx0, ty0, lane, 0);
}

// The full-frame launch of a frame that HAS deep bins (P.deep_k8 > 0; k_composite_tiles<4, true> otherwise), workgroups of four waves:
//   workgroups 0..7: wave 0 of each sorts one class of this frame's bin counts for the next frame (order_bins_wave);
//   then 16 per bin of the first P.deep_k8 positions of `order`: one deep strip each -- if the strip has P.deep_strip_min draws to shade;
//   then 4 per bin of the frame: sixteen one-wave strips as k_composite_tiles shades them, four to a workgroup (a strip the deep part took
//   leaves its wave idle).
// One launch, so that the deep strips' shaders and blenders run BESIDE the other strips' waves: as a launch of their own in front they
// added their whole duration (sweeps in profiles/r06_deep_strips.txt).  Workgroup b runs on XCD b % 8; every part's size is a multiple of
// 8, so position p keeps XCD p % 8 in both parts, like the strips of k_composite_tiles.
template <int kUnit>  // (a template so that only the translation unit that launches it holds the symbol)
__global__ __launch_bounds__(256, FDH_UNIFORM_WAVES) void k_composite_deep(const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts, CompositeParams P) {
  extern __shared__ uint32_t composite_lds[];
  constexpr int kStripsPerBin = kWgsPerBin * kWavesPerWg;  // 16
  // (which wave of the workgroup this is, as a value the compiler KNOWS to be wave-uniform: taken from the thread index alone it counted as
  // divergent, and with it every branch on whose turn a shading unit is -- the whole draw loop ran under exec masks)
  const int lane = threadIdx.x & 63, wg_wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  int blk = (int)blockIdx.x - 8;
  if (blk < 0) {
    if (wg_wave == 0) order_bins_wave(P.counts, P.order_next, P.bin_nx, P.bin_nx * P.bin_ny, composite_lds, lane, (int)blockIdx.x, P.deep_min, P.deep_out);
    return;
  }
  const int n_deep = P.deep_k8 * kStripsPerBin;
  const bool deep = blk < n_deep;
  int bin_local, sidx;
  if (deep) {
    const int q = blk >> 3;
    bin_local = (blk & 7) + 8 * (q / kStripsPerBin);
    sidx = q % kStripsPerBin;
  } else {
    blk -= n_deep;
    const int q = blk >> 3;
    bin_local = (blk & 7) + 8 * (q >> 2);
    sidx = 4 * (q & 3) + wg_wave;
  }
  if (bin_local >= P.bin_nx * P.bin_ny) return;
  const int j = sidx >> 2, wave = sidx & 3;
  const int sbit = j * 4 + wave;
  const int rc = P.order[bin_local];
  const int bly = rc >> 16, blx = rc & 0xffff;
  const int bin_x = P.bin_x0 + blx, bin_y = P.bin_y0 + bly;
  const int bin = bin_y * P.bins_x + bin_x;
  const int tx0 = bin_x * kBin + (j & 1) * kWgW;
  const int ty0 = bin_y * kBin + (j >> 1) * kWgH + wave * kTileH;
  if (tx0 >= P.W || ty0 >= P.H || ty0 + kTileH <= P.row_lo || ty0 >= P.row_hi) return;
  const bool is_deep = bin_local < P.deep_k8 && strip_shade_count(P, bin, sbit, lane) >= (uint32_t)P.deep_strip_min;
  if (!deep) {  // one wave, one strip
    if (!is_deep) composite_strip<4, true, 0>(P, draws, exts, composite_lds, bin, sidx, sbit, tx0, ty0, lane, 0);
    return;
  }
  if (!is_deep) return;  // (the whole workgroup: one strip, one answer -- its usual wave shades it)
  if (threadIdx.x < kDeepSlots + 2) composite_lds[kDeepSlots * (kDeepSlotFloats + kDeepHdr) + threadIdx.x] = 0u;  // ready[], consumed
  __syncthreads();
  // (the deep strips are the launch's critical path, and of a deep strip its blender: they go first where a SIMD has a choice)
  if (wg_wave == 0) {
    if (FDH_DEEP_PRIO) __builtin_amdgcn_s_setprio(3);
    composite_strip<4, true, 1>(P, draws, exts, composite_lds, bin, sidx, sbit, tx0, ty0, lane, 0);
  } else {
    if (FDH_DEEP_PRIO) __builtin_amdgcn_s_setprio(2);
    composite_strip<4, true, 2>(P, draws, exts, composite_lds, bin, sidx, sbit, tx0, ty0, lane, wg_wave - 1);
  }
}

template <int kPaths, bool kFull, int kRole>
__device__ __forceinline__ void composite_strip(const CompositeParams& P, const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts,
                                                uint32_t* composite_lds, const int bin, const int sidx, const int sbit, const int tx0, const int ty0,
                                                const int lane, const int shader_id) {
  static_assert(kRole == 0 || (kPaths == 4 && kFull), "deep strips: the no-clip build's full-frame launch only");
  constexpr int kStripsPerBin = kWgsPerBin * kWavesPerWg;  // 16
  constexpr int mslot = 0;
  constexpr bool kBlender = kRole == 1, kShader = kRole == 2;
  uint32_t (*mask_stack)[kMaskDepth][64] = reinterpret_cast<uint32_t (*)[kMaskDepth][64]>(composite_lds);
  const DeepRing ring(composite_lds);
  uint32_t rank = 0, unit = 0;  // deep strips: source terms / shading units (a draw, or a run of draws over one distance field) so far in the list
  const int tx1 = tx0 + kTileW, ty1 = ty0 + kTileH;
  const int px0 = tx0 + (lane & 7) * 4, py = ty0 + (lane >> 3);
  // The clip stack: levels 0 .. kMaskDepth - 1 in LDS; deeper nesting (the reference has no limit: one mask plane per level,
  // glcontext.nim:1886-1914) spills to a global plane the host sizes for the frame's deepest nest -- [level][strip][lane], every
  // slot written and read by this lane alone (agent-scope accesses: the read must not be served from a stale L1 line).
  const size_t spill_at = ((size_t)bin * kStripsPerBin + (size_t)sidx) * 64 + (size_t)lane;
  auto stack_put = [&](const int depth, const uint32_t v) __attribute__((always_inline)) {
    if (depth < kMaskDepth) mask_stack[mslot][depth][lane] = v;  // (wave-uniform)
    else __hip_atomic_store(P.mask_spill + (size_t)(depth - kMaskDepth) * P.spill_stride + spill_at, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto stack_get = [&](const int depth) __attribute__((always_inline)) -> uint32_t {
    if (depth < kMaskDepth) return mask_stack[mslot][depth][lane];
    return __hip_atomic_load(P.mask_spill + (size_t)(depth - kMaskDepth) * P.spill_stride + spill_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  constexpr bool kMasks = (kPaths & 4) == 0;  // <4>: a phase without clip / rect-mask operations -- no mask registers, no stack
#if FDH_TIMING
  const unsigned long long T0 = FDH_NOW(), W0 = wall_clock64();
  unsigned long long T_cull = 0, T_rec = 0, T_shade = 0, T_cnt = 0, n_draws_t = 0, T_cull_core = 0, n_core_t = 0, n_all_t = 0;
  unsigned long long T_mode[4] = {0, 0, 0, 0}, N_mode[4] = {0, 0, 0, 0};  // edge draws by mode: 3, 7, 9, 12
#endif
  // A DIRECT launch (round 6): a phase of at most 64 draws has no list -- k_bin_draws was not launched for the frame --, every strip's
  // wave makes the entries of its bin itself, lane i the one of the phase's draw i, with the two functions the bin kernel makes them
  // with.  A frame of a handful of draws (a dialog, the reference's 4-node test scene) is one launch less: its launches are latency,
  // 4 - 5 us each, not work.
  const bool direct = P.direct != 0;
  const uint32_t cnt = direct ? (uint32_t)P.direct_n : P.counts[bin];
#if FDH_TIMING
  T_cnt = FDH_NOW() - T0 + (cnt & 0u);
#endif
  if (!kFull && cnt == 0 && P.load_fb) return;  // nothing lands in this bin: the surface already holds the result

  const bool row_ok = py < P.H;
  const bool vec_ok = row_ok && px0 + 3 < P.W && (P.pitch & 3) == 0;  // whole 16-byte group inside the frame
  const size_t pix = (size_t)py * P.pitch + px0;
  F4 F0, F1, F2, F3;
  F0 = F1 = F2 = F3 = unpack255(P.clear_rgba8);
  if (!kFull && P.load_fb) {
    if (vec_ok) {
      const uint4 q = *reinterpret_cast<const uint4*>(P.fb + pix);
      F0 = unpack255(q.x); F1 = unpack255(q.y); F2 = unpack255(q.z); F3 = unpack255(q.w);
    } else if (row_ok) {
      if (px0 + 0 < P.W) F0 = unpack255(P.fb[pix + 0]);
      if (px0 + 1 < P.W) F1 = unpack255(P.fb[pix + 1]);
      if (px0 + 2 < P.W) F2 = unpack255(P.fb[pix + 2]);
      if (px0 + 3 < P.W) F3 = unpack255(P.fb[pix + 3]);
    }
  }
  float mk0 = 1.0f, mk1 = 1.0f, mk2 = 1.0f, mk3 = 1.0f;  // NfClipContent stack product (1 = maskTexEnabled false)
  float rm0 = 1.0f, rm1 = 1.0f, rm2 = 1.0f, rm3 = 1.0f;  // fast rect mask (atlas_rect_mask.frag), 1 when none
  int mask_depth = 0;
  bool rmask_on = false;  // wave-uniform: rm0..3 may differ from 1
  bool touched = false;
  const uint2* __restrict__ list = P.lists + (size_t)bin * P.stride;
  const float cy = (float)py + 0.5f;
  const float cx0 = (float)px0 + 0.5f;
  const float inv255 = 1.0f / 255.0f;

  // ---- deep strips: the ring between the strip's shaders and its blender (every store below is made by all 64 lanes with the same
  // value: a branch on the lane index would be the draw loop's only divergent one -- tools/lint_isa.py)
  auto deep_slot = [&](const uint32_t rk) __attribute__((always_inline)) -> uint32_t {  // shader: the slot of source term rk, once the blender has freed it
    if (rk >= (uint32_t)kDeepSlots) deep_wait_ge(ring.consumed, rk - (uint32_t)kDeepSlots + 1u);
    return rk % (uint32_t)kDeepSlots;
  };
  auto deep_publish = [&](const uint32_t slot, const uint32_t rk, const uint32_t tag, const uint32_t h1, const uint32_t h2, const uint32_t h3, const uint32_t h4, const uint32_t h5) __attribute__((always_inline)) {
    uint32_t* h = ring.hdr + slot * kDeepHdr;
    h[0] = tag; h[1] = h1; h[2] = h2; h[3] = h3; h[4] = h4; h[5] = h5;
    __hip_atomic_store(ring.ready + slot, rk + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto deep_nop = [&](const uint32_t rk) __attribute__((always_inline)) { deep_publish(deep_slot(rk), rk, DT_NOP, 0u, 0u, 0u, 0u, 0u); };
  // blender: source term rk out of its slot, blended into the strip
  auto deep_consume = [&](const uint32_t rk) __attribute__((always_inline)) {
    const uint32_t slot = rk % (uint32_t)kDeepSlots;
    const uint32_t* h = ring.hdr + slot * kDeepHdr;
    const float* v = ring.data + slot * kDeepSlotFloats + lane;
    // The slot's sequence number, its header and the first four payload floats are read in ONE round trip to LDS, then the number is
    // looked at: LDS returns a wave's reads in order, so values read after a number that says "published" are the published ones.
    // (Before: wait for the number, then the header, then the payload -- three dependent round trips per source term on the one wave
    // whose chain a deep strip's time is.)
    uint32_t tag, h1, h2, h3, h4, h5;
    float v0, v1, v2, v3;
    for (int spins = 0; spins < (1 << 20); spins++) {
      const uint32_t seq = __hip_atomic_load(ring.ready + slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      tag = h[0]; h1 = h[1]; h2 = h[2]; h3 = h[3]; h4 = h[4]; h5 = h[5];
      v0 = v[0]; v1 = v[64]; v2 = v[128]; v3 = v[192];
      asm volatile("" : "+v"(tag), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));  // (read HERE, after the number)
      if ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane(seq) - (rk + 1u)) >= 0) break;
      __builtin_amdgcn_s_sleep(1);
    }
    tag = __builtin_amdgcn_readfirstlane(tag);
    if (tag == DT_PACKED || tag == DT_PACKED_BLACK) {
      const f2 saa = {v0, v1}, sab = {v2, v3};
      const f2 Aa = saa * 255.0f, Ab = sab * 255.0f, iaa = 1.0f - saa, iab = 1.0f - sab;
      if (tag == DT_PACKED_BLACK) {
        blend_black(F0, Aa.x, iaa.x); blend_black(F1, Aa.y, iaa.y); blend_black(F2, Ab.x, iab.x); blend_black(F3, Ab.y, iab.y);
      } else {
        const f2 crg = {__uint_as_float(__builtin_amdgcn_readfirstlane(h1)), __uint_as_float(__builtin_amdgcn_readfirstlane(h2))};
        const float cb = __uint_as_float(__builtin_amdgcn_readfirstlane(h3));
        const f2 b1 = {cb, 1.0f};
        blend_pre(F0, crg * Aa.x, b1 * Aa.x, iaa.x); blend_pre(F1, crg * Aa.y, b1 * Aa.y, iaa.y);
        blend_pre(F2, crg * Ab.x, b1 * Ab.x, iab.x); blend_pre(F3, crg * Ab.y, b1 * Ab.y, iab.y);
      }
    } else if (tag == DT_GENERIC) {
      blend(F0, v0, v[4 * 64], v[8 * 64], v[12 * 64]); blend(F1, v1, v[5 * 64], v[9 * 64], v[13 * 64]);
      blend(F2, v2, v[6 * 64], v[10 * 64], v[14 * 64]); blend(F3, v3, v[7 * 64], v[11 * 64], v[15 * 64]);
    } else if (tag == DT_SELF17) {  // mode 17 over the live surface (blur radius <= 0.5): the source IS the strip's own texel (atlas.frag:381-388)
      const float k255 = 1.0f / 255.0f;
      blend(F0, F0.x * k255, F0.y * k255, F0.z * k255, F0.w * k255 * v0); blend(F1, F1.x * k255, F1.y * k255, F1.z * k255, F1.w * k255 * v1);
      blend(F2, F2.x * k255, F2.y * k255, F2.z * k255, F2.w * k255 * v2); blend(F3, F3.x * k255, F3.y * k255, F3.z * k255, F3.w * k255 * v3);
    } else if (tag == DT_UNIFORM_PRE) {
      const f2 c_rg = {__uint_as_float(__builtin_amdgcn_readfirstlane(h1)), __uint_as_float(__builtin_amdgcn_readfirstlane(h2))};
      const f2 c_ba = {__uint_as_float(__builtin_amdgcn_readfirstlane(h3)), __uint_as_float(__builtin_amdgcn_readfirstlane(h4))};
      const float ia = __uint_as_float(__builtin_amdgcn_readfirstlane(h5));
      blend_pre(F0, c_rg, c_ba, ia); blend_pre(F1, c_rg, c_ba, ia); blend_pre(F2, c_rg, c_ba, ia); blend_pre(F3, c_rg, c_ba, ia);
    }
    // (the slot is free once its values are in registers: the loads above have landed before the store below is made -- release)
    __hip_atomic_store(ring.consumed, rk + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  for (uint32_t base = 0; base < cnt; base += 64) {
#if FDH_TIMING
    const unsigned long long Tc0 = FDH_NOW();
#endif
    const uint32_t i = base + lane;
    // (no branch around the load: lanes past the end read the last entry and drop it)
    uint2 e;
    if (direct) {  // (wave-uniform; what is inside compiles to selects)
      const int by = bin / P.bins_x, bx = bin - by * P.bins_x;
      BinRec br;
      uint32_t word = 0, strips = 0;
      bool hit = bin_entry_head(P.binrec, P.direct_first + (int)min(i, cnt - 1u), bx * kBin, by * kBin, br, word, strips);
      bin_entry_tail(br, bx * kBin, by * kBin, hit, strips);
      e = make_uint2(word, hit ? strips : 0u);
    } else {
      e = list[min(i, cnt - 1u)];  // {draw index | flags, strips touched | strips inside the saturated core << 16}
    }
    const uint32_t idx = e.x;
    const uint32_t ey = i < cnt ? e.y : 0u;
    // this strip's state in the entry (fdh_types.h): (1, 0) touched, (1, 1) core, (0, 1) an edge strip wholly inside the draw's quad
    const uint32_t st = (ey >> sbit) & 0x10001u;
    unsigned long long m = __ballot(st != 0u);
    const unsigned long long m_core = __ballot(st == 0x10001u);
    const unsigned long long m_inq = __ballot(st == 0x10000u);  // every pixel of the strip is covered by the quad: no per-pixel test
    if (!kMasks || !P.has_masks) {
      // Occlusion: an opaque fill that covers the whole strip makes every earlier draw of the strip invisible.  (Only in
      // phases without clip / rect masks: a skipped push or pop would derail the mask stack.)
      const unsigned long long m_opaque = __ballot(st == 0x10001u && (idx & LE_OPAQUE) != 0u);
      if (m_opaque != 0) m &= ~((1ull << (63 - __builtin_clzll(m_opaque))) - 1ull);
    }
#if FDH_TIMING
    T_cull += FDH_NOW() - Tc0 + (m & 0ull);
#endif
    if (m == 0) continue;
    // Which straight-line path each surviving entry takes on this strip, decided for all 64 entries at once with vector
    // compares; the draw loop then tests ONE bit per decision (s_bitcmp1 + s_cbranch_scc) where it used to rebuild the answer per
    // draw out of four scalar booleans (s_cselect_b64 / s_and_b64 chains on the one scalar unit a CU has).
    //   m_plainc: the strip lies in the draw's saturated core and the draw has one colour -> a uniform blend, record not fetched
    //   m_simple: an edge strip of a draw with a packed edge path (list-entry path codes 1..8)
    // (each is one vector compare on the entry's flag word, combined with the strip masks above on the scalar side)
    const unsigned long long m_plainc = m_core & __ballot((int32_t)idx < 0);  // LE_PLAIN is the sign bit
    // (a deep strip's blender: the colours of the batch's one-colour core strips, lane i <-> entry i, in ONE vector load -- fetched draw by
    // draw through the scalar cache each was a round trip to L2 on the strip's critical wave)
    u32x4 plain_col = {0u, 0u, 0u, 0u};
    if (kBlender) plain_col = *reinterpret_cast<const u32x4*>(draws[idx & LE_INDEX].col);
    const uint32_t code_l = idx & (15u << LE_PATH_SHIFT);
    const unsigned long long m_simple = m & ~m_core & __ballot((kPaths & 3) == 0 ? code_l != 0u : (code_l - 1u) < (4u << LE_PATH_SHIFT));
    // sdRoundedBox (atlas.frag:51-69) of the lane's four pixels at height py for half extents (bx, by)
    auto dist4 = [&](const DrawRec& r, const f2 pxa, const f2 pxb, const float py_, const float bx, const float by, f2& da, f2& db) __attribute__((always_inline)) {
      const bool top = py_ > 0.0f;
      const float rR = top ? r.r[0] : r.r[1], rL = top ? r.r[2] : r.r[3];
      const float ay = __builtin_fabsf(py_) - by;
      const f2 rra = {pxa.x > 0.0f ? rR : rL, pxa.y > 0.0f ? rR : rL}, rrb = {pxb.x > 0.0f ? rR : rL, pxb.y > 0.0f ? rR : rL};
      const f2 axa = {__builtin_fabsf(pxa.x), __builtin_fabsf(pxa.y)}, axb = {__builtin_fabsf(pxb.x), __builtin_fabsf(pxb.y)};
      const f2 qxa = axa - bx + rra, qxb = axb - bx + rrb;
      const f2 qya = ay + rra, qyb = ay + rrb;
      // sdRoundedBox: min(max(q.x, q.y), 0) + length(max(q, 0)) - r.  With m = max(q.x, q.y): outside the corner cells (not both
      // components positive) length(max(q, 0)) = max(m, 0), and min(m, 0) + max(m, 0) = m exactly (one of the two is 0); in a corner
      // cell both components are positive, so max(q, 0) = q and the first term is 0.  So the distance is (corner ? |q| : m) - r,
      // bit for bit what the three-term form gives, in seven instructions per pixel instead of thirteen where no lane of the
      // strip sits in a corner cell (round 4).
      f2 ma = {__builtin_fmaxf(qxa.x, qya.x), __builtin_fmaxf(qxa.y, qya.y)}, mb = {__builtin_fmaxf(qxb.x, qyb.x), __builtin_fmaxf(qxb.y, qyb.y)};
      const f2 lowa = {__builtin_fminf(qxa.x, qya.x), __builtin_fminf(qxa.y, qya.y)}, lowb = {__builtin_fminf(qxb.x, qyb.x), __builtin_fminf(qxb.y, qyb.y)};
      if (__any(lowa.x > 0.0f || lowa.y > 0.0f || lowb.x > 0.0f || lowb.y > 0.0f)) {  // some lane sits in a corner cell
        const f2 sa2 = qxa * qxa + qya * qya, sb2 = qxb * qxb + qyb * qyb;
        ma.x = lowa.x > 0.0f ? fsqrt(sa2.x) : ma.x; ma.y = lowa.y > 0.0f ? fsqrt(sa2.y) : ma.y;
        mb.x = lowb.x > 0.0f ? fsqrt(sb2.x) : mb.x; mb.y = lowb.y > 0.0f ? fsqrt(sb2.y) : mb.y;
      }
      da = ma - rra; db = mb - rrb;
    };
    // elliptical corners (atlas.frag:96-115): the distance itself comes from the general routine, four pixels
    // unpacked; coverage and blend below stay packed
    auto dist4e = [&](const DrawRec& r, const f2 pxa, const f2 pxb, const float py_, const float bx, const float by, f2& oa, f2& ob) __attribute__((always_inline)) {
      const float px4[4] = {pxa.x, pxa.y, pxb.x, pxb.y};
      float d4[4];
      shape_distN<4>(true, px4, py_, bx, by, r.r[0], r.r[1], r.r[2], r.r[3], d4);
      oa = {d4[0], d4[1]}; ob = {d4[2], d4[3]};
    };
    // ---- the common edge strip: ONE colour, nothing clipping, mode fill / drop shadow / inner shadow / AA stroke (list-entry
    // path codes 1..8, k_bin_draws).  Written on float2 pairs -- pixels (0,1) and (2,3) of the lane side by side; same
    // formulas, same order of operations as the general path in shade() below.  Two halves: the distance field of the node's
    // shape at the lane's pixels (edge_geom), and what ONE draw makes of it -- coverage by mode, blend (edge_blend).  A node's
    // fill, stroke and inner shadows are consecutive draws over the same quad and the same shape
    // (renderRoundedShapeScaledCorners figrender.nim:806-873, renderInnerShadows :716-744): the draw loop evaluates the field
    // once for such a run (LE_SHARE) and calls edge_blend per draw with that draw's own few parameters.
    auto edge_geom = [&](const DrawRec& r, const bool inset, const bool ellip, f2& lxa, f2& lxb, float& pyy, f2& da, f2& db) __attribute__((always_inline)) {
      const float shx = inset ? r.p0 : r.p2, shy = inset ? r.p1 : r.p3;
      pyy = local_y_up(r, cy);
      float lx4[4];
      local_x4(r, cx0, lx4);
      lxa = {lx4[0], lx4[1]}; lxb = {lx4[2], lx4[3]};
      if ((kPaths & 3) == 0 && ellip) dist4e(r, lxa, lxb, pyy, shx, shy, da, db); else dist4(r, lxa, lxb, pyy, shx, shy, da, db);
    };
    // r: the run's geometry (quad, radii, AA factor, bounds); m_*: the draw's own sdfParams.zw, sdfFactors and colour
    // inq (wave-uniform, from the list entry): the strip lies wholly inside the quad's pixel bounds
    // (rk: a deep strip's shader puts the draw's source alphas into the ring as source term rk instead of blending them -- deep_consume is the rest)
    auto edge_blend = [&](const DrawRec& r, const uint32_t mode, const bool ellip, const bool inq, const float m_p2, const float m_p3, const float m_f0, const float m_f1,
                          const u32x4 m_col, const f2 lxa, const f2 lxb, const float pyy, const f2 da, const f2 db, F4& A0, F4& A1, F4& A2, F4& A3, const uint32_t rk) __attribute__((always_inline)) {
      f2 ala, alb;  // coverage
      if (mode == 3u) {
        ala = {cover_aa(da.x, r.aa), cover_aa(da.y, r.aa)}; alb = {cover_aa(db.x, r.aa), cover_aa(db.y, r.aa)};
      } else if (mode == 12u) {
        const float h = m_f0 * 0.5f;
        const f2 ea = da + h, eb = db + h;
        const f2 ga = {__builtin_fabsf(ea.x), __builtin_fabsf(ea.y)}, gb = {__builtin_fabsf(eb.x), __builtin_fabsf(eb.y)};
        ala = {cover_aa(ga.x - h, r.aa), cover_aa(ga.y - h, r.aa)}; alb = {cover_aa(gb.x - h, r.aa), cover_aa(gb.y - h, r.aa)};
        if (__all(ala.x == 0.0f && ala.y == 0.0f && alb.x == 0.0f && alb.y == 0.0f)) {  // inside the stroke: no-op
          if (kShader) deep_nop(rk);
          return;
        }
      } else if (mode == 9u) {  // atlas.frag:364-380 -- clip alpha of the node's own shape x the falloff inside the offset shape
        f2 sha, shb;
        if ((kPaths & 3) == 0 && ellip) dist4e(r, lxa - m_p2, lxb - m_p2, pyy + m_p3, r.p0, r.p1, sha, shb); else dist4(r, lxa - m_p2, lxb - m_p2, pyy + m_p3, r.p0, r.p1, sha, shb);
        const float spread = m_f1;
        const f2 sda = sha + spread, sdb = shb + spread;
        const float rs = frcp(__builtin_fmaxf(0.5f * m_f0, 0.5f));
        // falloff(sd) = sd < 0 ? min(exp2(-0.7213 z^2), 1) : 1 with z = sd rs, written as exp2(-0.7213 z'^2) with z' = min(sd, 0) rs:
        // the same operations on the same values where sd < 0; exp2(-0) = 1 where it is not; and v_exp_f32 of a non-positive
        // input is never above 1 (tools/microbench/exp2_le1.hip tries every such float), so the min never acted
        const f2 za = f2{__builtin_fminf(sda.x, 0.0f), __builtin_fminf(sda.y, 0.0f)} * rs, zb = f2{__builtin_fminf(sdb.x, 0.0f), __builtin_fminf(sdb.y, 0.0f)} * rs;
        const f2 ea = -0.72134752044f * za * za, eb = -0.72134752044f * zb * zb;
        ala = {cover_aa(da.x, r.aa) * fexp2(ea.x), cover_aa(da.y, r.aa) * fexp2(ea.y)};
        alb = {cover_aa(db.x, r.aa) * fexp2(eb.x), cover_aa(db.y, r.aa) * fexp2(eb.y)};
      } else {  // 7: atlas.frag:330-343
        const float spread = m_f1;
        const f2 sda = da - spread, sdb = db - spread;
        if (__all(sda.x <= 0.0f && sda.y <= 0.0f && sdb.x <= 0.0f && sdb.y <= 0.0f)) {
          ala = 1.0f; alb = 1.0f;
        } else {
          const float rs = frcp(__builtin_fmaxf(0.5f * m_f0, 0.5f));
          // (sd > 0 ? min(exp2(..), 1) : 1 as exp2 of max(sd, 0): see the inner shadow above)
          const f2 za = f2{__builtin_fmaxf(sda.x, 0.0f), __builtin_fmaxf(sda.y, 0.0f)} * rs, zb = f2{__builtin_fmaxf(sdb.x, 0.0f), __builtin_fmaxf(sdb.y, 0.0f)} * rs;
          const f2 ea = -0.72134752044f * za * za, eb = -0.72134752044f * zb * zb;
          ala = {fexp2(ea.x), fexp2(ea.y)};
          alb = {fexp2(eb.x), fexp2(eb.y)};
        }
      }
      const float cw = (float)(m_col.x >> 24) * inv255;  // (m_col: the draw's colour and, as floats, its r, g, b / 255: Context::prepare)
      f2 saa = ala * cw, sab = alb * cw;
      FDH_COUNT(inq ? 64 : 65);
      if ((m_col.y | m_col.z | m_col.w) == 0u) FDH_COUNT(71);
      if (!inq) {
        // coverage of the quad: unsigned (x - bx0) < width, width 0 on rows outside it
        const uint32_t xrel = (uint32_t)(px0 - (int)r.bx0);
        const uint32_t wcov = (py >= r.by0 && py < r.by1) ? (uint32_t)((int)r.bx1 - (int)r.bx0) : 0u;
        saa.x = xrel < wcov ? saa.x : 0.0f; saa.y = xrel + 1u < wcov ? saa.y : 0.0f;
        sab.x = xrel + 2u < wcov ? sab.x : 0.0f; sab.y = xrel + 3u < wcov ? sab.y : 0.0f;
      }
      if (kShader) {
        const uint32_t slot = deep_slot(rk);
        float* v = ring.data + slot * kDeepSlotFloats + lane;
        v[0] = saa.x; v[64] = saa.y; v[128] = sab.x; v[192] = sab.y;
        deep_publish(slot, rk, (m_col.y | m_col.z | m_col.w) == 0u ? DT_PACKED_BLACK : DT_PACKED, m_col.y, m_col.z, m_col.w, 0u, 0u);
        return;
      }
      const f2 Aa = saa * 255.0f, Ab = sab * 255.0f, iaa = 1.0f - saa, iab = 1.0f - sab;
      if ((m_col.y | m_col.z | m_col.w) == 0u) {
        // a black source (every drop shadow of the reference's scenes, most strokes): the colour terms are +0 and fma(F, 1 - sa, +0)
        // is the product F (1 - sa) itself (F >= 0, 1 - sa >= 0): three multiplies per pixel less, the same bits
        blend_black(A0, Aa.x, iaa.x); blend_black(A1, Aa.y, iaa.y); blend_black(A2, Ab.x, iab.x); blend_black(A3, Ab.y, iab.y);
        return;
      }
      const f2 crg = {__uint_as_float(m_col.y), __uint_as_float(m_col.z)};
      const float cb = __uint_as_float(m_col.w);
      { const f2 b1 = {cb, 1.0f}; blend_pre(A0, crg * Aa.x, b1 * Aa.x, iaa.x); blend_pre(A1, crg * Aa.y, b1 * Aa.y, iaa.y);
        blend_pre(A2, crg * Ab.x, b1 * Ab.x, iab.x); blend_pre(A3, crg * Ab.y, b1 * Ab.y, iab.y); }
    };
#if FDH_EDGE_CHECK
    auto simple_edge = [&](const DrawRec& r, const uint32_t mode, const bool ellip, F4& A0, F4& A1, F4& A2, F4& A3) __attribute__((always_inline)) {
      f2 lxa, lxb, da, db;
      float pyy;
      edge_geom(r, mode == 9u, ellip, lxa, lxb, pyy, da, db);
      edge_blend(r, mode, ellip, false, r.p2, r.p3, r.f0, r.f1, u32x4{r.col[0], r.col[1], r.col[2], r.col[3]}, lxa, lxb, pyy, da, db, A0, A1, A2, A3, 0u);
    };
#endif
    // One draw = one lambda call.  The record of the NEXT surviving draw is fetched (scalar loads) before the
    // current one is shaded, so the ~L2-latency of the fetch overlaps the shading arithmetic.
    auto shade = [&](const uint32_t d, const DrawRec& r, const bool core, const bool inq, const uint32_t rk) {
      const uint32_t om = r.op_mode;
      const uint32_t op = (om >> 12) & 15u;
      const uint32_t mode = om & 255u;
      touched = true;
      FDH_COUNT(0);
      if (kMasks && op == OP_MASK_POP) {
        mask_depth--;
        if (mask_depth > 0) {
          const uint32_t w = stack_get(mask_depth - 1);
          mk0 = (float)(w & 255u) * inv255; mk1 = (float)((w >> 8) & 255u) * inv255;
          mk2 = (float)((w >> 16) & 255u) * inv255; mk3 = (float)(w >> 24) * inv255;
        } else {
          mk0 = mk1 = mk2 = mk3 = 1.0f;
        }
        return;
      }
      if (kMasks && op == OP_RMASK_END) { rm0 = rm1 = rm2 = rm3 = 1.0f; rmask_on = false; return; }
      if (kMasks && op == OP_RMASK_BEGIN && r.inv_h == 0.0f) {
        // The fast rect mask under a transform without rotation (matY.x == 0: a row of pixels shares its local y), four pixels
        // at once: rectMaskAlpha atlas_rect_mask.frag:222-237, operation for operation what rect_mask_alpha() does per pixel.
        // (Rotated masks keep the one-pixel-slot path; this one lets a phase of rect-masked cells run on the <0> build: the
        // reference's own clip + rect-mask benchmark went through the 128-VGPR slot build for these alone.)
        float qx[4], dm[4];
#pragma unroll
        for (int k = 0; k < 4; k++) qx[k] = ((r.ox * (cx0 + (float)k) + r.oy * cy) + r.inv_w) - r.p0;
        const float qy = ((r.inv_h * cx0 + r.f0 * cy) + r.f1) - r.p1;
        shape_distN<4>((om & F_ELLIP) != 0u, qx, -qy, r.p2, r.p3, r.r[0], r.r[1], r.r[2], r.r[3], dm);
        rm0 = 1.0f - clamp01(r.aa * dm[0] + 0.5f); rm1 = 1.0f - clamp01(r.aa * dm[1] + 0.5f);
        rm2 = 1.0f - clamp01(r.aa * dm[2] + 0.5f); rm3 = 1.0f - clamp01(r.aa * dm[3] + 0.5f);
        rmask_on = true;
        return;
      }
      const bool atlas_mode = (mode == 0u) || (mode >= 13u && mode <= 16u);
      // (builds without the slot path and the atlas path only ever see `fast` draws: the host picks the build per phase from
      // exactly these properties, Context::submit -- no need to decode them again per draw)
      const bool fast = (kPaths & 11) == 0 || (!(om & F_GENERAL) && !atlas_mode && mode < 18u && (op == OP_DRAW || (kMasks && op == OP_MASK_PUSH)));
      // ---- axis-aligned atlas quads (glyphs, images at >= 1:1, MSDF / MTSDF): 4 pixels per lane in lock-step.  All
      // sixteen bilinear texel fetches of the lane are issued before any of them is used, so the wave pays the atlas
      // latency once per draw instead of once per pixel slot.  (Minified images, lod > 0, keep the trilinear slot path.)
      constexpr bool kSlow = (kPaths & 1) != 0, kAtlas = (kPaths & 2) != 0, kRot = (kPaths & 9) != 0;
      if (kAtlas && atlas_mode && !(om & F_GENERAL) && op == OP_DRAW && !(mode == 0u && r.aux2 > 0.0f && P.atlas.n_levels >= 2)) {
        const uint32_t fill_mode = (om >> 9) & 7u;
        const int S = P.atlas.size, msk = S - 1;
        const float fS = (float)S;
        const uint32_t* __restrict__ tex = P.atlas.level[0];
        if (mode == 0u && (om & F_TEXEL_1TO1) != 0u) {
          // ---- a glyph placed texel on pixel (figrender.nim:456-496): the bilinear fractions are 0 up to float noise (a GL
          // sampler's fixed-point coordinates snap them to 0), so a pixel IS its texel: the lane's four come in one 16-byte
          // run (4-byte aligned: the atlas origin of a glyph is arbitrary) instead of sixteen dword gathers and their filter
          // arithmetic.  Lanes outside the quad read texel (0, 0) and blend with alpha 0.
          const uint32_t xrel = (uint32_t)(px0 - (int)r.bx0);
          const bool rowc = py >= r.by0 && py < r.by1;
          const uint32_t wcov = rowc ? (uint32_t)((int)r.bx1 - (int)r.bx0) : 0u;
          const bool any_px = rowc & (px0 + 3 >= (int)r.bx0) & (px0 < (int)r.bx1);
          const int tx = px0 + (int)r.ext, ty = py + (int)r._pad;
          // (the atlas is a power of two wide: a shift, not a multiply -- an expensive arm would bring a divergent branch back)
          const uint32_t off_in = ((((uint32_t)ty) << (uint32_t)__builtin_ctz((uint32_t)S)) + (uint32_t)tx) << 2;
          const uint32_t off = any_px ? off_in : 0u;
          struct __attribute__((packed, aligned(4))) Run4 { uint32_t v[4]; };
          const Run4 run = *reinterpret_cast<const Run4*>(reinterpret_cast<const char*>(tex) + off);
          const F4 c0u = unpack255(r.col[0]);
          const bool solid = (om & F_SOLID) != 0u, masked = mask_depth > 0 || rmask_on;
          const bool lane_col = solid || (r.col[0] == r.col[1] && r.col[2] == r.col[3]);  // wave-uniform
          const float t = (cy - r.oy) * r.inv_h;
          F4 colL = {c0u.x * inv255, c0u.y * inv255, c0u.z * inv255, c0u.w * inv255};
          if (!solid && lane_col) {  // a vertical tint: one colour per lane (see the general atlas path below)
            const F4 br = unpack255(r.col[1]), tr = unpack255(r.col[2]), tl = unpack255(r.col[3]);
            const float s0 = (cx0 - r.ox) * r.inv_w;
            colL.x = tri_lerp(tl.x, c0u.x, br.x, tr.x, s0, t) * inv255;
            colL.y = tri_lerp(tl.y, c0u.y, br.y, tr.y, s0, t) * inv255;
            colL.z = tri_lerp(tl.z, c0u.z, br.z, tr.z, s0, t) * inv255;
            colL.w = tri_lerp(tl.w, c0u.w, br.w, tr.w, s0, t) * inv255;
          }
          auto texel_px = [&](const int k, F4& F, const float mk, const float rm) __attribute__((always_inline)) {
            const F4 a = unpack255(run.v[k]);
            F4 col = colL;
            if (!lane_col) {  // wave-uniform
              const F4 br = unpack255(r.col[1]), tr = unpack255(r.col[2]), tl = unpack255(r.col[3]);
              const float sk = (cx0 + (float)k - r.ox) * r.inv_w;
              col.x = tri_lerp(tl.x, c0u.x, br.x, tr.x, sk, t) * inv255;
              col.y = tri_lerp(tl.y, c0u.y, br.y, tr.y, sk, t) * inv255;
              col.z = tri_lerp(tl.z, c0u.z, br.z, tr.z, sk, t) * inv255;
              col.w = tri_lerp(tl.w, c0u.w, br.w, tr.w, sk, t) * inv255;
            }
            float sa = a.w * inv255 * col.w;
            if (masked) sa = sa * mk * rm;
            blend(F, a.x * inv255 * col.x, a.y * inv255 * col.y, a.z * inv255 * col.z, (xrel + (uint32_t)k) < wcov ? sa : 0.0f);
          };
          texel_px(0, F0, mk0, rm0); texel_px(1, F1, mk1, rm1); texel_px(2, F2, mk2, rm2); texel_px(3, F3, mk3, rm3);
          return;
        }
        const float uax = r.r[0], uay = r.r[1], utx = r.r[2], uty = r.r[3];
        const float t = (cy - r.oy) * r.inv_h;
        const float v = uay + (uty - uay) * t;
        const float ty_ = v * fS - 0.5f, fy = __builtin_floorf(ty_), ayf = ty_ - fy;
        const int y0 = (int)fy & msk, y1 = (y0 + 1) & msk;
        const uint32_t xrel = (uint32_t)(px0 - (int)r.bx0);
        const uint32_t wcov = (py >= r.by0 && py < r.by1) ? (uint32_t)((int)r.bx1 - (int)r.bx0) : 0u;
        float ushift = 0.0f;
        if (mode == 0u && (om & F_SUBPIXEL)) ushift = r.aux * frcp(__builtin_fmaxf(fS, 1.0f));  // wave-uniform
        float sK[4], uK[4], axK[4];
        uint32_t q00[4], q01[4], q10[4], q11[4];
        int fxi[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          sK[k] = (cx0 + (float)k - r.ox) * r.inv_w;
          uK[k] = uax + (utx - uax) * sK[k];
          const float tx_ = (uK[k] - ushift) * fS - 0.5f, fx = __builtin_floorf(tx_);
          axK[k] = tx_ - fx;
          fxi[k] = (int)fx;
        }
        const int fyi = (int)fy;
        // The strip's texel window.  The map pixel -> texel is linear, so the texel columns / rows the strip's 32 x 8 pixels touch
        // lie between those of its first and last pixel (+ 1 for the second bilinear tap).  When the window is small -- a glyph or
        // an MSDF image drawn at >= 0.5x: at most 64 x 12 texels -- the wave stages it in LDS with 16-byte runs (two or three
        // loads per lane) and every pixel takes its four taps from there: sixteen dword gathers per lane and draw kept the CU's
        // one texture-address unit busier than the arithmetic (config 4: a wave lived 13 us for 1.6 us of issue).
        const int wxa = __builtin_amdgcn_readlane(fxi[0], 0), wxb = __builtin_amdgcn_readlane(fxi[3], 7);
        const int wya = __builtin_amdgcn_readlane(fyi, 0), wyb = __builtin_amdgcn_readlane(fyi, 56);
        const int wx0 = min(wxa, wxb), wx1 = max(wxa, wxb) + 1, wy0 = min(wya, wyb), wy1 = max(wya, wyb) + 1;  // inclusive texel bounds
        const bool windowed = wx1 - wx0 < kWinCols && wy1 - wy0 < kWinRows && wx0 >= 0 && wy0 >= 0 && wx1 + 3 < S && wy1 < S;  // wave-uniform
        if (windowed) {
          uint32_t* const win = composite_lds + (P.has_masks ? kMaskDepth * 64 : 256);
          {  // 16 lanes x 16 bytes per window row, four rows per pass; lanes past the window repeat its last run / row (no branch)
            const int lr = lane >> 4, lc = (lane & 15) * 4;
            const int cc = min(lc, (wx1 - wx0) & ~3);
            struct __attribute__((packed, aligned(4))) Run4 { uint32_t v[4]; };
#pragma unroll
            for (int pass = 0; pass < kWinRows / 4; pass++) {
              const int row = min(pass * 4 + lr, wy1 - wy0);
              const Run4 run = *reinterpret_cast<const Run4*>(tex + (((uint32_t)(wy0 + row)) << (uint32_t)__builtin_ctz((uint32_t)S)) + (uint32_t)(wx0 + cc));
              uint4 q4 = {run.v[0], run.v[1], run.v[2], run.v[3]};
              *reinterpret_cast<uint4*>(win + (pass * 4 + lr) * kWinStride + lc) = q4;
            }
          }
          __builtin_amdgcn_wave_barrier();
          const int ly = min(max(fyi - wy0, 0), kWinRows - 2);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const int lx = min(max(fxi[k] - wx0, 0), kWinCols - 2);
            const uint32_t* w0 = win + ly * kWinStride + lx;
            q00[k] = w0[0]; q01[k] = w0[1]; q10[k] = w0[kWinStride]; q11[k] = w0[kWinStride + 1];
          }
          __builtin_amdgcn_wave_barrier();  // (the next draw of this strip overwrites the window)
        } else {
          // texel addresses as 32-bit byte offsets from the (scalar) level pointer: one shift and two adds per column instead of
          // sign extensions and 64-bit adds (a level is at most 16384^2 x 4 bytes = 1 GiB)
          const char* __restrict__ texb = reinterpret_cast<const char*>(tex);
          const uint32_t row0 = ((uint32_t)y0 * (uint32_t)S) << 2, row1 = ((uint32_t)y1 * (uint32_t)S) << 2;
          auto texel = [&](uint32_t off) __attribute__((always_inline)) { return *reinterpret_cast<const uint32_t*>(texb + off); };
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const uint32_t x0 = (uint32_t)(fxi[k] & msk), x1 = (x0 + 1u) & (uint32_t)msk;
            q00[k] = texel(row0 + (x0 << 2)); q01[k] = texel(row0 + (x1 << 2));
            q10[k] = texel(row1 + (x0 << 2)); q11[k] = texel(row1 + (x1 << 2));
          }
        }
        const bool solid = (om & F_SOLID) != 0u;
        const bool masked = mask_depth > 0 || rmask_on;
        const bool msdf = mode != 0u;
        const bool is_mtsdf = (mode == 14u || mode == 16u), is_stroke = (mode == 15u || mode == 16u);
        float spr = 1.0f;
        if (msdf) {  // wave-uniform (three v_rcp a coverage glyph has no use for)
          const float unit = r.f0 * frcp(r.p0);  // pxRange / atlas size (atlas.frag:45-49)
          const float fw_u = __builtin_fabsf((utx - uax) * r.inv_w), fw_v = __builtin_fabsf((uty - uay) * r.inv_h);
          spr = __builtin_fmaxf(0.5f * (unit * frcp(fw_u) + unit * frcp(fw_v)), 1.0f);
        }
        const F4 c0u = unpack255(r.col[0]);
        // The vertex-colour term once per lane where it cannot differ between the lane's four pixels: one colour, or a
        // vertical gradient (BL == BR and TR == TL -- what a text tint is).  With equal colours on both sides, tri_lerp's two
        // triangle formulas reduce to the same fma(bl - tl, t, tl) whatever s is (the s terms multiply an exact 0), so
        // this is the value every pixel computed before, bit for bit.
        const bool lane_col = solid || (r.col[0] == r.col[1] && r.col[2] == r.col[3]);  // wave-uniform
        F4 colL = {c0u.x * inv255, c0u.y * inv255, c0u.z * inv255, c0u.w * inv255};
        if (!solid && lane_col) {
          const F4 br = unpack255(r.col[1]), tr = unpack255(r.col[2]), tl = unpack255(r.col[3]);
          colL.x = tri_lerp(tl.x, c0u.x, br.x, tr.x, sK[0], t) * inv255;
          colL.y = tri_lerp(tl.y, c0u.y, br.y, tr.y, sK[0], t) * inv255;
          colL.z = tri_lerp(tl.z, c0u.z, br.z, tr.z, sK[0], t) * inv255;
          colL.w = tri_lerp(tl.w, c0u.w, br.w, tr.w, sK[0], t) * inv255;
        }
        const bool msdf3 = msdf && !is_mtsdf;  // the distance is the median of r, g, b: the alpha channel is not sampled
        auto pixel = [&](const int k, F4& F, const float mk, const float rm) __attribute__((always_inline)) {
          const F4 a = unpack255(q00[k]), b = unpack255(q01[k]), c = unpack255(q10[k]), d = unpack255(q11[k]);
          const float ax = axK[k];
          F4 col = colL;
          if (!lane_col) {  // wave-uniform
            const F4 br = unpack255(r.col[1]), tr = unpack255(r.col[2]), tl = unpack255(r.col[3]);
            col.x = tri_lerp(tl.x, c0u.x, br.x, tr.x, sK[k], t) * inv255;
            col.y = tri_lerp(tl.y, c0u.y, br.y, tr.y, sK[k], t) * inv255;
            col.z = tri_lerp(tl.z, c0u.z, br.z, tr.z, sK[k], t) * inv255;
            col.w = tri_lerp(tl.w, c0u.w, br.w, tr.w, sK[k], t) * inv255;
          }
          float sr, sg, sb, sa;
          if (msdf3) {  // wave-uniform.  atlas.frag:296-318; median3(k x, k y, k z) == k median3(x, y, z) exactly for k > 0
            const float bx_ = mixf(a.x, b.x, ax) * (1.0f - ayf) + mixf(c.x, d.x, ax) * ayf;
            const float by_ = mixf(a.y, b.y, ax) * (1.0f - ayf) + mixf(c.y, d.y, ax) * ayf;
            const float bz_ = mixf(a.z, b.z, ax) * (1.0f - ayf) + mixf(c.z, d.z, ax) * ayf;
            const F4 fc = eval_fill_rec(r, col, fill_mode, uK[k], v);
            const float sd = median3(bx_, by_, bz_) * inv255;
            const float spd = spr * (sd - r.f1);
            const float alpha = is_stroke ? clamp01(__builtin_fmaxf(r.p1, 0.0f) * 0.5f - __builtin_fabsf(spd) + 0.5f) : clamp01(spd + 0.5f);
            sr = fc.x; sg = fc.y; sb = fc.z; sa = fc.w * alpha;
          } else {
            F4 tx;  // GL_LINEAR, 0..1
            tx.x = (mixf(a.x, b.x, ax) * (1.0f - ayf) + mixf(c.x, d.x, ax) * ayf) * inv255;
            tx.y = (mixf(a.y, b.y, ax) * (1.0f - ayf) + mixf(c.y, d.y, ax) * ayf) * inv255;
            tx.z = (mixf(a.z, b.z, ax) * (1.0f - ayf) + mixf(c.z, d.z, ax) * ayf) * inv255;
            tx.w = (mixf(a.w, b.w, ax) * (1.0f - ayf) + mixf(c.w, d.w, ax) * ayf) * inv255;
            if (!msdf) {  // atlas.frag:284-295
              sr = tx.x * col.x; sg = tx.y * col.y; sb = tx.z * col.z; sa = tx.w * col.w;
            } else {  // atlas.frag:296-318 (MTSDF: the distance is in alpha)
              const F4 fc = eval_fill_rec(r, col, fill_mode, uK[k], v);
              const float sd = is_mtsdf ? tx.w : median3(tx.x, tx.y, tx.z);
              const float spd = spr * (sd - r.f1);
              const float alpha = is_stroke ? clamp01(__builtin_fmaxf(r.p1, 0.0f) * 0.5f - __builtin_fabsf(spd) + 0.5f) : clamp01(spd + 0.5f);
              sr = fc.x; sg = fc.y; sb = fc.z; sa = fc.w * alpha;
            }
          }
          if (masked) sa = sa * mk * rm;
          blend(F, sr, sg, sb, (xrel + (uint32_t)k) < wcov ? sa : 0.0f);
        };
        pixel(0, F0, mk0, rm0); pixel(1, F1, mk1, rm1); pixel(2, F2, mk2, rm2); pixel(3, F3, mk3, rm3);
        return;
      }
      // Two-triangle coverage and barycentrics of a rotated / skewed quad for the lane's four pixels (make_frag()'s arithmetic, 32-bit:
      // F_EDGE32).  The quad is the reference's triangles (3,0,1) = (TL, BL, BR) and (2,3,1) = (TR, TL, BR) over per-vertex ceil'd
      // corners (glcontext.nim:418-429); a pixel belongs to the first whose three edge functions -- exact integers in half-pixel
      // units, top-left rule -- admit its centre.  An edge value is a scalar base per strip + two v_mad_i32_i24 per lane + one add per
      // further pixel; ownership is folded into the base (E - 1 >= 0 <=> E > 0), so a triangle's inside test is one v_or3 and a sign
      // test.  Out: L0..L2 = the hit triangle's barycentrics (edge value x 1 / (E0 + E1 + E2)), T1 = it is the second triangle.
      // `exact`: the barycentrics as the ORACLE's rasteriser forms them -- (float)(E x 1 / (E0 + E1 + E2)) in DOUBLE precision -- instead of
      // float(E) x float(1 / sum).  The two differ in the last bit now and then, which no shading path cares about but one: the
      // bezier distance's closed-form cubic amplifies a last-bit difference of its input into pixels (see sd_bezierN).
      auto tri_bary = [&](const QuadExt& q, float (&L0)[4], float (&L1)[4], float (&L2)[4], bool (&T1)[4], bool (&cov)[4], const bool exact) __attribute__((always_inline)) {
        const int X0 = 2 * tx0 + 1, Y0 = 2 * ty0 + 1;
        const int dxl = 8 * (lane & 7), dyl = 2 * (lane >> 3);
        int eb[2][3], a2[2][3], nb[2][3];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
          for (int k = 0; k < 3; k++) {
            const int a = q.e[t][k].a, b = q.e[t][k].b;
            nb[t][k] = (int)(((q.own >> (t * 3 + k)) & 1u) ^ 1u);
            const int base = a * X0 + b * Y0 + (int)(uint32_t)(uint64_t)q.e[t][k].c - nb[t][k];  // (scalar)
            eb[t][k] = __mul24(b, dyl) + (__mul24(a, dxl) + base);
            a2[t][k] = 2 * a;
          }
        const bool valid0 = q.inv_sum[0] != 0.0f, valid1 = q.inv_sum[1] != 0.0f;
        const bool rowc = py >= r.by0 && py < r.by1;
        // (exact) E0 + E1 + E2 is the same at every point of the plane: one double-precision reciprocal per triangle and lane.  The oracle
        // divides in pixel units, w = E / 4 and 1 / (sum / 4): powers of two, the same quotient bit for bit.
        double invd0 = 0.0, invd1 = 0.0;
        if (exact) {
          const int s0 = eb[0][0] + eb[0][1] + eb[0][2] + nb[0][0] + nb[0][1] + nb[0][2], s1 = eb[1][0] + eb[1][1] + eb[1][2] + nb[1][0] + nb[1][1] + nb[1][2];
          invd0 = 1.0 / (double)(valid0 ? s0 : 1);
          invd1 = 1.0 / (double)(valid1 ? s1 : 1);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int e00 = eb[0][0] + k * a2[0][0], e01 = eb[0][1] + k * a2[0][1], e02 = eb[0][2] + k * a2[0][2];
          const int e10 = eb[1][0] + k * a2[1][0], e11 = eb[1][1] + k * a2[1][1], e12 = eb[1][2] + k * a2[1][2];
          const bool in0 = valid0 && ((e00 | e01 | e02) >= 0), in1 = valid1 && ((e10 | e11 | e12) >= 0);
          const bool use1 = !in0 && in1;
          T1[k] = use1;
          cov[k] = rowc && px0 + k >= r.bx0 && px0 + k < r.bx1 && (in0 || in1);
          const int E0 = (use1 ? e10 : e00) + (use1 ? nb[1][0] : nb[0][0]), E1 = (use1 ? e11 : e01) + (use1 ? nb[1][1] : nb[0][1]),
                    E2 = (use1 ? e12 : e02) + (use1 ? nb[1][2] : nb[0][2]);
          if (exact) {  // (compile-time at every call site) oracle: w = edge function in pixel units (= E / 4), l = (float)(w / (w0 + w1 + w2))
            const double inv = use1 ? invd1 : invd0;
            L0[k] = (float)((double)E0 * inv); L1[k] = (float)((double)E1 * inv); L2[k] = (float)((double)E2 * inv);
          } else {
            const float is = use1 ? q.inv_sum[1] : q.inv_sum[0];
            L0[k] = (float)E0 * is; L1[k] = (float)E1 * is; L2[k] = (float)E2 * is;
          }
        }
      };
      if (kSlow && FDH_ROT_ATLAS4 && (om & F_GENERAL) != 0u && (om & F_EDGE32) != 0u && atlas_mode && op == OP_DRAW) {
        // ---- rotated / skewed atlas quads (glyphs, images, MSDF under a rotated transform), four pixels per lane: the two-triangle
        // coverage and barycentrics of the SDF block below, uv interpolated between the quad's atlas corners, then the sampling
        // and shading shade_one() does per pixel slot (atlas.frag:284-318) -- unrolled, so the lane's sixteen (trilinear: thirty-two)
        // texel fetches are in flight together.
        FDH_COUNT(59);
        const QuadExt& q = exts[r.ext];
        float L0[4], L1[4], L2[4];
        bool T1[4], cov[4];
        tri_bary(q, L0, L1, L2, T1, cov, false);
        const bool solid = (om & F_SOLID) != 0u;
        const F4 cBL = unpack255(r.col[0]), cBR = unpack255(r.col[1]), cTR = unpack255(r.col[2]), cTL = unpack255(r.col[3]);
        const float uax = r.r[0], uay = r.r[1], utx = r.r[2], uty = r.r[3];
        const uint32_t fill_mode = (om >> 9) & 7u;
        const bool is_mtsdf = (mode == 14u || mode == 16u), is_stroke = (mode == 15u || mode == 16u);
        float sr[4], sg[4], sb[4], sa[4];
        // Round 5.  An MSDF image, or magnified / 1:1 under both triangles (lod <= 0: any glyph row, any image not shrunk): every pixel samples level 0
        // with GL_LINEAR -- wave-uniform, so no per-lane branch stands between the lane's sixteen texel fetches (atlas_sample() below
        // decides per pixel, and four dependent round trips per strip-draw made the 10 000-glyph rotated frame 4.5 x the upright one).
        // And the texels any covered pixel can touch lie in the quad's own atlas rectangle (uv is a convex combination of the corners'):
        // when that rectangle fits the strip's LDS window -- 64 x 12, 32 x 24 or 16 x 48 texels: every glyph -- the wave stages it with
        // 16-byte runs and the taps come from LDS, as on the upright path.
        if (mode != 0u || P.atlas.n_levels < 2 || (!(q.lod[0] > 0.0f) && !(q.lod[1] > 0.0f))) {  // (MSDF: textureLod(.., 0.0), atlas.frag:296-318)
          const int S = P.atlas.size, msk = S - 1;
          const float fS = (float)S;
          const uint32_t* __restrict__ tex = P.atlas.level[0];
          const bool shifted = mode == 0u && (om & F_SUBPIXEL) != 0u;
          float ushift = 0.0f;
          if (shifted) ushift = r.aux * frcp(__builtin_fmaxf(fS, 1.0f));  // (the same expression as the per-pixel form below)
          float uK[4], vK[4], axK[4], ayK[4];
          int fxi[4], fyi[4];
          F4 colK[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const bool use1 = T1[k];
            const float l0 = L0[k], l1 = L1[k], l2 = L2[k];
            const float u0 = use1 ? utx : uax, v0 = uay, u1 = uax, v1 = use1 ? uay : uty;
            uK[k] = l0 * u0 + l1 * u1 + l2 * utx;
            vK[k] = l0 * v0 + l1 * v1 + l2 * uty;
            colK[k] = {cBL.x * inv255, cBL.y * inv255, cBL.z * inv255, cBL.w * inv255};
            if (!solid) {
              const float c0x = use1 ? cTR.x : cTL.x, c0y = use1 ? cTR.y : cTL.y, c0z = use1 ? cTR.z : cTL.z, c0w = use1 ? cTR.w : cTL.w;
              const float c1x = use1 ? cTL.x : cBL.x, c1y = use1 ? cTL.y : cBL.y, c1z = use1 ? cTL.z : cBL.z, c1w = use1 ? cTL.w : cBL.w;
              colK[k].x = (l0 * c0x + l1 * c1x + l2 * cBR.x) * inv255;
              colK[k].y = (l0 * c0y + l1 * c1y + l2 * cBR.y) * inv255;
              colK[k].z = (l0 * c0z + l1 * c1z + l2 * cBR.z) * inv255;
              colK[k].w = (l0 * c0w + l1 * c1w + l2 * cBR.w) * inv255;
            }
            float us = uK[k];
            if (shifted) us -= ushift;
            const float x = us * fS - 0.5f, y = vK[k] * fS - 0.5f;  // (atlas_sample: u S - 0.5)
            const float fx = __builtin_floorf(x), fy = __builtin_floorf(y);
            axK[k] = x - fx; ayK[k] = y - fy;
            fxi[k] = (int)fx; fyi[k] = (int)fy;
          }
          // the quad's atlas rectangle in texels, a texel of slack on every side (the barycentrics sum to 1 only up to rounding)
          const float ulo = __builtin_fminf(uax, utx) - __builtin_fabsf(ushift), uhi = __builtin_fmaxf(uax, utx) + __builtin_fabsf(ushift);
          const float vlo = __builtin_fminf(uay, uty), vhi = __builtin_fmaxf(uay, uty);
          const int wx0 = __builtin_amdgcn_readfirstlane((int)__builtin_floorf(ulo * fS - 0.5f)) - 1, wx1 = __builtin_amdgcn_readfirstlane((int)__builtin_floorf(uhi * fS - 0.5f)) + 2;
          const int wy0 = __builtin_amdgcn_readfirstlane((int)__builtin_floorf(vlo * fS - 0.5f)) - 1, wy1 = __builtin_amdgcn_readfirstlane((int)__builtin_floorf(vhi * fS - 0.5f)) + 2;
          const int WW = wx1 - wx0 + 1, WH = wy1 - wy0 + 1;  // texels
          const int wsh = WW <= 16 ? 4 : WW <= 32 ? 5 : 6;    // log2 of the row stride in dwords: 768 dwords as 48 x 16, 24 x 32 or 12 x 64
          const bool windowed = WW <= 64 && WH <= (768 >> wsh) && wx0 >= 0 && wy0 >= 0 && wx1 + 3 < S && wy1 < S;  // wave-uniform
          uint32_t q00[4], q01[4], q10[4], q11[4];
          if (windowed) {
            uint32_t* const win = composite_lds + (P.has_masks ? kMaskDepth * 64 : 256);
            {
              const int lpr_sh = wsh - 2;  // lanes per window row = stride / 4
              const int lr = lane >> lpr_sh, lc = (lane & ((1 << lpr_sh) - 1)) * 4, rpp = 64 >> lpr_sh;
              const int cc = min(lc, (WW - 1) & ~3);
              struct __attribute__((packed, aligned(4))) Run4 { uint32_t v[4]; };
              for (int row0 = 0; row0 < WH; row0 += rpp) {  // (wave-uniform trip count: two passes for a 12 x 20 glyph)
                const int row = min(row0 + lr, WH - 1);
                const Run4 run = *reinterpret_cast<const Run4*>(tex + (((uint32_t)(wy0 + row)) << (uint32_t)__builtin_ctz((uint32_t)S)) + (uint32_t)(wx0 + cc));
                uint4 q4 = {run.v[0], run.v[1], run.v[2], run.v[3]};
                *reinterpret_cast<uint4*>(win + ((row0 + lr) << wsh) + lc) = q4;
              }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 4; k++) {  // (a pixel outside the quad may point anywhere: clamped into the window, blended with alpha 0)
              const int lx = min(max(fxi[k] - wx0, 0), WW - 2), ly = min(max(fyi[k] - wy0, 0), WH - 2);
              const uint32_t* w0 = win + (ly << wsh) + lx;
              q00[k] = w0[0]; q01[k] = w0[1]; q10[k] = w0[1 << wsh]; q11[k] = w0[(1 << wsh) + 1];
            }
            __builtin_amdgcn_wave_barrier();  // (the next draw of this strip overwrites the window)
          } else {
            const char* __restrict__ texb = reinterpret_cast<const char*>(tex);
            auto texel = [&](uint32_t off) __attribute__((always_inline)) { return *reinterpret_cast<const uint32_t*>(texb + off); };
#pragma unroll
            for (int k = 0; k < 4; k++) {  // GL_REPEAT, as atlas_bilinear()
              const uint32_t x0 = (uint32_t)(fxi[k] & msk), x1 = (x0 + 1u) & (uint32_t)msk, y0 = (uint32_t)(fyi[k] & msk), y1 = (y0 + 1u) & (uint32_t)msk;
              const uint32_t row0 = (y0 * (uint32_t)S) << 2, row1 = (y1 * (uint32_t)S) << 2;
              q00[k] = texel(row0 + (x0 << 2)); q01[k] = texel(row0 + (x1 << 2));
              q10[k] = texel(row1 + (x0 << 2)); q11[k] = texel(row1 + (x1 << 2));
            }
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const F4 t = bilinear_of(q00[k], q01[k], q10[k], q11[k], axK[k], ayK[k]);
            if (mode == 0u) {  // wave-uniform; atlas.frag:284-295
              sr[k] = t.x * colK[k].x; sg[k] = t.y * colK[k].y; sb[k] = t.z * colK[k].z; sa[k] = t.w * colK[k].w;
            } else {  // atlas.frag:296-318
              const bool use1 = T1[k];
              const float fwu = use1 ? q.fw_u[1] : q.fw_u[0], fwv = use1 ? q.fw_v[1] : q.fw_v[0];
              const F4 fc = eval_fill_rec(r, colK[k], fill_mode, uK[k], vK[k]);
              const float sd = is_mtsdf ? t.w : median3(t.x, t.y, t.z);
              const float unit = r.f0 * frcp(r.p0);  // pxRange / atlas size (atlas.frag:45-49)
              const float spr = __builtin_fmaxf(0.5f * (unit * frcp(fwu) + unit * frcp(fwv)), 1.0f);
              const float spd = spr * (sd - r.f1);
              const float alpha = is_stroke ? clamp01(__builtin_fmaxf(r.p1, 0.0f) * 0.5f - __builtin_fabsf(spd) + 0.5f) : clamp01(spd + 0.5f);
              sr[k] = fc.x; sg[k] = fc.y; sb[k] = fc.z; sa[k] = fc.w * alpha;
            }
          }
          blend(F0, sr[0], sg[0], sb[0], cov[0] ? sa[0] * mk0 * rm0 : 0.0f);
          blend(F1, sr[1], sg[1], sb[1], cov[1] ? sa[1] * mk1 * rm1 : 0.0f);
          blend(F2, sr[2], sg[2], sb[2], cov[2] ? sa[2] * mk2 * rm2 : 0.0f);
          blend(F3, sr[3], sg[3], sb[3], cov[3] ? sa[3] * mk3 * rm3 : 0.0f);
          return;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const bool use1 = T1[k];
          const float l0 = L0[k], l1 = L1[k], l2 = L2[k];
          // triangle 0 = (TL, BL, BR), triangle 1 = (TR, TL, BR): make_frag()'s corner table
          const float u0 = use1 ? utx : uax, v0 = uay, u1 = uax, v1 = use1 ? uay : uty;
          float u = l0 * u0 + l1 * u1 + l2 * utx;
          const float v = l0 * v0 + l1 * v1 + l2 * uty;
          F4 col = {cBL.x * inv255, cBL.y * inv255, cBL.z * inv255, cBL.w * inv255};
          if (!solid) {
            const float c0x = use1 ? cTR.x : cTL.x, c0y = use1 ? cTR.y : cTL.y, c0z = use1 ? cTR.z : cTL.z, c0w = use1 ? cTR.w : cTL.w;
            const float c1x = use1 ? cTL.x : cBL.x, c1y = use1 ? cTL.y : cBL.y, c1z = use1 ? cTL.z : cBL.z, c1w = use1 ? cTL.w : cBL.w;
            col.x = (l0 * c0x + l1 * c1x + l2 * cBR.x) * inv255;
            col.y = (l0 * c0y + l1 * c1y + l2 * cBR.y) * inv255;
            col.z = (l0 * c0z + l1 * c1z + l2 * cBR.z) * inv255;
            col.w = (l0 * c0w + l1 * c1w + l2 * cBR.w) * inv255;
          }
          const float fwu = use1 ? q.fw_u[1] : q.fw_u[0], fwv = use1 ? q.fw_v[1] : q.fw_v[0], lod = use1 ? q.lod[1] : q.lod[0];
          if (mode == 0u) {  // wave-uniform; atlas.frag:284-295
            if (om & F_SUBPIXEL) u -= r.aux * frcp(__builtin_fmaxf((float)P.atlas.size, 1.0f));
            const F4 t = atlas_sample(P.atlas, u, v, lod);
            sr[k] = t.x * col.x; sg[k] = t.y * col.y; sb[k] = t.z * col.z; sa[k] = t.w * col.w;
          } else {  // atlas.frag:296-318
            const F4 fc = eval_fill_rec(r, col, fill_mode, u, v);
            const F4 t = atlas_sample(P.atlas, u, v, 0.0f);  // textureLod(atlasTex, uv, 0.0)
            const float sd = is_mtsdf ? t.w : median3(t.x, t.y, t.z);
            const float unit = r.f0 * frcp(r.p0);  // pxRange / atlas size (atlas.frag:45-49)
            const float spr = __builtin_fmaxf(0.5f * (unit * frcp(fwu) + unit * frcp(fwv)), 1.0f);
            const float spd = spr * (sd - r.f1);
            const float alpha = is_stroke ? clamp01(__builtin_fmaxf(r.p1, 0.0f) * 0.5f - __builtin_fabsf(spd) + 0.5f) : clamp01(spd + 0.5f);
            sr[k] = fc.x; sg[k] = fc.y; sb[k] = fc.z; sa[k] = fc.w * alpha;
          }
        }
        blend(F0, sr[0], sg[0], sb[0], cov[0] ? sa[0] * mk0 * rm0 : 0.0f);
        blend(F1, sr[1], sg[1], sb[1], cov[1] ? sa[1] * mk1 * rm1 : 0.0f);
        blend(F2, sr[2], sg[2], sb[2], cov[2] ? sa[2] * mk2 * rm2 : 0.0f);
        blend(F3, sr[3], sg[3], sb[3], cov[3] ? sa[3] * mk3 * rm3 : 0.0f);
        return;
      }
      if (kRot && (om & F_GENERAL) != 0u && (om & F_EDGE32) != 0u && !atlas_mode && mode < 18u && (op == OP_DRAW || (kMasks && op == OP_MASK_PUSH))) {
        // ---- rotated / skewed SDF quads, 4 pixels per lane in lock-step (round 4; before: one pixel slot at a time through
        // shade_one(), 15x the time of the same tree unrotated).  The quad is the reference's two triangles (3,0,1) / (2,3,1) over
        // per-vertex ceil'd corners (glcontext.nim:418-429); a pixel belongs to the first whose three edge functions -- exact
        // integers in half-pixel units, top-left rule -- admit its centre, and its uv / vertex colour are that triangle's
        // barycentric interpolation, as in make_frag().  F_EDGE32 (Recorder::emit_corners): every edge value any pixel of the frame
        // can produce fits 32 bits and the coefficients fit 24, so the strip's scalar base + two v_mad_i32_i24 per edge replace the
        // 64-bit arithmetic; quads beyond that keep the slot path.  Ownership is folded into the base (E - 1 >= 0 <=> E > 0).
        FDH_COUNT(60);
        if (core) FDH_COUNT(63);
        if (core && (mode == 9u || mode == 11u || mode == 12u)) return;
        const QuadExt& q = exts[r.ext];
        float L0[4], L1[4], L2[4];
        bool T1[4], cov[4];
        tri_bary(q, L0, L1, L2, T1, cov, false);
        const bool solid = (om & F_SOLID) != 0u;
        const F4 cBL = unpack255(r.col[0]), cBR = unpack255(r.col[1]), cTR = unpack255(r.col[2]), cTL = unpack255(r.col[3]);
        const float qhx = r.p0, qhy = r.p1;
        float u[4], v[4], lx[4], nly[4];
        F4 col[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          // tri 0 = (TL, BL, BR): u = l2, v = l1 + l2;  tri 1 = (TR, TL, BR): u = l0 + l2, v = l2   (uv corners are 0 / 1: the
          // products of make_frag()'s sums are exact)
          const bool use1 = T1[k];
          const float la = use1 ? L0[k] : L1[k], lb = L2[k];
          const float sum = la + lb;
          u[k] = use1 ? sum : lb;
          v[k] = use1 ? lb : sum;
          if (solid) {
            col[k] = {cBL.x * inv255, cBL.y * inv255, cBL.z * inv255, cBL.w * inv255};
          } else {
            const float l0 = L0[k], l1 = L1[k];
            const float c0x = use1 ? cTR.x : cTL.x, c0y = use1 ? cTR.y : cTL.y, c0z = use1 ? cTR.z : cTL.z, c0w = use1 ? cTR.w : cTL.w;
            const float c1x = use1 ? cTL.x : cBL.x, c1y = use1 ? cTL.y : cBL.y, c1z = use1 ? cTL.z : cBL.z, c1w = use1 ? cTL.w : cBL.w;
            col[k].x = (l0 * c0x + l1 * c1x + lb * cBR.x) * inv255;
            col[k].y = (l0 * c0y + l1 * c1y + lb * cBR.y) * inv255;
            col[k].z = (l0 * c0z + l1 * c1z + lb * cBR.z) * inv255;
            col[k].w = (l0 * c0w + l1 * c1w + lb * cBR.w) * inv255;
          }
          lx[k] = (u[k] - 0.5f) * 2.0f * qhx;
          nly[k] = -((v[k] - 0.5f) * 2.0f * qhy);
        }
        const bool ellip = (om & F_ELLIP) != 0u;
        const uint32_t fill_mode = (om >> 9) & 7u;
        const bool inset = mode == 9u && op == OP_DRAW;
        const float shx = inset ? qhx : r.p2, shy = inset ? qhy : r.p3;
        const float spread = fill_mode == 0u ? r.f1 : 0.0f;
        float dist[4];
        if (core) { dist[0] = dist[1] = dist[2] = dist[3] = -1.0e30f; }  // (wave-uniform: the coverage term is saturated on this strip, QuadExt::core)
        else shape_distNy<4, true>(ellip, lx, nly, shx, shy, r.r[0], r.r[1], r.r[2], r.r[3], dist);
        if (kMasks && op == OP_MASK_PUSH) {  // mask.frag:186-234, as on the axis-aligned path below
          float mk[4] = {mk0, mk1, mk2, mk3};
          uint32_t packed = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            float a = (1.0f - clamp01(r.aa * dist[k] + 0.5f)) * col[k].w * mk[k];
            a = cov[k] ? a : 0.0f;
            const float qq = __builtin_rintf(a * a * 255.0f);
            packed |= (uint32_t)qq << (8 * k);
            mk[k] = qq * inv255;
          }
          mk0 = mk[0]; mk1 = mk[1]; mk2 = mk[2]; mk3 = mk[3];
          stack_put(mask_depth, packed);
          mask_depth++;
          return;
        }
        float alpha[4];
        switch (mode) {  // wave-uniform; atlas.frag:337-393, operation for operation what shade_one() does per pixel
          case 11u: {
            const float h = r.f0 * 0.5f;
#pragma unroll
            for (int k = 0; k < 4; k++) alpha[k] = (__builtin_fabsf(dist[k] + h) - h) < 0.0f ? 1.0f : 0.0f;
            break;
          }
          case 12u: {
            const float h = r.f0 * 0.5f;
#pragma unroll
            for (int k = 0; k < 4; k++) alpha[k] = 1.0f - clamp01(r.aa * (__builtin_fabsf(dist[k] + h) - h) + 0.5f);
            break;
          }
          case 7u: {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const float sd = dist[k] - spread;
              const float sp = __builtin_fminf(shadow_profile(sd, r.f0), 1.0f);
              alpha[k] = sd > 0.0f ? sp : 1.0f;
            }
            break;
          }
          case 8u: {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const float inside = 1.0f - clamp01(r.aa * dist[k] + 0.5f);
              const float sd = dist[k] - spread;
              const float sp = __builtin_fminf(shadow_profile(sd, r.f0), 1.0f);
              alpha[k] = sd >= 0.0f ? sp : inside;
            }
            break;
          }
          case 9u: {
            float sx[4], sy[4], shd[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { sx[k] = lx[k] - r.p2; sy[k] = nly[k] + r.p3; }
            shape_distNy<4, true>(ellip, sx, sy, qhx, qhy, r.r[0], r.r[1], r.r[2], r.r[3], shd);
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const float clip_a = 1.0f - clamp01(r.aa * dist[k] + 0.5f);
              const float sd = shd[k] + spread;
              const float sp = __builtin_fminf(shadow_profile(sd, r.f0), 1.0f);
              const float ia = sd < 0.0f ? sp : 1.0f;
              alpha[k] = clip_a * ia;
            }
            break;
          }
          default: {
#pragma unroll
            for (int k = 0; k < 4; k++) alpha[k] = 1.0f - clamp01(r.aa * dist[k] + 0.5f);
            break;
          }
        }
        float sr[4], sg[4], sb[4], sa[4];
        if (mode == 17u) {  // atlas.frag:381-388: the blurred backdrop at this fragment's own pixel (clamped addresses, no branch)
          F4 b[4] = {F0, F1, F2, F3};
          if (!(om & F_SELF_BACKDROP)) {
            const size_t rowp = (size_t)min(py, P.H - 1) * P.pitch;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const F4 t = unpack255(P.backdrop[rowp + min(px0 + k, P.W - 1)]);
              const bool in = row_ok && px0 + k < P.W;
              b[k].x = in ? t.x : b[k].x; b[k].y = in ? t.y : b[k].y; b[k].z = in ? t.z : b[k].z; b[k].w = in ? t.w : b[k].w;
            }
          }
#pragma unroll
          for (int k = 0; k < 4; k++) { sr[k] = b[k].x * inv255; sg[k] = b[k].y * inv255; sb[k] = b[k].z * inv255; sa[k] = b[k].w * inv255 * alpha[k]; }
        } else if (fill_mode == 0u) {
#pragma unroll
          for (int k = 0; k < 4; k++) { sr[k] = col[k].x; sg[k] = col[k].y; sb[k] = col[k].z; sa[k] = col[k].w * alpha[k]; }
        } else {
          const F4 mc = unpack255(r.mid), sc = unpack255(r.stop);
          const F4 m01 = {mc.x * inv255, mc.y * inv255, mc.z * inv255, mc.w * inv255}, s01 = {sc.x * inv255, sc.y * inv255, sc.z * inv255, sc.w * inv255};
          const float mid = __builtin_fminf(__builtin_fmaxf(r.f1, 0.01f), 0.99f);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const F4 fc = eval_fill_nb(col[k], m01, s01, fill_mode, mid, u[k], v[k]);
            sr[k] = fc.x; sg[k] = fc.y; sb[k] = fc.z; sa[k] = fc.w * alpha[k];
          }
        }
        blend(F0, sr[0], sg[0], sb[0], cov[0] ? sa[0] * mk0 * rm0 : 0.0f);
        blend(F1, sr[1], sg[1], sb[1], cov[1] ? sa[1] * mk1 * rm1 : 0.0f);
        blend(F2, sr[2], sg[2], sb[2], cov[2] ? sa[2] * mk2 * rm2 : 0.0f);
        blend(F3, sr[3], sg[3], sb[3], cov[3] ? sa[3] * mk3 * rm3 : 0.0f);
        return;
      }
      if (kSlow && FDH_BEZIER4 && mode >= 18u && mode <= 20u && ((om & F_GENERAL) == 0u || (om & F_EDGE32) != 0u) && op == OP_DRAW) {
        // ---- quadratic-bezier strokes on upright quads (drawQuadraticBezierSdf, modes 18 - 20: atlas.frag:121-209, 321-336), four
        // pixels per lane.  The quad is the span's bounding box; most of it is far from the curve.  The curve lies in the hull of its
        // control points, inside the box aligned with its chord AC that reaches min(0, b.f) .. max(|AC|, b.f) along it and 0 .. b.g / 2
        // across (b = B - A): a pixel farther from that box than sqrt 2 (half width + 0.5 / aa) has coverage exactly 0 in every
        // mode (the square cap of mode 20 reaches that far past an end point), and a strip of such pixels skips the cubic.
        const float Ax = r.p2, Ay = r.p3, Bx = r.r[0], By = r.r[1], Cx = r.r[2], Cy = r.r[3];
        const float hw = __builtin_fmaxf(r.f0, 0.0f) * 0.5f;
        const bool general = (om & F_GENERAL) != 0u;  // wave-uniform: the stroke under a rotated transform (uv from the two-triangle barycentrics)
        float lx[4], lyv[4], u[4], vv[4];
        bool cov[4];
        F4 colv[4];
        const F4 c0 = unpack255(r.col[0]);
        if (general) {
          float L0[4], L1[4], L2[4];
          bool T1[4];
          tri_bary(exts[r.ext], L0, L1, L2, T1, cov, true);
          const F4 cBR = unpack255(r.col[1]), cTR = unpack255(r.col[2]), cTL = unpack255(r.col[3]);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const bool use1 = T1[k];
            const float la = use1 ? L0[k] : L1[k], lb = L2[k], sum = la + lb;
            u[k] = use1 ? sum : lb;
            vv[k] = use1 ? lb : sum;
            lx[k] = (u[k] - 0.5f) * 2.0f * r.p0;
            lyv[k] = (vv[k] - 0.5f) * 2.0f * r.p1;
            colv[k] = {c0.x * inv255, c0.y * inv255, c0.z * inv255, c0.w * inv255};
            if (!(om & F_SOLID)) {
              const float l0 = L0[k], l1 = L1[k];
              const float c0x = use1 ? cTR.x : cTL.x, c0y = use1 ? cTR.y : cTL.y, c0z = use1 ? cTR.z : cTL.z, c0w = use1 ? cTR.w : cTL.w;
              const float c1x = use1 ? cTL.x : c0.x, c1y = use1 ? cTL.y : c0.y, c1z = use1 ? cTL.z : c0.z, c1w = use1 ? cTL.w : c0.w;
              colv[k].x = (l0 * c0x + l1 * c1x + lb * cBR.x) * inv255; colv[k].y = (l0 * c0y + l1 * c1y + lb * cBR.y) * inv255;
              colv[k].z = (l0 * c0z + l1 * c1z + lb * cBR.z) * inv255; colv[k].w = (l0 * c0w + l1 * c1w + lb * cBR.w) * inv255;
            }
          }
        } else {
          const bool rowc = py >= r.by0 && py < r.by1;
          const float t = (cy - r.oy) * r.inv_h;
          local_x4(r, cx0, lx);
          const float ly = -local_y_up(r, cy);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            u[k] = (cx0 + (float)k - r.ox) * r.inv_w;
            vv[k] = t;
            lyv[k] = ly;
            cov[k] = rowc && px0 + k >= r.bx0 && px0 + k < r.bx1;
            colv[k] = {c0.x * inv255, c0.y * inv255, c0.z * inv255, c0.w * inv255};
            if (!(om & F_SOLID)) {  // wave-uniform
              const F4 br = unpack255(r.col[1]), tr = unpack255(r.col[2]), tl = unpack255(r.col[3]);
              colv[k].x = tri_lerp(tl.x, c0.x, br.x, tr.x, u[k], t) * inv255; colv[k].y = tri_lerp(tl.y, c0.y, br.y, tr.y, u[k], t) * inv255;
              colv[k].z = tri_lerp(tl.z, c0.z, br.z, tr.z, u[k], t) * inv255; colv[k].w = tri_lerp(tl.w, c0.w, br.w, tr.w, u[k], t) * inv255;
            }
          }
        }
        float fx, fy, sx, sy, ex, ey;  // (wave-uniform: the record's alone)
        safe_normalize(Cx - Ax, Cy - Ay, 1.0f, 0.0f, fx, fy);
        {
          const float bf = (Bx - Ax) * fx + (By - Ay) * fy, bg = (By - Ay) * fx - (Bx - Ax) * fy;
          const float lac = (Cx - Ax) * fx + (Cy - Ay) * fy;
          const float x_lo = __builtin_fminf(0.0f, bf), x_hi = __builtin_fmaxf(lac, bf), y_lo = __builtin_fminf(0.0f, 0.5f * bg), y_hi = __builtin_fmaxf(0.0f, 0.5f * bg);
          const float ocx = 0.5f * (x_lo + x_hi), ohx = 0.5f * (x_hi - x_lo), ocy = 0.5f * (y_lo + y_hi), ohy = 0.5f * (y_hi - y_lo);
          const float reach = 1.41422f * (hw + 0.5f * frcp(r.aa)) + 0.01f;
          bool near = false;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float rx = lx[k] - Ax, ry = lyv[k] - Ay;
            const float X = rx * fx + ry * fy, Y = ry * fx - rx * fy;
            const float db = __builtin_fmaxf(__builtin_fabsf(X - ocx) - ohx, __builtin_fabsf(Y - ocy) - ohy);
            near = near || (cov[k] && db < reach);
          }
          if (!__any(near)) { FDH_COUNT(62); return; }
        }
        FDH_COUNT(61);
        float dist[4];
        if (general) sd_bezierN<4, true>(lx, lyv, Ax, Ay, Bx, By, Cx, Cy, dist);  // (wave-uniform)
        else sd_bezierN<4, false>(lx, lyv, Ax, Ay, Bx, By, Cx, Cy, dist);
        float alpha[4];
        if (mode == 18u) {
#pragma unroll
          for (int k = 0; k < 4; k++) alpha[k] = 1.0f - clamp01(r.aa * (dist[k] - hw) + 0.5f);
        } else {  // bezierStrokeSd atlas.frag:179-209
          safe_normalize(Bx - Ax, By - Ay, fx, fy, sx, sy);
          safe_normalize(Cx - Bx, Cy - By, fx, fy, ex, ey);
          const float trim = mode == 20u ? hw : 0.0f;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float start_proj = (lx[k] - Ax) * sx + (lyv[k] - Ay) * sy, end_proj = (lx[k] - Cx) * ex + (lyv[k] - Cy) * ey;
            float tube = dist[k];
            if (mode == 20u) {  // wave-uniform
              const float ta = __builtin_fminf(tube, __builtin_fabsf((lx[k] - Ax) * sy - (lyv[k] - Ay) * sx));
              tube = start_proj < 0.0f ? ta : tube;
              const float tb = __builtin_fminf(tube, __builtin_fabsf((lx[k] - Cx) * ey - (lyv[k] - Cy) * ex));
              tube = end_proj > 0.0f ? tb : tube;
            }
            const float cap = __builtin_fmaxf(-start_proj - trim, end_proj - trim);
            alpha[k] = 1.0f - clamp01(r.aa * __builtin_fmaxf(tube - hw, cap) + 0.5f);
          }
        }
        const uint32_t fill_mode = (om >> 9) & 7u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const F4 fc = eval_fill_rec(r, colv[k], fill_mode, u[k], vv[k]);
          const float mkk = k == 0 ? mk0 : k == 1 ? mk1 : k == 2 ? mk2 : mk3, rmk = k == 0 ? rm0 : k == 1 ? rm1 : k == 2 ? rm2 : rm3;
          F4& F = k == 0 ? F0 : k == 1 ? F1 : k == 2 ? F2 : F3;
          blend(F, fc.x, fc.y, fc.z, cov[k] ? fc.w * alpha[k] * mkk * rmk : 0.0f);
        }
        return;
      }
      if (!kSlow && !fast) return;  // unreachable: the host picks kSlow = true for any phase holding such a draw
      if (kSlow && !fast) {
        FDH_COUNT(1);
        // ---- one pixel slot at a time, state rotated so slot 0 is always the live one
        uint32_t packed = 0;
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
          const int px = px0 + k;
          if (op == OP_RMASK_BEGIN) {
            rmask_on = true;
            rm0 = rect_mask_alpha(draws[d], (float)px + 0.5f, cy);
          } else {
            const bool in_frame = row_ok && px < P.W;
            const Src s = shade_one(draws + d, exts, &P.atlas, P.backdrop, pix + k, in_frame, px, py, F0);
            if (op == OP_MASK_PUSH) {
              float a = s.covered ? s.a * mk0 : 0.0f;
              const float q = __builtin_rintf(a * a * 255.0f);
              packed |= (uint32_t)q << (8 * k);
              mk0 = q * inv255;
            } else {
              float a = s.a * mk0 * rm0;
              a = s.covered ? a : 0.0f;
              blend(F0, s.r, s.g, s.b, a);
            }
          }
          { const F4 t = F0; F0 = F1; F1 = F2; F2 = F3; F3 = t; }
          { const float t = mk0; mk0 = mk1; mk1 = mk2; mk2 = mk3; mk3 = t; }
          { const float t = rm0; rm0 = rm1; rm1 = rm2; rm2 = rm3; rm3 = t; }
        }
        if (op == OP_MASK_PUSH) { stack_put(mask_depth, packed); mask_depth++; }
        return;
      }

      // ---- fast path: axis-aligned SDF draw / clip push, 4 pixels per lane in lock-step
      const bool ellip = (om & F_ELLIP) != 0u;
      const uint32_t fill_mode = (om >> 9) & 7u;
      // `core`: the strip lies in the draw's saturated core (DrawRec::ix0..iy1, decided per strip by k_bin_draws): the
      // whole strip has coverage alpha 1 (annular strokes, alpha 0 there, never get this far).
#ifdef FDH_ABLATE_SHADING  // ablation build (make variant): records are fetched, nothing is shaded
      { F0.x += core ? 1e-30f : 0.0f; return; }
#endif
      if (core) {
        if (mode == 9u || mode == 11u || mode == 12u) { FDH_COUNT(32); if (kShader) deep_nop(rk); return; }
        if (op == OP_DRAW && (om & F_SOLID) && fill_mode == 0u && mode != 17u) {
          FDH_COUNT(33);
          const F4 c0 = unpack255(r.col[0]);
          const float sa = c0.w * inv255;
          if (mask_depth == 0 && !rmask_on) {  // one source term for the whole strip
            const float A = 255.0f * sa, ia = 1.0f - sa;
            const f2 c_rg = {c0.x * inv255 * A, c0.y * inv255 * A}, c_ba = {c0.z * inv255 * A, A};
            if (kShader) {
              deep_publish(deep_slot(rk), rk, DT_UNIFORM_PRE, __float_as_uint(c_rg.x), __float_as_uint(c_rg.y), __float_as_uint(c_ba.x), __float_as_uint(c_ba.y), __float_as_uint(ia));
              return;
            }
            blend_pre(F0, c_rg, c_ba, ia); blend_pre(F1, c_rg, c_ba, ia); blend_pre(F2, c_rg, c_ba, ia); blend_pre(F3, c_rg, c_ba, ia);
          } else {
            const float cr = c0.x * inv255, cg = c0.y * inv255, cb = c0.z * inv255;
            blend(F0, cr, cg, cb, sa * mk0 * rm0); blend(F1, cr, cg, cb, sa * mk1 * rm1);
            blend(F2, cr, cg, cb, sa * mk2 * rm2); blend(F3, cr, cg, cb, sa * mk3 * rm3);
          }
          return;
        }
        FDH_COUNT(34);
      }
#ifdef FDH_ABLATE_EDGE  // ablation build (make variant): edge strips are skipped, core strips shaded
      if (!core) return;
#endif
      FDH_COUNT(8 + (mode & 31u));
      if (ellip) FDH_COUNT(2);
      if (!(om & F_SOLID)) FDH_COUNT(3);
      if (fill_mode != 0u) FDH_COUNT(4);
      const float t = (cy - r.oy) * r.inv_h;  // v of the quad (uv = (0,0)-(1,1) for SDF quads)
      const bool rowc = py >= r.by0 && py < r.by1;
      // (wave-uniform) every pixel of the strip is covered by the quad: the strip lies in the draw's saturated core, which lies inside
      // the quad (set_saturated_core), or k_bin_draws found it inside the quad's pixel bounds (BR_BOX_EXACT)
      const bool all_cov = core || inq;
      float u[4];
      bool cov[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        u[k] = (cx0 + (float)k - r.ox) * r.inv_w;
        cov[k] = rowc && px0 + k >= r.bx0 && px0 + k < r.bx1;
      }
      F4 col[4];
      {
        const F4 c0 = unpack255(r.col[0]);
        if (om & F_SOLID) {
#pragma unroll
          for (int k = 0; k < 4; k++) col[k] = {c0.x * inv255, c0.y * inv255, c0.z * inv255, c0.w * inv255};
        } else {
          const F4 br = unpack255(r.col[1]), tr = unpack255(r.col[2]), tl = unpack255(r.col[3]);
          // Which of the quad's two triangles interpolates (s > t: the upper one) is the same for every pixel of most strips -- the
          // diagonal crosses few of them: then only that triangle's expression is evaluated (two FMAs per channel instead of four and a select)
          const bool up_all = __all(u[0] > t && u[1] > t && u[2] > t && u[3] > t), low_all = __all(!(u[0] > t) && !(u[1] > t) && !(u[2] > t) && !(u[3] > t));
          if (up_all) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              col[k].x = tri_upper(tl.x, br.x, tr.x, u[k], t) * inv255; col[k].y = tri_upper(tl.y, br.y, tr.y, u[k], t) * inv255;
              col[k].z = tri_upper(tl.z, br.z, tr.z, u[k], t) * inv255; col[k].w = tri_upper(tl.w, br.w, tr.w, u[k], t) * inv255;
            }
          } else if (low_all) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              col[k].x = tri_lower(tl.x, c0.x, br.x, u[k], t) * inv255; col[k].y = tri_lower(tl.y, c0.y, br.y, u[k], t) * inv255;
              col[k].z = tri_lower(tl.z, c0.z, br.z, u[k], t) * inv255; col[k].w = tri_lower(tl.w, c0.w, br.w, u[k], t) * inv255;
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              col[k].x = tri_lerp(tl.x, c0.x, br.x, tr.x, u[k], t) * inv255;
              col[k].y = tri_lerp(tl.y, c0.y, br.y, tr.y, u[k], t) * inv255;
              col[k].z = tri_lerp(tl.z, c0.z, br.z, tr.z, u[k], t) * inv255;
              col[k].w = tri_lerp(tl.w, c0.w, br.w, tr.w, u[k], t) * inv255;
            }
          }
        }
      }
      const float qhx = r.p0, qhy = r.p1;
      const bool inset = mode == 9u && op == OP_DRAW;
      const float shx = inset ? qhx : r.p2, shy = inset ? qhy : r.p3;
      float lx[4], dist[4];
      local_x4(r, cx0, lx);
      const float ly = -local_y_up(r, cy);
      const float spread = fill_mode == 0u ? r.f1 : 0.0f;
      // Tile classification.  Along a row the rounded-box distance is quasi-convex (its sub-level sets are
      // intervals), so if the two OUTER pixels of every lane's 4-pixel run give the same saturated alpha, the two
      // inner ones do too -- exactly, not approximately.  cls 1: alpha == 1 on the whole 32x8 strip (shape interior);
      // cls 2: alpha == 0 (deep inside a stroke): the draw is a no-op for this strip.
      int cls = 0;
      if (core) {
        cls = 1;
#pragma unroll
        for (int k = 0; k < 4; k++) dist[k] = -1.0e30f;
      } else if (!ellip) {
        const float lxo[2] = {lx[0], lx[3]};
        float d2[2];
        shape_distN<2>(false, lxo, -ly, shx, shy, r.r[0], r.r[1], r.r[2], r.r[3], d2);
        dist[0] = d2[0];
        dist[3] = d2[1];
        const float dm = __builtin_fmaxf(d2[0], d2[1]);
        bool one = false, zero = false;
        if (op == OP_MASK_PUSH || mode == 3u || mode == 17u) one = __builtin_fmaf(-dm, r.aa, 0.5f) >= 1.0f;
        else if (mode == 7u) one = dm - spread <= 0.0f;
        else if (mode == 12u) { const float h = r.f0 * 0.5f; zero = (dm + h < 0.0f) & (__builtin_fmaf(-(-(dm + h) - h), r.aa, 0.5f) <= 0.0f); }
        if (__all(one)) cls = 1;
        else if (__all(zero)) cls = 2;
        if (cls == 1) FDH_COUNT(5);
        if (cls == 2) FDH_COUNT(6);
        if (cls == 1 && r.bx0 <= tx0 && r.bx1 >= tx1 && r.by0 <= ty0 && r.by1 >= ty1) FDH_COUNT(7);
        if (cls == 1) FDH_COUNT(40 + (mode & 15u));
        if (cls == 2) { if (kShader) deep_nop(rk); return; }
        if (cls == 0) {
          const float lxi[2] = {lx[1], lx[2]};
          shape_distN<2>(false, lxi, -ly, shx, shy, r.r[0], r.r[1], r.r[2], r.r[3], d2);
          dist[1] = d2[0];
          dist[2] = d2[1];
        } else {
          dist[1] = dist[2] = dm;
        }
      } else {
        shape_distN<4>(true, lx, -ly, shx, shy, r.r[0], r.r[1], r.r[2], r.r[3], dist);
      }

      if (kMasks && op == OP_MASK_PUSH) {
        // mask.frag:186-234 drawn through the blender into a cleared R8 plane: stored = q8(a*a), a = shape*parent
        float mk[4] = {mk0, mk1, mk2, mk3};
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float a = cover_aa(dist[k], r.aa) * col[k].w * mk[k];
          a = cov[k] ? a : 0.0f;
          const float q = __builtin_rintf(a * a * 255.0f);
          packed |= (uint32_t)q << (8 * k);
          mk[k] = q * inv255;
        }
        mk0 = mk[0]; mk1 = mk[1]; mk2 = mk[2]; mk3 = mk[3];
        stack_put(mask_depth, packed);
        mask_depth++;
        return;
      }

      // ---- OP_DRAW: atlas.frag main():252-405
      float alpha[4];
      if (cls == 1) {  // wave-uniform: saturated coverage
#pragma unroll
        for (int k = 0; k < 4; k++) alpha[k] = 1.0f;
      } else switch (mode) {  // wave-uniform
        case 11u: {
          const float h = r.f0 * 0.5f;
#pragma unroll
          for (int k = 0; k < 4; k++) alpha[k] = (__builtin_fabsf(dist[k] + h) - h) < 0.0f ? 1.0f : 0.0f;
          break;
        }
        case 12u: {
          const float h = r.f0 * 0.5f;
#pragma unroll
          for (int k = 0; k < 4; k++) alpha[k] = cover_aa(__builtin_fabsf(dist[k] + h) - h, r.aa);
          break;
        }
        case 7u: {
#pragma unroll
          for (int k = 0; k < 4; k++) alpha[k] = shadow_profile(__builtin_fmaxf(dist[k] - spread, 0.0f), r.f0);  // (= sd > 0 ? min(profile, 1) : 1: edge_blend)
          break;
        }
        case 8u: {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float inside = cover_aa(dist[k], r.aa);
            const float sd = dist[k] - spread;
            const float sp = __builtin_fminf(shadow_profile(sd, r.f0), 1.0f);
            alpha[k] = sd >= 0.0f ? sp : inside;
          }
          break;
        }
        case 9u: {  // atlas.frag:364-380
          float sx[4], shd[4];
#pragma unroll
          for (int k = 0; k < 4; k++) sx[k] = lx[k] - r.p2;
          shape_distN<4>(ellip, sx, -ly + r.p3, qhx, qhy, r.r[0], r.r[1], r.r[2], r.r[3], shd);
#pragma unroll
          for (int k = 0; k < 4; k++) alpha[k] = cover_aa(dist[k], r.aa) * shadow_profile(__builtin_fminf(shd[k] + spread, 0.0f), r.f0);  // (edge_blend)
          break;
        }
        default: {  // ClipAA / BackdropBlur / others: atlas.frag:389-393
#pragma unroll
          for (int k = 0; k < 4; k++) alpha[k] = cover_aa(dist[k], r.aa);
          break;
        }
      }
      float sr[4], sg[4], sb[4], sa[4];
      if (mode == 17u) {  // atlas.frag:381-388: the blurred backdrop at this fragment's own pixel
        F4 b[4] = {F0, F1, F2, F3};
        if (!(om & F_SELF_BACKDROP)) {
          if (__all(vec_ok)) {  // (wave-uniform on purpose: see )
            const uint4 q = *reinterpret_cast<const uint4*>(P.backdrop + pix);
            b[0] = unpack255(q.x); b[1] = unpack255(q.y); b[2] = unpack255(q.z); b[3] = unpack255(q.w);
          } else {  // a strip on the frame's right or bottom edge: clamped addresses, no branch
            int here = 0;  // (opaque: keeps the address arithmetic of this rare branch from being hoisted into every strip's prologue)
            asm volatile("" : "+s"(here));
            const size_t rowp = (size_t)min(py + here, P.H - 1) * P.pitch;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const F4 t = unpack255(P.backdrop[rowp + min(px0 + k + here, P.W - 1)]);
              const bool in = row_ok && px0 + k + here < P.W;
              b[k].x = in ? t.x : b[k].x; b[k].y = in ? t.y : b[k].y; b[k].z = in ? t.z : b[k].z; b[k].w = in ? t.w : b[k].w;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { sr[k] = b[k].x * inv255; sg[k] = b[k].y * inv255; sb[k] = b[k].z * inv255; sa[k] = b[k].w * inv255 * alpha[k]; }
      } else if (fill_mode == 0u) {
#pragma unroll
        for (int k = 0; k < 4; k++) { sr[k] = col[k].x; sg[k] = col[k].y; sb[k] = col[k].z; sa[k] = col[k].w * alpha[k]; }
      } else {
        const F4 mc = unpack255(r.mid), sc = unpack255(r.stop);
        const F4 m01 = {mc.x * inv255, mc.y * inv255, mc.z * inv255, mc.w * inv255}, s01 = {sc.x * inv255, sc.y * inv255, sc.z * inv255, sc.w * inv255};
        const float mid = __builtin_fminf(__builtin_fmaxf(r.f1, 0.01f), 0.99f);
        // evalFillColor (atlas.frag:233-250) picks the stop pair (start, mid) or (mid, end) per pixel by t <= mid.  A 32 x 8 strip
        // nearly always lies on ONE side of the middle stop: then the pair is the same for the whole strip and the eight selects per
        // pixel (and the select between the two weights) are not needed -- the operands they would have picked are used as they are
        float tt[4];
        bool lo_all = true, lo_none = true;
#pragma unroll
        for (int k = 0; k < 4; k++) { tt[k] = fill_t(fill_mode, u[k], t); lo_all = lo_all && tt[k] <= mid; lo_none = lo_none && !(tt[k] <= mid); }
        FDH_COUNT(__all(lo_all) ? 68 : __all(lo_none) ? 69 : 70);
        if (__all(lo_all)) {
          const float rw = frcp(mid);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float w = tt[k] * rw;
            sr[k] = mixf(col[k].x, m01.x, w); sg[k] = mixf(col[k].y, m01.y, w); sb[k] = mixf(col[k].z, m01.z, w); sa[k] = mixf(col[k].w, m01.w, w) * alpha[k];
          }
        } else if (__all(lo_none)) {
          const float rw = frcp(1.0f - mid);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const float w = (tt[k] - mid) * rw;
            sr[k] = mixf(m01.x, s01.x, w); sg[k] = mixf(m01.y, s01.y, w); sb[k] = mixf(m01.z, s01.z, w); sa[k] = mixf(m01.w, s01.w, w) * alpha[k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const F4 fc = eval_fill_nb(col[k], m01, s01, fill_mode, mid, u[k], t);
            sr[k] = fc.x; sg[k] = fc.y; sb[k] = fc.z; sa[k] = fc.w * alpha[k];
          }
        }
      }
      // mask multiply (atlas.frag:401-404), rect mask (atlas_rect_mask.frag:425); outside the quad alpha is forced
      // to 0: blending with alpha 0 leaves the integer texel exactly as it is
      // (the masked alphas are formed BEFORE the uniform branch, for both of its sides: as arms of the selects below they became a
      // divergent branch around two multiplies in the builds with masks -- tools/lint_isa.py)
      const float am[4] = {sa[0] * mk0 * rm0, sa[1] * mk1 * rm1, sa[2] * mk2 * rm2, sa[3] * mk3 * rm3};
      FDH_COUNT(all_cov ? 66 : 67);
      if (kShader) {  // a deep strip's shader: the source term goes into the ring, the blender does what follows (deep_consume)
        const uint32_t slot = deep_slot(rk);
        float* v = ring.data + slot * kDeepSlotFloats + lane;
        if (mode == 17u && (om & F_SELF_BACKDROP) != 0u) {  // the source is the strip's own texel, which only the blender holds: the coverage goes over
#pragma unroll
          for (int k = 0; k < 4; k++) { const float a = alpha[k]; v[64 * k] = all_cov ? a : (cov[k] ? a : 0.0f); }
          deep_publish(slot, rk, DT_SELF17, 0u, 0u, 0u, 0u, 0u);
          return;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          v[64 * k] = sr[k]; v[64 * (4 + k)] = sg[k]; v[64 * (8 + k)] = sb[k];
          v[64 * (12 + k)] = all_cov ? am[k] : (cov[k] ? am[k] : 0.0f);
        }
        deep_publish(slot, rk, DT_GENERIC, 0u, 0u, 0u, 0u, 0u);
        return;
      }
      if (all_cov) {
        blend(F0, sr[0], sg[0], sb[0], am[0]); blend(F1, sr[1], sg[1], sb[1], am[1]);
        blend(F2, sr[2], sg[2], sb[2], am[2]); blend(F3, sr[3], sg[3], sb[3], am[3]);
        return;
      }
      blend(F0, sr[0], sg[0], sb[0], cov[0] ? am[0] : 0.0f);
      blend(F1, sr[1], sg[1], sb[1], cov[1] ? am[1] : 0.0f);
      blend(F2, sr[2], sg[2], sb[2], cov[2] ? am[2] : 0.0f);
      blend(F3, sr[3], sg[3], sb[3], cov[3] ? am[3] : 0.0f);
    };
    while (m != 0) {
      const int bit = __builtin_ctzll(m);
      const unsigned long long one = 1ull << bit;  // (one shift serves the removal and every class test below)
      m &= ~one;
      const uint32_t word = __builtin_amdgcn_readlane(idx, bit);
      const uint32_t d = word & LE_INDEX;
#if FDH_TIMING
      n_all_t++;
#endif
      const bool unclipped = !kMasks || (mask_depth == 0 && !rmask_on);
#ifdef FDH_ABLATE_PATHS  // ablation builds (tools/ablate_paths.sh): what each path of the draw loop costs, by leaving it out
      if ((FDH_ABLATE_PATHS & 1) && unclipped && (m_plainc & one) != 0ull) { touched = true; continue; }
      if (unclipped && (m_simple & one) != 0ull && ((FDH_ABLATE_PATHS >> (1 + ((((word >> LE_PATH_SHIFT) & 15u) - 1u) & 3u))) & 1)) { touched = true; continue; }
      if ((FDH_ABLATE_PATHS & 32) && !(unclipped && ((m_plainc | m_simple) & one) != 0ull) && (m_core & one) != 0ull) { touched = true; continue; }
      if ((FDH_ABLATE_PATHS & 64) && !(unclipped && ((m_plainc | m_simple) & one) != 0ull) && (m_core & one) == 0ull) { touched = true; continue; }
#endif
      if (unclipped && (m_plainc & one) != 0ull) {
        if (kShader) continue;  // (a deep strip: the blender's own)
        // One colour, coverage 1, nothing clipping: the whole strip gets the same source term.  Only the colour is
        // fetched (4 bytes instead of the 128-byte record) and nothing of the record is decoded.
        // (col[1..3] of such a record hold c / 255 as floats: Context::prepare)
        u32x4 c4;
        if (kBlender) {
          c4 = u32x4{(uint32_t)__builtin_amdgcn_readlane(plain_col.x, bit), (uint32_t)__builtin_amdgcn_readlane(plain_col.y, bit),
                     (uint32_t)__builtin_amdgcn_readlane(plain_col.z, bit), (uint32_t)__builtin_amdgcn_readlane(plain_col.w, bit)};
        } else {
          c4 = *reinterpret_cast<const u32x4*>(draws[d].col);
          asm volatile("" : "+s"(c4));
        }
        const float sa = (float)(c4.x >> 24) * inv255, A = 255.0f * sa, ia = 1.0f - sa;
        const f2 c_rg = {__uint_as_float(c4.y) * A, __uint_as_float(c4.z) * A}, c_ba = {__uint_as_float(c4.w) * A, A};
        blend_pre(F0, c_rg, c_ba, ia); blend_pre(F1, c_rg, c_ba, ia); blend_pre(F2, c_rg, c_ba, ia); blend_pre(F3, c_rg, c_ba, ia);
        touched = true;
        FDH_COUNT(35);
        continue;
      }
      if (kBlender) {  // a deep strip's blender: every other draw's source term comes out of the ring, in list order
        touched = true;
        deep_consume(rank);
        rank++;
        continue;
      }
#if FDH_SIMPLE_EDGE && !defined(FDH_ABLATE_SHADING) && !defined(FDH_ABLATE_EDGE)
      // the path code rides in the list entry: the branch is taken on a value that is already in an SGPR, and the record
      // is fetched whole, once, behind it
      const uint32_t code = (word >> LE_PATH_SHIFT) & 15u;
      if (unclipped && (m_simple & one) != 0ull) {
        // (a deep strip's shaders take the shading units -- this draw and the run that shares its field -- in turn; the others' are
        // walked over on the list words alone, counting their source terms)
        const bool mine = !kShader || unit % 3u == (uint32_t)shader_id;
        unit++;
        if (kShader && !mine) {
          rank++;
          uint32_t wcur = word, dcur = d;
          while ((wcur & LE_SHARE) != 0u && m != 0) {
            const int nb = __builtin_ctzll(m);
            const uint32_t w2 = __builtin_amdgcn_readlane(idx, nb);
            const unsigned long long one2 = 1ull << nb;
            if ((w2 & LE_INDEX) != dcur + 1u || (m_simple & one2) == 0ull || (((w2 >> LE_PATH_SHIFT) & 15u) > 4u) != (code > 4u)) break;
            m &= ~one2;
            dcur++;
            wcur = w2;
            rank++;
          }
          continue;
        }
        const DrawRec r = load_rec_whole(draws + d);
        const uint32_t c4 = (code - 1u) & 3u;
        const uint32_t mode = c4 == 0u ? 3u : c4 == 1u ? 7u : c4 == 2u ? 9u : 12u;
        touched = true;
        FDH_COUNT(48 + code);
#if FDH_EDGE_CHECK
        F4 S0 = F0, S1 = F1, S2 = F2, S3 = F3;
        simple_edge(r, mode, code > 4u, S0, S1, S2, S3);
        shade(d, r, false, false, 0u);
        {
          const bool ellip = code > 4u;
          const float gS[16] = {S0.x, S0.y, S0.z, S0.w, S1.x, S1.y, S1.z, S1.w, S2.x, S2.y, S2.z, S2.w, S3.x, S3.y, S3.z, S3.w};
          const float gF[16] = {F0.x, F0.y, F0.z, F0.w, F1.x, F1.y, F1.z, F1.w, F2.x, F2.y, F2.z, F2.w, F3.x, F3.y, F3.z, F3.w};
#pragma unroll
          for (int e = 0; e < 16; e++)
            if (__float_as_uint(gS[e]) != __float_as_uint(gF[e])) {
              const unsigned int i = atomicAdd(&g_edge_bad_n, 1u);
              if (i < 4096u) { unsigned int* o = g_edge_bad + 8 * i; o[0] = mode; o[1] = blockIdx.x; o[2] = d; o[3] = (unsigned)e; o[4] = (unsigned)lane; o[5] = (ellip ? 1u : 0u) | ((unsigned)(reinterpret_cast<uintptr_t>(P.fb) >> 12) << 1); o[6] = __float_as_uint(gS[e]); o[7] = __float_as_uint(gF[e]); }
            }
        }
#else
        {
          const bool ellip = code > 4u;
          f2 lxa, lxb, da, db;
          float pyy;
          edge_geom(r, mode == 9u, ellip, lxa, lxb, pyy, da, db);
          const bool inq = (m_inq & one) != 0ull;
          edge_blend(r, mode, ellip, inq, r.p2, r.p3, r.f0, r.f1, u32x4{r.col[0], r.col[1], r.col[2], r.col[3]}, lxa, lxb, pyy, da, db, F0, F1, F2, F3, rank);
          rank++;
          // The draws that follow over the same quad and shape (the node's stroke, its inner shadows: LE_SHARE on the entry of
          // the draw before them) reuse the field: their entries are taken off the list here.  All conditions are wave-uniform.
          uint32_t wcur = word, dcur = d;
          while ((wcur & LE_SHARE) != 0u && m != 0) {
            const int nb = __builtin_ctzll(m);
            const uint32_t w2 = __builtin_amdgcn_readlane(idx, nb);
            const uint32_t code2 = (w2 >> LE_PATH_SHIFT) & 15u;
            const unsigned long long one2 = 1ull << nb;
            if ((w2 & LE_INDEX) != dcur + 1u || (m_simple & one2) == 0ull || (code2 > 4u) != ellip) break;
            m &= ~one2;
            dcur++;
            wcur = w2;
            // the member's own parameters: sdfParams.zw + sdfFactors (16 bytes at offset 32) and its colour (offset 64)
            const u32x4* __restrict__ mp = reinterpret_cast<const u32x4*>(draws + dcur);
            u32x4 q = mp[2];
            u32x4 mcol = mp[4];
            asm volatile("" : "+s"(q), "+s"(mcol));
            const uint32_t c42 = (code2 - 1u) & 3u;
            const uint32_t mode2 = c42 == 0u ? 3u : c42 == 1u ? 7u : c42 == 2u ? 9u : 12u;
            FDH_COUNT(57);
            edge_blend(r, mode2, ellip, (m_inq & one2) != 0ull, __uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w), mcol, lxa, lxb, pyy, da, db, F0, F1, F2, F3, rank);
            rank++;
          }
        }
#endif
        continue;
      }
#endif
      if (kShader) {  // (a unit of one draw)
        const bool mine = unit % 3u == (uint32_t)shader_id;
        unit++;
        if (!mine) { rank++; continue; }
      }
      const DrawRec r = load_rec_whole(draws + d);
      const bool core = (m_core & one) != 0ull;
#if FDH_TIMING
      const unsigned long long Ts0 = FDH_NOW() + (r.op_mode & 0u);
#endif
      shade(d, r, core, (m_inq & one) != 0ull, rank);
      rank++;
#if FDH_TIMING
      {
        const unsigned long long Ts1 = FDH_NOW() + (__builtin_amdgcn_readfirstlane(__float_as_uint(F0.x + F1.x + F2.x + F3.x)) & 0u);
        const uint32_t md = r.op_mode & 255u;
        const int slot = md == 3u ? 0 : md == 7u ? 1 : md == 9u ? 2 : 3;
        if (!core) { T_mode[slot] += Ts1 - Ts0; N_mode[slot]++; }
        T_shade += Ts1 - Ts0; n_draws_t++;
      }
#endif
    }
  }
#if FDH_TIMING
  if (lane == 0 && blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) < 65536) {  // (one row per wave: the four of a k_composite_deep workgroup side by side)
    const unsigned long long T1 = FDH_NOW();
    unsigned long long* row = g_wave_times + 16 * ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    row[0] = T1 - T0; row[1] = W0 + (T_cnt & 0ull); row[2] = wall_clock64() + (T_cull & 0ull);  // (wall clock, 100 MHz, one counter for the device: tools/wave_timeline.py)
    row[3] = T_rec; row[4] = T_shade; row[5] = n_all_t; row[6] = 1 + kRole; row[7] = T_cull_core * 1024 + n_core_t;  // (row[5]: list entries walked; row[6]: 1 a strip's one wave, 2 a deep strip's blender, 3 one of its shaders)
    for (int i = 0; i < 4; i++) { row[8 + i] = T_mode[i]; row[12 + i] = N_mode[i]; }
  }
#endif
  // The store address is derived again from an (opaque) lane index: kept from the prologue it held three VGPRs across
  // the whole draw loop, the three that stood between the no-clip build and six waves per SIMD.
  if (kShader) return;  // (a deep strip's shaders hold no texels)
  int lane_e = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_e));
  const int px0e = tx0 + (lane_e & 7) * 4, pye = ty0 + (lane_e >> 3);
  if (!(touched || kFull || !P.load_fb) || pye < P.row_lo || pye >= P.row_hi) return;
  const bool row_ok_e = pye < P.H;
  const size_t pixe = (size_t)pye * P.pitch + px0e;
  if (row_ok_e && px0e + 3 < P.W && (P.pitch & 3) == 0) {
    // (a strip of the launch that starts the frame on which nothing landed -- wave-uniform -- is the clear colour as it is: no packing)
    uint4 o = {P.clear_rgba8, P.clear_rgba8, P.clear_rgba8, P.clear_rgba8};
    if (!kFull || touched) o = uint4{pack255(F0), pack255(F1), pack255(F2), pack255(F3)};
    *reinterpret_cast<uint4*>(P.fb + pixe) = o;
  } else if (row_ok_e) {
    if (px0e + 0 < P.W) P.fb[pixe + 0] = pack255(F0);
    if (px0e + 1 < P.W) P.fb[pixe + 1] = pack255(F1);
    if (px0e + 2 < P.W) P.fb[pixe + 2] = pack255(F2);
    if (px0e + 3 < P.W) P.fb[pixe + 3] = pack255(F3);
  }
}

#if FDH_TU == 0
// ------------------------------------------------------------------ blur (blur.frag:11-32 as a merged FIR)
// blur.frag takes 17 bilinear taps at i*step px.  The step is constant, so tap i has the same bilinear fraction at
// every pixel and the pass is a fixed FIR over integer offsets (BlurTaps, built on the host).  Each thread produces
// FOUR consecutive outputs along the filter direction: every staged texel is unpacked once (4 x v_cvt_f32_ubyte)
// and feeds up to four accumulators (float2 pairs: r, g and b, a), instead of being re-read and re-unpacked per tap.
__device__ __forceinline__ void unpack2(uint32_t c, f2& rg, f2& ba) {
  rg.x = (float)(c & 255u);
  rg.y = (float)((c >> 8) & 255u);
  ba.x = (float)((c >> 16) & 255u);
  ba.y = (float)(c >> 24);
}
__device__ __forceinline__ uint32_t pack2(f2 rg, f2 ba) {
  // v_cvt_pk_u8_f32: round-to-nearest-even conversion to 0..255 dropped into one byte of the destination dword --
  // four instructions for a texel instead of 4 x v_rndne + 4 x v_cvt_u32 + 3 x v_lshl_or
  uint32_t o = __builtin_amdgcn_cvt_pk_u8_f32(rg.x, 0, 0u);
  o = __builtin_amdgcn_cvt_pk_u8_f32(rg.y, 1, o);
  o = __builtin_amdgcn_cvt_pk_u8_f32(ba.x, 2, o);
  return __builtin_amdgcn_cvt_pk_u8_f32(ba.y, 3, o);
}

// NOUT = consecutive outputs per thread along the filter direction: 8 for large regions (every staged texel is unpacked
// once per 8 outputs), 2 for small ones (a 360x240 backdrop is ~50 workgroups at NOUT = 8: a few long serial waves on an
// empty machine; at NOUT = 2 four times as many waves each run a four times shorter chain).

// One thread's kBlurOut consecutive outputs of the merged FIR.  `tex(j)` returns window texel j (output p sees it at
// offset j - p - reach); every texel is unpacked once and feeds the accumulators pair by pair.  The first and last
// kBlurOut - 1 window texels reach only some of the outputs (the rest would multiply the zero padding of `dense`):
// those two triangles are peeled with the in-range (p, j) pairs spelled out at compile time.
template <int kBlurOut, int kUnroll = 1, typename Tex>
__device__ __forceinline__ void fir_outputs(const float* __restrict__ d, int reach, Tex tex, f2 (&rg)[kBlurOut], f2 (&ba)[kBlurOut]) {
#pragma unroll
  for (int p = 0; p < kBlurOut; p++) { rg[p] = 0.0f; ba[p] = 0.0f; }
  const int nwin = kBlurOut + 2 * reach;
  // head: window texels 0 .. kBlurOut-2, texel j reaches outputs p <= j
#pragma unroll
  for (int j = 0; j < kBlurOut - 1; j++) {
    f2 trg, tba;
    unpack2(tex(j), trg, tba);
#pragma unroll
    for (int p = 0; p <= j; p++) {
      const float c = d[j - p + kBlurPad];
      rg[p] += trg * c;
      ba[p] += tba * c;
    }
  }
  // body: every output is in range (kUnroll > 1: several texels' LDS reads and coefficient loads are in flight at once -- a
  // workgroup with one wave per SIMD has nothing else to hide their latency behind; the sums keep their order)
#pragma unroll kUnroll
  for (int j = kBlurOut - 1; j < nwin - (kBlurOut - 1); j++) {
    f2 trg, tba;
    unpack2(tex(j), trg, tba);
#pragma unroll
    for (int p = 0; p < kBlurOut; p++) {
      const float c = d[j - p + kBlurPad];
      rg[p] += trg * c;
      ba[p] += tba * c;
    }
  }
  // tail: window texel nwin-1-i (i = kBlurOut-2 .. 0) reaches outputs p >= kBlurOut-1-i
#pragma unroll
  for (int i = kBlurOut - 2; i >= 0; i--) {
    const int j = nwin - 1 - i;
    f2 trg, tba;
    unpack2(tex(j), trg, tba);
#pragma unroll
    for (int p = kBlurOut - 1 - i; p < kBlurOut; p++) {
      const float c = d[j - p + kBlurPad];
      rg[p] += trg * c;
      ba[p] += tba * c;
    }
  }
}

template <int kBlurOut>
__global__ __launch_bounds__(256) void k_blur_h(BlurParams P) {
  constexpr int kBlurHW = 64 * kBlurOut;  // one wave = 64 * NOUT consecutive pixels of one row, 4 rows per workgroup
  constexpr int kBlurHLine = kBlurHW + 2 * kMaxBlurReach + kBlurOut;
  __shared__ __attribute__((aligned(16))) uint32_t lines[4][kBlurHLine];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int y = P.y0 + blockIdx.y * 4 + wave;
  const int xs = P.x0 + blockIdx.x * kBlurHW;
  const int reach = P.taps.reach;
  const int wpx = min(kBlurHW, P.x1 - xs);  // pixels this block produces
  const int span = wpx + 2 * reach;
  uint32_t* line = lines[wave];
  if (y < P.y1) {
    const uint32_t* __restrict__ row = P.src + (size_t)y * P.pitch;
    if (xs - reach >= 0 && xs - reach + span + kBlurOut <= P.W) {  // wave-uniform: nothing to clamp
      const uint32_t* __restrict__ p = row + (xs - reach);
      for (int i = lane; i < span + kBlurOut; i += 64) line[i] = p[i];
    } else {
      for (int i = lane; i < span + kBlurOut; i += 64) {
        int x = xs - reach + i;
        x = x < 0 ? 0 : (x > P.W - 1 ? P.W - 1 : x);  // clamp-to-edge (glcontext.nim:214-215)
        line[i] = row[x];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();  // a wave reads back only the line it staged itself: no workgroup barrier
  const int x = xs + lane * kBlurOut;
  if (y >= P.y1 || x >= P.x1) return;
  f2 rg[kBlurOut], ba[kBlurOut];
  const uint32_t* __restrict__ win = line + lane * kBlurOut;
  fir_outputs<kBlurOut>(P.taps.dense, reach, [&](int j) { return win[j]; }, rg, ba);
  uint32_t ov[kBlurOut];
#pragma unroll
  for (int p = 0; p < kBlurOut; p++) ov[p] = pack2(rg[p], ba[p]);
  uint32_t* out = P.dst + (size_t)y * P.pitch + x;
  if (kBlurOut % 4 == 0 && x + kBlurOut - 1 < P.x1 && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
#pragma unroll
    for (int g = 0; g < kBlurOut / 4; g++)
      reinterpret_cast<uint4*>(out)[g] = make_uint4(ov[(4 * g) % kBlurOut], ov[(4 * g + 1) % kBlurOut], ov[(4 * g + 2) % kBlurOut], ov[(4 * g + 3) % kBlurOut]);
  } else if (kBlurOut % 2 == 0 && x + kBlurOut - 1 < P.x1 && ((reinterpret_cast<uintptr_t>(out) & 7) == 0)) {
#pragma unroll
    for (int g = 0; g < kBlurOut / 2; g++) reinterpret_cast<uint2*>(out)[g] = make_uint2(ov[(2 * g) % kBlurOut], ov[(2 * g + 1) % kBlurOut]);
  } else {
#pragma unroll
    for (int p = 0; p < kBlurOut; p++) if (x + p < P.x1) out[p] = ov[p];
  }
}

// vertical pass: one workgroup = 64 columns x 32 rows (taller tiles cut the halo re-read but cost LDS occupancy: measured
// slower); a lane owns a column, each wave produces 8 consecutive rows.  With fuse_draw >= 0 the consuming mode-17 quad
// is blended in place; tiles inside the quad's saturated core (DrawRec::ix0..iy1) skip the coverage evaluation.
constexpr int kBlurVW = 64;
template <int kBlurOut, int kVWaves>
__global__ __launch_bounds__(64 * kVWaves) void k_blur_v(BlurParams P, const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts) {
  constexpr int kBlurVH = kVWaves * kBlurOut;  // rows per workgroup: each wave produces kBlurOut of them
  extern __shared__ uint32_t tile[];  // (kBlurVH + 2*reach) rows x 64 columns
  // XCD-aware tile order: workgroup b runs on XCD b % 8.  Tiles are sequenced band by band (a band = 8 tile columns,
  // walked row by row) and every XCD takes one contiguous eighth of that sequence, so the 2*reach halo rows a tile
  // shares with the tiles above and below it are still in THAT XCD's L2 (row-major order put vertical neighbours on
  // different XCDs: every halo row was fetched from HBM twice) and all XCDs get the same number of tiles.
  const int ntx = (P.x1 - P.x0 + kBlurVW - 1) / kBlurVW, nty = (P.y1 - P.y0 + kBlurVH - 1) / kBlurVH;
  const int total = ntx * nty, per = (total + 7) >> 3;
  const int q = blockIdx.x >> 3;
  const int item = (blockIdx.x & 7) * per + q;
  if (q >= per || item >= total) return;
  const int full = ntx >> 3, in_full = full * 8 * nty;
  int band, rem, bw;
  if (item < in_full) { band = item / (8 * nty); rem = item - band * 8 * nty; bw = 8; }
  else { band = full; rem = item - in_full; bw = ntx - full * 8; }
  const int tyi = rem / bw, txi = band * 8 + rem - tyi * bw;
  const int xs = P.x0 + txi * kBlurVW;
  const int ys = P.y0 + tyi * kBlurVH;
  const int reach = P.taps.reach;
  const int rows = kBlurVH + 2 * reach;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = xs + lane;
  const int xc = x > P.W - 1 ? P.W - 1 : x;
  if (ys - reach >= 0 && ys - reach + rows <= P.H) {  // workgroup-uniform: no row to clamp, the row pointer just advances
    const uint32_t* __restrict__ p = P.src + (size_t)(ys - reach + wave) * P.pitch + xc;
    const size_t step = (size_t)kVWaves * P.pitch;
#pragma unroll 6
    for (int rr = wave; rr < rows; rr += kVWaves) tile[rr * kBlurVW + lane] = p[(size_t)((rr - wave) / kVWaves) * step];
  } else {
    for (int rr = wave; rr < rows; rr += kVWaves) {
      int y = ys - reach + rr;
      y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);  // clamp-to-edge
      tile[rr * kBlurVW + lane] = P.src[(size_t)y * P.pitch + xc];
    }
  }
  __syncthreads();
  const int ry = wave * kBlurOut;  // first of this wave's output rows, relative to ys
  const int y = ys + ry;
  if (x >= P.x1 || y >= P.y1) return;
  f2 rg[kBlurOut], ba[kBlurOut];
  const uint32_t* __restrict__ col = tile + ry * kBlurVW + lane;  // window row j is tile row ry + j
  fir_outputs<kBlurOut>(P.taps.dense, reach, [&](int j) { return col[j * kBlurVW]; }, rg, ba);
  if (P.fuse_draw < 0) {
#pragma unroll
    for (int p = 0; p < kBlurOut; p++)
      if (y + p < P.y1) P.dst[(size_t)(y + p) * P.pitch + x] = pack2(rg[p], ba[p]);
    return;
  }
  // atlas.frag:381-388 on the blurred texel just produced, blended over the live surface (first draw of the phase)
  const DrawRec r = load_rec(draws + P.fuse_draw);
  const bool core = xs >= r.ix0 && xs + kBlurVW <= r.ix1 && ys >= r.iy0 && ys + kBlurVH <= r.iy1;  // coverage alpha == 1
  const float k = 1.0f / 255.0f;
#pragma unroll
  for (int p = 0; p < kBlurOut; p++) {
    if (y + p >= P.y1) break;
    const size_t pix = (size_t)(y + p) * P.pitch + x;
    float alpha = 1.0f;
    if (!core) {  // workgroup-uniform
      const Frag f = make_frag(r, exts, x, y + p);
      if (!f.covered) continue;
      const float lx = (f.u - 0.5f) * 2.0f * r.p0, ly = (f.v - 0.5f) * 2.0f * r.p1;
      const float dist = shape_dist((r.op_mode & F_ELLIP) != 0u, lx, -ly, r.p2, r.p3, r.r[0], r.r[1], r.r[2], r.r[3]);
      alpha = 1.0f - clamp01(r.aa * dist + 0.5f);
    }
    if (__all(__builtin_rintf(ba[p].y) == 255.0f && alpha == 1.0f)) {  // opaque backdrop under full coverage: the blend is a
      P.dst[pix] = pack2(rg[p], ba[p]);                                 // replacement (bit-identical: 1 - sa is 0 to 1e-7 and
      continue;                                                         // every term an integer <= 255)
    }
    const F4 b = {__builtin_rintf(rg[p].x), __builtin_rintf(rg[p].y), __builtin_rintf(ba[p].x), __builtin_rintf(ba[p].y)};
    F4 F = unpack255(P.dst[pix]);
    const float sa = b.w * k * alpha, A = 255.0f * sa;
    const f2 brg = {b.x, b.y};
    blend_pre(F, brg * k * A, f2{b.z * k * A, A}, 1.0f - sa);  // = blend(F, b.rgb / 255, sa)
    P.dst[pix] = pack255(F);
  }
}

// ------------------------------------------------------------------ a small region: both passes in one kernel
// A 360 x 240 backdrop (the demo's own blur node) took two launches of 5 + 7 us -- latency, not work: < 1 % of the HBM peak -- and
// a third for the composite behind them.  Here a workgroup produces a 32 x 16 tile of the BLURRED SNAPSHOT: it stages the tile's
// (32 + 2 reach) x (16 + 2 reach) source window in LDS (clamp-to-edge), filters its 16 + 2 reach rows horizontally into LDS --
// rounded to RGBA8 exactly as the horizontal pass stores its intermediate texture (glcontext.nim:1743-1786) -- and filters
// those vertically.  Same per-output sums in the same order as k_blur_h<2> / k_blur_v<2, .> (fir_outputs<2>): the snapshot is
// the two-pass one bit for bit.  It goes to the backdrop surface, out of place (a tile's neighbours still read the live surface
// around it), and the phase's compositor launch samples it for the mode-17 quad like any other draw.
// 1024 threads per workgroup: the tile's 832 horizontal tasks (radius 18) are ONE fir_outputs per thread and its 256 vertical
// tasks one more -- with 256 threads a thread ran 3.25 + 1 of them back to back, each a chain of 38 dependent LDS reads (15 us
// for the 360 x 240 node against 5 + 7 for the two launches).
constexpr int kSmallTW = 32, kSmallTH = 16, kSmallThreads = 1024;
__global__ __launch_bounds__(kSmallThreads) void k_blur_small(BlurParams P) {
  extern __shared__ uint32_t small_lds[];
  const int reach = P.taps.reach;
  const int in_w = kSmallTW + 2 * reach, rows = kSmallTH + 2 * reach;
  uint32_t* in = small_lds;                    // [rows][in_w]
  uint32_t* hres = small_lds + rows * in_w;    // [rows][kSmallTW]
  const int ntx = (P.x1 - P.x0 + kSmallTW - 1) / kSmallTW;
  const int ty = (int)blockIdx.x / ntx, tx = (int)blockIdx.x - ty * ntx;
  const int xs = P.x0 + tx * kSmallTW, ys = P.y0 + ty * kSmallTH;
  // Staging: every thread asks for ALL its texels of the window, then stores them (at most kSmallStage each: (32 + 48) x (16 + 48)
  // texels for the widest eligible filter).  Written as a row loop inside a column loop, each thread fetched and stored one texel
  // after the other -- up to seven dependent round trips to memory at the head of a kernel that is little else.
  constexpr int kSmallStage = 5;
  const int n_in = rows * in_w;
  uint32_t texel[kSmallStage];
#pragma unroll
  for (int k = 0; k < kSmallStage; k++) {
    const int i = min((int)threadIdx.x + k * kSmallThreads, n_in - 1);
    const int rr = i / in_w, cc = i - rr * in_w;
    int y = ys - reach + rr, x = xs - reach + cc;
    y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);  // clamp-to-edge (glcontext.nim:214-215)
    x = x < 0 ? 0 : (x > P.W - 1 ? P.W - 1 : x);
    texel[k] = P.src[(size_t)y * P.pitch + x];
  }
#pragma unroll
  for (int k = 0; k < kSmallStage; k++) {
    const int i = (int)threadIdx.x + k * kSmallThreads;
    if (i < n_in) in[i] = texel[k];
  }
  __syncthreads();
  // horizontal: a thread produces two consecutive outputs of one row
  for (int t = threadIdx.x; t < rows * (kSmallTW / 2); t += kSmallThreads) {
    const int rr = t / (kSmallTW / 2), c = t - rr * (kSmallTW / 2);
    f2 rg[2], ba[2];
    const uint32_t* __restrict__ win = in + rr * in_w + 2 * c;
    fir_outputs<2, 6>(P.taps.dense, reach, [&](int j) { return win[j]; }, rg, ba);
    hres[rr * kSmallTW + 2 * c] = pack2(rg[0], ba[0]);
    hres[rr * kSmallTW + 2 * c + 1] = pack2(rg[1], ba[1]);
  }
  __syncthreads();
  // vertical: a thread produces two consecutive rows of one column
  for (int t = threadIdx.x; t < kSmallTW * (kSmallTH / 2); t += kSmallThreads) {
    const int pr = t / kSmallTW, c = t - pr * kSmallTW;
    const int x = xs + c, y = ys + 2 * pr;
    if (x >= P.x1 || y >= P.y1) continue;
    f2 rg[2], ba[2];
    const uint32_t* __restrict__ col = hres + (2 * pr) * kSmallTW + c;
    fir_outputs<2, 6>(P.taps.dense, reach, [&](int j) { return col[j * kSmallTW]; }, rg, ba);
    P.dst[(size_t)y * P.pitch + x] = pack2(rg[0], ba[0]);
    if (y + 1 < P.y1) P.dst[(size_t)(y + 1) * P.pitch + x] = pack2(rg[1], ba[1]);
  }
}

// ------------------------------------------------------------------ blur on the matrix pipe (large regions)
// The FIR is the one contraction on the path: 32 consecutive outputs of a line are a banded Toeplitz matrix (32 x (32 + 2 reach))
// times the line's texels.  As packed-FMA code it ran at ~85 % of the VALU issue rate and 28 % of the HBM roofline; on the
// matrix pipe (v_mfma_f32_32x32x16_f16, f32 accumulate) the arithmetic drops under the memory time.
//   * texels need no conversion: a byte b in the low bits of a half IS the subnormal b * 2^-24, and the matrix pipe honours
//     f16 subnormals (tools/microbench/mfma_f16_probe.hip) -- one v_perm_b32 builds two operand halves of one channel;
//   * weights: the taps at scale 2^10 as ONE f16 each (round 5: FDH_MX_LO below; rounds 2 - 4 split them hi + lo, 22 bits, two
//     MFMAs per operand); every product with an 8-bit texel is exact in f32, so out = acc * 2^14;
//   * operands: A[i][k] = Toeplitz weights (i = output inside the block), B[k][j] = texels (j = lane & 31: a column for the
//     vertical pass, a row for the horizontal one; k = 16 texels along the filter direction per MFMA, lane group g = lane >> 5
//     holds k = 8 g .. 8 g + 7); D[i][j]: lane (j, g), register r <-> i = (r & 3) + 8 (r >> 2) + 4 g.
//   A wave walks T blocks of 32 outputs along the filter direction; a block reads NK k-steps (16 NK >= 32 + 2 reach) and
//   shares all but two of them with the block before it.
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
union H8Bits { h8 v; uint32_t u[4]; };

// the two operand halves (texels 2q, 2q + 1 of the lane's eight) of channel c: bytes (R[2q].c, 0, R[2q+1].c, 0)
template <int c> __device__ __forceinline__ h8 mx_frag(const uint32_t (&R)[8]) {
  constexpr uint32_t sel = 0x0c000c00u | (uint32_t)c | ((uint32_t)(4 + c) << 16);
  H8Bits o;
#pragma unroll
  for (int q = 0; q < 4; q++) o.u[q] = __builtin_amdgcn_perm(R[2 * q + 1], R[2 * q], sel);
  return o.v;
}
// ring slots of a wave: the NK k-steps of the block being multiplied + the two the next block adds; the vertical pass keeps
// two more that are idle for the length of an iteration -- its fused composite moves alphas through them
// Round 5: ONE f16 fragment per k-step.  The weights were hi + lo halves (22 bits, two MFMAs per operand and k-step) so that the
// matrix-pipe passes reproduced the float FIR to the bit; the matrix pipe is what bounds k_blur_fx (26 M of its 70 M SIMD-cycles
// with nothing overlapping them), and the low halves are half of that.  The weights are now the taps rounded to f16 at scale 2^10
// with the rounding error carried from tap to tap (fdh_context.cpp, build_mx_weights): 11 bits each, their sum kept, the error an
// alternating pattern that smooth content cancels.  Against the exact taps 0.07 - 0.35 % of a UI-like frame's texels move by one LSB
// (the suite's oracle bars -- at most 1 LSB, at most 0.5 % of the pixels -- are unchanged and met).
// -DFDH_MX_LO=1 (make variant) restores the second MFMA per operand, for fragments built with both halves (the switch lives in
// fdh_types.h: the host side builds the fragments, the device side multiplies them, and the two must agree).
#ifndef FDH_MX_H_EXTRA
#define FDH_MX_H_EXTRA 0
#endif
#ifndef FDH_MX_V_EXTRA
#define FDH_MX_V_EXTRA 0  // (2: rounds 1 - 3 kept two idle slots for the fused composite's alphas -- 18 KB per wave, eight waves per CU)
#endif
#ifndef FDH_MX_WG_H
#define FDH_MX_WG_H 1     // waves per WORKGROUP of k_blur_mx, horizontal / vertical pass (mx_wg).  Each wave works alone (own ring, no barrier); what
#endif                    // a workgroup decides is WHERE they run: its waves are neighbours along the filter direction on ONE CU, so the halo they
#ifndef FDH_MX_WG_V       // share (38 of every 166 texels at radius 18) comes out of that CU's vector cache.  Measured with four (4K, same box, both
#define FDH_MX_WG_V 1     // builds): the horizontal pass 19.7 -> 17.4 us -- and the vertical one behind it 19.5 -> 20.7 (20.2 with four vertical
#endif                    // neighbours per workgroup); the sum 39.2 -> 38.0, `value` the same within the boxes' noise.  Left at one.
#ifndef FDH_MX_WAVES
#define FDH_MX_WAVES 2    // waves per SIMD the passes are compiled for.  (3 -- eleven 14-KB rings fit a CU, 2720 waves of three blocks in one
                          // round -- measured 21.3 / 22.5 us against 19.9 / 19.6 for the two passes at 4K: more waves per SIMD do not help, the
                          // passes are paced by the memory system, profiles/r04_blur_notes.txt)
#endif
constexpr int mx_ring_slots(int nk, bool vertical) { return nk + 2 + (vertical ? FDH_MX_V_EXTRA : FDH_MX_H_EXTRA); }
// waves per workgroup of an instantiation: FDH_MX_WG_H / _V where two such workgroups' rings fit a CU's LDS (eight waves per CU,
// two per SIMD, as with one-wave workgroups); one for the wide filters, whose rings would leave room for a single workgroup
constexpr int mx_wg(int nk, bool vertical) {
  const int want = vertical ? FDH_MX_WG_V : FDH_MX_WG_H;
  return 2 * want * mx_ring_slots(nk, vertical) * 2 <= 160 ? want : 1;
}
constexpr float kMxScale = 16384.0f;  // 2^24 (subnormal texels) / 2^10 (weight scale)
constexpr int kMxSlot = 512;          // dwords of one k-step in LDS: 16 texels along the filter x 32 lines
// LDS-DMA: 16 (or 4) bytes per lane from `src` to LDS byte address `lds` + 16 (4) * lane.  Written as inline assembly on
// purpose: after the builtin form hipcc drains vmcnt to 0 before the next LDS read, which also waits for the stores just
// issued; the kernel below places its own counted waits.  (M0 = LDS address; one wait state between s_mov m0 and its use.)
// M0 is a reserved register for hipcc (a clobber on it is ignored, -Winline-asm): the block saves and restores it.
__device__ __forceinline__ void lds_dma16(const void* src, uint32_t lds) {
  uint32_t m0_saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved) : "v"(src), "s"(lds) : "memory");
}
__device__ __forceinline__ void lds_dma4(const void* src, uint32_t lds) {
  uint32_t m0_saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved) : "v"(src), "s"(lds) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One wave walks `T` blocks of 32 outputs along the filter direction over 32 lines (columns for the vertical pass, rows
// for the horizontal one).  Texels reach LDS by LDS-DMA (global_load_lds, no registers, full 16-byte pieces of whole
// 64/128-byte runs) into a ring of NK + 4 k-step slots: while block b is multiplied, the two k-steps block b + 2 adds
// are in flight and the stores of block b - 1 drain.  vmcnt is one in-order counter for loads and stores, so the wait at
// the top of an iteration allows exactly the batch issued last and nothing is issued between that batch and the wait.
// Both passes end with lane = x, accumulator register = y inside a 32 x 32 pixel block (the horizontal pass multiplies
// texels x weights, the vertical one weights x texels), so every store is a 128-byte run.
//
// Two specialisations (horizontal, vertical).  Round 1 shipped ONE merged body with a run-time direction flag because, with
// several contexts in flight, frames came out with wrong texels when the passes were two kernels.  The wrong texels were
// never produced here: they were compositor pixels misread by packed-FP32 instructions while these kernels' v_mfma shared
// the SIMD (tools/microbench/pk_vs_mfma.hip); the merged body merely ran slowly enough to hide it.
// The library is now built without packed-FP32 instructions (csrc/Makefile, tools/lint_isa.py).
#ifndef FDH_MX_CHECK
#define FDH_MX_CHECK 0  // experiment builds only: every texel a wave reads from its LDS ring is compared with global memory
#endif
#if FDH_MX_CHECK
__device__ unsigned int g_mx_bad_n;
__device__ unsigned int g_mx_bad[4096 * 8];
#endif
template <int NK, bool kV>
__global__ __launch_bounds__(64 * mx_wg(NK, kV), FDH_MX_WAVES) void k_blur_mx(BlurParams P, const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts, int T) {
  constexpr int R = mx_ring_slots(NK, kV);
  extern __shared__ __attribute__((aligned(16))) uint32_t ring_wg[];  // R slots per wave
  constexpr int kWG = mx_wg(NK, kV);
  const int wave_in_wg = kWG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  uint32_t* const ring = ring_wg + wave_in_wg * (R * kMxSlot);
  // blocks sit at absolute multiples of 32 along the filter direction: a pixel's sum is then grouped into MFMAs the same
  // way whatever region or stripe it is rendered in (stripes of a frame must reproduce the full frame bit for bit)
  const int a_lo = kV ? P.y0 : P.x0, a_hi = kV ? P.y1 : P.x1, a0 = a_lo & ~31;
  const int l0 = kV ? (P.x0 & ~31) : P.y0, l_hi = kV ? P.x1 : P.y1;
  const int n_along = (a_hi - a0 + 32 * T - 1) / (32 * T), n_lines = (l_hi - l0 + 31) >> 5;
  const int total = n_along * n_lines, per = (total + 7) >> 3, q = (int)(blockIdx.x >> 3) * kWG + wave_in_wg, item = (blockIdx.x & 7) * per + q;
  if (q >= per || item >= total) return;  // every XCD takes a contiguous eighth of the sequence: neighbours along the filter share an L2
  // Sequence: horizontal pass, along the rows (neighbours in x run together and share their halo); vertical pass, bands of
  // 16 strips (2 KB of every row) walked segment row by segment row -- the waves in flight on an XCD then read each row
  // in 2-KB runs, not in 128-byte pieces 15 KB apart (one DRAM page per piece), and vertical neighbours still share an L2.
  int sl, sa;
  if (kV) {
    constexpr int kBand = 16;
    const int per_band = kBand * n_along, band = item / per_band, rem = item - band * per_band;
    const int bw = min(kBand, n_lines - band * kBand);
    if (kWG > 1) {  // strip by strip inside a band: the waves of a workgroup are vertical neighbours and share halo ROWS
      const int st = rem / n_along;
      sa = rem - st * n_along;
      sl = band * kBand + st;
    } else {
      sa = rem / bw;
      sl = band * kBand + rem - sa * bw;
    }
  } else {
    sl = item / n_along;
    sa = item - sl * n_along;
  }
  const int lane = threadIdx.x & 63, g = lane >> 5, j = lane & 31;
  const int reach = P.taps.reach;
  const int as = a0 + 32 * T * sa, lb = l0 + 32 * sl;
  const int n_blocks = min(T, (a_hi - as + 31) >> 5);
  const int w0 = as - reach, w0a = kV ? w0 : (w0 & ~3);  // horizontal: window start moved back to a 16-byte boundary (mx_delta)
  // Toeplitz weight fragments (fdh_context.cpp, build_mx_weights): fragment m of the lane that carries output j of a block
  // holds, for texel 16 m + 8 g + t of the block's window, the tap that texel meets at that output -- the same for every
  // wave of the launch, so it is built once on the host and fetched here as 2 NK coalesced 16-byte loads
#if FDH_TIMING  // per-wave phase times in shader cycles (tools/mx_wave_times.py)
  const unsigned long long T0 = FDH_NOW();
  unsigned long long T_wait = 0, T_st = 0, T_mma = 0, T_epi = 0, T_iss = 0;
#endif
  const uint32_t ring_lds = (uint32_t)reinterpret_cast<uintptr_t>(ring);

  // LDS-DMA of k-step s (texels w0a + 16 s .. + 15 along the filter, 32 lines) into slot s % R; returns the instructions issued
  // Per-lane source pointers of the two DMA instructions of a k-step, without the k-step's own (uniform) offset:
  // computed once, a k-step adds a scalar.  V: (column piece, row 8 h + r) -- the row is clamped per k-step instead.
  // H: row 16 h + r of the block's 32 rows, 16-byte piece c ^ swizzle.
  const uint32_t* hsrc[2];
  if (!kV) {
    const int r = lane >> 2, c = lane & 3;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int row = 16 * h + r, y = min(lb + row, P.y1 - 1);  // rows past the region read a valid row and store nothing
      hsrc[h] = P.src + (size_t)y * P.pitch + 4 * (c ^ ((row >> 2) & 3));
    }
  }
  const int vxch = kV ? min(lb + 4 * (lane & 7), P.W - 4) : 0;  // (columns past the frame are never stored)
  int issue_slot = 0;  // ring slot of the next k-step to be issued
  auto issue = [&](int s) -> int {
    const uint32_t slot = ring_lds + (uint32_t)issue_slot * (kMxSlot * 4u);  // LDS byte address
    issue_slot = issue_slot + 1 == R ? 0 : issue_slot + 1;
    if (kV) {  // slot image [16 rows][32 px]; an instruction = 8 rows x 128 bytes
      const int r = lane >> 3;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        int y = w0a + 16 * s + 8 * h + r;
        y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);  // clamp-to-edge (glcontext.nim:214-215)
        lds_dma16(P.src + (size_t)y * P.pitch + vxch, slot + h * 1024u);
      }
      return 2;
    }
    // slot image [32 rows][16 px], the four 16-byte pieces of a row XOR-swizzled by (row >> 2) & 3 so that the
    // ds_read_b128 of sixteen lanes (rows) hits sixteen different bank groups
    const int xb = w0a + 16 * s;
    if (xb >= 0 && xb + 16 <= P.W) {  // wave-uniform: an instruction = 16 rows x 64 bytes
      lds_dma16(hsrc[0] + xb, slot);
      lds_dma16(hsrc[1] + xb, slot + 1024u);
      return 2;
    }
    const int rr = lane >> 4, pp = lane & 15;  // the k-step crosses a frame edge: one texel per lane, clamped
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int row = 4 * i + rr, y = min(lb + row, P.y1 - 1);
      int gx = xb + 4 * ((pp >> 2) ^ ((row >> 2) & 3)) + (pp & 3);
      gx = gx < 0 ? 0 : (gx > P.W - 1 ? P.W - 1 : gx);
      lds_dma4(P.src + (size_t)y * P.pitch + gx, slot + i * 256u);
    }
    return 8;
  };
  auto wait_for_all_but = [&](int n) {  // (allowing fewer than were issued last is always safe)
    if (n >= 16) wait_vm<16>(); else if (n >= 10) wait_vm<10>(); else if (n >= 4) wait_vm<4>(); else wait_vm<0>();
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int s = 0; s < NK; s++) issue(s);
  int last_batch = 0;
  if (n_blocks > 1) { last_batch = issue(NK); last_batch += issue(NK + 1); }
  // The weight fragments are fetched behind the first k-steps and waited for HERE, with a wait the compiler can see.  (Left
  // to itself it keeps `s_waitcnt vmcnt(3 .. 0)` for them in front of the MFMAs of every block -- it cannot know they landed
  // long ago -- and since vmcnt counts every memory operation of the wave, those waits drained the prefetch of the next
  // block and the stores of the last one in every iteration.)
  h8 whi[NK];
#if FDH_MX_LO
  h8 wlo[NK];
#endif
#pragma unroll
  for (int m = 0; m < NK; m++) {
    H8Bits a;
    const uint4 va = P.mx_w[(2 * m) * 64 + lane];
    a.u[0] = va.x; a.u[1] = va.y; a.u[2] = va.z; a.u[3] = va.w;
    whi[m] = a.v;
#if FDH_MX_LO
    H8Bits b;
    const uint4 vb = P.mx_w[(2 * m + 1) * 64 + lane];
    b.u[0] = vb.x; b.u[1] = vb.y; b.u[2] = vb.z; b.u[3] = vb.w;
    wlo[m] = b.v;
#endif
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) (expcnt, lgkmcnt untouched)
  last_batch = 0;                      // (everything issued so far has landed)
  // the consuming quad's saturated core; the rest of its record is fetched by the few blocks on its border (32 fewer
  // SGPRs held through the walk: the vertical pass was spilling them into VGPR lanes)
  int core_x0 = 0, core_y0 = 0, core_x1 = 0, core_y1 = 0;
  if (kV && P.fuse_draw >= 0) { const DrawRec* q = draws + P.fuse_draw; core_x0 = q->ix0; core_y0 = q->iy0; core_x1 = q->ix1; core_y1 = q->iy1; }
#if FDH_TIMING
  const unsigned long long T_pro = FDH_NOW() - T0 + (__builtin_amdgcn_readfirstlane(whi[0][0] != whi[0][1]) & 0u);
#endif
  uint32_t pend[16];
  uint32_t pmask = 0;
  int pbx = 0, pby = 0;
  auto store_pending = [&]() {
    // a row's address = uniform row pointer (scalar arithmetic) + one 32-bit per-lane byte offset: no vector address
    // arithmetic per store; a block with every pixel live (wave-uniform test) stores without execution masks
    const uint32_t lane_off = ((uint32_t)(4 * g) * (uint32_t)P.pitch + (uint32_t)j) * 4u;
    char* base = reinterpret_cast<char*>(P.dst + (size_t)pby * P.pitch + pbx);
    if (__all(pmask == 0xffffu)) {
#pragma unroll
      for (int rr = 0; rr < 16; rr++)
        *reinterpret_cast<uint32_t*>(base + (size_t)((rr & 3) + 8 * (rr >> 2)) * P.pitch * 4u + lane_off) = pend[rr];
    } else if (__any(pmask != 0u)) {
#pragma unroll
      for (int rr = 0; rr < 16; rr++)
        if ((pmask >> rr) & 1u) *reinterpret_cast<uint32_t*>(base + (size_t)((rr & 3) + 8 * (rr >> 2)) * P.pitch * 4u + lane_off) = pend[rr];
    }
    pmask = 0;
  };
  int slot0 = 0;  // (2 b) % R
#pragma unroll 1
  for (int b = 0; b < n_blocks; b++) {
#if FDH_TIMING
    const unsigned long long Ta = FDH_NOW();
#endif
    wait_for_all_but(last_batch);  // the k-steps of block b have landed; the stores issued an iteration ago have drained
#if FDH_TIMING
    const unsigned long long Tb = FDH_NOW();
#endif
    store_pending();               // block b - 1
#if FDH_TIMING
    const unsigned long long Tc = FDH_NOW();
#endif
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[c][e] = 0.0f;
#pragma unroll
    for (int m = 0; m < NK; m++) {
      const uint32_t* slot = ring + (slot0 + m >= R ? slot0 + m - R : slot0 + m) * kMxSlot;
      uint32_t t8[8];
      if (kV) {  // (rows in the order of an accumulator tile's registers: mx_krow)
#pragma unroll
        for (int t = 0; t < 8; t++) t8[t] = slot[((t & 3) + 8 * (t >> 2) + 4 * g) * 32 + j];
      } else {
        const uint4* row4 = reinterpret_cast<const uint4*>(slot + j * 16);
        const int sw = (j >> 2) & 3;
        const uint4 lo4 = row4[(2 * g) ^ sw], hi4 = row4[(2 * g + 1) ^ sw];
        t8[0] = lo4.x; t8[1] = lo4.y; t8[2] = lo4.z; t8[3] = lo4.w; t8[4] = hi4.x; t8[5] = hi4.y; t8[6] = hi4.z; t8[7] = hi4.w;
      }
#if FDH_MX_CHECK
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const int pos = w0a + 16 * (2 * b + m) + (kV ? (t & 3) + 8 * (t >> 2) + 4 * g : 8 * g + t);
        uint32_t want;
        if (kV) {
          const int yy = pos < 0 ? 0 : (pos > P.H - 1 ? P.H - 1 : pos);
          want = P.src[(size_t)yy * P.pitch + min(lb + 4 * (j >> 2), P.W - 4) + (j & 3)];
        } else {
          const int xx = pos < 0 ? 0 : (pos > P.W - 1 ? P.W - 1 : pos);
          want = P.src[(size_t)min(lb + j, P.y1 - 1) * P.pitch + xx];
        }
        if (want != t8[t]) {
          const unsigned int i = atomicAdd(&g_mx_bad_n, 1u);
          if (i < 4096u) {
            unsigned int* o = g_mx_bad + 8 * i;
            o[0] = kV ? 1u : 0u; o[1] = blockIdx.x; o[2] = (unsigned)b | ((unsigned)n_blocks << 16); o[3] = (unsigned)m; o[4] = (unsigned)lane; o[5] = (unsigned)t; o[6] = t8[t]; o[7] = want;
          }
        }
      }
#endif
      const h8 f0 = mx_frag<0>(t8), f1 = mx_frag<1>(t8), f2_ = mx_frag<2>(t8), f3 = mx_frag<3>(t8);
      if (kV) {  // weights x texels: D[output row][column]
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[m], f0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[m], f1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[m], f2_, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi[m], f3, acc[3], 0, 0, 0);
#if FDH_MX_LO
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[m], f0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[m], f1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[m], f2_, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo[m], f3, acc[3], 0, 0, 0);
#endif
      } else {   // texels x weights: D[row][output column]
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0, whi[m], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1, whi[m], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2_, whi[m], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f3, whi[m], acc[3], 0, 0, 0);
#if FDH_MX_LO
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0, wlo[m], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1, wlo[m], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2_, wlo[m], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f3, wlo[m], acc[3], 0, 0, 0);
#endif
      }
    }
#if FDH_TIMING
    const unsigned long long Td = FDH_NOW() + (__builtin_amdgcn_readfirstlane(__float_as_uint(acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0])) & 0u);
#endif
    // the block's 32 x 32 pixels: lane = x, register rr = row (rr & 3) + 8 (rr >> 2) + 4 g
    const int bx = kV ? lb : as + 32 * b, by = kV ? as + 32 * b : lb;
    const int x = bx + j;
    pbx = bx; pby = by;
    const bool x_ok = x >= P.x0 && x < P.x1;
#pragma unroll
    for (int rr = 0; rr < 16; rr++)  // (scalar multiplies: the four channels sit in four accumulator tiles, a packed multiply would need two moves first)
      pend[rr] = pack2(f2{acc[0][rr] * kMxScale, acc[1][rr] * kMxScale}, f2{acc[2][rr] * kMxScale, acc[3][rr] * kMxScale});
    if (bx >= P.x0 && bx + 32 <= P.x1 && by >= P.y0 && by + 32 <= P.y1) {  // wave-uniform: the whole block lies in the region
      pmask = 0xffffu;
    } else {
#pragma unroll
      for (int rr = 0; rr < 16; rr++) {
        const int y = by + (rr & 3) + 8 * (rr >> 2) + 4 * g;
        if (x_ok && y >= P.y0 && y < P.y1) pmask |= 1u << rr;
      }
    }
    if (kV && P.fuse_draw >= 0) {
      // atlas.frag:381-388 on the blurred texel just produced, blended over the live surface (first draw of the phase).
      // `alpha16[rr]`: the quad's coverage at the lane's pixel of row rr; pixels with an opaque backdrop under full
      // coverage are plain replacements (the blend is exact there) and need nothing more.
      const bool core = bx >= core_x0 && bx + 32 <= core_x1 && by >= core_y0 && by + 32 <= core_y1;  // coverage alpha == 1 (wave-uniform)
      // the common block -- inside the core, every blurred texel opaque -- is done: one AND chain and one ballot decide it
      bool replace_all = false;
      if (core) {
        uint32_t conj = pend[0];
#pragma unroll
        for (int rr = 1; rr < 16; rr++) conj &= pend[rr];
        replace_all = __all((conj >> 24) == 255u);
      }
      if (!replace_all) {
      uint32_t blend_mask = 0;
      // Two ring slots nothing is in flight into until the end of this iteration.  With FDH_MX_V_EXTRA = 2: the two idle slots in
      // front of block b's.  Without them (round 4): block b's OWN first two k-steps -- its products are done, block b + 1 starts two
      // k-steps further on, and the k-steps of block b + 2 are only issued into them behind this epilogue.
      uint32_t* sc0 = ring + (FDH_MX_V_EXTRA >= 2 ? (slot0 >= 2 ? slot0 - 2 : slot0 - 2 + R) : slot0) * kMxSlot;
      uint32_t* sc1 = ring + (FDH_MX_V_EXTRA >= 2 ? (slot0 >= 1 ? slot0 - 1 : slot0 - 1 + R) : (slot0 + 1 >= R ? slot0 + 1 - R : slot0 + 1)) * kMxSlot;
      if (core) {
#pragma unroll
        for (int rr = 0; rr < 16; rr++) if (((pmask >> rr) & 1u) && (pend[rr] >> 24) != 255u) blend_mask |= 1u << rr;
      } else {
        // A block on the quad's border (a few hundred of 8100 at 4K).  Its rows and columns inside the core have
        // alpha == 1; the others are evaluated densely, one row (lane = x) or one column (lane = y) per lane group and
        // step, and reach the accumulator layout (lane = x, register = row) through LDS.  (Evaluating in the accumulator
        // layout costs 16 sparse steps per block: the two border strips then ran 2.5x longer than every other wave.)
        const uint32_t valid = pmask;
        pmask = 0;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) (rr < 8 ? sc0 : sc1)[(rr & 7) * 64 + lane] = __float_as_uint(((valid >> rr) & 1u) ? 1.0f : -1.0f);
        __builtin_amdgcn_wave_barrier();
        const DrawRec r = load_rec(draws + P.fuse_draw);
        // (rows / columns of the block inside the region, [ry0, ry1) x [rx0, rx1), less the core's [ra, rb) x [ca, cb): see k_blur_fx)
        const int ry0 = min(max(P.y0 - by, 0), 32), ry1 = max(ry0, min(P.y1 - by, 32)), rx0 = min(max(P.x0 - bx, 0), 32), rx1 = max(rx0, min(P.x1 - bx, 32));
        const int ra = min(max(core_y0 - by, ry0), ry1), rb = max(ra, min(max(core_y1 - by, ry0), ry1));
        const int ca = min(max(core_x0 - bx, rx0), rx1), cb = max(ca, min(max(core_x1 - bx, rx0), rx1));
        const int nra = ra - ry0, nca = ca - rx0, nr = nra + ry1 - rb, nc = nca + rx1 - cb;
#pragma unroll 1
        for (int u0 = 0; u0 < nr + nc; u0 += 2) {
          const int u = u0 + g;
          int dx, dy;
          if (u < nr) { dy = u < nra ? ry0 + u : rb + (u - nra); dx = j; }
          else { const int v = u - nr; dx = v < nca ? rx0 + v : cb + (v - nca); dy = j; }
          if (u >= nr + nc) continue;
          const int ex = bx + dx, ey = by + dy;
          float alpha = -1.0f;
          if (ex >= P.x0 && ex < P.x1 && ey >= P.y0 && ey < P.y1) {
            const Frag f = make_frag(r, exts, ex, ey);
            if (f.covered) {
              const float lx = (f.u - 0.5f) * 2.0f * r.p0, ly = (f.v - 0.5f) * 2.0f * r.p1;
              const float dist = shape_dist((r.op_mode & F_ELLIP) != 0u, lx, -ly, r.p2, r.p3, r.r[0], r.r[1], r.r[2], r.r[3]);
              alpha = 1.0f - clamp01(r.aa * dist + 0.5f);
            }
          }
          const int er = (dy & 3) + 4 * (dy >> 3), el = dx + 32 * ((dy >> 2) & 1);  // accumulator register and lane of (dx, dy)
          (er < 8 ? sc0 : sc1)[(er & 7) * 64 + el] = __float_as_uint(alpha);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
          const float al = __uint_as_float((rr < 8 ? sc0 : sc1)[(rr & 7) * 64 + lane]);
          if (al >= 0.0f) {
            pmask |= 1u << rr;
            if (al != 1.0f || (pend[rr] >> 24) != 255u) blend_mask |= 1u << rr;
          }
        }
      }
      if (__any(blend_mask != 0u)) {  // (never on an opaque surface inside the quad)
        uint32_t dstv[16];
#pragma unroll
        for (int rr = 0; rr < 16; rr++)  // all the loads first: sixteen dependent round trips otherwise
          dstv[rr] = ((blend_mask >> rr) & 1u) ? P.dst[(size_t)(by + (rr & 3) + 8 * (rr >> 2) + 4 * g) * P.pitch + x] : 0u;
        const float k = 1.0f / 255.0f;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
          if (!((blend_mask >> rr) & 1u)) continue;
          const float alpha = core ? 1.0f : __uint_as_float((rr < 8 ? sc0 : sc1)[(rr & 7) * 64 + lane]);
          const F4 bl = unpack255(pend[rr]);
          F4 Fd = unpack255(dstv[rr]);
          const float sa = bl.w * k * alpha, A = 255.0f * sa;
          // = blend(F, b.rgb / 255, sa) (blend_pre's arithmetic, FMA for FMA), kept scalar: in an earlier arrangement of this
          // block (coverage evaluated in a rolled 16-step loop) the packed f2 form gave red = 0 in lanes 48-63 of a few dozen
          // wavefronts per 4K frame while the scalar form was exact; the cause was never found and the current arrangement
          // is exact either way.  The block runs for a few hundred of 8100 blocks: its cost does not matter.
          const float ia = 1.0f - sa;
          Fd.x = __builtin_rintf(__builtin_fmaf(Fd.x, ia, bl.x * k * A));
          Fd.y = __builtin_rintf(__builtin_fmaf(Fd.y, ia, bl.y * k * A));
          Fd.z = __builtin_rintf(__builtin_fmaf(Fd.z, ia, bl.z * k * A));
          Fd.w = __builtin_rintf(__builtin_fmaf(Fd.w, ia, A));
          pend[rr] = pack255(Fd);
        }
        // (a wait the compiler can see: otherwise it assumes one of these masked loads may still be in flight when their
        // registers are reused at the top of the walk and puts a vmcnt(0) there -- in front of EVERY block's stores)
        __builtin_amdgcn_s_waitcnt(0x0F70);
      }
      __builtin_amdgcn_wave_barrier();
      }
    }
#if FDH_TIMING
    const unsigned long long Te = FDH_NOW() + (__builtin_amdgcn_readfirstlane(pend[0] + pend[15]) & 0u);
#endif
    // the ring slots block b - 1 gave up take the two k-steps block b + 2 adds
    __builtin_amdgcn_sched_barrier(0);
    last_batch = 0;
    if (b + 2 < n_blocks) { last_batch = issue(2 * b + NK + 2); last_batch += issue(2 * b + NK + 3); }
    slot0 = slot0 + 2 >= R ? slot0 + 2 - R : slot0 + 2;
#if FDH_TIMING
    const unsigned long long Tf = FDH_NOW();
    T_wait += Tb - Ta; T_st += Tc - Tb; T_mma += Td - Tc; T_epi += Te - Td; T_iss += Tf - Te;
#endif
  }
  store_pending();
#if FDH_TIMING
  if (lane == 0 && blockIdx.x < 32768) {  // rows 0.. : horizontal pass, rows 32768.. : vertical pass (the compositor's rows are overwritten)
    unsigned long long* row = g_wave_times + 16 * ((size_t)blockIdx.x + (kV ? 32768 : 0));
    row[0] = FDH_NOW() - T0; row[1] = T_pro; row[2] = T_wait; row[3] = T_st; row[4] = T_mma; row[5] = T_epi; row[6] = kV ? 3 : 2; row[7] = T_iss; row[8] = n_blocks; row[9] = T0;
  }
#endif
}

// ------------------------------------------------------------------ both passes of a full-frame node in ONE kernel
// A backdrop blur that covers the whole frame moved every texel four times: H read + H write (the reference's RGBA8
// intermediate texture, glcontext.nim:1743-1786), V read + V write.  Here the intermediate never leaves the wave's REGISTERS.
// One wave owns a strip 32 columns wide and walks DOWN it in blocks of 32 rows: it filters 32 new rows horizontally (the
// k_blur_mx<., false> product: texels x Toeplitz weights) and rounds them to RGBA8 exactly as the H pass stores them; two blocks
// behind, the vertical product (k_blur_mx<., true>: weights x texels) takes them as its operand, and the epilogue -- scale, RGBA8,
// the fused mode-17 composite, 128-byte row stores -- is k_blur_mx's.
//   * The chain (round 5).  The horizontal product leaves a 32 x 32 tile with lane = column, register 8 s + e = row
//     16 s + (e & 3) + 8 (e >> 2) + 4 g: for the vertical product -- B operand: lane = column, element e of lane group g = some row of
//     a 16-row k-step -- that IS two k-steps of operand, if the vertical weights are laid out for that row order (mx_krow; the
//     two-pass vertical kernel reads its rows in the same order, so the sums are grouped alike).  Rounds 3 - 4 wrote the rounded
//     tile to an LDS ring and read it back texel by texel: 16 + 40 LDS operations, 128 VALU for scale + pack and 80 v_perm for
//     the operand halves per block.  Now: one FMA per value (acc * 2^14 + 1.5 * 2^23: round to nearest even, the integer in the
//     low mantissa byte -- what v_cvt_pk_u8_f32 gives for these values, which lie in [0, 255.001]) and one v_perm per PAIR
//     builds the two f16 subnormals; the vertical product's operands cost nothing more.  No H ring: 12 KB of LDS less per wave.
//   * Registers: HB H-blocks (HB = 3 for radius 18: 96 VGPRs) rotate through a stash indexed by block number mod HB -- the three
//     phases are three copies of "round into the stash + vertical MFMAs", chosen by a uniform branch; everything else is one
//     copy.  The vertical weights live in LDS (10 KB, read as 16-byte fragments next to their MFMAs) so that the stash fits
//     beside the accumulators at two waves per SIMD; the horizontal ones stay in registers.
//   * Same sums, same grouping: H blocks sit at absolute multiples of 32 in x with the H pass's k-step alignment, V blocks at
//     absolute multiples of 32 in y with windows starting at y - reach: the result is the two-pass result bit for bit.
//   * H-blocks start at rows congruent to -reach mod 32, so a V window starts on an H-block boundary and the V weight table of
//     the two-pass kernel is used unchanged.
//   * Out of place: src is the surface the phase before left, dst another one (Context::launch_frame alternates the two and
//     starts so that the frame ends in the context's own surface); the fused composite blends over src's texel.
//   * A segment of T blocks re-filters HB - 1 extra H-blocks of halo: T is chosen so that every wave of the launch is
//     resident at once (20 KB of LDS per wave at radius 18: eight waves per CU).
// Bytes: (1 + halo) x 4 A read (the x halo comes out of L2) + 4 A written, against 16 A for the two passes.
constexpr int fx_vblocks(int nkv) { return ((nkv - 1) >> 1) + 1; }               // H-blocks one V block reads
// Waves per workgroup: x-neighbours of one segment row; they share the weight fragments in LDS (and, in L2, their source halo).
// (Eight -- a CU's worth -- with the halves of the workgroup kept one segment of the block apart by an s_barrier per segment, so that one
// wave of a SIMD multiplies while the other rounds, packs and stores: built and measured in round 5, 36.6 - 40.2 us against 33.1: every
// segment then lasts as long as the slowest of EIGHT waves' memory waits.  Not kept.)
constexpr int kFxWaves = 4;
constexpr int fx_wg_ksteps(int nkh) { return nkh + 2 * (kFxWaves - 1); }  // k-steps of the window the workgroup's kFxWaves strips share: 32 columns = 2 k-steps per strip
// 2-KB LDS slots per WORKGROUP: both weight tables + the shared source window of one H-block row, twice (block i is read while block i + 1 lands)
constexpr int fx_slots(int nkh, int nkv) { return nkh + nkv + 2 * fx_wg_ksteps(nkh); }
constexpr float kMxMagic = 12582912.0f;  // 1.5 * 2^23: x + this, as f32, is round-to-nearest-even(x) in the low mantissa bits (|x| < 2^22)
template <int NKH, int NKV>
__global__ __launch_bounds__(64 * kFxWaves, 2) void k_blur_fx(BlurParams P, const uint4* __restrict__ w_v, const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts, int T) {
  constexpr int HB = fx_vblocks(NKV);      // V block b reads H-blocks b .. b + HB - 1
  constexpr int NKW = fx_wg_ksteps(NKH);   // k-steps of the workgroup's shared source window
  extern __shared__ __attribute__((aligned(16))) uint32_t ring[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint4* const hw = reinterpret_cast<const uint4*>(ring);                   // [2 NKH fragments][64 lanes] x 16 bytes
  const uint4* const vw = reinterpret_cast<const uint4*>(ring + NKH * kMxSlot);  // [2 NKV fragments][64 lanes] x 16 bytes
  // The source window.  The workgroup's kFxWaves strips are x-neighbours: their horizontal windows overlap by all but two k-steps,
  // so ONE window of NKW = NKH + 2 (kFxWaves - 1) k-steps serves them all -- wave w reads k-steps 2 w .. 2 w + NKH - 1 of it -- and
  // each wave fetches a quarter of it: 5.5 LDS-DMA pieces per wave and block at radius 18 instead of 10.  (An LDS-DMA piece holds its
  // wave for 100 - 270 cycles until the memory pipeline has taken it, and a CU's pieces go through at about one per 100 cycles:
  // tools/fx_wave_times.py, MI355X_MICROARCH.md "ldsdma-fill" -- at ten pieces per wave and block the FILL was what a block cost.)
  // Two copies: block i is multiplied out of one while block i + 1 lands in the other, one s_barrier per block.
  // Slot image [32 rows][16 px], 16-byte pieces XOR-swizzled (as the H pass).
  uint32_t* const src_base = ring + (NKH + NKV) * kMxSlot;
  const int n_strips = (P.x1 - (P.x0 & ~31) + 31) >> 5, n_sg = (n_strips + kFxWaves - 1) / kFxWaves;
  const int y_first = P.y0 & ~31;
  const int n_seg = (P.y1 - y_first + 32 * T - 1) / (32 * T);
  const int total = n_sg * n_seg, per = (total + 7) >> 3, q = (int)(blockIdx.x >> 3), item = (blockIdx.x & 7) * per + q;
  if (q >= per || item >= total) return;  // (the whole workgroup) every XCD takes a contiguous eighth of the row-major sequence: a band of the frame
  const int seg = item / n_sg, sg = item - seg * n_sg;
  const int strip = kFxWaves * sg + wave;
  const bool active = strip < n_strips;   // a wave past the region's last strip still fetches its share of the window and meets the barriers
  const int lane = threadIdx.x & 63, g = lane >> 5, j = lane & 31;
  const int reach = P.taps.reach;
  const int xb = (P.x0 & ~31) + 32 * strip;  // the strip's columns
  const int ys = y_first + 32 * T * seg;      // first output row of the segment
  const int n_blocks = min(T, (P.y1 - ys + 31) >> 5);
  const int ws = ys - reach;                  // first row of H-block 0
  const int w0a = ((P.x0 & ~31) + 32 * kFxWaves * sg - reach) & ~3;  // the shared window's start, moved back to a 16-byte boundary (mx_delta); strip w's own window starts 32 w further on
  const uint32_t ring_lds = (uint32_t)reinterpret_cast<uintptr_t>(ring);
  const uint32_t src_lds = ring_lds + (uint32_t)((NKH + NKV) * kMxSlot * 4);

  // This wave's share of H-block row i's window: k-steps wave, wave + kFxWaves, .. into copy i & 1.  Returns nothing: the wait at the
  // top of the next block allows exactly the stores issued behind it (vmcnt retires in order).
  auto issue_block = [&](int i) __attribute__((always_inline)) {
    const int r = lane >> 2, c = lane & 3;
    const uint32_t* dma_row[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int row = 16 * h + r;
      int y = ws + 32 * i + row;
      y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);  // clamp-to-edge (glcontext.nim:214-215): a clamped row filters to a clamped H row
      dma_row[h] = P.src + (size_t)y * P.pitch + 4 * (c ^ ((row >> 2) & 3));
    }
    const uint32_t copy = src_lds + (uint32_t)((i & 1) * NKW) * (kMxSlot * 4u);
#pragma unroll
    for (int s0 = 0; s0 < NKW; s0 += kFxWaves) {
      const int s = s0 + wave;
      if (s >= NKW) break;
      const uint32_t slot = copy + (uint32_t)s * (kMxSlot * 4u);
      const int xk = w0a + 16 * s;
      if (xk >= 0 && xk + 16 <= P.W) {  // wave-uniform
        lds_dma16(dma_row[0] + xk, slot);
        lds_dma16(dma_row[1] + xk, slot + 1024u);
      } else {  // the k-step crosses a frame edge: one texel per lane, clamped (a rolled loop: this code exists several times)
        const int rr = lane >> 4, pp = lane & 15;
#pragma unroll 1
        for (int e = 0; e < 8; e++) {
          const int row = 4 * e + rr;
          int y = ws + 32 * i + row;
          y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);
          int gx = xk + 4 * ((pp >> 2) ^ ((row >> 2) & 3)) + (pp & 3);
          gx = gx < 0 ? 0 : (gx > P.W - 1 ? P.W - 1 : gx);
          lds_dma4(P.src + (size_t)y * P.pitch + gx, slot + e * 256u);
        }
      }
    }
  };
  auto wait_for_all_but = [&](int n) __attribute__((always_inline)) {
    if (n >= 16) wait_vm<16>(); else if (n >= 3) wait_vm<3>(); else if (n == 2) wait_vm<2>(); else if (n == 1) wait_vm<1>(); else wait_vm<0>();
    __builtin_amdgcn_sched_barrier(0);
  };

#if FDH_TIMING  // per-wave phase times in shader cycles (tools/fx_wave_times.py)
  const unsigned long long T0 = FDH_NOW();
  unsigned long long T_wait = 0, T_h = 0, T_v = 0, T_epi = 0, T_st = 0, T_cv = 0, T_cv_mark = 0, T_dma = 0;
#endif
  issue_block(0);
  // the weight fragments of both products go to LDS as they lie in memory (lane-linear 16-byte pieces: exactly what the DMA
  // writes), every wave of the workgroup fetching its share -- the one point at which the waves meet
  // -- the horizontal table first: the first block's product needs it; the vertical one is first read HB - 1 blocks later, so this
  // wave's pieces of it (`v_pieces`, the youngest in the queue) may still be in flight at the first barrier: the second block's wait
  // covers them
  int v_pieces = 0;
  {
    constexpr int NF = 2 * (NKH + NKV);
#pragma unroll
    for (int f0 = 0; f0 < NF; f0 += kFxWaves) {
      const int f = f0 + wave;
      if (f < NF) {
        lds_dma16((f < 2 * NKH ? P.mx_w + f * 64 : w_v + (f - 2 * NKH) * 64) + lane, ring_lds + (uint32_t)f * 1024u);
        if (f >= 2 * NKH) v_pieces++;
      }
    }
  }
#if FDH_TIMING
  const unsigned long long T_pro = FDH_NOW() - T0;
#endif
  int core_x0 = 0, core_y0 = 0, core_x1 = 0, core_y1 = 0;
  if (P.fuse_draw >= 0) { const DrawRec* qd = draws + P.fuse_draw; core_x0 = qd->ix0; core_y0 = qd->iy0; core_x1 = qd->ix1; core_y1 = qd->iy1; }

  // the stash: k-step 2 p + s of it = rows 16 s .. 16 s + 15 of the H-block whose number is p mod HB, one operand (four VGPRs of
  // f16 pairs) per channel.  Indexed by constants only (phase<PH>): it lives in registers.
  uint32_t stash[2 * HB][4][4];  // [k-step][channel][VGPR of f16 pairs]
  auto operand = [](const uint32_t (&u)[4]) { H8Bits o; o.u[0] = u[0]; o.u[1] = u[1]; o.u[2] = u[2]; o.u[3] = u[3]; return o.v; };
  f32x16 acc[4];
  // phase PH = (H-block number) mod HB: round the horizontal product into the stash, then -- `vertical` -- multiply the block
  // that this H-block completes
  auto phase = [&](auto ph_tag, bool vertical) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_tag)::value;
    // (an opaque marker that differs per phase: without it the optimizer hoists the rounding -- identical in the three copies -- in
    // front of the branch and turns "which stash slot" into 96 v_cndmask per block)
    asm volatile("; k_blur_fx: phase %0" ::"n"(PH));
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
          const float x0 = __builtin_fmaf(acc[c][8 * s2 + 2 * qq], kMxScale, kMxMagic), x1 = __builtin_fmaf(acc[c][8 * s2 + 2 * qq + 1], kMxScale, kMxMagic);
          const uint32_t pr = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x0c040c00u);
          stash[2 * PH + s2][c][qq] = pr;
        }
    // (pinned here: the half of the H-block this V block does not read would otherwise be rounded BEHIND the vertical product --
    // the optimizer sinks it towards its first use -- with its 32 accumulator registers alive all the way)
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int qq = 0; qq < 4; qq++) asm volatile("" : "+v"(stash[2 * PH + s2][c][qq]));
#if FDH_TIMING
    T_cv_mark = FDH_NOW() + (__builtin_amdgcn_readfirstlane(stash[2 * PH][0][0] ^ stash[2 * PH + 1][3][3]) & 0u);
#endif
    if (!vertical) return;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[c][e] = 0.0f;
    constexpr int first = (PH + 1) % HB;  // the V block's first H-block: number i - (HB - 1), i.e. (PH + 1) mod HB
    // (the weight fragments one k-step ahead, a scheduling fence per k-step: left to itself the scheduler fetches all 2 NKV
    // fragments first -- 40 registers on top of the stash and the accumulators -- and spills the stash)
    uint4 va = vw[lane], vb = vw[64 + lane];
#pragma unroll
    for (int m = 0; m < NKV; m++) {
      const int slot = 2 * ((first + (m >> 1)) % HB) + (m & 1);
      H8Bits whi, wlo;
      whi.u[0] = va.x; whi.u[1] = va.y; whi.u[2] = va.z; whi.u[3] = va.w; wlo.u[0] = vb.x; wlo.u[1] = vb.y; wlo.u[2] = vb.z; wlo.u[3] = vb.w;
      if (m + 1 < NKV) { va = vw[(2 * m + 2) * 64 + lane]; vb = vw[(2 * m + 3) * 64 + lane]; }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi.v, operand(stash[slot][0]), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi.v, operand(stash[slot][1]), acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi.v, operand(stash[slot][2]), acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi.v, operand(stash[slot][3]), acc[3], 0, 0, 0);
#if FDH_MX_LO
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo.v, operand(stash[slot][0]), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo.v, operand(stash[slot][1]), acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo.v, operand(stash[slot][2]), acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo.v, operand(stash[slot][3]), acc[3], 0, 0, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int n_hblocks = n_blocks + HB - 1;
  int stores_behind = v_pieces;  // memory instructions issued after this block's DMA batch that may stay out (vmcnt retires in issue order): the V block's stores; for block 0 the vertical weights' pieces
  auto iteration = [&](auto ph_tag, int i) __attribute__((always_inline)) {
    // this H-block's texels have landed -- its batch is older than the stores of the V block issued after it, which are NOT
    // waited for (they would cost a store round trip per iteration)
#if FDH_TIMING
    const unsigned long long Ta = FDH_NOW();
#endif
    wait_for_all_but(stores_behind);  // this wave's pieces of block row i have landed (the stores issued behind them may stay out) ...
    stores_behind = 0;
    __builtin_amdgcn_s_barrier();       // ... and so have the other waves'; every wave has also finished reading block row i - 1's copy,
    __builtin_amdgcn_sched_barrier(0);  // into which the next row's pieces go now: they have a whole block's time to land
    if (i + 1 < n_hblocks) issue_block(i + 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!active) return;
    const uint32_t* const src_ring = src_base + ((i & 1) * NKW + 2 * wave) * kMxSlot;  // this strip's NKH k-steps of the window
#if FDH_TIMING
    const unsigned long long Tb = FDH_NOW();
#endif
    // ---- horizontal product of H-block i: rows ws + 32 i .. + 31, columns xb .. xb + 31
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[c][e] = 0.0f;
    {
      // A software pipeline two k-steps deep, its order pinned with scheduling groups.  Left to the scheduler (round 5, first form:
      // loads one k-step ahead in the source, a fence per k-step) the LDS reads of k-step m + 1 sat behind all but the last one or
      // two MFMAs of k-step m and their v_perm straight behind the last: the wave waited out the LDS latency and the sixteen
      // v_perm five times per block with its matrix pipe idle -- 1.04 us per product against 0.5 for the vertical one, which has
      // no operands to build (tools/fx_wave_times.py).  Now, per k-step m: the LDS reads of k-step m + 2 FIRST, then four MFMAs,
      // the sixteen v_perm of k-step m + 1 (its texels were asked for a whole k-step ago), four MFMAs.
      const int sw = (j >> 2) & 3;
      auto texels = [&](int m, uint4& lo4, uint4& hi4) __attribute__((always_inline)) {
        const uint4* rowq = reinterpret_cast<const uint4*>(src_ring + m * kMxSlot + j * 16);
        lo4 = rowq[(2 * g) ^ sw]; hi4 = rowq[(2 * g + 1) ^ sw];
      };
      uint4 tl[3], th[3], wa[2], wb[2];  // raw texels of k-steps m, m + 1, m + 2 (rotating); weight fragments of k-steps m, m + 1
      h8 fr[2][4];                       // operand halves of k-steps m, m + 1
      texels(0, tl[0], th[0]);
      wa[0] = hw[lane]; wb[0] = hw[64 + lane];
      if (NKH > 1) { texels(1, tl[1], th[1]); wa[1] = hw[2 * 64 + lane]; wb[1] = hw[3 * 64 + lane]; }
      {
        const uint32_t t8[8] = {tl[0].x, tl[0].y, tl[0].z, tl[0].w, th[0].x, th[0].y, th[0].z, th[0].w};
        fr[0][0] = mx_frag<0>(t8); fr[0][1] = mx_frag<1>(t8); fr[0][2] = mx_frag<2>(t8); fr[0][3] = mx_frag<3>(t8);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < NKH; m++) {
        const int cur = m & 1, nxt = cur ^ 1;
        H8Bits whi, wlo;
        whi.u[0] = wa[cur].x; whi.u[1] = wa[cur].y; whi.u[2] = wa[cur].z; whi.u[3] = wa[cur].w;
        wlo.u[0] = wb[cur].x; wlo.u[1] = wb[cur].y; wlo.u[2] = wb[cur].z; wlo.u[3] = wb[cur].w;
        if (m + 2 < NKH) texels(m + 2, tl[(m + 2) % 3], th[(m + 2) % 3]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][0], whi.v, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][1], whi.v, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][2], whi.v, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][3], whi.v, acc[3], 0, 0, 0);
        if (m + 1 < NKH) {
          const uint4 &a4 = tl[(m + 1) % 3], &b4 = th[(m + 1) % 3];
          const uint32_t t8[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
          fr[nxt][0] = mx_frag<0>(t8); fr[nxt][1] = mx_frag<1>(t8); fr[nxt][2] = mx_frag<2>(t8); fr[nxt][3] = mx_frag<3>(t8);
        }
#if FDH_MX_LO
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][0], wlo.v, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][1], wlo.v, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][2], wlo.v, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[cur][3], wlo.v, acc[3], 0, 0, 0);
#endif
        if (m + 2 < NKH) { wa[cur] = hw[(2 * m + 4) * 64 + lane]; wb[cur] = hw[(2 * m + 5) * 64 + lane]; }  // (this k-step's weights are in the MFMAs' hands)
        // the order above, made binding: DS reads (texels m + 2) | 4 MFMA | 16 VALU | [4 MFMA of the low halves |] DS reads (weights m + 2)
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
#if FDH_MX_LO
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#else
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#if FDH_TIMING
    const unsigned long long Tc = FDH_NOW() + (__builtin_amdgcn_readfirstlane(__float_as_uint(acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0])) & 0u);
    T_wait += Tb - Ta; T_h += Tc - Tb;
#endif
    const int b = i - (HB - 1);  // the V block whose last H-block this is
    const int bx = xb, by = ys + 32 * b;
    const bool core = bx >= core_x0 && bx + 32 <= core_x1 && by >= core_y0 && by + 32 <= core_y1;  // coverage alpha == 1 (wave-uniform)
    __builtin_amdgcn_sched_barrier(0);
#if FDH_TIMING
    const unsigned long long Tc2 = FDH_NOW();
#endif
    phase(ph_tag, b >= 0);
#if FDH_TIMING
    const unsigned long long Td = FDH_NOW() + (__builtin_amdgcn_readfirstlane(__float_as_uint(acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0])) & 0u);
    T_v += Td - Tc; T_cv += T_cv_mark - Tc2; T_dma += Tc2 - Tc;
#endif
    if (b < 0) return;
    const int x = bx + j;
    const bool x_ok = x >= P.x0 && x < P.x1;
    uint32_t pend[16];
    uint32_t pmask = 0;
#pragma unroll
    for (int rr = 0; rr < 16; rr++)
      pend[rr] = pack2(f2{acc[0][rr] * kMxScale, acc[1][rr] * kMxScale}, f2{acc[2][rr] * kMxScale, acc[3][rr] * kMxScale});
    if (bx >= P.x0 && bx + 32 <= P.x1 && by >= P.y0 && by + 32 <= P.y1) {
      pmask = 0xffffu;
    } else {
#pragma unroll
      for (int rr = 0; rr < 16; rr++) {
        const int y = by + (rr & 3) + 8 * (rr >> 2) + 4 * g;
        if (x_ok && y >= P.y0 && y < P.y1) pmask |= 1u << rr;
      }
    }
    if (P.fuse_draw >= 0) {
      // atlas.frag:381-388 on the blurred texel just produced, blended over the live texel (src: dst is another surface, so
      // every pixel of the region is written: where the quad does not cover, the live texel passes through the blend unchanged)
      bool replace_all = false;
      if (core) {
        uint32_t conj = pend[0];
#pragma unroll
        for (int rr = 1; rr < 16; rr++) conj &= pend[rr];
        replace_all = __all((conj >> 24) == 255u);
      }
      if (!replace_all) {
        // `al16[rr]`: the quad's coverage at this lane's pixel of row rr.  A block on the quad's border (a few hundred of 8100 at 4K):
        // its rows and columns inside the core have alpha == 1; the others are evaluated densely -- one row (lane = x) or one column
        // (lane = y) per lane group and step, as in k_blur_mx -- and each lane then FETCHES the alphas of its own sixteen pixels
        // from the lanes that evaluated them (ds_bpermute: the LDS crossbar, no LDS memory; rounds 3 - 4 went through two idle ring
        // slots, which the shared source window no longer has).
        const uint32_t valid = pmask;
        float al16[16];
#pragma unroll
        for (int rr = 0; rr < 16; rr++) al16[rr] = 1.0f;
        if (!core) {
          const DrawRec r = load_rec(draws + P.fuse_draw);
          // the block's rows / columns INSIDE THE REGION, [ry0, ry1) x [rx0, rx1), less those in the core's, [ra, rb) x [ca, cb): what is left
          // above / below (left / right of) the core is evaluated.  (Rounds 3 - 4 evaluated every row of the block outside the core's:
          // the frame's last block row -- 2160 = 67.5 blocks -- spent seventeen steps on a block with ONE row on the quad's edge and
          // sixteen below the frame; those waves lived 32 - 35 us against 24, and the kernel lasts as long as its slowest wave.)
          const int ry0 = min(max(P.y0 - by, 0), 32), ry1 = max(ry0, min(P.y1 - by, 32)), rx0 = min(max(P.x0 - bx, 0), 32), rx1 = max(rx0, min(P.x1 - bx, 32));
          const int ra = min(max(core_y0 - by, ry0), ry1), rb = max(ra, min(max(core_y1 - by, ry0), ry1));
          const int ca = min(max(core_x0 - bx, rx0), rx1), cb = max(ca, min(max(core_x1 - bx, rx0), rx1));
          const int nra = ra - ry0, nca = ca - rx0, nr = nra + ry1 - rb, nc = nca + rx1 - cb;
#pragma unroll 1
          for (int u0 = 0; u0 < nr + nc; u0 += 2) {
            const int u = min(u0 + g, nr + nc - 1);  // (an odd count: the second lane group repeats the last step)
            int dx, dy;
            if (u < nr) { dy = u < nra ? ry0 + u : rb + (u - nra); dx = j; }
            else { const int v = u - nr; dx = v < nca ? rx0 + v : cb + (v - nca); dy = j; }
            const int ex = bx + dx, ey = by + dy;
            float alpha = 0.0f;  // (outside the quad the live texel stays: a blend with alpha 0; outside the region nothing is stored)
            if (ex >= P.x0 && ex < P.x1 && ey >= P.y0 && ey < P.y1) {
              const Frag f = make_frag(r, exts, ex, ey);
              if (f.covered) {
                const float lx = (f.u - 0.5f) * 2.0f * r.p0, ly = (f.v - 0.5f) * 2.0f * r.p1;
                const float dist = shape_dist((r.op_mode & F_ELLIP) != 0u, lx, -ly, r.p2, r.p3, r.r[0], r.r[1], r.r[2], r.r[3]);
                alpha = 1.0f - clamp01(r.aa * dist + 0.5f);
              }
            }
            const int abits = (int)__float_as_uint(alpha);
#pragma unroll
            for (int rr = 0; rr < 16; rr++) {
              const int y = (rr & 3) + 8 * (rr >> 2) + 4 * g;  // this lane's pixel (j, y): which step evaluated it, in which lane?
              int su = -2, sl = 0;  // (a pixel outside the region is never stored: its alpha stays whatever it is)
              if (y >= ry0 && y < ry1 && j >= rx0 && j < rx1) {
                if (y < ra || y >= rb) { su = y < ra ? y - ry0 : nra + (y - rb); sl = j; }
                else if (j < ca || j >= cb) { su = nr + (j < ca ? j - rx0 : nca + (j - cb)); sl = y; }
              }
              const int v = __builtin_amdgcn_ds_bpermute(4 * (sl + 32 * (su & 1)), abits);
              if ((su & ~1) == u0) al16[rr] = __uint_as_float((uint32_t)v);
            }
          }
        }
        uint32_t blend_mask = 0;
#pragma unroll
        for (int rr = 0; rr < 16; rr++) {
          if (!((valid >> rr) & 1u)) continue;
          if (al16[rr] != 1.0f || (pend[rr] >> 24) != 255u) blend_mask |= 1u << rr;
        }
        if (__any(blend_mask != 0u)) {
          uint32_t dstv[16];
#pragma unroll
          for (int rr = 0; rr < 16; rr++)  // all the loads first
            dstv[rr] = ((blend_mask >> rr) & 1u) ? P.src[(size_t)(by + (rr & 3) + 8 * (rr >> 2) + 4 * g) * P.pitch + x] : 0u;
          const float k = 1.0f / 255.0f;
#pragma unroll
          for (int rr = 0; rr < 16; rr++) {
            if (!((blend_mask >> rr) & 1u)) continue;
            const float alpha = al16[rr];
            const F4 bl = unpack255(pend[rr]);
            F4 Fd = unpack255(dstv[rr]);
            const float sa = bl.w * k * alpha, A = 255.0f * sa, ia = 1.0f - sa;
            Fd.x = __builtin_rintf(__builtin_fmaf(Fd.x, ia, bl.x * k * A));
            Fd.y = __builtin_rintf(__builtin_fmaf(Fd.y, ia, bl.y * k * A));
            Fd.z = __builtin_rintf(__builtin_fmaf(Fd.z, ia, bl.z * k * A));
            Fd.w = __builtin_rintf(__builtin_fmaf(Fd.w, ia, A));
            pend[rr] = pack255(Fd);
          }
          __builtin_amdgcn_s_waitcnt(0x0F70);
        }
      }
    }
#if FDH_TIMING
    const unsigned long long Te = FDH_NOW() + (__builtin_amdgcn_readfirstlane(pend[0] + pend[15]) & 0u);
    T_epi += Te - Td;
#endif
    __builtin_amdgcn_sched_barrier(0);
    // the block's rows: uniform row pointer + one per-lane byte offset
    {
      const uint32_t lane_off = ((uint32_t)(4 * g) * (uint32_t)P.pitch + (uint32_t)j) * 4u;
      char* base = reinterpret_cast<char*>(P.dst + (size_t)by * P.pitch + bx);
      if (__all(pmask == 0xffffu)) {
#pragma unroll
        for (int rr = 0; rr < 16; rr++)
          *reinterpret_cast<uint32_t*>(base + (size_t)((rr & 3) + 8 * (rr >> 2)) * P.pitch * 4u + lane_off) = pend[rr];
        stores_behind = 16;  // exactly sixteen store instructions
      } else if (__any(pmask != 0u)) {
#pragma unroll
        for (int rr = 0; rr < 16; rr++)
          if ((pmask >> rr) & 1u) *reinterpret_cast<uint32_t*>(base + (size_t)((rr & 3) + 8 * (rr >> 2)) * P.pitch * 4u + lane_off) = pend[rr];
        // (an unknown number: the next wait drains everything)
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#if FDH_TIMING
    T_st += FDH_NOW() - Te;
#endif
  };
#pragma unroll 1
  for (int i = 0; i < n_hblocks; i += HB) {
    iteration(std::integral_constant<int, 0>{}, i);
    if (HB >= 2) { if (i + 1 >= n_hblocks) break; iteration(std::integral_constant<int, HB >= 2 ? 1 : 0>{}, i + 1); }
    if (HB >= 3) { if (i + 2 >= n_hblocks) break; iteration(std::integral_constant<int, HB >= 3 ? 2 : 0>{}, i + 2); }
  }
#if FDH_TIMING
  if (lane == 0) {
    const size_t w_id = (size_t)blockIdx.x * kFxWaves + wave;
    if (w_id < 32768) {
      unsigned long long* row = g_wave_times + 16 * w_id;
      row[0] = FDH_NOW() - T0; row[1] = T_pro; row[2] = T_wait; row[3] = T_h; row[4] = T_v; row[5] = T_epi; row[6] = 7; row[7] = T_st; row[8] = n_hblocks; row[9] = T0; row[10] = n_blocks; row[11] = T_cv; row[12] = T_dma;
    }
  }
#endif
}

// ------------------------------------------------------------------ glyph images on their way into the atlas
// FreeType's default 5-tap LCD filter as the reference applies it to a rasterised glyph before the upload
// (common/textrasters/pixie_raster.nim:12-43): weights 8, 77, 86, 77, 8 over x - 2 .. x + 2 with the column clamped to the image,
// per channel (sum + 128) >> 8.  Integer arithmetic: bit-exact with the oracle's restatement.
__global__ void k_lcd_filter(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const int wt[5] = {8, 77, 86, 77, 8};
  int sr = 0, sg = 0, sb = 0, sa = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const int sx = min(max(x + i - 2, 0), w - 1);
    const uint32_t p = src[(size_t)y * w + sx];
    sr += (int)(p & 255u) * wt[i]; sg += (int)((p >> 8) & 255u) * wt[i]; sb += (int)((p >> 16) & 255u) * wt[i]; sa += (int)(p >> 24) * wt[i];
  }
  dst[(size_t)y * w + x] = (uint32_t)(((sr + 128) >> 8) & 255) | ((uint32_t)(((sg + 128) >> 8) & 255) << 8) |
                           ((uint32_t)(((sb + 128) >> 8) & 255) << 16) | ((uint32_t)(((sa + 128) >> 8) & 255) << 24);
}
// one mip step of updateSubImage (textures.nim:106-119) = pixie's Image.minifyBy2 on premultiplied RGBA8: box sum div 4; an odd
// extent rounds the result size up and the extra column / row / corner carry half / half / quarter coverage (the arithmetic
// is pinned by the reference's data/img1.flippy: minify_by2_host in fdh_context.cpp spells it out)
__global__ void k_minify2(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int sw, int sh) {
  const int nw = (sw + 1) >> 1, nh = (sh + 1) >> 1;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= nw || y >= nh) return;
  const bool col_pair = 2 * x + 1 < sw, row_pair = 2 * y + 1 < sh;
  const int x0 = col_pair ? 2 * x : sw - 1, x1 = col_pair ? 2 * x + 1 : sw - 1, y0 = row_pair ? 2 * y : sh - 1, y1 = row_pair ? 2 * y + 1 : sh - 1;
  const uint32_t a = src[(size_t)y0 * sw + x0], b = src[(size_t)y0 * sw + x1], c = src[(size_t)y1 * sw + x0], d = src[(size_t)y1 * sw + x1];
  uint32_t o = 0;
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const uint32_t ca = (a >> k) & 255u, cb = (b >> k) & 255u, cc = (c >> k) & 255u, cd = (d >> k) & 255u;
    uint32_t v;
    if (col_pair && row_pair) v = (ca + cb + cc + cd) >> 2;
    else if (row_pair) v = ((ca * 127u + cc * 128u) / 255u) * 128u / 255u;  // last column: rows 2y, 2y + 1
    else if (col_pair) v = ((ca * 127u + cb * 128u) / 255u) * 128u / 255u;  // last row: columns 2x, 2x + 1
    else v = ca * 64u / 255u;
    o |= v << k;
  }
  dst[(size_t)y * nw + x] = o;
}
// a w x h image into the rectangle (x, y) of one atlas level (LS texels wide); texels outside the level are dropped
__global__ void k_atlas_blit(uint32_t* __restrict__ level, int LS, int x, int y, const uint32_t* __restrict__ src, int w, int h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= w || j >= h) return;
  const int tx = x + i, ty = y + j;
  if (tx >= 0 && ty >= 0 && tx < LS && ty < LS) level[(size_t)ty * LS + tx] = src[(size_t)j * w + i];
}
// Glyph outline -> coverage: exact-area scanline accumulation (oracle/figdraw_oracle.c, raster_row_line, operation for operation:
// no FMA contraction here so that both produce the same floats).  One lane owns one pixel row: it walks every line segment in
// order and adds the signed areas of the part inside its row to the row's accumulation cells (global scratch, w + 2 floats per
// row: nobody else touches them), then a running sum along the row turns areas into coverage.  Out: premultiplied white.
__device__ __forceinline__ void raster_row_line(float* acc, int w, int y, float x0, float y0, float x1, float y1) {
#pragma clang fp contract(off)
  if (y0 == y1) return;
  float dir = 1.0f;
  if (y0 > y1) { float t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; dir = -1.0f; }
  const float ya = y0 > (float)y ? y0 : (float)y, yb = y1 < (float)(y + 1) ? y1 : (float)(y + 1);
  if (!(yb > ya)) return;
  const float dxdy = (x1 - x0) / (y1 - y0);
  const float xa = x0 + (ya - y0) * dxdy, xb = x0 + (yb - y0) * dxdy;
  const float d = (yb - ya) * dir;
  float xl = xa < xb ? xa : xb, xr = xa < xb ? xb : xa;
  if (xl < 0.0f) xl = 0.0f;
  if (xr < 0.0f) xr = 0.0f;
  if (xl > (float)w) xl = (float)w;
  if (xr > (float)w) xr = (float)w;
  const float x0floor = __builtin_floorf(xl);
  const int x0i = (int)x0floor;
  const float x1ceil = __builtin_ceilf(xr);
  const int x1i = (int)x1ceil;
  if (x1i <= x0i + 1) {
    const float xmf = 0.5f * (xl + xr) - x0floor;
    acc[x0i] += d - d * xmf;
    if (x0i + 1 <= w) acc[x0i + 1] += d * xmf;
  } else {
    const float s = 1.0f / (xr - xl);
    const float x0f = xl - x0floor;
    const float a0 = 0.5f * s * (1.0f - x0f) * (1.0f - x0f);
    const float x1f = xr - x1ceil + 1.0f;
    const float am = 0.5f * s * x1f * x1f;
    acc[x0i] += d * a0;
    if (x1i == x0i + 2) {
      acc[x0i + 1] += d * (1.0f - a0 - am);
    } else {
      const float a1 = s * (1.5f - x0f);
      acc[x0i + 1] += d * (a1 - a0);
      for (int xi = x0i + 2; xi < x1i - 1; xi++) acc[xi] += d * s;
      const float a2 = a1 + (float)(x1i - x0i - 3) * s;
      acc[x1i - 1] += d * (1.0f - a2 - am);
    }
    if (x1i <= w) acc[x1i] += d * am;
  }
}
__global__ __launch_bounds__(64) void k_rasterize_lines(const float4* __restrict__ lines, int n, int w, int h, float* __restrict__ scratch, uint32_t* __restrict__ out) {
#pragma clang fp contract(off)
  const int y = blockIdx.x * 64 + threadIdx.x;
  if (y >= h) return;
  float* acc = scratch + (size_t)y * (w + 2);
  for (int x = 0; x < w + 2; x++) acc[x] = 0.0f;
  for (int i = 0; i < n; i++) {
    const float4 l = lines[i];
    raster_row_line(acc, w, y, l.x, l.y, l.z, l.w);
  }
  float sum = 0.0f;
  for (int x = 0; x < w; x++) {
    sum += acc[x];
    float c = __builtin_fabsf(sum);
    if (c > 1.0f) c = 1.0f;
    const uint32_t v = (uint32_t)(c * 255.0f + 0.5f);
    out[(size_t)y * w + x] = v * 0x01010101u;
  }
}
void launch_rasterize_lines(hipStream_t s, const float4* lines, int n, int w, int h, float* scratch, uint32_t* out) {
  if (w > 0 && h > 0) hipLaunchKernelGGL(k_rasterize_lines, dim3((h + 63) / 64), dim3(64), 0, s, lines, n, w, h, scratch, out);
}
void launch_lcd_filter(hipStream_t s, const uint32_t* src, uint32_t* dst, int w, int h) {
  if (w > 0 && h > 0) hipLaunchKernelGGL(k_lcd_filter, dim3((w + 63) / 64, h), dim3(64), 0, s, src, dst, w, h);
}
void launch_minify2(hipStream_t s, const uint32_t* src, uint32_t* dst, int sw, int sh) {
  const int nw = (sw + 1) / 2, nh = (sh + 1) / 2;
  if (sw > 0 && sh > 0) hipLaunchKernelGGL(k_minify2, dim3((nw + 63) / 64, nh), dim3(64), 0, s, src, dst, sw, sh);
}
void launch_atlas_blit(hipStream_t s, uint32_t* level, int LS, int x, int y, const uint32_t* src, int w, int h) {
  if (w > 0 && h > 0) hipLaunchKernelGGL(k_atlas_blit, dim3((w + 63) / 64, h), dim3(64), 0, s, level, LS, x, y, src, w, h);
}

__global__ void k_fill_u32(uint32_t* p, uint32_t v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

#endif  // FDH_TU == 0

#if FDH_TU == 0
// ------------------------------------------------------------------ launch wrappers (called from fdh_context.cpp)
// Per-kernel timing (fdh_profile): with a pair of events set, the next launch goes through hipExtLaunchKernelGGL, which stamps
// them from the dispatch's own start / end timestamps -- the kernel's execution time as rocprofv3 reports it.  (Events
// recorded around a launch also count the gap to the neighbouring dispatches: +2..5 us on a 20 us kernel.)
void launch_composite_uniform(hipStream_t s, hipEvent_t e0, hipEvent_t e1, int grid, size_t lds, const DrawRec* draws, const QuadExt* exts, const CompositeParams& P, int paths);  // FDH_TU 1
static thread_local hipEvent_t t_prof_start = nullptr, t_prof_stop = nullptr;
static thread_local bool t_prof_used = false;
void set_launch_events(hipEvent_t start, hipEvent_t stop) { t_prof_start = start; t_prof_stop = stop; t_prof_used = false; }
bool launch_events_used() { return t_prof_used; }  // false: the launch_* call between had nothing to launch
#define FDH_LAUNCH(kern, grid, block, lds, stream, ...)                                                                  \
  do {                                                                                                                   \
    if (t_prof_start) { hipExtLaunchKernelGGL(kern, grid, block, lds, stream, t_prof_start, t_prof_stop, 0, __VA_ARGS__); t_prof_used = true; } \
    else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                                \
  } while (0)
void launch_bin(hipStream_t s, const BinParams& P) {
  const int n = P.sub_n ? P.sub_first[P.sub_n] : P.n_phases * P.bins_x * P.bins_y;
  if (n <= 0) return;
  if (P.refine) FDH_LAUNCH(k_bin_draws<true>, dim3(n), dim3(64), 0, s, P);
  else FDH_LAUNCH(k_bin_draws<false>, dim3(n), dim3(64), 0, s, P);
}
void launch_composite(hipStream_t s, const DrawRec* draws, const QuadExt* exts, CompositeParams P) {
  const int n = P.bin_nx * P.bin_ny * kWgsPerBin;
  if (n <= 0) return;
  P.n_wg = n;
  const int bins8 = (P.bin_nx * P.bin_ny + 7) / 8;  // bins per XCD
  const int grid = 8 * bins8 * kWgsPerBin * kWavesPerWg + (P.order_next ? 8 : 0);  // 16 strips per bin, one wavefront each (+ the sorting one)
  const dim3 blk(64);
  // FDH_FORCE_KERNEL_PATHS=3 (or 2, or 1): run a more general build than the phase needs -- a test hook: every build must give
  // the same pixels (tests/test_hip_parity.py)
  static const int force = [] { const char* e = std::getenv("FDH_FORCE_KERNEL_PATHS"); return e ? std::atoi(e) : 0; }();
  if (force == 3) P.has_slow = 1;
  if (force == 8) P.has_rot = 1;
  if (force == 2) P.has_atlas = 1;
  if (force == 1) P.has_masks = 1;  // (the build with mask registers and the 4-KB stack, even where no clip is open)
  const size_t lds = (P.has_masks ? sizeof(uint32_t) * kMaskDepth * 64 : sizeof(uint32_t) * 256) +
                     ((P.has_atlas || P.has_slow || (P.has_rot && P.has_atlas)) ? sizeof(uint32_t) * kWinRows * kWinStride : 0);  // + the texel window of the atlas path
  const bool full = P.load_fb == 0;  // the launch that starts a frame (k_composite_tiles<., true>)
  // deep strips (k_composite_deep): only beside the no-clip build's full-frame launch, and only with the order at hand
  if (!(full && P.order && P.order_next && !P.has_slow && !P.has_rot && !P.has_atlas && !P.has_masks)) P.deep_k8 = 0;
  P.deep_k8 = std::min(P.deep_k8 & ~7, 8 * bins8);
#define FDH_COMPOSITE(paths) \
  do { if (full) FDH_LAUNCH((k_composite_tiles<paths, true>), dim3(grid), blk, lds, s, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P); \
       else FDH_LAUNCH((k_composite_tiles<paths, false>), dim3(grid), blk, lds, s, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P); } while (0)
#if FDH_SPLIT_UNIFORM
  if (P.has_slow || (P.has_rot && P.has_atlas)) FDH_COMPOSITE(3);
  else if (P.has_rot) FDH_COMPOSITE(8);
  else { launch_composite_uniform(s, t_prof_start, t_prof_stop, grid, lds, draws, exts, P, P.has_atlas ? 2 : P.has_masks ? 0 : 4); if (t_prof_start) t_prof_used = true; }
#else
  if (P.has_slow || (P.has_rot && P.has_atlas)) FDH_COMPOSITE(3);
  else if (P.has_rot) FDH_COMPOSITE(8);
  else if (P.has_atlas) FDH_COMPOSITE(2);
  else if (!P.has_masks) {
    if (P.deep_k8 > 0 && P.order && P.order_next) {  // a frame with deep bins: the launch of four-wave workgroups
      const dim3 g(8 + P.deep_k8 * 16 + 8 * bins8 * kWgsPerBin);
      FDH_LAUNCH((k_composite_deep<0>), g, dim3(256), sizeof(uint32_t) * kDeepLdsDwords, s, draws, exts, P);
    } else FDH_COMPOSITE(4);
  }
  else FDH_COMPOSITE(0);
#endif
#undef FDH_COMPOSITE
}
// small regions: fewer outputs per thread -> more, shorter waves (see NOUT above)
#ifndef FDH_BLUR_NOUT
#define FDH_BLUR_NOUT 0  // 0: pick 8, 10 or 12 outputs per thread per launch (blur_pick_nout); otherwise that value everywhere
#endif
#ifndef FDH_BLUR_VWAVES
#define FDH_BLUR_VWAVES 4  // waves per V-pass workgroup: the tile is 64 columns x (waves * outputs) rows
#endif
// FDH_FORCE_BLUR_PATH=1|2|3 (a test hook, tests/test_hip_parity.py): every blur on the 2-outputs-per-thread passes / the
// many-outputs passes / the matrix-pipe passes, whatever the region size
static int blur_forced_path() { static const int v = [] { const char* e = std::getenv("FDH_FORCE_BLUR_PATH"); return e ? std::atoi(e) : 0; }(); return v; }
static bool blur_small(const BlurParams& P) {
  if (blur_forced_path()) return blur_forced_path() == 1;
  // tools/blur_size_sweep.py, full-frame blur(18), H + V: 640x360 16.4 us on these passes / 17.3 on the matrix pipe,
  // 960x540 20.8 / 16.3, 1280x720 26.1 / 18.9
  return (P.node_pixels > 0 ? P.node_pixels : (long long)(P.x1 - P.x0) * (P.y1 - P.y0)) < 384 * 1024;
}
// Outputs per thread for a large region: more outputs share each unpacked texel (4 converts per texel and n outputs
// against the 2 * taps pair FMAs every output needs anyway), but the extent along the pass is cut into units of
// `quantum * n` and the last unit of every row / column runs with idle lanes.  3840 px in 512-px waves is 7.5 waves per row
// (1/16 of the pass wasted); in 768-px waves it is exactly 5.  Cost model = padded extent x instructions per output.
static int blur_pick_nout(int extent, int quantum, int reach) {
  if (FDH_BLUR_NOUT) return FDH_BLUR_NOUT;
  int best = 8;
  double best_cost = 1e300;
  for (int n : {8, 10, 12}) {
    const int unit = quantum * n;
    const double padded = (double)((extent + unit - 1) / unit) * unit;
    const double per_output = 2.0 * (2 * reach + 1) + 4.0 * (n + 2 * reach) / n + 6.0;
    const double cost = padded * per_output;
    if (cost < best_cost) { best_cost = cost; best = n; }
  }
  return best;
}
template <int NOUT> static void launch_blur_h_n(hipStream_t s, const BlurParams& P) {
  dim3 grid((P.x1 - P.x0 + 64 * NOUT - 1) / (64 * NOUT), (P.y1 - P.y0 + 3) / 4);
  FDH_LAUNCH(k_blur_h<NOUT>, grid, dim3(256), 0, s, P);
}
template <int NOUT, int WAVES> static void launch_blur_v_n(hipStream_t s, const BlurParams& P, const DrawRec* draws, const QuadExt* exts) {
  const int ntx = (P.x1 - P.x0 + kBlurVW - 1) / kBlurVW, nty = (P.y1 - P.y0 + WAVES * NOUT - 1) / (WAVES * NOUT);
  dim3 grid(8 * ((ntx * nty + 7) / 8));  // 8 XCDs x an eighth of the tile sequence each
  const size_t lds = (size_t)(WAVES * NOUT + 2 * P.taps.reach) * kBlurVW * sizeof(uint32_t);
  FDH_LAUNCH((k_blur_v<NOUT, WAVES>), grid, dim3(64 * WAVES), lds, s, P, draws, exts);
}
// Matrix-pipe passes: NK k-steps of 16 texels must cover a block's 32 + 2 reach window (+ up to 3 texels of alignment
// for the horizontal pass); T blocks per wave, as many as still leave every SIMD a couple of waves.
#ifndef FDH_BLUR_MX
#define FDH_BLUR_MX 1
#endif
// Blocks per wave: the smallest T for which every wave of the launch is resident at once.  A wave lives for the whole pass
// (prologue + T blocks), so a second, partly filled round of waves costs a whole wave lifetime.  Waves per CU = what the
// runtime's occupancy query gives the instantiation with its LDS ring (NK + 4 slots of 2 KB per wave; registers: three waves
// per SIMD for the narrow filters, two for the widest ones).  (4K, NK = 5: T = 4, 2040 waves for 2048 slots; leaving a
// tenth of the slots free, T = 5, measured 1 us slower on the vertical pass.)
static int mx_pick_t(long long per_cu, long long outputs_along, long long lines) {
#if defined(FDH_MX_T)  // experiment builds only (make variant DEFS=-DFDH_MX_T=n): a fixed T
  return FDH_MX_T;
#endif
  // (two waves per SIMD at most: with the horizontal pass's smaller ring eleven fit a CU, and T = 3 with 2720 shorter waves
  // measured 24.6 us against 23.4 for the two blur launches of the bench frame)
  const long long slots = 256 * std::min<long long>(per_cu, 4 * FDH_MX_WAVES);
  const long long along_blocks = (outputs_along + 31) / 32, line_groups = (lines + 31) / 32;
  for (int t = 1; t < 64; t++) if (line_groups * ((along_blocks + t - 1) / t) <= slots) return t;
  return 64;
}
template <int NK, bool kV> static void launch_blur_mx(hipStream_t s, const BlurParams& P, const DrawRec* draws, const QuadExt* exts) {
  constexpr size_t lds = (size_t)mx_ring_slots(NK, kV) * kMxSlot * sizeof(uint32_t);
  constexpr int kWG = mx_wg(NK, kV);
  static const int per_cu = [] {  // waves resident per CU, asked once per instantiation
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_blur_mx<NK, kV>, 64 * kWG, lds * kWG) != hipSuccess || n <= 0) n = std::min<int>(4 * FDH_MX_WAVES, 160 / (mx_ring_slots(NK, kV) * 2)) / kWG;
    return n * kWG;  // (waves)
  }();
  const int t = kV ? mx_pick_t(per_cu, P.y1 - P.y0, P.x1 - P.x0) : mx_pick_t(per_cu, P.x1 - P.x0, P.y1 - P.y0);
  const int a_lo = kV ? P.y0 : P.x0, a_hi = kV ? P.y1 : P.x1, l_lo = kV ? (P.x0 & ~31) : P.y0, l_hi = kV ? P.x1 : P.y1;
  const int total = ((a_hi - (a_lo & ~31) + 32 * t - 1) / (32 * t)) * ((l_hi - l_lo + 31) / 32);
  FDH_LAUNCH((k_blur_mx<NK, kV>), dim3(8 * ((total + 8 * kWG - 1) / (8 * kWG))), dim3(64 * kWG), lds * kWG, s, P, draws, exts, t);
}
template <bool kV> static bool launch_blur_mx_nk(hipStream_t s, const BlurParams& P, const DrawRec* draws, const QuadExt* exts) {
  // LDS-DMA moves 16-byte pieces: rows have to start on 16-byte boundaries
  if (!P.mx_w || (P.pitch & 3) || (reinterpret_cast<uintptr_t>(P.src) & 15) || P.W < 4) return false;
  const int nk = mx_nk(P.taps.reach, kV);
  switch (nk) {
    case 3: launch_blur_mx<3, kV>(s, P, draws, exts); return true;
    case 4: launch_blur_mx<4, kV>(s, P, draws, exts); return true;
    case 5: launch_blur_mx<5, kV>(s, P, draws, exts); return true;
    case 6: launch_blur_mx<6, kV>(s, P, draws, exts); return true;
    case 7: launch_blur_mx<7, kV>(s, P, draws, exts); return true;
    case 8: launch_blur_mx<8, kV>(s, P, draws, exts); return true;
    case 9: launch_blur_mx<9, kV>(s, P, draws, exts); return true;
    case 10: launch_blur_mx<10, kV>(s, P, draws, exts); return true;
    case 11: launch_blur_mx<11, kV>(s, P, draws, exts); return true;  // reach 66 = the widest filter (radius clamp 64)
    default: return false;
  }
}
// Both passes in one kernel (k_blur_fx): instantiated for the filter widths whose two rings fit five waves' worth of LDS per CU
// (NKH <= 6: tap reach <= 22, blur radius <= ~21); wider filters keep the two-pass route.
template <int NKH, int NKV> static void launch_blur_fx(hipStream_t s, const BlurParams& P, const uint4* w_v, const DrawRec* draws, const QuadExt* exts) {
  constexpr size_t lds = (size_t)fx_slots(NKH, NKV) * kMxSlot * sizeof(uint32_t);  // per workgroup of kFxWaves waves
  static const int wg_per_cu = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_blur_fx<NKH, NKV>, 64 * kFxWaves, lds) != hipSuccess || n <= 0) n = std::min<int>(8 / kFxWaves, (int)(160 * 1024 / lds));
    return n;
  }();
  const int n_strips = (P.x1 - (P.x0 & ~31) + 31) >> 5, n_sg = (n_strips + kFxWaves - 1) / kFxWaves, blocks = (P.y1 - (P.y0 & ~31) + 31) >> 5;
  const long long slots = 256LL * std::min(wg_per_cu, 8 / kFxWaves);  // workgroups resident at once
  int t = 2;  // (a one-block segment would filter three H-blocks per output block)
  while (t < 64 && (long long)n_sg * ((blocks + t - 1) / t) > slots) t++;
#if defined(FDH_FX_T)  // experiment builds only (make variant DEFS=-DFDH_FX_T=n; tools/fx_t_sweep.sh): a fixed T
  t = FDH_FX_T;
#endif
  const int total = n_sg * ((blocks + t - 1) / t), per = (total + 7) / 8;
  FDH_LAUNCH((k_blur_fx<NKH, NKV>), dim3(8 * per), dim3(64 * kFxWaves), lds, s, P, w_v, draws, exts, t);
}
bool blur_fused_supported(int reach, int W, int pitch) {
  const int nkh = mx_nk(reach, false), nkv = mx_nk(reach, true);
  return FDH_BLUR_MX && (pitch & 3) == 0 && W >= 4 && nkh >= 3 && nkh <= 6 && (nkh == nkv || nkh == nkv + 1);
}
bool launch_blur_fused(hipStream_t s, const BlurParams& P, const uint4* w_v, const DrawRec* draws, const QuadExt* exts) {
  if (P.x1 <= P.x0 || P.y1 <= P.y0) return true;
  if (!P.mx_w || !w_v || !blur_fused_supported(P.taps.reach, P.W, P.pitch) || (reinterpret_cast<uintptr_t>(P.src) & 15)) return false;
  const int nkh = mx_nk(P.taps.reach, false), nkv = mx_nk(P.taps.reach, true);
#define FDH_FX(a, b) if (nkh == a && nkv == b) { launch_blur_fx<a, b>(s, P, w_v, draws, exts); return true; }
  FDH_FX(3, 3) FDH_FX(4, 3) FDH_FX(4, 4) FDH_FX(5, 4) FDH_FX(5, 5) FDH_FX(6, 5) FDH_FX(6, 6)
#undef FDH_FX
  return false;
}
// one kernel for a small region (k_blur_small): the region sizes the small-region passes take, filters of reach <= 24 (the tile's
// source window and its horizontal result stay under 20 KB of LDS; a wider filter re-filters too many halo rows per 16-row tile)
bool blur_one_kernel_ok(int w, int h, int reach) {
  if (blur_forced_path() || w <= 0 || h <= 0) return false;
  return (long long)w * h < 384 * 1024 && reach <= 24;
}
void launch_blur_small(hipStream_t s, const BlurParams& P) {
  if (P.x1 <= P.x0 || P.y1 <= P.y0) return;
  const int reach = P.taps.reach, ntx = (P.x1 - P.x0 + kSmallTW - 1) / kSmallTW, nty = (P.y1 - P.y0 + kSmallTH - 1) / kSmallTH;
  const size_t lds = (size_t)(kSmallTH + 2 * reach) * (size_t)(2 * kSmallTW + 2 * reach) * sizeof(uint32_t);
  FDH_LAUNCH(k_blur_small, dim3(ntx * nty), dim3(kSmallThreads), lds, s, P);
}
void launch_blur_h(hipStream_t s, const BlurParams& P) {
  if (P.x1 <= P.x0 || P.y1 <= P.y0) return;
  if (blur_small(P)) { launch_blur_h_n<2>(s, P); return; }
  if (FDH_BLUR_MX && blur_forced_path() != 2 && launch_blur_mx_nk<false>(s, P, nullptr, nullptr)) return;
  switch (blur_pick_nout(P.x1 - P.x0, 64, P.taps.reach)) {
    case 12: launch_blur_h_n<12>(s, P); break;
    case 10: launch_blur_h_n<10>(s, P); break;
    case 16: launch_blur_h_n<16>(s, P); break;
    default: launch_blur_h_n<8>(s, P); break;
  }
}
void launch_blur_v(hipStream_t s, const BlurParams& P, const DrawRec* draws, const QuadExt* exts) {
  if (P.x1 <= P.x0 || P.y1 <= P.y0) return;
  if (blur_small(P)) { launch_blur_v_n<2, 4>(s, P, draws, exts); return; }
  if (FDH_BLUR_MX && blur_forced_path() != 2 && launch_blur_mx_nk<true>(s, P, draws, exts)) return;
  switch (blur_pick_nout(P.y1 - P.y0, FDH_BLUR_VWAVES, P.taps.reach)) {
    case 12: launch_blur_v_n<12, FDH_BLUR_VWAVES>(s, P, draws, exts); break;
    case 10: launch_blur_v_n<10, FDH_BLUR_VWAVES>(s, P, draws, exts); break;
    case 16: launch_blur_v_n<16, FDH_BLUR_VWAVES>(s, P, draws, exts); break;
    default: launch_blur_v_n<8, FDH_BLUR_VWAVES>(s, P, draws, exts); break;
  }
}
// Frame upload as a kernel on the render stream: the sources are pinned host memory mapped into the device's address space,
// read over the host link.  (hipMemcpyAsync hands the copy to another engine; the round trip of dependencies between that
// engine and the compute queue cost ~60 us of idle GPU per frame.)
//
// k_upload_frame GATHERS the frame block: the recording threads leave the frame as PIECES -- runs of DrawRecs, BinRecs and quad
// extensions in each thread's own pinned arrays (fdh_context.h: Lane) -- and the table in the kernel arguments says where every
// run goes in the dense device arrays.  One wavefront per 1-KB unit of a run (the whole table sits in SGPRs / the scalar cache:
// no dependent round trip over the host link before the data's own).  A piece's extension indices are lane-relative; the copy
// re-bases them.  The 4-byte bin boxes the bin kernel scans are derived on the way: the lane that copies the first 8 bytes of
// a BinRec -- its pixel bounds -- writes the draw's box too, so the host never stores or sends them.  (First version: a second
// role in this kernel read the BinRecs again at the source, per 256 draws, for boxes and chunk boxes: its dependent reads over the
// host link made the launch 20 us long; the chunk boxes now come from the host, a few dozen bytes.)
__global__ __launch_bounds__(64) void k_upload_frame(uint8_t* __restrict__ dst, UploadTable T) {
  const uint32_t lane = threadIdx.x;
  // blockIdx.y = the run, blockIdx.x = the 1-KB unit inside it (the grid is as wide as the longest run; the workgroups past a
  // shorter run's end leave at once).  First version: a linear grid and a search of the run table for the unit's run -- one
  // scalar load and a wait per table entry, ~30 entries for a frame recorded by the walk pool: half of the launch's 4.5 us.
  const UploadRun R = T.run[blockIdx.y];
  const uint32_t at = blockIdx.x * 1024u;
  if (at >= R.bytes) return;
  if (R.kind == 1u) {  // BinRecs in 8-byte units (24 bytes each: a piece starts 8-byte aligned)
    const uint32_t ush = 6u + T.binbox_shift;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t o = at + (lane + 64u * h) * 8u;
      if (o >= R.bytes) continue;
      const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const uint8_t*>(R.src) + o);
      *reinterpret_cast<uint2*>(dst + R.dst_off + o) = v;
      const uint32_t rel = R.dst_off - T.bins_off + o;  // byte offset in the BinRec array
      if (rel % 24u != 0u) continue;
      const uint32_t draw = rel / 24u;
      const int x0 = (int)(int16_t)(v.x & 0xffffu), y0 = (int)(int16_t)(v.x >> 16), x1 = (int)(int16_t)(v.y & 0xffffu), y1 = (int)(int16_t)(v.y >> 16);
      uint32_t q = 0x7f7f7f7fu;  // x0 = y0 = 127, x1 = y1 = 0: never hits
      if (x1 > x0 && y1 > y0)
        q = (uint32_t)(x0 >> ush) | ((uint32_t)(y0 >> ush) << 8) | ((127u - (uint32_t)((x1 - 1) >> ush)) << 16) | ((127u - (uint32_t)((y1 - 1) >> ush)) << 24);
      uint32_t* box = reinterpret_cast<uint32_t*>(dst + T.box_off);
      box[draw] = q;
      if (draw + 1u == T.n_draws)  // the array is read four draws at a time: pad the last group
        for (uint32_t k = draw + 1u; (k & 3u) != 0u; k++) box[k] = 0x7f7f7f7fu;
    }
    return;
  }
  const uint32_t o = at + lane * 16u;
  if (o >= R.bytes) return;
  uint4 v = *reinterpret_cast<const uint4*>(static_cast<const uint8_t*>(R.src) + o);
  if (R.kind == 2u && (o & 127u) == 0u && (v.x & F_GENERAL)) v.y += R.ext_add;  // a record's first 16 bytes: op_mode, ext
  *reinterpret_cast<uint4*>(dst + R.dst_off + o) = v;
}
void launch_upload_frame(hipStream_t s, void* dst, const UploadTable& T) {
  if (T.copy_units == 0) return;
  uint32_t widest = 1;
  for (uint32_t r = 0; r < T.n_runs; r++) widest = std::max(widest, (T.run[r].bytes + 1023u) / 1024u);
  hipLaunchKernelGGL(k_upload_frame, dim3(widest, T.n_runs), dim3(64), 0, s, static_cast<uint8_t*>(dst), T);
}
void launch_fill(hipStream_t s, uint32_t* p, uint32_t v, size_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_fill_u32, dim3(1024), dim3(256), 0, s, p, v, n);
}

#if FDH_EDGE_CHECK
void debug_edge_bad(unsigned int* n, unsigned int* out, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(n, HIP_SYMBOL(g_edge_bad_n), sizeof(unsigned int));
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_edge_bad), sizeof(unsigned int) * 4096 * 8);
  if (reset) { const unsigned int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_edge_bad_n), &z, sizeof z); }
}
#endif
#if FDH_MX_CHECK
void debug_mx_bad(unsigned int* n, unsigned int* out, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(n, HIP_SYMBOL(g_mx_bad_n), sizeof(unsigned int));
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mx_bad), sizeof(unsigned int) * 4096 * 8);
  if (reset) { const unsigned int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mx_bad_n), &z, sizeof z); }
}
#endif
#if FDH_STATS
void debug_wave_times(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), sizeof(unsigned long long) * 16 * 65536);
  void* p = nullptr;  // cleared after every read: the next read then holds exactly the launches in between
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wave_times)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 16 * 65536);
  (void)hipDeviceSynchronize();  // (the memset is asynchronous, the contexts' streams do not wait for the null stream: without this it clears rows of the NEXT launch)
}
void debug_counters(unsigned long long out[128], bool reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_counters), 128 * sizeof(unsigned long long));
  if (reset) { unsigned long long z[128] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_counters), z, sizeof z); }
}
#endif

#else  // FDH_TU 1: the one launcher of this unit
template <int kPaths, bool kFull>
static void launch_uniform2(hipStream_t s, hipEvent_t e0, hipEvent_t e1, int grid, size_t lds, const DrawRec* draws, const QuadExt* exts, const CompositeParams& P) {
  if (kPaths == 4 && kFull && P.deep_k8 > 0 && P.order && P.order_next) {  // a frame with deep bins: the launch of four-wave workgroups
    const int bins8 = (P.bin_nx * P.bin_ny + 7) / 8;
    const dim3 g(8 + P.deep_k8 * 16 + 8 * bins8 * kWgsPerBin);
    if (e0) hipExtLaunchKernelGGL((k_composite_deep<1>), g, dim3(256), sizeof(uint32_t) * kDeepLdsDwords, s, e0, e1, 0, draws, exts, P);
    else hipLaunchKernelGGL((k_composite_deep<1>), g, dim3(256), sizeof(uint32_t) * kDeepLdsDwords, s, draws, exts, P);
    return;
  }
  if (e0) hipExtLaunchKernelGGL((k_composite_tiles<kPaths, kFull>), dim3(grid), dim3(64), lds, s, e0, e1, 0, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P);
  else hipLaunchKernelGGL((k_composite_tiles<kPaths, kFull>), dim3(grid), dim3(64), lds, s, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P);
}
template <int kPaths>
static void launch_uniform(hipStream_t s, hipEvent_t e0, hipEvent_t e1, int grid, size_t lds, const DrawRec* draws, const QuadExt* exts, const CompositeParams& P) {
  if (P.load_fb == 0) launch_uniform2<kPaths, true>(s, e0, e1, grid, lds, draws, exts, P);  // the launch that starts a frame
  else launch_uniform2<kPaths, false>(s, e0, e1, grid, lds, draws, exts, P);
}
void launch_composite_uniform(hipStream_t s, hipEvent_t e0, hipEvent_t e1, int grid, size_t lds, const DrawRec* draws, const QuadExt* exts, const CompositeParams& P, int paths) {
  if (paths == 2) launch_uniform<2>(s, e0, e1, grid, lds, draws, exts, P);
  else if (paths == 0) launch_uniform<0>(s, e0, e1, grid, lds, draws, exts, P);
  else launch_uniform<4>(s, e0, e1, grid, lds, draws, exts, P);
}
#endif  // FDH_TU
}  // namespace fdh
