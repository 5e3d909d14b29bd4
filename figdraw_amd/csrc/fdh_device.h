// fdh_device.h -- what the gfx950 kernels of libfigdraw_hip share: the device helpers that restate the reference's shader functions
// (src/figdraw/opengl/glsl/atlas.frag: sdRoundedBox, sdEllipticalRoundedBox, shadowProfile, sdBezier, the atlas sampler), the strip masks and
// the two ends of a list entry's making (k_bin_upload.hip, k_composite.hip), a fragment of a quad and the blend of one draw into a texel
// (k_composite.hip and the blur passes that composite their node's quad), the profile-mode launch macro, and the instrumented builds'
// counters.  One translation unit per kernel family includes it (csrc/Makefile); `make variant SINGLE=1` compiles them all as one
// (fdh_kernels_all.hip) so that the device-side counters are one set.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>

#include "fdh_kernels.h"

#include <cstddef>
#include <cstdlib>

namespace fdh {

// Build switches of the instrumented builds (never in the product build).
#ifndef FDH_STATS
#define FDH_STATS 0  // `make stats`: per-strip draw classification counters (tools/strip_stats.py)
#endif
#ifndef FDH_TIMING
#define FDH_TIMING 0  // `make variant SINGLE=1 DEFS="-DFDH_STATS=1 -DFDH_TIMING=1"`: per-wave phase times (shader cycles) instead of counts
#endif
#ifndef FDH_SPLIT_UNIFORM
#define FDH_SPLIT_UNIFORM 0  // the Makefile's product and variant builds set 1: k_composite_tiles<0|2|4> live in a unit of their own (k_composite.hip)
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

// Per-kernel timing (fdh_profile): with a pair of events set, the next launch goes through hipExtLaunchKernelGGL, which stamps
// them from the dispatch's own start / end timestamps -- the kernel's execution time as rocprofv3 reports it.  (Events
// recorded around a launch also count the gap to the neighbouring dispatches: +2..5 us on a 20 us kernel.)
// (one set per host thread, shared by the units: set_launch_events / launch_events_used are in k_bin_upload.hip)
inline thread_local hipEvent_t t_prof_start = nullptr, t_prof_stop = nullptr;
inline thread_local bool t_prof_used = false;
#define FDH_LAUNCH(kern, grid, block, lds, stream, ...)                                                                  \
  do {                                                                                                                   \
    if (t_prof_start) { hipExtLaunchKernelGGL(kern, grid, block, lds, stream, t_prof_start, t_prof_stop, 0, __VA_ARGS__); t_prof_used = true; } \
    else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                                \
  } while (0)
// FDH_FORCE_BLUR_PATH=1|2|3 (a test hook, tests/test_hip_parity.py): every blur on the 2-outputs-per-thread passes / the
// many-outputs passes / the matrix-pipe passes, whatever the region size
inline int blur_forced_path() { static const int v = [] { const char* e = std::getenv("FDH_FORCE_BLUR_PATH"); return e ? std::atoi(e) : 0; }(); return v; }

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
  return m & (0u - (uint32_t)(x1 > x0 && y1 > y0));
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
// (written without branches, for the same reason -- and without `?:` on anything but flags: the compiler turned selects between
// computed masks back into branches in one build of the compositor; bit masks it cannot)
__device__ __forceinline__ void bin_entry_tail(const BinRec& r, int x0, int y0, bool& hit, uint32_t& strips) {
  const uint32_t m_core = 0u - (uint32_t)((r.flags & BR_HAS_CORE) != 0u), m_removed = 0u - (uint32_t)((r.flags & BR_CORE_REMOVED) != 0u),
                 m_exact = 0u - (uint32_t)((r.flags & BR_BOX_EXACT) != 0u);  // all ones / zero
  const uint32_t core = strip_mask_inside(r.ix0 - x0, r.iy0 - y0, r.ix1 - x0, r.iy1 - y0) & strips & m_core;
  // alpha == 0 on the core (stroke interior) or too small to change an 8-bit channel (deep inside an inner shadow): those strips leave
  // the entry, and the entry goes with them if none is left; any other core is marked in the high half
  const uint32_t s1 = (strips & ~(core & m_removed)) | ((core & ~m_removed) << 16);
  const uint32_t m_gone = m_core & m_removed & (0u - (uint32_t)((s1 & 0xffffu) == 0u));
  // edge strips wholly inside the quad's pixel bounds: state (0, 1) -- fdh_types.h, BR_BOX_EXACT (draws with a core only: as round 5 had it)
  const BBox b = r.box;
  const uint32_t inq = strip_mask_inside(b.x0 - x0, b.y0 - y0, b.x1 - x0, b.y1 - y0) & s1 & ~(s1 >> 16) & 0xffffu & m_core & m_exact & ~m_gone;
  strips = (s1 & ~inq) | (inq << 16);
  hit = hit && m_gone == 0u;
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

// RGBA8 <-> two float pairs (r, g) / (b, a) holding 0..255 (the blur passes; the V passes' fused composite)
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
}  // namespace fdh
