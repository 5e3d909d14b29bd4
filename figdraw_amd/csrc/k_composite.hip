// k_composite.hip -- the fused tile compositor: atlas.frag's main() (glsl/atlas.frag:252-405), the fixed-function blend and the RGBA8
// re-quantisation per draw (utils/glutils.nim:150-154), clip and rect masks, as ONE launch per phase of a frame.
//
//   k_composite_tiles  one single-wave workgroup per 32x8-pixel strip (four 8x8 tiles side by side, four pixels per lane)
//                      walks the bin's list in painter's order keeping RGBA in registers, re-quantising to RGBA8 after every
//                      draw like the GL framebuffer does, and stores the strip once.  Five builds picked per phase: <4> SDF draws
//                      only, <0> + clip masks, <2> + the 4-wide atlas path, <8> + rotated SDF quads, <3> + everything (one pixel slot at a time)
//   k_composite_deep   the full-frame launch of a frame with deep lists (round 6): workgroups of four waves, the strips of the deepest
//                      bins shaded by three waves and blended by a fourth
//
// Two translation units from this one file (csrc/Makefile).  FDH_TU 0 (this file as it is): the builds <3> and <8> and the launcher.
// FDH_TU 1 (k_composite_uniform.hip): <0>, <2>, <4>, k_composite_deep and their launcher only, compiled with
// -structurizecfg-skip-uniform-regions.  hipcc structurizes EVERY region of a kernel's control flow, uniform branches
// included; in the draw loop that turns each wave-uniform branch into a predicate in an SGPR pair (s_cselect_b64 / s_and_b64
// / s_cbranch_vccnz where one s_cbranch_scc would do) and keeps the texels that merge at the loop latch out of the
// registers they came from (eight v_mov_b64 per draw).  With the switch, a region whose branches are all wave-uniform is
// left as the branches it is: phase 0 of the bench frame 38 -> 34 us, and with the registers that freed, six waves per SIMD.
// The switch is NOT safe for code that nests uniform branches inside divergent ones: with it the slot path <3> and the blur
// passes come out wrong (measured: 23 of the 57 GPU parity tests fail), so it stays confined to kernels whose draw loop
// nest holds no divergent branch at all -- tools/lint_isa.py checks exactly that on the built code object after every
// build, and the note at shadow_profile() says how the source keeps it so.  Instrumented builds (FDH_STATS, FDH_TIMING:
// device-side counters) are single-unit builds (`make variant SINGLE=1`).
#include "fdh_device.h"

#ifndef FDH_TU
#define FDH_TU 0
#endif

namespace fdh {

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
// The builds without the one-pixel-slot path (phases made only of axis-aligned SDF draws, clips, axis-aligned atlas quads:
// no rotated quads, no bezier strokes, no rect-mask setup) need no scratch: 80 - 96 VGPRs, five or six waves per SIMD.
constexpr int kFastWaves = 5;
constexpr int kAtlasWaves = 5;  // waves per SIMD of the atlas build <2>
constexpr int kSlowWaves = 4;  // waves per SIMD of the build with every path <3>
// (<19> = <3> at three: a phase with atlas quads off the 4-wide path -- rotated glyph rows, minified images -- holds sixteen to thirty-two
// texels per lane and spills 47 registers at 128; config 11 243 -> 223 us at 168 registers, while the curves of config 10, all arithmetic,
// lose 7 % of their waves' overlap there (136 -> 146 us): the phase's content picks the form, fdh_record.cpp's has_slow_atlas)
constexpr int kSlowAtlasWaves = 3;
constexpr int kRotWaves = 4;  // waves per SIMD of the rotated-quad build <8>
constexpr int kUniformWaves = 6;  // waves per SIMD of the no-clip build <4>: 80 VGPRs, no spills
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

template <int kPaths, bool kFull, int kRole, bool kDirect = false>
__device__ __forceinline__ void composite_strip(const CompositeParams& P, const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts,
                                                uint32_t* composite_lds, const int bin, const int sidx, const int sbit, const int tx0, const int ty0,
                                                const int lane, const int shader_id);

template <int kPaths, bool kFull, bool kDirect = false>
__global__ __launch_bounds__(64, (kPaths & 16) ? kSlowAtlasWaves : (kPaths & 1) ? kSlowWaves : kPaths == 4 ? kUniformWaves : (kPaths & 2) ? kAtlasWaves : (kPaths & 8) ? kRotWaves : kFastWaves) void k_composite_tiles(
    // the sixteen dwords a wave needs before anything else, as leading scalar arguments: with kernel-argument preloading
    // (-amdgpu-kernarg-preload-count, csrc/Makefile) they arrive in SGPRs with the wave instead of through a first s_load
    const int* __restrict__ a_order, int* __restrict__ a_order_next, const uint32_t* __restrict__ a_counts, const uint2* __restrict__ a_lists,
    int a_bin_x0, int a_bin_y0, int a_bin_nx, int a_bin_ny, int a_bins_x, int a_stride, int a_row_lo, int a_row_hi,
    const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts, const CompositeParams P_in) {
  // (a copy, so that the preloaded arguments can take their fields' places -- and the atlas read through the ORIGINAL: a mip level is picked by
  // a run-time index, and an indexed member of a struct the kernel writes to keeps the WHOLE struct in scratch memory: the <3> build read
  // every field of its 312-byte copy from there, tools/kernel_regs.py)
  CompositeParams P = P_in;
  [[maybe_unused]] const AtlasView& atlas_view = P_in.atlas;
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
    order_bins_wave(P.counts, P.order_next, P.bin_nx, P.bin_nx * P.bin_ny, &mask_stack[0][0][0], threadIdx.x, (int)blockIdx.x, kPaths == 4 ? P.deep_min : 0, kPaths == 4 ? P.deep_out : nullptr);
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
  // The strip's shading, k_composite_strip.inc: as a CALL in the uniform-regions unit and as TEXT in the other.  Same source either way, but
  // the register allocator lands elsewhere: called, the bench frame's <4, true> has no spilled scalar register (as text: nine), and as text the
  // <3> build keeps round 5's size and scratch (called: 9 KB more code than the instruction cache likes, config 10 19 % slower) --
  // same-box A/B against round 5's library in profiles/r06_ab_r05g.txt.
#if FDH_TU == 1
  composite_strip<kPaths, kFull, 0, kDirect>(P, draws, exts, composite_lds, bin, sidx, sbit, tx0, ty0, lane, 0);
#else
  constexpr int kRole = 0;
  constexpr int shader_id = 0;
  {  // (a scope of its own: the body names some of what the mapping above has named)
#include "k_composite_strip.inc"
  }
#endif
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
__global__ __launch_bounds__(256, kUniformWaves) void k_composite_deep(const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts, CompositeParams P) {
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
  // (tried: s_setprio 3 / 2 for a deep strip's blender / shaders -- the launch takes the same time, profiles/r06_deep_strips.txt)
  if (wg_wave == 0) composite_strip<4, true, 1>(P, draws, exts, composite_lds, bin, sidx, sbit, tx0, ty0, lane, 0);
  else composite_strip<4, true, 2>(P, draws, exts, composite_lds, bin, sidx, sbit, tx0, ty0, lane, wg_wave - 1);
}

template <int kPaths, bool kFull, int kRole, bool kDirect>
__device__ __forceinline__ void composite_strip(const CompositeParams& P, const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts,
                                                uint32_t* composite_lds, const int bin, const int sidx, const int sbit, const int tx0, const int ty0,
                                                const int lane, const int shader_id) {
  [[maybe_unused]] const AtlasView& atlas_view = P.atlas;
#include "k_composite_strip.inc"
}

#if FDH_TU == 0
void launch_composite_uniform(hipStream_t s, hipEvent_t e0, hipEvent_t e1, int grid, size_t lds, const DrawRec* draws, const QuadExt* exts, const CompositeParams& P, int paths);  // FDH_TU 1
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
  if (force == 3) { P.has_slow = 1; P.has_slow_atlas = 0; }  // (<3> itself, also where the phase would take its 168-register form)
  if (force == 19) { P.has_slow = 1; P.has_slow_atlas = 1; }
  if (force == 8) P.has_rot = 1;
  if (force == 2) P.has_atlas = 1;
  if (force == 1) P.has_masks = 1;  // (the build with mask registers and the 4-KB stack, even where no clip is open)
  const size_t lds = (P.has_masks ? sizeof(uint32_t) * kMaskDepth * 64 : sizeof(uint32_t) * 256) +
                     ((P.has_atlas || P.has_slow || (P.has_rot && P.has_atlas)) ? sizeof(uint32_t) * kWinRows * kWinStride : 0);  // + the texel window of the atlas path
  const bool full = P.load_fb == 0;  // the launch that starts a frame (k_composite_tiles<., true>)
  // deep strips (k_composite_deep): only beside the no-clip build's full-frame launch, and only with the order at hand
  if (!(full && P.order && P.order_next && !P.has_slow && !P.has_rot && !P.has_atlas && !P.has_masks)) P.deep_k8 = 0;
  P.deep_k8 = std::min(P.deep_k8 & ~7, 8 * bins8);
#define FDH_COMPOSITE_DIRECT(paths) \
  do { if (full) FDH_LAUNCH((k_composite_tiles<paths, true, true>), dim3(grid), blk, lds, s, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P); \
       else FDH_LAUNCH((k_composite_tiles<paths, false, true>), dim3(grid), blk, lds, s, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P); } while (0)
#define FDH_COMPOSITE(paths) \
  do { if (full) FDH_LAUNCH((k_composite_tiles<paths, true>), dim3(grid), blk, lds, s, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P); \
       else FDH_LAUNCH((k_composite_tiles<paths, false>), dim3(grid), blk, lds, s, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P); } while (0)
#if FDH_SPLIT_UNIFORM
  if (P.has_slow || (P.has_rot && P.has_atlas)) { if (P.has_slow_atlas) FDH_COMPOSITE(19); else FDH_COMPOSITE(3); }
  else if (P.has_rot) FDH_COMPOSITE(8);
  else { launch_composite_uniform(s, t_prof_start, t_prof_stop, grid, lds, draws, exts, P, P.has_atlas ? 2 : P.has_masks ? 0 : 4); if (t_prof_start) t_prof_used = true; }
#else
  if (P.has_slow || (P.has_rot && P.has_atlas)) { if (P.has_slow_atlas) FDH_COMPOSITE(19); else FDH_COMPOSITE(3); }
  else if (P.has_rot) FDH_COMPOSITE(8);
  else if (P.direct) { if (P.has_atlas) FDH_COMPOSITE_DIRECT(2); else if (!P.has_masks) FDH_COMPOSITE_DIRECT(4); else FDH_COMPOSITE_DIRECT(0); }
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
#undef FDH_COMPOSITE_DIRECT
}
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
  if (P.direct) {  // a frame without a bin launch: the builds whose waves make their list entries themselves
    if (e0) hipExtLaunchKernelGGL((k_composite_tiles<kPaths, kFull, true>), dim3(grid), dim3(64), lds, s, e0, e1, 0, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P);
    else hipLaunchKernelGGL((k_composite_tiles<kPaths, kFull, true>), dim3(grid), dim3(64), lds, s, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P);
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
