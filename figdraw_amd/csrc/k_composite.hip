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
__global__ __launch_bounds__(64, (kPaths & 1) ? kSlowWaves : kPaths == 4 ? kUniformWaves : (kPaths & 2) ? kAtlasWaves : (kPaths & 8) ? kRotWaves : kFastWaves) void k_composite_tiles(
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
  composite_strip<kPaths, kFull, 0, kDirect>(P, draws, exts, composite_lds, bin, sidx, sbit, tx0, ty0, lane, 0);
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
  // (a build of its own, kDirect: as a run-time branch of every build the entry code cost the bench frame's launch sixteen spilled
  // scalar registers and 1.6 us of its 26 -- same-box A/B against round 5's library, profiles/r06_ab_r05g.txt)
  constexpr bool direct = kDirect;
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
      if (kSlow && (om & F_GENERAL) != 0u && (om & F_EDGE32) != 0u && atlas_mode && op == OP_DRAW) {
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
      if (kSlow && mode >= 18u && mode <= 20u && ((om & F_GENERAL) == 0u || (om & F_EDGE32) != 0u) && op == OP_DRAW) {
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
        continue;
      }
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
#define FDH_COMPOSITE_DIRECT(paths) \
  do { if (full) FDH_LAUNCH((k_composite_tiles<paths, true, true>), dim3(grid), blk, lds, s, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P); \
       else FDH_LAUNCH((k_composite_tiles<paths, false, true>), dim3(grid), blk, lds, s, P.order, P.order_next, P.counts, P.lists, P.bin_x0, P.bin_y0, P.bin_nx, P.bin_ny, P.bins_x, P.stride, P.row_lo, P.row_hi, draws, exts, P); } while (0)
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
