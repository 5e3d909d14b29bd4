// fdh_types.h -- records shared by the host-side context and the gfx950 kernels.
#pragma once
#include <stdint.h>

// How the matrix-pipe blur passes hold a tap: 0 = ONE f16 at scale 2^10, rounding error carried to the next tap out (the product build,
// round 5: quantise_taps_f16 in fdh_context.cpp); 1 = hi + lo, 22 bits, two MFMAs per operand (rounds 2 - 4; `make variant
// DEFS=-DFDH_MX_LO=1`, what tools/blur_weights_pin.py measures the product build against).  ONE definition: the host side builds the weight
// fragments and the kernels multiply them, an object built without the define would silently drop the lo halves.
#ifndef FDH_MX_LO
#define FDH_MX_LO 0
#endif

namespace fdh {

// Tile geometry: the shading unit is the 8x8 pixel tile; one wavefront (64 lanes) shades four of them side by
// side in lock-step (a 32x8 strip, lane <-> 4 adjacent pixels of one row), so a row of the strip is one 128-byte
// line of the RGBA8 surface.  A workgroup stacks 4 waves (32x32 pixels); a coarse bin is 64x64 pixels = 4 workgroups.
constexpr int kTile = 8;
constexpr int kTileW = 4 * kTile;  // a wavefront covers four side-by-side 8x8 sub-tiles: 32 x 8 pixels, 4 pixels per lane
constexpr int kTileH = kTile;
constexpr int kWavesPerWg = 4;
constexpr int kWgW = kTileW;                // 32
constexpr int kWgH = kTileH * kWavesPerWg;  // 32: the four waves of a workgroup are stacked vertically
constexpr int kBin = 64;
constexpr int kWgsPerBin = (kBin / kWgW) * (kBin / kWgH);  // 4
constexpr int kMaskDepth = 16;                             // per-lane clip stack levels kept in LDS (4 KB per wavefront); deeper levels spill to a global plane
// a bin whose list holds at least this many draws gets quarter-strip waves in the next frames' full-frame launch (k_composite_tiles;
// FDH_DEEP_MIN overrides, 0 turns the quarter strips off)
constexpr int kDeepMinDefault = 24;
constexpr int kDeepMinInFlight = 40;      // ... when the device has other contexts (frames in flight beside this one's): fdh_context.cpp
constexpr int kDeepStripMinDefault = 16;  // ... and of such a bin's strips, those with at least this many draws to shade (FDH_DEEP_STRIP_MIN)
constexpr int kMaxMips = 14;
constexpr int kMaxBlurTaps = 36;

// op codes (DrawRec::op_mode bits 12..15)
enum : uint32_t { OP_DRAW = 0, OP_MASK_PUSH = 1, OP_MASK_POP = 2, OP_RMASK_BEGIN = 3, OP_RMASK_END = 4 };

// DrawRec::op_mode layout
//   bits 0..7   SdfMode (figbackend.nim:36-52)
//   bit  8      elliptical radii (SdfEllipticalRadiiFlag, glcontext.nim:992)
//   bits 9..11  fill mode 0..4 (glcontext.nim:987-991)
//   bits 12..15 op
//   bit  16     general quad: `ext` indexes a QuadExt (rotated / mirrored / skewed transforms)
//   bit  17     all four vertex colours equal
//   bit  18     mode 17 samples the live framebuffer (blur radius <= 0.5: the snapshot is unblurred)
//   bit  19     subpixel positioning enabled (mode 0)
constexpr uint32_t F_ELLIP = 1u << 8;
constexpr uint32_t F_GENERAL = 1u << 16;
constexpr uint32_t F_SOLID = 1u << 17;
constexpr uint32_t F_SELF_BACKDROP = 1u << 18;
constexpr uint32_t F_SUBPIXEL = 1u << 19;
// bit 20: an axis-aligned mode-0 quad that maps atlas texels to pixels 1:1 at integer offsets (what renderText emits for every
// glyph, figrender.nim:456-496): pixel (x, y) shows texel (x + tdx, y + tdy) -- DrawRec::ext / _pad hold tdx / tdy as int32.  The
// bilinear fractions are 0 up to float noise (GL's fixed-point sampler snaps them to 0): the compositor fetches a lane's four
// texels as one 16-byte run instead of sixteen gathers.
constexpr uint32_t F_TEXEL_1TO1 = 1u << 20;
// bit 21 (with F_GENERAL): the quad's edge functions fit 32-bit arithmetic at every pixel a strip of the frame can hold -- |a|, |b| <
// 2^23 and |a| (2 W + 129) + |b| (2 H + 129) + |c| < 2^31 -- so the compositor's 4-wide path for rotated SDF quads may evaluate them
// with a scalar base per strip and v_mad_i32_i24 per lane
constexpr uint32_t F_EDGE32 = 1u << 21;

// One BackendContext draw call, 128 bytes, read with wave-uniform (scalar) loads.
struct alignas(16) DrawRec {
  uint32_t op_mode;
  uint32_t ext;        // QuadExt index when F_GENERAL
  float ox, oy;        // ceil'd quad origin (axis-aligned path)      | RMASK: matX.xy
  float inv_w, inv_h;  // 1 / ceil'd quad extent                      | RMASK: matX.z, matY.x
  float p0, p1;        // sdfParams.xy (quad half extents; msdf: atlas size, stroke width) | RMASK: centre
  float p2, p3;        // sdfParams.zw (shape half extents; inset: shadow offset)           | RMASK: half extents
  float f0, f1;        // sdfFactors (factor, spread | midPos | msdf pxRange, threshold)     | RMASK: matY.yz
  float r[4];          // sdfRadii (TR,BR,TL,BL) -- atlas modes: uvAt.xy, uvTo.xy
  uint32_t col[4];     // vertex colours BL,BR,TR,TL, RGBA8 little-endian (r in the low byte)
  uint32_t mid, stop;  // 3-stop gradient colours
  float aa;            // SDF AA factor in effect for this draw
  float aux;           // mode 0: subpixel shift ; atlas modes: texture LOD (log2 rho) in aux2
  float aux2;
  int16_t bx0, by0, bx1, by1;  // covered pixel bounds, clipped to the frame: [bx0,bx1) x [by0,by1)
  // Saturated core of an axis-aligned SDF draw, in (unclipped) pixel bounds, empty when unknown: every pixel centre in
  // [ix0,ix1) x [iy0,iy1) has coverage alpha == 1 (fills, clip pushes, drop-shadow bodies, blur composites) or, for
  // the annular stroke modes 11/12, alpha == 0 -- or, for an inner shadow (mode 9), alpha too small to change any 8-bit
  // channel.  Conservative by a pixel; lets a strip be classified by a bit test.
  int16_t ix0, iy0, ix1, iy1;
  // axis-aligned SDF quads: local-frame steps per pixel, kx = 2 p0 inv_w, ky = 2 p1 inv_h (atlas.frag:252-262: p = (uv - 0.5) * 2 *
  // quadHalfExtents): a lane's pixels 1..3 are pixel 0's local x plus multiples of kx
  float kx, ky;
  uint32_t _pad;
};
static_assert(sizeof(DrawRec) == 128, "DrawRec must be 128 bytes");

// Extension for non-axis-aligned quads: the two triangles (3,0,1) and (2,3,1) of glcontext.nim:418-429
// over the four per-vertex ceil'd positions.  Edge functions are exact integers in half-pixel units.
struct alignas(16) QuadExt {
  struct Edge { int32_t a, b; int64_t c; };  // E = a*X + b*Y + c with X = 2*px+1, Y = 2*py+1 (64-bit accumulate)
  Edge e[2][3];            // [tri][edge]; edge k is opposite vertex k; interior is E >= 0
  float inv_sum[2];        // 1 / (E0+E1+E2)  (0 => degenerate triangle)
  uint32_t own;            // bit (tri*3+edge): edge owns pixel centres lying exactly on it (top-left rule)
  float fw_u[2], fw_v[2];  // |du/dx|+|du/dy|, |dv/dx|+|dv/dy| per triangle (msdf fwidth)
  float lod[2];            // log2(rho) per triangle (atlas mip selection: GL derives rho from per-fragment derivatives,
                           // which are constant on a triangle)
  // The saturated core of a rotated SDF draw (BR_GENERAL | BR_HAS_CORE), for k_bin_draws: core[] = {xl, xr, yb, yt}, a rectangle in
  // the shader's local frame (y up) on which the coverage term is saturated -- what DrawRec::ix0..iy1 is for an upright quad --,
  // already shrunk by the slack that absorbs float rounding; lm[t] = the affine map of triangle t from (X, Y) = (2 px + 1, 2 py + 1)
  // to that frame: lx = lm[t][0] X + lm[t][1] Y + lm[t][2], ly = lm[t][3] X + lm[t][4] Y + lm[t][5].  A strip that lies inside the
  // quad and whose four corner pixels map into core[] under BOTH triangles' maps is a core strip (the maps are affine, the
  // rectangle convex: every pixel of the strip then maps into it whichever triangle it belongs to).
  float _pad[3];
  float core[4];  // (byte 144: core and lm are read as 16-byte pieces)
  float lm[2][6];
};
static_assert(sizeof(QuadExt) == 208 && offsetof(QuadExt, core) == 144 && offsetof(QuadExt, lm) == 160, "QuadExt: 208 bytes, core at 144, lm at 160");

struct alignas(8) BBox { int16_t x0, y0, x1, y1; };

// What k_bin_draws needs of a draw, in one 24-byte piece (built on the host at submit): the clipped pixel bounds, the
// saturated core, and the parts of a list entry that do not depend on the bin -- flags and path code in their entry
// positions (bits 25..31: LE_SHARE, LE_PATH, LE_OPAQUE, LE_PLAIN), bit 0: the draw has a core worth testing (axis-aligned SDF draw or
// clip push), bit 1: its core strips are REMOVED from the entry (stroke interiors, deep inside an inner shadow).  Before, a
// hit walked bounds -> mode word -> core -> colours: three dependent round trips per batch of hits.
struct alignas(8) BinRec { BBox box; int16_t ix0, iy0, ix1, iy1; uint32_t flags; uint32_t pad; };
static_assert(sizeof(BinRec) == 24, "BinRec must be 24 bytes");
constexpr uint32_t BR_HAS_CORE = 1u, BR_CORE_REMOVED = 2u;
// bit 2: a rotated / skewed SDF draw (F_GENERAL | F_EDGE32, OP_DRAW): its pixel bounds are the quad's bounding box; k_bin_draws drops the
// strips that lie outside one of the quad's four outer edges (half of the box of a quad rotated by 30 degrees)
constexpr uint32_t BR_GENERAL = 4u;
// bit 3: a quadratic-bezier stroke on an upright quad (modes 18 - 20): its bounds are the span's bounding box; k_bin_draws drops the
// strips farther from the curve's chord-aligned box than any pixel with coverage can be (the same test the compositor applies per strip)
constexpr uint32_t BR_CURVE = 8u;
// bit 4: an upright SDF draw whose bounds ARE its quad's pixel bounds (DrawRec::bx0..by1): a strip that lies inside them is covered by
// the quad at every pixel, and k_bin_draws says so in the entry (see the strip states below) -- the compositor then skips the
// per-pixel quad test (eight compares / selects per strip-draw on the packed edge paths, sixteen on the generic one)
constexpr uint32_t BR_BOX_EXACT = 16u;
// A list entry's second word holds two bits per strip of the bin, bit s and bit 16 + s:  (1, 0) touched;  (1, 1) touched and inside the
// draw's saturated core;  (0, 1) touched, not core, and wholly inside the quad (BR_BOX_EXACT draws only);  (0, 0) not touched.
constexpr uint32_t LE_PLAIN = 1u << 31;  // axis-aligned SDF draw with ONE colour: on its core strips it is a uniform blend
constexpr uint32_t LE_OPAQUE = 1u << 30;  // a fill whose source alpha is 255 everywhere: on its core strips it REPLACES the surface
// bits 26..29: which straight-line shading path the draw's EDGE strips can take, decided on the host so that the
// compositor branches on the list entry (already in an SGPR) and fetches the record once, instead of fetching the mode word, waiting,
// decoding it and only then fetching the rest.  0: the general path; 1..4: one colour, no gradient, OP_DRAW, mode 3 / 7 / 9 /
// 12 with circular corners; 5..8: the same with elliptical corners
constexpr int LE_PATH_SHIFT = 26;
// bit 25: the NEXT draw (index + 1) is drawn over the same quad with the same shape -- a node's fill, then its stroke, then its
// inner shadows (figrender.nim:806-873, 716-744) -- so on a strip where both take a packed edge path the compositor evaluates the
// distance field once for the run
constexpr uint32_t LE_SHARE = 1u << 25;
constexpr uint32_t LE_INDEX = LE_SHARE - 1u;

constexpr int kMaxBlurReach = 66;
constexpr int kBlurPad = 15;  // >= (largest outputs-per-thread) - 1
struct BlurTaps {  // merged FIR of blur.frag:19-29 for one radius: out = sum coef[k] * src[x + off[k]]
  int n;
  int reach;  // max |off|
  int off[kMaxBlurTaps];
  float coef[kMaxBlurTaps];
  // the same filter as a dense coefficient array over offsets -reach..+reach, padded with kBlurPad zeros on both
  // sides: dense[kBlurPad + reach + off].  Used by the 8-outputs-per-thread kernels (every staged texel is
  // unpacked once and feeds up to eight accumulators).
  float dense[2 * kMaxBlurReach + 1 + 2 * kBlurPad + 16];  // + slack read (never used) by the pipelined loops
};

// Matrix-pipe blur passes (k_blur_mx): a block of 32 outputs reads NK k-steps of 16 texels; they must cover its
// 32 + 2 reach window (+ up to 3 texels of alignment slack for the horizontal pass)
constexpr int kMxMaxNK = 11;  // reach 66 = the widest filter (radius clamp 64)
inline int mx_nk(int reach, bool vertical) { return (32 + 2 * reach + (vertical ? 0 : 3) + 15) / 16; }
inline int mx_delta(int reach, bool vertical) { return vertical ? 0 : ((-reach) % 4 + 4) % 4; }  // window start -> 16-byte boundary
// Row of a 16-texel k-step that element e (0..7) of lane group g (0, 1) carries in the matrix-pipe operands.  Horizontal pass: 8 g + e.
// Vertical pass: (e & 3) + 8 (e >> 2) + 4 g -- register 8 s + e of a 32 x 32 f32 accumulator tile holds row 16 s + that of the tile, so
// an accumulator tile IS two k-steps of the next product's operand in this order.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int mx_krow(int g, int e, bool vertical) { return vertical ? (e & 3) + 8 * (e >> 2) + 4 * g : 8 * g + e; }
constexpr size_t mx_table_bytes(int nk) { return (size_t)nk * 2 * 64 * 16; }  // [k-step][hi, lo][lane] x 8 halves

struct AtlasView {
  const uint32_t* level[kMaxMips];
  int size;
  int n_levels;
};

}  // namespace fdh
