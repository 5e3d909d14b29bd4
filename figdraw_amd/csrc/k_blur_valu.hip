// k_blur_valu.hip -- the separable backdrop blur of glsl/blur.frag (:11-32: 17 bilinear taps, a fixed FIR over integer offsets) with VALU
// FMAs: k_blur_h / k_blur_v for small regions and pitches the matrix-pipe passes cannot take, k_blur_small (both passes of a small region
// in one kernel), and the launchers that pick between them and the matrix-pipe passes of k_blur_mx.hip.
#include "fdh_device.h"

namespace fdh {
bool launch_blur_mx_pass(bool vertical, hipStream_t s, const BlurParams& P, const DrawRec* draws, const QuadExt* exts);  // k_blur_mx.hip
// ------------------------------------------------------------------ blur (blur.frag:11-32 as a merged FIR)
// blur.frag takes 17 bilinear taps at i*step px.  The step is constant, so tap i has the same bilinear fraction at
// every pixel and the pass is a fixed FIR over integer offsets (BlurTaps, built on the host).  Each thread produces
// FOUR consecutive outputs along the filter direction: every staged texel is unpacked once (4 x v_cvt_f32_ubyte)
// and feeds up to four accumulators (float2 pairs: r, g and b, a), instead of being re-read and re-unpacked per tap.

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
// a third for the composite behind them.  Here a workgroup produces a kSmallTW x kSmallTH tile of the BLURRED SNAPSHOT: it stages the tile's
// (TW + 2 reach) x (TH + 2 reach) source window in LDS (clamp-to-edge), filters its TH + 2 reach rows horizontally into LDS --
// rounded to RGBA8 exactly as the horizontal pass stores its intermediate texture (glcontext.nim:1743-1786) -- and filters
// those vertically.  Same per-output sums in the same order as k_blur_h<2> / k_blur_v<2, .> (fir_outputs<2>): the snapshot is
// the two-pass one bit for bit.  It goes to the backdrop surface, out of place (a tile's neighbours still read the live surface
// around it), and the phase's compositor launch samples it for the mode-17 quad like any other draw.
// 1024 threads per workgroup: the tile's horizontal tasks (radius 18: 832 for the first, 32 x 16 tile) are ONE fir_outputs per thread and
// its vertical tasks one more -- with 256 threads a thread ran 3.25 + 1 of them back to back, each a chain of 38 dependent LDS reads (15 us
// for the 360 x 240 node against 5 + 7 for the two launches).
// The tile is 16 x 24 (round 6; 32 x 16 before).  Wall clocks of every wave (tools/small_blur_times.py) put the kernel's 7.9 us at 2.0 us
// until the window is in LDS, then 5.5 us of ONE CU's arithmetic per tile: 832 horizontal tasks of ~560 instructions (a 16-row tile
// filters 16 + 2 reach = 54 rows: 3.4 x what it keeps) on sixteen waves, then 256 vertical ones.  At 16 x 24 a tile has 496 horizontal
// tasks (eight waves, two per SIMD) for 384 outputs, the 360 x 240 node is 230 tiles for 256 CUs, and the window's rows are at most 64
// texels wide, so a wave stages a row with a lane per column (no division on the way to the first load: -0.7 us).  Same sums per output:
// the frames of every shape tried are equal bit for bit.  Measured (rocprofv3, 300 bench frames, same box): 32 x 16 7.91 us, 16 x 32 7.42,
// 12 x 32 7.10, 16 x 24 7.05 -- with four horizontal outputs per thread 7.44 --, 16 x 24 staged by rows 6.24, that with one vertical
// output per thread 6.27 - 6.57.
constexpr int kSmallTW = 16, kSmallTH = 24, kSmallHOut = 2, kSmallVOut = 2, kSmallThreads = 1024;
__global__ __launch_bounds__(kSmallThreads) void k_blur_small(BlurParams P) {
  extern __shared__ uint32_t small_lds[];
#if FDH_TIMING  // tools/small_blur_times.py: wall clocks of every wave (100 MHz, one counter for the device)
  const unsigned long long W_in = wall_clock64();
  unsigned long long W_asked = 0, W_staged = 0, W_h = 0, W_hb = 0, W_v = 0;
#endif
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
  // (a window row of at most 64 texels -- every filter the 16-wide tile takes: a wave per row, a lane per column, no division by the
  // window's width on the way to the first load; the general form below spent 1.0 of the kernel's 7 us before its loads were out)
  const bool by_rows = in_w <= 64 && rows <= kSmallStage * (kSmallThreads / 64);
  if (by_rows) {
    const int wv = (int)threadIdx.x >> 6, ln = (int)threadIdx.x & 63;
    int x = xs - reach + min(ln, in_w - 1);
    x = x < 0 ? 0 : (x > P.W - 1 ? P.W - 1 : x);
#pragma unroll
    for (int k = 0; k < kSmallStage; k++) {
      int y = ys - reach + min(wv + k * (kSmallThreads / 64), rows - 1);
      y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);  // clamp-to-edge (glcontext.nim:214-215)
      texel[k] = P.src[(size_t)y * P.pitch + x];
    }
  } else {
#pragma unroll
    for (int k = 0; k < kSmallStage; k++) {
      const int i = min((int)threadIdx.x + k * kSmallThreads, n_in - 1);
      const int rr = i / in_w, cc = i - rr * in_w;
      int y = ys - reach + rr, x = xs - reach + cc;
      y = y < 0 ? 0 : (y > P.H - 1 ? P.H - 1 : y);  // clamp-to-edge (glcontext.nim:214-215)
      x = x < 0 ? 0 : (x > P.W - 1 ? P.W - 1 : x);
      texel[k] = P.src[(size_t)y * P.pitch + x];
    }
  }
#if FDH_TIMING
  W_asked = wall_clock64();
#endif
  if (by_rows) {
    const int wv = (int)threadIdx.x >> 6, ln = (int)threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < kSmallStage; k++) {
      const int rr = wv + k * (kSmallThreads / 64);
      if (ln < in_w && rr < rows) in[rr * in_w + ln] = texel[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < kSmallStage; k++) {
      const int i = (int)threadIdx.x + k * kSmallThreads;
      if (i < n_in) in[i] = texel[k];
    }
  }
  __syncthreads();
#if FDH_TIMING
  W_staged = wall_clock64();
#endif
  // horizontal: a thread produces two consecutive outputs of one row
  for (int t = threadIdx.x; t < rows * (kSmallTW / kSmallHOut); t += kSmallThreads) {
    const int rr = t / (kSmallTW / kSmallHOut), c = t - rr * (kSmallTW / kSmallHOut);
    f2 rg[kSmallHOut], ba[kSmallHOut];
    const uint32_t* __restrict__ win = in + rr * in_w + kSmallHOut * c;
    fir_outputs<kSmallHOut, 6>(P.taps.dense, reach, [&](int j) { return win[j]; }, rg, ba);
#pragma unroll
    for (int o = 0; o < kSmallHOut; o++) hres[rr * kSmallTW + kSmallHOut * c + o] = pack2(rg[o], ba[o]);
  }
#if FDH_TIMING
  W_h = wall_clock64();
#endif
  __syncthreads();
#if FDH_TIMING
  W_hb = wall_clock64();
#endif
  // vertical: a thread produces kSmallVOut consecutive rows of one column
  for (int t = threadIdx.x; t < kSmallTW * (kSmallTH / kSmallVOut); t += kSmallThreads) {
    const int pr = t / kSmallTW, c = t - pr * kSmallTW;
    const int x = xs + c, y = ys + kSmallVOut * pr;
    if (x >= P.x1 || y >= P.y1) continue;
    f2 rg[kSmallVOut], ba[kSmallVOut];
    const uint32_t* __restrict__ col = hres + (kSmallVOut * pr) * kSmallTW + c;
    fir_outputs<kSmallVOut, 6>(P.taps.dense, reach, [&](int j) { return col[j * kSmallTW]; }, rg, ba);
#pragma unroll
    for (int o = 0; o < kSmallVOut; o++)
      if (y + o < P.y1) P.dst[(size_t)(y + o) * P.pitch + x] = pack2(rg[o], ba[o]);
  }
#if FDH_TIMING
  W_v = wall_clock64();
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) {
    unsigned long long* row = g_wave_times + 16 * ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6));
    row[1] = W_in; row[2] = W_asked; row[3] = W_staged; row[4] = W_h; row[5] = W_hb; row[7] = W_v; row[6] = 8;
  }
#endif
}

// small regions: fewer outputs per thread -> more, shorter waves (see NOUT above)
constexpr int kBlurVWaves = 4;  // waves per V-pass workgroup: the tile is 64 columns x (waves * outputs) rows
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
  if (blur_forced_path() != 2 && launch_blur_mx_pass(false, s, P, nullptr, nullptr)) return;
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
  if (blur_forced_path() != 2 && launch_blur_mx_pass(true, s, P, draws, exts)) return;
  switch (blur_pick_nout(P.y1 - P.y0, kBlurVWaves, P.taps.reach)) {
    case 12: launch_blur_v_n<12, kBlurVWaves>(s, P, draws, exts); break;
    case 10: launch_blur_v_n<10, kBlurVWaves>(s, P, draws, exts); break;
    case 16: launch_blur_v_n<16, kBlurVWaves>(s, P, draws, exts); break;
    default: launch_blur_v_n<8, kBlurVWaves>(s, P, draws, exts); break;
  }
}
}  // namespace fdh
