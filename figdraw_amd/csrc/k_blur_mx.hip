// k_blur_mx.hip -- the backdrop blur on the matrix pipe, for regions of 0.4 Mpx and more: the merged FIR of glsl/blur.frag (:11-32) as a
// banded Toeplitz product (v_mfma_f32_32x32x16_f16), texels staged by LDS-DMA into a per-wave ring.  k_blur_mx<NK, kV>: one pass;
// k_blur_fx<NKH, NKV>: both passes of a full-frame node in one out-of-place kernel.
#include "fdh_device.h"

namespace fdh {
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
constexpr int kMxWaves = 2;  // waves per SIMD the passes are compiled for.  (3 -- eleven 14-KB rings fit a CU, 2720 waves of three blocks in one
                          // round -- measured 21.3 / 22.5 us against 19.9 / 19.6 for the two passes at 4K: more waves per SIMD do not help, the
                          // passes are paced by the memory system, profiles/r04_blur_notes.txt)
constexpr int mx_ring_slots(int nk, bool /*vertical*/) { return nk + 2; }
// waves per workgroup of an instantiation.  One: each wave works alone (own ring, no barrier); what a workgroup decides is WHERE they
// run.  (Round 4 measured two and four neighbours per workgroup, so that the texels two horizontal neighbours share come out of one
// CU's vector cache: the sum of the two passes 39.2 -> 38.0 us, `value` the same within the boxes' noise.  Left at one.)
constexpr int mx_wg(int /*nk*/, bool /*vertical*/) { return 1; }
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
template <int NK, bool kV>
__global__ __launch_bounds__(64 * mx_wg(NK, kV), kMxWaves) void k_blur_mx(BlurParams P, const DrawRec* __restrict__ draws, const QuadExt* __restrict__ exts, int T) {
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
      // Two ring slots nothing is in flight into until the end of this iteration: block b's OWN first two k-steps -- its products
      // are done, block b + 1 starts two k-steps further on, and the k-steps of block b + 2 are only issued into them behind this epilogue.
      uint32_t* sc0 = ring + slot0 * kMxSlot;
      uint32_t* sc1 = ring + (slot0 + 1 >= R ? slot0 + 1 - R : slot0 + 1) * kMxSlot;
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

// Matrix-pipe passes: NK k-steps of 16 texels must cover a block's 32 + 2 reach window (+ up to 3 texels of alignment
// for the horizontal pass); T blocks per wave, as many as still leave every SIMD a couple of waves.
// Blocks per wave: the smallest T for which every wave of the launch is resident at once.  A wave lives for the whole pass
// (prologue + T blocks), so a second, partly filled round of waves costs a whole wave lifetime.  Waves per CU = what the
// runtime's occupancy query gives the instantiation with its LDS ring (NK + 4 slots of 2 KB per wave; registers: three waves
// per SIMD for the narrow filters, two for the widest ones).  (4K, NK = 5: T = 4, 2040 waves for 2048 slots; leaving a
// tenth of the slots free, T = 5, measured 1 us slower on the vertical pass.)
static int mx_pick_t(long long per_cu, long long outputs_along, long long lines) {
  // (two waves per SIMD at most: with the horizontal pass's smaller ring eleven fit a CU, and T = 3 with 2720 shorter waves
  // measured 24.6 us against 23.4 for the two blur launches of the bench frame)
  const long long slots = 256 * std::min<long long>(per_cu, 4 * kMxWaves);
  const long long along_blocks = (outputs_along + 31) / 32, line_groups = (lines + 31) / 32;
  for (int t = 1; t < 64; t++) if (line_groups * ((along_blocks + t - 1) / t) <= slots) return t;
  return 64;
}
template <int NK, bool kV> static void launch_blur_mx(hipStream_t s, const BlurParams& P, const DrawRec* draws, const QuadExt* exts) {
  constexpr size_t lds = (size_t)mx_ring_slots(NK, kV) * kMxSlot * sizeof(uint32_t);
  constexpr int kWG = mx_wg(NK, kV);
  static const int per_cu = [] {  // waves resident per CU, asked once per instantiation
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_blur_mx<NK, kV>, 64 * kWG, lds * kWG) != hipSuccess || n <= 0) n = std::min<int>(4 * kMxWaves, 160 / (mx_ring_slots(NK, kV) * 2)) / kWG;
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
  const int total = n_sg * ((blocks + t - 1) / t), per = (total + 7) / 8;
  FDH_LAUNCH((k_blur_fx<NKH, NKV>), dim3(8 * per), dim3(64 * kFxWaves), lds, s, P, w_v, draws, exts, t);
}
bool blur_fused_supported(int reach, int W, int pitch) {
  const int nkh = mx_nk(reach, false), nkv = mx_nk(reach, true);
  return (pitch & 3) == 0 && W >= 4 && nkh >= 3 && nkh <= 6 && (nkh == nkv || nkh == nkv + 1);
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
bool launch_blur_mx_pass(bool vertical, hipStream_t s, const BlurParams& P, const DrawRec* draws, const QuadExt* exts) {
  return vertical ? launch_blur_mx_nk<true>(s, P, draws, exts) : launch_blur_mx_nk<false>(s, P, draws, exts);
}
}  // namespace fdh
