// k_bin_upload.hip -- the two small launches in front of a frame's compositing: k_upload_frame gathers the frame's pieces out of the
// recording threads' staging mirrors into the dense device arrays, k_bin_draws builds every (phase, bin)'s draw list in painter's order.
#include "fdh_device.h"

namespace fdh {
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
    // (a phase of at most kAhead chunks -- the bench frame's 707 draws are three -- walks them all: their boxes are fetched together
    // anyway, and the chunk boxes in front were one more dependent round trip of a launch that is four of them)
    const bool few = c_last - (first >> 8) < kAhead;  // (wave-uniform)
    unsigned long long live = few ? __ballot(cl <= c_last) : __ballot(cl <= c_last && binbox_hits(P.chunkbox[min(cl, c_last)], U));
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
void set_launch_events(hipEvent_t start, hipEvent_t stop) { t_prof_start = start; t_prof_stop = stop; t_prof_used = false; }
bool launch_events_used() { return t_prof_used; }  // false: the launch_* call between had nothing to launch
void launch_bin(hipStream_t s, const BinParams& P) {
  const int n = P.sub_n ? P.sub_first[P.sub_n] : P.n_phases * P.bins_x * P.bins_y;
  if (n <= 0) return;
  if (P.refine) FDH_LAUNCH(k_bin_draws<true>, dim3(n), dim3(64), 0, s, P);
  else FDH_LAUNCH(k_bin_draws<false>, dim3(n), dim3(64), 0, s, P);
}
void launch_upload_frame(hipStream_t s, void* dst, const UploadTable& T) {
  if (T.copy_units == 0) return;
  uint32_t widest = 1;
  for (uint32_t r = 0; r < T.n_runs; r++) widest = std::max(widest, (T.run[r].bytes + 1023u) / 1024u);
  hipLaunchKernelGGL(k_upload_frame, dim3(widest, T.n_runs), dim3(64), 0, s, static_cast<uint8_t*>(dst), T);
}
}  // namespace fdh
