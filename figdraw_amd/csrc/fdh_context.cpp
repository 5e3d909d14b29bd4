// fdh_context.cpp -- host side: BackendContext calls -> draw records -> GPU submission.
//
// What glcontext.nim does with ten vertex streams and a batch flush, this does with one 128-byte record
// per call.  The record carries exactly what the reference's vertex attributes carry (ceil-snapped quad,
// un-snapped half extents, packed radii, mode word, factors, colours) so the kernels can restate the
// fragment shaders.  Clip masks become push/pop records evaluated analytically per pixel, backdrop
// blurs split the list into phases (a blur is a global barrier in painter's order, glcontext.nim:1788-1841).
#include "fdh_context.h"

#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace fdh {

void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw Error(FDH_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

static inline float nim_round(float x) { return x >= 0.0f ? std::floor(x + 0.5f) : -std::floor(-x + 0.5f); }  // Nim math.round
static inline float clampf(float x, float lo, float hi) { return !(x >= lo) ? lo : (x > hi ? hi : x); }  // (a NaN comes out as lo)
static inline uint32_t pack_color(FdhColor c) { return (uint32_t)c.r | ((uint32_t)c.g << 8) | ((uint32_t)c.b << 16) | ((uint32_t)c.a << 24); }

static Aff aff_mul(const Aff& m, const Aff& n) {
  Aff r;
  r.a = m.a * n.a + m.c * n.b;
  r.b = m.b * n.a + m.d * n.b;
  r.c = m.a * n.c + m.c * n.d;
  r.d = m.b * n.c + m.d * n.d;
  r.tx = m.a * n.tx + m.c * n.ty + m.tx;
  r.ty = m.b * n.tx + m.d * n.ty + m.ty;
  return r;
}

// when each device context of the process last submitted a frame (steady-clock ns; 0 = never): Context::prepare asks whether
// frames of OTHER contexts are in flight
static constexpr int kSubmitSlots = 64;
static std::atomic<int64_t> g_last_submit_ns[kSubmitSlots];
static std::atomic<int> g_next_submit_slot{0};

// ------------------------------------------------------------------ lifetime
Context::Context(int atlas_size, float pixel_scale, int device, uint32_t flags) : device_(device), flags_(flags), pixel_scale_(pixel_scale) {
  host_only_ = (flags & FDH_CREATE_RECORD_ONLY) != 0;
  if (host_only_) {  // a call recorder: the front-end and the atlas packer run, nothing is drawn, no device is touched
    initial_atlas_size_ = atlas_size > 0 ? atlas_size : 1024;
    alloc_atlas(initial_atlas_size_);
    return;
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Error(FDH_ERR_NO_DEVICE, "no HIP device visible (libfigdraw_hip has no CPU fallback)");
  if (device < 0 || device >= n) throw Error(FDH_ERR_NO_DEVICE, "HIP device ordinal out of range");
  FDH_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  FDH_HIP(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    throw Error(FDH_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
  FDH_HIP(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
  stream_ = own_stream_;
  for (auto& e : ev_) FDH_HIP(hipEventCreate(&e));
  for (auto& e : staging_ev_) FDH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  initial_atlas_size_ = atlas_size > 0 ? atlas_size : 1024;  // newContext default, glcontext.nim:255-261
  alloc_atlas(initial_atlas_size_);
  submit_slot_ = g_next_submit_slot.fetch_add(1) % kSubmitSlots;
  static const bool env_sync = [] { const char* e = std::getenv("FDH_SYNC_SUBMIT"); return e && std::atoi(e) != 0; }();
  if (!(flags & FDH_CREATE_SYNC_SUBMIT) && !env_sync) worker_ = std::thread([this] { worker_main(); });
}

Context::~Context() {
  if (host_only_) return;
  try { comm_destroy(); } catch (...) {}
  if (worker_.joinable()) {
    { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
    cv_job_.notify_all();
    worker_.join();
  }
  (void)hipSetDevice(device_);
  if (stream_) (void)hipStreamSynchronize(stream_);
  for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
  for (auto& e : ev_pool_) (void)hipEventDestroy(e);
  for (auto& l : atlas_levels_) if (l) (void)hipFree(l);
  if (fb_) (void)hipFree(fb_);
  if (backdrop_) (void)hipFree(backdrop_);
  if (blur_tmp_) (void)hipFree(blur_tmp_);
  if (alt_) (void)hipFree(alt_);
  if (dbg_snap_) (void)hipFree(dbg_snap_);
  d_frame_.release(); d_lists_.release(); d_counts_.release(); d_order_[0].release(); d_order_[1].release();
  glyph_a_.release(); glyph_b_.release(); glyph_lines_.release(); glyph_acc_.release(); d_mask_spill_.release();
  for (auto& b : staging_) b.release();
  for (auto& e : staging_ev_) if (e) (void)hipEventDestroy(e);
  if (own_stream_) (void)hipStreamDestroy(own_stream_);
}

void Context::set_stream(void* s) {
  need_device("set_stream");
  drain();
  FDH_HIP(hipStreamSynchronize(stream_));
  stream_ = s ? (hipStream_t)s : own_stream_;
}
void Context::need_device(const char* what) const {
  if (host_only_) throw Error(FDH_ERR_NO_DEVICE, std::string(what) + ": this context was created with FDH_CREATE_RECORD_ONLY (it records calls, it draws nothing)");
}
// ------------------------------------------------------------------ the submit thread
// end_frame prepares the frame on the calling thread, hands the LaunchJob over and returns; this thread issues the launches
// (Context::issue: ~20 us of HIP runtime calls per bench frame that used to sit between the caller's tree walks).  One job at
// a time; the caller only waits when it has the NEXT frame prepared before this one's launches are out.
// Both sides spin briefly before they sleep: at 10 000 frames/s a futex round trip per hand-over would be a tenth of a frame.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}
void Context::worker_main() {
  (void)hipSetDevice(device_);
  for (;;) {
    bool have = false;
    for (int spin = 0; spin < 4000 && !have; spin++) { have = pending_.load(std::memory_order_acquire); if (!have) cpu_relax(); }
    if (!have) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_job_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) || quit_; });
      if (!pending_.load(std::memory_order_acquire)) return;  // quit_
    }
    try {
      issue(job_);
    } catch (...) {
      worker_error_ = std::current_exception();
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      pending_.store(false, std::memory_order_release);
    }
    cv_done_.notify_all();
  }
}
void Context::drain() {
  if (!worker_.joinable()) return;
  for (int spin = 0; spin < 4000 && pending_.load(std::memory_order_acquire); spin++) cpu_relax();
  if (pending_.load(std::memory_order_acquire)) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return !pending_.load(std::memory_order_acquire); });
  }
  if (worker_error_) {
    std::exception_ptr e = worker_error_;
    worker_error_ = nullptr;
    std::rethrow_exception(e);
  }
}

void Context::sync() {
  if (host_only_) return;
  drain();
  FDH_HIP(hipSetDevice(device_));
  FDH_HIP(hipStreamSynchronize(stream_));
}

// ------------------------------------------------------------------ atlas (glcontext.nim:536-641, textures.nim:88-119)
void Context::alloc_atlas(int size) {
  int s = 1;
  while (s < size) s <<= 1;  // the samplers mask coordinates: keep the atlas a power of two
  for (auto& l : atlas_levels_) { if (l) (void)hipFree(l); l = nullptr; }
  atlas_size_ = s;
  n_levels_ = 0;
  for (int ls = s; ls >= 1 && n_levels_ < kMaxMips; ls >>= 1) {
    if (!host_only_) {
      FDH_HIP(hipMalloc((void**)&atlas_levels_[n_levels_], (size_t)ls * ls * 4));
      FDH_HIP(hipMemsetAsync(atlas_levels_[n_levels_], 0, (size_t)ls * ls * 4, stream_));
    }
    n_levels_++;
    if (ls == 1) break;
  }
  heights_.assign((size_t)s, 0);
  entries_.clear();
  atlas_epoch_++;  // cached draw records of image nodes carry atlas positions (RetainedRoot::atlas_epoch)
}
void Context::reset_atlas(int minimum_size) {
  sync();
  int s = initial_atlas_size_;
  while (s < minimum_size) s *= 2;  // plannedAtlasSize
  alloc_atlas(s);
}
int64_t Context::atlas_packed_area() const {
  int64_t a = 0;
  for (auto h : heights_) a += h;
  return a;
}
void Context::find_empty_rect(int w, int h, int* ox, int* oy) {  // glcontext.nim:541-579
  for (;;) {
    const int S = atlas_size_, M = atlas_margin_;
    const int iw = w + M * 2, ih = h + M * 2;
    int lowest = S, at = 0;
    for (int i = 0; i < S; i++) {
      int v = heights_[i];
      if (v < lowest) {
        bool fit = true;
        for (int j = 0; j <= iw; j++) {
          if (i + j >= S) { fit = false; break; }
          if ((int)heights_[i + j] > v) { fit = false; break; }
        }
        if (fit) { lowest = v; at = i; }
      }
    }
    if (lowest + ih > S) {
      if (S >= 16384) throw Error(FDH_ERR_ATLAS_FULL, "atlas full at 16384^2");
      sync();
      alloc_atlas(S * 2);  // grow(): resetImageAtlas(atlasSize * 2) drops every entry (glcontext.nim:536-539)
      continue;
    }
    for (int j = at; j < at + iw; j++) heights_[j] = (uint16_t)(lowest + ih + M * 2);
    *ox = at + M;
    *oy = lowest + M;
    return;
  }
}
void Context::upload_atlas_rect(int level, int x, int y, int w, int h, const uint8_t* rgba) {
  const int LS = atlas_size_ >> level;
  if (x < 0 || y < 0 || x + w > LS || y + h > LS || w <= 0 || h <= 0 || host_only_) return;
  FDH_HIP(hipMemcpy2D(atlas_levels_[level] + (size_t)y * LS + x, (size_t)LS * 4, rgba, (size_t)w * 4, (size_t)w * 4, h,
                      hipMemcpyHostToDevice));  // synchronous: image uploads are rare and the source is pageable
}
// pixie's Image.minifyBy2 on premultiplied RGBA8 (the arithmetic the reference's data/img1.flippy pins: its stored levels are this
// chain): box SUM div 4; an odd extent rounds the result size up, the extra column / row holding mix(a, b, 0.5) * 0.5 of the last
// source column / row (mix = (127 a + 128 b) div 255, * 0.5 = (128 v) div 255) and the extra corner the last texel * 0.25 =
// (64 v) div 255.  k_minify2 is the device form of the same step.
static void minify_by2_host(const uint8_t* src, int w, int h, uint8_t* dst) {
  const int nw = (w + 1) / 2, nh = (h + 1) / 2;
  auto at = [&](int x, int y, int k) -> unsigned { return src[((size_t)y * w + x) * 4 + k]; };
  for (int y = 0; y < nh; y++) {
    const bool row_pair = 2 * y + 1 < h;
    for (int x = 0; x < nw; x++) {
      const bool col_pair = 2 * x + 1 < w;
      for (int k = 0; k < 4; k++) {
        unsigned v;
        if (col_pair && row_pair) v = (at(2 * x, 2 * y, k) + at(2 * x + 1, 2 * y, k) + at(2 * x + 1, 2 * y + 1, k) + at(2 * x, 2 * y + 1, k)) >> 2;
        else if (row_pair) v = ((at(w - 1, 2 * y, k) * 127u + at(w - 1, 2 * y + 1, k) * 128u) / 255u) * 128u / 255u;
        else if (col_pair) v = ((at(2 * x, h - 1, k) * 127u + at(2 * x + 1, h - 1, k) * 128u) / 255u) * 128u / 255u;
        else v = at(w - 1, h - 1, k) * 64u / 255u;
        dst[((size_t)y * nw + x) * 4 + k] = (uint8_t)v;
      }
    }
  }
}
void Context::put_levels(int x, int y, int w, int h, const uint8_t* rgba) {
  // updateSubImage: level chain by repeated minifyBy2 while width > 1 and height > 1 (textures.nim:106-119).
  std::vector<uint8_t> cur(rgba, rgba + (size_t)w * h * 4), nxt;
  int cw = w, ch = h, lx = x, ly = y, level = 0;
  while (cw > 1 && ch > 1 && level < n_levels_) {
    upload_atlas_rect(level, lx, ly, cw, ch, cur.data());
    const int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
    nxt.assign((size_t)nw * nh * 4, 0);
    minify_by2_host(cur.data(), cw, ch, nxt.data());
    cur.swap(nxt);
    cw = nw; ch = nh; lx /= 2; ly /= 2; level++;
  }
}
// the ink boxes of an image whose texels the host holds (AtlasEntry): one pass, sixteen running boxes
static void measure_ink(AtlasEntry& e, const uint8_t* rgba) {
  // Glyph- and icon-sized images only (a 48 x 48 MSDF cell, a 20 px glyph): that is where a draw covers a fraction of its quad, and
  // the pass stays in the microseconds.  A photograph is opaque to its edges and would cost a pass over megapixels for nothing.
  e.has_ink = false;
  if (e.w > 128 || e.h > 128) return;
  for (int k = 0; k < kInkLevels; k++) e.ink_a[k] = e.ink_rgb[k] = InkBox{32767, 32767, 0, 0};
  auto grow = [](InkBox& b, int x, int y) {
    b.x0 = (int16_t)std::min<int>(b.x0, x); b.y0 = (int16_t)std::min<int>(b.y0, y);
    b.x1 = (int16_t)std::max<int>(b.x1, x + 1); b.y1 = (int16_t)std::max<int>(b.y1, y + 1);
  };
  for (int y = 0; y < e.h; y++) {
    const uint8_t* row = rgba + (size_t)y * e.w * 4;
    for (int x = 0; x < e.w; x++) {
      const int a = row[4 * x + 3], m = std::max<int>(row[4 * x], std::max<int>(row[4 * x + 1], row[4 * x + 2]));
      // levels the value exceeds: t = 0, 16, .. below it.  The boxes are nested (level k's holds level k + 1's): a texel inside
      // the highest one it counts for is inside them all.
      const int la = std::min((a + 15) >> 4, kInkLevels), lm = std::min((m + 15) >> 4, kInkLevels);
      auto inside = [&](const InkBox& b) { return x >= b.x0 && x < b.x1 && y >= b.y0 && y < b.y1; };
      if (la > 0 && !inside(e.ink_a[la - 1])) for (int k = 0; k < la; k++) grow(e.ink_a[k], x, y);
      if (lm > 0 && !inside(e.ink_rgb[lm - 1])) for (int k = 0; k < lm; k++) grow(e.ink_rgb[k], x, y);
    }
  }
  for (int k = 0; k < kInkLevels; k++) {  // nothing above the level: an empty box at the origin
    if (e.ink_a[k].x1 <= e.ink_a[k].x0) e.ink_a[k] = InkBox{0, 0, 0, 0};
    if (e.ink_rgb[k].x1 <= e.ink_rgb[k].x0) e.ink_rgb[k] = InkBox{0, 0, 0, 0};
  }
  e.has_ink = true;
}
// The record just emitted (an upright atlas quad sampling level 0 of `e`) covers nothing outside the image of the ink box at
// level `level_t` (values <= level_t give coverage exactly 0 for this draw): its pixel bounds shrink to that image.  A bilinear
// sample at texel coordinate t reads texels floor(t) and floor(t) + 1, the sub-pixel shift moves t by less than one texel: the box
// is widened by three texels and the pixel range by one pixel on every side, far beyond any rounding of the linear map.
void Context::shrink_to_ink(const AtlasEntry& e, bool use_alpha, int level_t) {
  static const bool enabled = [] { const char* v = std::getenv("FDH_INK_BOUNDS"); return !v || std::atoi(v) != 0; }();
  if (!enabled || !e.has_ink || level_t < 0 || recs_.empty()) return;
  DrawRec& r = recs_.back();
  BBox& b = bboxes_.back();
  if ((r.op_mode & F_GENERAL) || b.x1 <= b.x0 || b.y1 <= b.y0) return;
  const InkBox ib = (use_alpha ? e.ink_a : e.ink_rgb)[std::min(level_t / 16, kInkLevels - 1)];
  if (ib.x1 <= ib.x0 || ib.y1 <= ib.y0) { b = BBox{0, 0, 0, 0}; r.bx0 = r.by0 = r.bx1 = r.by1 = 0; return; }  // nothing in the image reaches the level
  const double S = (double)atlas_size_;
  auto range = [&](double ua, double ut, double o, double inv, double lo_t, double hi_t, int& p0, int& p1) {
    // texel coordinate at pixel centre c: t(c) = (ua + (ut - ua) (c - o) inv) S - 0.5; pixels whose t lies in [lo_t - 3, hi_t + 2]
    const double A = (ut - ua) * inv * S, B = ua * S - 0.5 - A * o;
    if (!(std::fabs(A) > 1e-12)) return;
    double c0 = ((lo_t - 3.0) - B) / A, c1 = ((hi_t + 2.0) - B) / A;
    if (c0 > c1) std::swap(c0, c1);
    if (!(c0 > -1.0e6 && c1 < 1.0e6)) return;
    p0 = std::max(p0, (int)std::floor(c0 - 0.5) - 1);
    p1 = std::min(p1, (int)std::ceil(c1 - 0.5) + 2);
  };
  int x0 = b.x0, x1 = b.x1, y0 = b.y0, y1 = b.y1;
  range(r.r[0], r.r[2], r.ox, r.inv_w, (double)(e.x + ib.x0), (double)(e.x + ib.x1), x0, x1);
  range(r.r[1], r.r[3], r.oy, r.inv_h, (double)(e.y + ib.y0), (double)(e.y + ib.y1), y0, y1);
  if (x1 <= x0 || y1 <= y0) { x0 = y0 = x1 = y1 = 0; }
  b = BBox{(int16_t)x0, (int16_t)y0, (int16_t)x1, (int16_t)y1};
  r.bx0 = b.x0; r.by0 = b.y0; r.bx1 = b.x1; r.by1 = b.y1;
}

void Context::put_image(int64_t key, int w, int h, const uint8_t* rgba, int out_rect[4]) {
  if (w <= 0 || h <= 0 || !rgba) throw Error(FDH_ERR_INVALID, "put_image: empty image");
  if (!host_only_) FDH_HIP(hipSetDevice(device_));
  int x, y;
  find_empty_rect(w, h, &x, &y);
  AtlasEntry ent{x, y, w, h};
  measure_ink(ent, rgba);
  entries_[key] = ent;
  atlas_epoch_++;
  sync();  // a frame in flight may still sample the atlas
  put_levels(x, y, w, h, rgba);
  if (out_rect) { out_rect[0] = x; out_rect[1] = y; out_rect[2] = w; out_rect[3] = h; }
}
// A rasterised glyph on its way into the atlas, processed on the device: optional LCD filter (applyLcdFilter, common/
// textrasters/pixie_raster.nim:12-43, what renderPixieGlyph does between fillText and loadGlyphImage :83-91), then the
// level chain of updateSubImage (textures.nim:106-119) -- every step a kernel on the context's stream.
void Context::put_glyph_image(int64_t key, int w, int h, const uint8_t* rgba, uint32_t flags, int out_rect[4]) {
  if (w <= 0 || h <= 0 || !rgba) throw Error(FDH_ERR_INVALID, "put_glyph_image: empty image");
  if (flags & ~(uint32_t)FDH_GLYPH_LCD_FILTER) throw Error(FDH_ERR_INVALID, "put_glyph_image: unknown flag");
  int x, y;
  find_empty_rect(w, h, &x, &y);
  entries_[key] = AtlasEntry{x, y, w, h, false, {}, {}};
  atlas_epoch_++;
  if (out_rect) { out_rect[0] = x; out_rect[1] = y; out_rect[2] = w; out_rect[3] = h; }
  if (host_only_) return;
  FDH_HIP(hipSetDevice(device_));
  sync();  // a frame in flight may still sample the atlas
  const size_t n = (size_t)w * h;
  glyph_a_.reserve(n);
  glyph_b_.reserve(n);
  FDH_HIP(hipMemcpyAsync(glyph_a_.ptr, rgba, n * 4, hipMemcpyHostToDevice, stream_));
  glyph_to_atlas(glyph_a_.ptr, glyph_b_.ptr, w, h, x, y, flags);
}
// device image -> (LCD filter) -> atlas level chain, all on the context's stream; waits for it (the caller's buffers are free after)
void Context::glyph_to_atlas(uint32_t* cur, uint32_t* nxt, int w, int h, int x, int y, uint32_t flags) {
  if (flags & FDH_GLYPH_LCD_FILTER) { launch_lcd_filter(stream_, cur, nxt, w, h); std::swap(cur, nxt); }
  int cw = w, ch = h, lx = x, ly = y, level = 0;
  while (cw > 1 && ch > 1 && level < n_levels_) {
    launch_atlas_blit(stream_, atlas_levels_[level], atlas_size_ >> level, lx, ly, cur, cw, ch);
    const int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
    launch_minify2(stream_, cur, nxt, cw, ch);
    std::swap(cur, nxt);
    cw = nw; ch = nh; lx /= 2; ly /= 2; level++;
  }
  FDH_HIP(hipStreamSynchronize(stream_));
  FDH_HIP(hipGetLastError());
}

// generateGlyph's job (common/fontglyphs.nim:61-106) with an own rasteriser in pixie's place: a glyph OUTLINE (quadratic segments in
// pixel units of the w x h image, y down; cx = NaN marks a straight line) becomes coverage on the device and goes into the atlas.
// The curves are flattened here on the host (chord error <= 0.025 px; the same float formula as oracle/figdraw_oracle.c,
// fo_flatten_outline), the area accumulation runs in k_rasterize_lines.  pixie's texels are third-party and unpinned
// (SURVEY.md 8c): parity is defined against the oracle's restatement of the same published algorithm.
static int flatten_count(const float* q) {
  const float ddx = q[0] - 2.0f * q[2] + q[4], ddy = q[1] - 2.0f * q[3] + q[5];
  const float dev = std::sqrt(ddx * ddx + ddy * ddy);
  const int n = (int)std::ceil(std::sqrt(dev * 10.0f));  // error of n chords = dev / (4 n^2) <= 0.025 px
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}
void Context::put_glyph_outline(int64_t key, int w, int h, const float* segs, int n, uint32_t flags, int out_rect[4]) {
  if (w <= 0 || h <= 0 || w > 4096 || h > 4096) throw Error(FDH_ERR_INVALID, "put_glyph_outline: image size must be in 1..4096");
  if (n < 0 || (n > 0 && !segs)) throw Error(FDH_ERR_INVALID, "put_glyph_outline: bad outline");
  if (flags & ~(uint32_t)FDH_GLYPH_LCD_FILTER) throw Error(FDH_ERR_INVALID, "put_glyph_outline: unknown flag");
  std::vector<float> lines;
  lines.reserve((size_t)n * 16);
  for (int i = 0; i < n; i++) {
    const float* q = segs + 6 * (size_t)i;
    if (q[2] != q[2]) { lines.insert(lines.end(), {q[0], q[1], q[4], q[5]}); continue; }
    const int k = flatten_count(q);
    float px = q[0], py = q[1];
    for (int j = 1; j <= k; j++) {
      const float t = (float)j / (float)k, u = 1.0f - t;
      const float x = j == k ? q[4] : (u * u) * q[0] + (2.0f * u * t) * q[2] + (t * t) * q[4];
      const float y = j == k ? q[5] : (u * u) * q[1] + (2.0f * u * t) * q[3] + (t * t) * q[5];
      lines.insert(lines.end(), {px, py, x, y});
      px = x; py = y;
    }
  }
  int x, y;
  find_empty_rect(w, h, &x, &y);
  entries_[key] = AtlasEntry{x, y, w, h, false, {}, {}};
  atlas_epoch_++;
  if (out_rect) { out_rect[0] = x; out_rect[1] = y; out_rect[2] = w; out_rect[3] = h; }
  if (host_only_) return;
  FDH_HIP(hipSetDevice(device_));
  sync();
  const size_t npx = (size_t)w * h, m = lines.size() / 4;
  glyph_a_.reserve(npx);
  glyph_b_.reserve(npx);
  glyph_lines_.reserve(std::max<size_t>(lines.size(), 4));
  glyph_acc_.reserve((size_t)h * (w + 2));
  if (m) FDH_HIP(hipMemcpyAsync(glyph_lines_.ptr, lines.data(), lines.size() * sizeof(float), hipMemcpyHostToDevice, stream_));
  launch_rasterize_lines(stream_, reinterpret_cast<const float4*>(glyph_lines_.ptr), (int)m, w, h, glyph_acc_.ptr, glyph_a_.ptr);
  glyph_to_atlas(glyph_a_.ptr, glyph_b_.ptr, w, h, x, y, flags);  // (synchronises: `lines` stays alive until then)
}
// Flippy: figdraw's mip-mapped image container (common/formatflippy.nim:77-149).  Layout: "flip", u32 version (1), then per
// mip level "mip!", u32 width, u32 height, u32 zlen, and a raw-snappy block holding straight RGBA8.  The reference
// converts every texel to pixie's premultiplied ColorRGBX on load and uploads level l at (x >> l, y >> l)
// (putFlippy glcontext.nim:610-620) instead of rebuilding the chain with minifyBy2.
static std::vector<uint8_t> snappy_uncompress(const uint8_t* in, size_t n) {
  size_t i = 0, len = 0;
  for (int shift = 0;; shift += 7) {
    if (i >= n || shift > 35) throw Error(FDH_ERR_INVALID, "flippy: bad snappy length");
    const uint8_t c = in[i++];
    len |= (size_t)(c & 0x7f) << shift;
    if (c < 0x80) break;
  }
  std::vector<uint8_t> out;
  out.reserve(len);
  auto need = [&](size_t k) { if (i + k > n) throw Error(FDH_ERR_INVALID, "flippy: truncated snappy block"); };
  while (i < n) {
    const uint8_t tag = in[i++];
    const int t = tag & 3;
    if (t == 0) {  // literal
      size_t l = tag >> 2;
      if (l < 60) l += 1;
      else {
        const int nb = (int)l - 59;
        need(nb);
        l = 0;
        for (int k = 0; k < nb; k++) l |= (size_t)in[i + k] << (8 * k);
        l += 1;
        i += nb;
      }
      need(l);
      out.insert(out.end(), in + i, in + i + l);
      i += l;
    } else {  // copy with 1-, 2- or 4-byte offset
      size_t l, off;
      if (t == 1) { need(1); l = ((tag >> 2) & 7) + 4; off = ((size_t)(tag >> 5) << 8) | in[i]; i += 1; }
      else if (t == 2) { need(2); l = (tag >> 2) + 1; off = in[i] | ((size_t)in[i + 1] << 8); i += 2; }
      else { need(4); l = (tag >> 2) + 1; off = in[i] | ((size_t)in[i + 1] << 8) | ((size_t)in[i + 2] << 16) | ((size_t)in[i + 3] << 24); i += 4; }
      if (off == 0 || off > out.size()) throw Error(FDH_ERR_INVALID, "flippy: bad snappy copy offset");
      for (size_t k = 0; k < l; k++) out.push_back(out[out.size() - off]);
    }
  }
  if (out.size() != len) throw Error(FDH_ERR_INVALID, "flippy: snappy length mismatch");
  return out;
}
void Context::put_mips(int64_t key, int n, const int* ws, const int* hs, const uint8_t* const* premul_rgba, int out_rect[4]) {
  // putFlippy glcontext.nim:610-620: level l goes to (x >> l, y >> l) with the size the container stored for it
  if (n <= 0 || !ws || !hs || !premul_rgba) throw Error(FDH_ERR_INVALID, "put_mips: no mip levels");
  for (int l = 0; l < n; l++)
    if (ws[l] <= 0 || hs[l] <= 0 || !premul_rgba[l]) throw Error(FDH_ERR_INVALID, "put_mips: bad mip level");
  if (!host_only_) FDH_HIP(hipSetDevice(device_));
  int rx = 0, ry = 0;
  find_empty_rect(ws[0], hs[0], &rx, &ry);
  entries_[key] = AtlasEntry{rx, ry, ws[0], hs[0], false, {}, {}};
  atlas_epoch_++;
  if (out_rect) { out_rect[0] = rx; out_rect[1] = ry; out_rect[2] = ws[0]; out_rect[3] = hs[0]; }
  sync();
  for (int l = 0; l < n && l < n_levels_; l++) upload_atlas_rect(l, rx >> l, ry >> l, ws[l], hs[l], premul_rgba[l]);
}
void Context::put_flippy(int64_t key, const uint8_t* data, size_t n, int out_rect[4]) {
  auto u32 = [&](size_t at) { return (uint32_t)data[at] | ((uint32_t)data[at + 1] << 8) | ((uint32_t)data[at + 2] << 16) | ((uint32_t)data[at + 3] << 24); };
  if (!data || n < 8 || std::memcmp(data, "flip", 4) != 0) throw Error(FDH_ERR_INVALID, "Invalid Flippy header");
  if (u32(4) != 1) throw Error(FDH_ERR_INVALID, "Invalid Flippy version");
  std::vector<std::vector<uint8_t>> mips;
  std::vector<int> ws, hs;
  size_t i = 8;
  while (i < n) {
    if (i + 16 > n || std::memcmp(data + i, "mip!", 4) != 0) throw Error(FDH_ERR_INVALID, "Invalid Flippy sub header");
    const int w = (int)u32(i + 4), h = (int)u32(i + 8);
    const size_t z = u32(i + 12);
    i += 16;
    if (i + z > n || w <= 0 || h <= 0) throw Error(FDH_ERR_INVALID, "Flippy read error");
    std::vector<uint8_t> px = snappy_uncompress(data + i, z);
    i += z;
    if (px.size() != (size_t)w * h * 4) throw Error(FDH_ERR_INVALID, "Flippy mip size mismatch");
    for (size_t k = 0; k < (size_t)w * h; k++) {  // ColorRGBA -> premultiplied ColorRGBX
      const unsigned a = px[4 * k + 3];
      px[4 * k + 0] = (uint8_t)((px[4 * k + 0] * a) / 255);
      px[4 * k + 1] = (uint8_t)((px[4 * k + 1] * a) / 255);
      px[4 * k + 2] = (uint8_t)((px[4 * k + 2] * a) / 255);
    }
    mips.push_back(std::move(px));
    ws.push_back(w);
    hs.push_back(h);
  }
  if (mips.empty()) throw Error(FDH_ERR_INVALID, "Flippy has no mip levels");
  std::vector<const uint8_t*> ptrs;
  for (auto& m : mips) ptrs.push_back(m.data());
  put_mips(key, (int)mips.size(), ws.data(), hs.data(), ptrs.data(), out_rect);
}
void Context::update_image(int64_t key, int w, int h, const uint8_t* rgba) {  // glcontext.nim:591-604
  auto it = entries_.find(key);
  if (it == entries_.end()) throw Error(FDH_ERR_INVALID, "update_image: unknown key");
  if (it->second.w != w || it->second.h != h) throw Error(FDH_ERR_INVALID, "update_image: size mismatch");
  if (!rgba) throw Error(FDH_ERR_INVALID, "update_image: null image");
  sync();
  measure_ink(it->second, rgba);  // the new texels have bounds of their own (draws shrink to them: shrink_to_ink) ...
  atlas_epoch_++;                 // ... and records cached for retained scenes hold the old ones
  put_levels(it->second.x, it->second.y, w, h, rgba);
}

// ------------------------------------------------------------------ transforms (glcontext.nim:1991-2024)

// ------------------------------------------------------------------ call recorder
// The reference's own front-end tests (tests/ttransform.nim, tests/trenderfragments.nim) hand the renderer a RecordingBackend
// and assert on the calls it receives.  fdh_record_begin / fdh_record_json give the same view of THIS library's front-end
// (fdh_frontend.cpp): every BackendContext-level call between the two, as a JSON array of [name, args...] -- the format
// oracle/figdraw_oracle.c records and oracle/ref_swiftshader.py replays.
namespace {
struct Rec {
  std::string& s; bool& first; const bool on;
  Rec(std::string& s_, bool& first_, bool on_, const char* name, size_t* mark, bool* mark_first) : s(s_), first(first_), on(on_) {
    if (!on) return;
    *mark = s.size(); *mark_first = first;  // (a call that turns out to be culled is taken back: Context::rec_drop_last)
    s += first ? "[\"" : ",\n[\""; s += name; s += "\""; first = false;
  }
  ~Rec() { if (on) s += "]"; }
  Rec& f(double v) { if (on) { char b[40]; std::snprintf(b, sizeof b, ",%.9g", v); s += b; } return *this; }
  Rec& i(long long v) { if (on) { char b[32]; std::snprintf(b, sizeof b, ",%lld", v); s += b; } return *this; }
  Rec& fv(const float* v, int n) {
    if (on) { s += ",["; for (int k = 0; k < n; k++) { char b[40]; std::snprintf(b, sizeof b, "%s%.9g", k ? "," : "", (double)v[k]); s += b; } s += "]"; }
    return *this;
  }
  Rec& col(FdhColor c) { if (on) { char b[48]; std::snprintf(b, sizeof b, ",[%d,%d,%d,%d]", c.r, c.g, c.b, c.a); s += b; } return *this; }
  Rec& cols(const FdhColor c[4]) {
    if (on) { s += ",["; for (int k = 0; k < 4; k++) { char b[48]; std::snprintf(b, sizeof b, "%s[%d,%d,%d,%d]", k ? "," : "", c[k].r, c[k].g, c[k].b, c[k].a); s += b; } s += "]"; }
    return *this;
  }
  Rec& fill(const FdhFill& fl) {
    if (on) {
      char b[200];
      std::snprintf(b, sizeof b, ",{\"kind\":%d,\"axis\":%d,\"start\":[%d,%d,%d,%d],\"mid\":[%d,%d,%d,%d],\"stop\":[%d,%d,%d,%d],\"mid_pos\":%d}", fl.kind, fl.axis,
                    fl.start.r, fl.start.g, fl.start.b, fl.start.a, fl.mid.r, fl.mid.g, fl.mid.b, fl.mid.a, fl.stop.r, fl.stop.g, fl.stop.b, fl.stop.a, fl.mid_pos);
      s += b;
    }
    return *this;
  }
};
}  // namespace
struct RecPause {  // a backend method that calls other backend methods records only itself
  bool& on; const bool was;
  explicit RecPause(bool& o) : on(o), was(o) { on = false; }
  ~RecPause() { on = was; }
};
#define FDH_REC(name) Rec rec_scope_(rec_, rec_first_, rec_on_, name, &rec_mark_, &rec_mark_first_); rec_scope_
// cull mode 2 (culling while the recorder runs): the draw call just recorded left no record -- it leaves no entry either
#define FDH_CULLED() do { culled_draws_++; if (rec_on_) { rec_.resize(rec_mark_); rec_first_ = rec_mark_first_; } } while (0)
void Context::record_begin() { rec_on_ = true; rec_first_ = true; rec_ = "["; }
const char* Context::record_json() {
  if (!rec_on_) return "[]";
  rec_ += "\n]";
  rec_on_ = false;
  return rec_.c_str();
}
void Context::set_aa(float aa) { { FDH_REC("set_aa_factor").f(aa); } aa_ = aa; }
void Context::set_subpixel_shift(float s) { { FDH_REC("set_text_subpixel_shift").f(s); } subpixel_shift_ = s; }  // setTextSubpixelShift figbackend.nim:663-686

void Context::save_transform() { { FDH_REC("save_transform"); } mats_.push_back(mat_); }
void Context::restore_transform() {
  { FDH_REC("restore_transform"); }
  if (mats_.empty()) throw Error(FDH_ERR_INVALID, "restoreTransform: empty transform stack");
  mat_ = mats_.back();
  mats_.pop_back();
}
void Context::translate(float x, float y) { { FDH_REC("translate").f(x).f(y); } Aff t; t.tx = x; t.ty = y; mat_ = aff_mul(mat_, t); }
void Context::rotate(float a) {
  { FDH_REC("rotate").f(a); }
  Aff r;  // vmath rotateZ: column 0 = (cos, -sin), column 1 = (sin, cos); pinned by tests/expected/render_line_rect.png
  r.a = std::cos(a); r.b = -std::sin(a); r.c = -r.b; r.d = r.a;
  mat_ = aff_mul(mat_, r);
}
void Context::scale(float sx, float sy) { { FDH_REC("scale").f(sx).f(sy); } Aff s; s.a = sx; s.d = sy; mat_ = aff_mul(mat_, s); }
void Context::apply_transform(const float m[16]) {  // column-major Mat4; `mat * vec3(x, y, 0)` uses its 2D affine part
  { FDH_REC("apply_transform").fv(m, 16); }
  Aff n;
  n.a = m[0]; n.b = m[1]; n.c = m[4]; n.d = m[5]; n.tx = m[12]; n.ty = m[13];
  mat_ = aff_mul(mat_, n);
}
bool Context::transform_mirrors_y() const { return mat_.a * mat_.d - mat_.b * mat_.c < 0.0f; }

// ------------------------------------------------------------------ frame
void Context::ensure_surfaces() {
  if (host_only_ || (surf_w_ == W_ && surf_h_ == H_ && fb_)) return;
  drain();  // the frame in submission still renders into the old surfaces
  FDH_HIP(hipStreamSynchronize(stream_));
  if (fb_) FDH_HIP(hipFree(fb_));
  if (backdrop_) FDH_HIP(hipFree(backdrop_));
  if (blur_tmp_) FDH_HIP(hipFree(blur_tmp_));
  if (alt_) { FDH_HIP(hipFree(alt_)); alt_ = nullptr; }
  if (dbg_snap_) { FDH_HIP(hipFree(dbg_snap_)); dbg_snap_ = nullptr; }
  const size_t n = (size_t)W_ * H_;
  FDH_HIP(hipMalloc((void**)&fb_, n * 4));
  FDH_HIP(hipMalloc((void**)&backdrop_, n * 4));
  FDH_HIP(hipMalloc((void**)&blur_tmp_, n * 4));
  FDH_HIP(hipMemsetAsync(fb_, 0, n * 4, stream_));
  surf_w_ = W_;
  surf_h_ = H_;
}

void Context::begin_frame(int w, int h, bool clear, const float rgba[4]) {  // glcontext.nim:2080-2092, 1951-1980
  { FDH_REC("begin_frame").i(clear ? 1 : 0).fv(rgba, 4); }
  if (frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has already been called.");
  if (w <= 0 || h <= 0 || w > 16384 || h > 16384) throw Error(FDH_ERR_INVALID, "beginFrame: frame size must be in 1..16384");
  t_begin_frame_ = std::chrono::steady_clock::now();
  if (!host_only_) FDH_HIP(hipSetDevice(device_));
  W_ = w;
  H_ = h;
  ensure_surfaces();
  clear_ = clear;
  if (clear) {
    auto q = [](float v) { return (uint32_t)std::floor(clampf(v, 0.0f, 1.0f) * 255.0f + 0.5f); };
    clear_rgba8_ = q(rgba[0]) | (q(rgba[1]) << 8) | (q(rgba[2]) << 16) | (q(rgba[3]) << 24);
  }
  frame_begun_ = true;
  mask_begun_ = false;
  mask_depth_ = 0;
  rect_masks_.clear();
  open_ops_.clear();
  recs_.clear();
  bboxes_.clear();
  exts_.clear();
  phases_.clear();
  blurs_.clear();
  fragments_ = 0;
  rec_diff_upload_ = false;
  phases_.push_back(Phase{});
  // rows a draw has to reach: the frame's, or -- under fdh_set_stripe, when the front-end has told how far the scene's blur nodes
  // reach (render_frame: the per-call path cannot know what is still to come) -- the stripe's, widened by that reach
  cull_y0_ = 0; cull_y1_ = H_;
  if (stripe_y1_ > stripe_y0_ && pending_reach_ >= 0) {
    cull_y0_ = std::max(0, std::min(H_, stripe_y0_) - pending_reach_);
    cull_y1_ = std::min(H_, std::max(0, stripe_y1_) + pending_reach_);
  }
  pending_reach_ = -1;
  culled_draws_ = 0;
}

void Context::push_rec(const DrawRec& r, const BBox& b) {
  recs_.push_back(r);
  bboxes_.push_back(b);
  phases_.back().count++;
}

static inline bool bbox_empty(const BBox& b) { return b.x1 <= b.x0 || b.y1 <= b.y0; }
static inline void bbox_union(BBox& a, const BBox& b) {
  if (bbox_empty(b)) return;
  if (bbox_empty(a)) { a = b; return; }
  a.x0 = std::min(a.x0, b.x0); a.y0 = std::min(a.y0, b.y0); a.x1 = std::max(a.x1, b.x1); a.y1 = std::max(a.y1, b.y1);
}

// The part of an axis-aligned SDF quad where the coverage term is saturated (DrawRec::ix0..iy1).  Works in the
// shader's local frame (atlas.frag:252-262: p = (uv - 0.5) * 2 * quadHalfExtents, y up) and maps back to pixels.
// {dist <= -e} of sdRoundedBox(b, r) is the rounded box (b - e, max(r - e, 0)); an axis-aligned rectangle whose
// corners are pulled in by (1 - 1/sqrt 2) r per corner lies inside it.  Elliptical corners use an approximate
// distance (atlas.frag:71-79), so there the core stays out of the corner cells, where the distance is the plain
// box distance max(|p| - b).  One pixel of slack on every side absorbs all float rounding.
static void set_saturated_core(DrawRec& r, float w_px, float h_px) {
  r.ix0 = r.iy0 = r.ix1 = r.iy1 = 0;
  const uint32_t mode = r.op_mode & 255u, op = (r.op_mode >> 12) & 15u;
  const uint32_t fill_mode = (r.op_mode >> 9) & 7u;
  if (!(op == OP_DRAW || op == OP_MASK_PUSH) || !(r.aa > 0.0f)) return;
  double e;  // core = {dist <= -e}
  if (op == OP_MASK_PUSH || mode == FDH_SDF_CLIP_AA || mode == FDH_SDF_BACKDROP_BLUR) e = 0.5 / r.aa;
  else if (mode == FDH_SDF_DROP_SHADOW) e = std::max(0.0, -(double)(fill_mode == 0u ? r.f1 : 0.0f));
  else if (mode == FDH_SDF_ANNULAR || mode == FDH_SDF_ANNULAR_AA) e = std::max(0.0, (double)r.f0) + 0.5 / r.aa;
  else if (mode == FDH_SDF_INSET_SHADOW && op == OP_DRAW) {
    // Inner shadow: far enough inside the (offset) shape the profile exp(-z^2/2) is below 0.49/255, so the blend cannot
    // move any 8-bit channel whatever the colours are (|sa (255 c - F)| < 0.5): the draw is a no-op there, like the
    // inside of a stroke.  z > 3.7 leaves a margin over the exact 3.54.
    const double sigma = std::max(0.5 * (double)r.f0, 0.5);
    e = std::max(0.0, 3.7 * sigma + (double)(fill_mode == 0u ? r.f1 : 0.0f));
  } else return;
  const bool inset = mode == FDH_SDF_INSET_SHADOW;
  const double qhx = r.p0, qhy = r.p1, bx = inset ? qhx : (double)r.p2, by = inset ? qhy : (double)r.p3;
  if (!(qhx > 0.0 && qhy > 0.0 && bx > 0.0 && by > 0.0)) return;
  double crx[4], cry[4];  // TR, BR, TL, BL as in DrawRec::r
  for (int k = 0; k < 4; k++) {
    const double sel = r.r[k];
    if (!(r.op_mode & F_ELLIP)) { crx[k] = cry[k] = std::max(sel, 0.0); continue; }
    if (sel < 0.0) { crx[k] = cry[k] = -sel - 1.0; continue; }
    const double pv = std::floor(sel + 0.5), hi = std::floor(pv / 4096.0);
    crx[k] = (pv - 4096.0 * hi) * bx / 4095.0;
    cry[k] = hi * by / 4095.0;
  }
  enum { TR = 0, BR = 1, TL = 2, BL = 3 };
  double xl, xr, yb, yt;  // local frame, y up
  if (!(r.op_mode & F_ELLIP)) {
    const double k = 0.2929;
    auto rr = [&](int i) { return std::max(crx[i] - e, 0.0); };
    xr = (bx - e) - k * std::max(rr(TR), rr(BR));
    xl = -(bx - e) + k * std::max(rr(TL), rr(BL));
    yt = (by - e) - k * std::max(rr(TR), rr(TL));
    yb = -(by - e) + k * std::max(rr(BR), rr(BL));
  } else {
    // horizontal band (full width, between the corner cells) or vertical band, whichever is larger
    const double hx0 = -(bx - e), hx1 = bx - e;
    const double hy1 = std::min(by - e, by - std::max(cry[TR], cry[TL])), hy0 = -std::min(by - e, by - std::max(cry[BR], cry[BL]));
    const double vy0 = -(by - e), vy1 = by - e;
    const double vx1 = std::min(bx - e, bx - std::max(crx[TR], crx[BR])), vx0 = -std::min(bx - e, bx - std::max(crx[TL], crx[BL]));
    const double ah = std::max(hx1 - hx0, 0.0) * std::max(hy1 - hy0, 0.0), av = std::max(vx1 - vx0, 0.0) * std::max(vy1 - vy0, 0.0);
    if (ah >= av) { xl = hx0; xr = hx1; yb = hy0; yt = hy1; } else { xl = vx0; xr = vx1; yb = vy0; yt = vy1; }
  }
  if (inset) { xl += r.p2; xr += r.p2; yb -= r.p3; yt -= r.p3; }  // the shadow shape sits at (p2, -p3) in the quad's frame
  if (!(xr > xl && yt > yb)) return;
  // local -> pixel centres: cx = ox + w_px * (x / (2 qhx) + 0.5), cy = oy + h_px * (0.5 - y / (2 qhy))
  // slack: what float rounding in the kernels' coordinate arithmetic can move a pixel centre against the level set (~2e-3 px at
  // 4K, 8e-3 at 16K), with room.  (It was a whole pixel: a quad ending on the frame edge -- the full-frame backdrop blur -- then
  // kept its outermost pixel ring out of the core although the coverage is exactly 1 there (centre 0.5 px inside, threshold
  // 0.5 / aa = 0.417): every block on the frame border took the vertical blur pass's slow path.)
#ifndef FDH_CORE_SLACK
#define FDH_CORE_SLACK (1.0 / 16.0)
#endif
  const double slack = FDH_CORE_SLACK;
  const double cxl = r.ox + w_px * (xl / (2.0 * qhx) + 0.5) + slack, cxr = r.ox + w_px * (xr / (2.0 * qhx) + 0.5) - slack;
  const double cyt = r.oy + h_px * (0.5 - yt / (2.0 * qhy)) + slack, cyb = r.oy + h_px * (0.5 - yb / (2.0 * qhy)) - slack;
  double ix0 = std::ceil(cxl - 0.5), ix1 = std::floor(cxr - 0.5) + 1.0, iy0 = std::ceil(cyt - 0.5), iy1 = std::floor(cyb - 0.5) + 1.0;
  ix0 = std::max(ix0, (double)r.ox); ix1 = std::min(ix1, (double)r.ox + w_px);  // stay inside the quad (coverage)
  iy0 = std::max(iy0, (double)r.oy); iy1 = std::min(iy1, (double)r.oy + h_px);
  auto c16 = [](double v) { return (int16_t)std::min(std::max(v, -32768.0), 32767.0); };
  if (!(ix1 > ix0 && iy1 > iy0)) return;
  r.ix0 = c16(ix0); r.iy0 = c16(iy0); r.ix1 = c16(ix1); r.iy1 = c16(iy1);
}

// Quad emission: ceil(ctx.mat * corner) per vertex, order BL,BR,TR,TL (glcontext.nim:1498-1509), then either the
// axis-aligned fast form or the two-triangle general form.
bool Context::emit_quad(DrawRec& r, float x0, float y0, float x1, float y1, int64_t* fragments) {
  const float vx[4] = {x0, x1, x1, x0}, vy[4] = {y1, y1, y0, y0};  // BL, BR, TR, TL
  return emit_quad_pts(r, vx, vy, fragments);
}

// The pixel bounds emit_quad_pts gives a quad over `rect` (same arithmetic), grown by `pad`: does it reach a row / column the
// frame will produce?  The scene front-end asks before it opens a clip: content under a mask that lies outside is invisible.
bool Context::rect_visible(const float rect[4], float pad) const {
  if (!(rect[2] > 0.0f) || !(rect[3] > 0.0f)) return false;
  const float vx[4] = {rect[0], rect[0] + rect[2], rect[0] + rect[2], rect[0]}, vy[4] = {rect[1] + rect[3], rect[1] + rect[3], rect[1], rect[1]};
  float minx = 0, maxx = 0, miny = 0, maxy = 0;
  for (int i = 0; i < 4; i++) {
    const float px = std::ceil(mat_.a * vx[i] + mat_.c * vy[i] + mat_.tx), py = std::ceil(mat_.b * vx[i] + mat_.d * vy[i] + mat_.ty);
    if (i == 0) { minx = maxx = px; miny = maxy = py; }
    else { minx = std::min(minx, px); maxx = std::max(maxx, px); miny = std::min(miny, py); maxy = std::max(maxy, py); }
  }
  const float lim = 1.0e6f;
  if (!(minx > -lim && maxx < lim && miny > -lim && maxy < lim)) return pad > 0.0f;  // (such a quad is recorded with empty bounds; an analytic mask has no quad)
  BBox b;
  b.x0 = (int16_t)clampf(minx - pad, 0.0f, (float)W_); b.x1 = (int16_t)clampf(maxx + pad, 0.0f, (float)W_);
  b.y0 = (int16_t)clampf(miny - pad, 0.0f, (float)H_); b.y1 = (int16_t)clampf(maxy + pad, 0.0f, (float)H_);
  return bbox_visible(b);
}

// Four pre-transform vertices in the reference's vertex order 0..3 (triangles (3,0,1) and (2,3,1), glcontext.nim:418-429).
bool Context::emit_quad_pts(DrawRec& r, const float vx[4], const float vy[4], int64_t* fragments) {
  float px[4], py[4];
  for (int i = 0; i < 4; i++) {
    px[i] = std::ceil(mat_.a * vx[i] + mat_.c * vy[i] + mat_.tx);
    py[i] = std::ceil(mat_.b * vx[i] + mat_.d * vy[i] + mat_.ty);
  }
  float minx = px[0], maxx = px[0], miny = py[0], maxy = py[0];
  for (int i = 1; i < 4; i++) {
    minx = std::min(minx, px[i]); maxx = std::max(maxx, px[i]);
    miny = std::min(miny, py[i]); maxy = std::max(maxy, py[i]);
  }
  const float lim = 1.0e6f;  // keep the integer edge functions far from overflow
  BBox b{0, 0, 0, 0};
  // A draw that reaches no pixel the frame will produce leaves no trace (it would never be binned).  Clip pushes stay: their
  // bounds grow to their content's, and a push that is not there would let that content through.
  const bool cullable = ((r.op_mode >> 12) & 15u) == OP_DRAW && culling();
  if (!(minx > -lim && maxx < lim && miny > -lim && maxy < lim)) {
    if (cullable) { FDH_CULLED(); return false; }
    r.bx0 = r.by0 = r.bx1 = r.by1 = 0; push_rec(r, b); return true;
  }
  b.x0 = (int16_t)clampf(minx, 0.0f, (float)W_); b.x1 = (int16_t)clampf(maxx, 0.0f, (float)W_);
  b.y0 = (int16_t)clampf(miny, 0.0f, (float)H_); b.y1 = (int16_t)clampf(maxy, 0.0f, (float)H_);
  if (cullable && !bbox_visible(b)) { FDH_CULLED(); return false; }
  r.bx0 = b.x0; r.by0 = b.y0; r.bx1 = b.x1; r.by1 = b.y1;
  const bool aligned = px[3] == px[0] && px[2] == px[1] && py[3] == py[2] && py[0] == py[1] && px[1] > px[0] && py[0] > py[3];
  if (aligned) {
    r.ox = px[3];
    r.oy = py[3];
    r.inv_w = 1.0f / (px[1] - px[0]);
    r.inv_h = 1.0f / (py[0] - py[3]);
    r.kx = 2.0f * r.p0 * r.inv_w;  // (meaningful for SDF quads, where p0, p1 are the quad's half extents)
    r.ky = 2.0f * r.p1 * r.inv_h;
    set_saturated_core(r, px[1] - px[0], py[0] - py[3]);
  } else {
    QuadExt q;
    std::memset(&q, 0, sizeof q);
    static const int TRI[2][3] = {{3, 0, 1}, {2, 3, 1}};  // glcontext.nim:418-429
    const uint32_t mode = r.op_mode & 255u;
    const bool atlas_mode = mode == 0u || (mode >= 13u && mode <= 16u);
    const float uax = atlas_mode ? r.r[0] : 0.0f, uay = atlas_mode ? r.r[1] : 0.0f, utx = atlas_mode ? r.r[2] : 1.0f, uty = atlas_mode ? r.r[3] : 1.0f;
    const float vu[4] = {uax, utx, utx, uax}, vv[4] = {uty, uty, uay, uay};
    for (int t = 0; t < 2; t++) {
      long long X[3], Y[3];
      for (int k = 0; k < 3; k++) { X[k] = 2 * (long long)px[TRI[t][k]]; Y[k] = 2 * (long long)py[TRI[t][k]]; }
      // edge k is opposite vertex k: from vertex (k+1)%3 to vertex (k+2)%3
      long long area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);
      const long long sgn = area2 >= 0 ? 1 : -1;
      for (int k = 0; k < 3; k++) {
        const int i0 = (k + 1) % 3, i1 = (k + 2) % 3;
        long long A = -(Y[i1] - Y[i0]) * sgn, B = (X[i1] - X[i0]) * sgn;
        long long C = -(B * Y[i0]) - (A * X[i0]);
        q.e[t][k].a = (int32_t)A; q.e[t][k].b = (int32_t)B; q.e[t][k].c = C;
        // top-left rule in image orientation (y down): owns iff top edge (horizontal, interior below) or left edge
        bool own;
        if (Y[i0] == Y[i1]) own = Y[k] > Y[i0];
        else {
          double tt = (double)(Y[k] - Y[i0]) / (double)(Y[i1] - Y[i0]);
          double ex = (double)X[i0] + tt * (double)(X[i1] - X[i0]);
          own = (double)X[k] > ex;
        }
        if (own) q.own |= 1u << (t * 3 + k);
      }
      if (area2 != 0) {
        // E0+E1+E2 is the same at every point: |area2| (each E_k equals it at vertex k, where the other two vanish)
        q.inv_sum[t] = (float)(1.0 / (double)(area2 * sgn));
        const double e1x = (double)(px[TRI[t][1]] - px[TRI[t][0]]), e1y = (double)(py[TRI[t][1]] - py[TRI[t][0]]);
        const double e2x = (double)(px[TRI[t][2]] - px[TRI[t][0]]), e2y = (double)(py[TRI[t][2]] - py[TRI[t][0]]);
        const double det = e1x * e2y - e1y * e2x;
        const double du1 = vu[TRI[t][1]] - vu[TRI[t][0]], du2 = vu[TRI[t][2]] - vu[TRI[t][0]];
        const double dv1 = vv[TRI[t][1]] - vv[TRI[t][0]], dv2 = vv[TRI[t][2]] - vv[TRI[t][0]];
        const double dudx = (du1 * e2y - du2 * e1y) / det, dudy = (du2 * e1x - du1 * e2x) / det;
        const double dvdx = (dv1 * e2y - dv2 * e1y) / det, dvdy = (dv2 * e1x - dv1 * e2x) / det;
        q.fw_u[t] = (float)(std::fabs(dudx) + std::fabs(dudy));
        q.fw_v[t] = (float)(std::fabs(dvdx) + std::fabs(dvdy));
        const double S = (double)atlas_size_;
        const double rho = std::max(std::sqrt(dudx * dudx + dvdx * dvdx), std::sqrt(dudy * dudy + dvdy * dvdy)) * S;
        q.lod[t] = rho > 0.0 ? (float)std::log2(rho) : 0.0f;
      } else {
        q.fw_u[t] = q.fw_v[t] = 1.0f;
      }
    }
    r.op_mode |= F_GENERAL;
    r.ext = (uint32_t)exts_.size();
    exts_.push_back(q);
  }
  if (fragments) *fragments += (int64_t)(b.x1 - b.x0) * (b.y1 - b.y0);
  for (auto idx : open_ops_) bbox_union(bboxes_[idx], b);  // clip pushes only need to reach tiles their content touches
  push_rec(r, b);
  return true;
}

// radii packing: glcontext.nim:745-817
static float clamp_radius(float r, float m) { return r <= 0.0f ? 0.0f : nim_round(std::max(1.0f, std::min(r, m))); }
static bool rounded_radii_vec(const float rx[4], const float ry[4], float hx, float hy, float out[4]) {
  enum { TL = 0, TR = 1, BL = 2, BR = 3 };
  bool circular = true;
  for (int i = 0; i < 4; i++) circular = circular && rx[i] == ry[i];
  static const int order[4] = {TR, BR, TL, BL};
  if (circular) {
    const float m = std::min(hx, hy);
    for (int k = 0; k < 4; k++) out[k] = clamp_radius(rx[order[k]], m);
    return false;
  }
  const float cm = std::min(hx, hy);
  for (int k = 0; k < 4; k++) {
    const int i = order[k];
    const float cx = clamp_radius(rx[i], hx), cy = clamp_radius(ry[i], hy);
    if (rx[i] == ry[i]) out[k] = -(clamp_radius(rx[i], cm) + 1.0f);
    else if (cx == cy) out[k] = -(cx + 1.0f);
    else {
      const float qx = nim_round(clampf(cx / std::max(hx, 0.000001f), 0.0f, 1.0f) * 4095.0f);
      const float qy = nim_round(clampf(cy / std::max(hy, 0.000001f), 0.0f, 1.0f) * 4095.0f);
      out[k] = qx + qy * 4096.0f;
    }
  }
  return true;
}

static void fill_sdf_rec(DrawRec& r, const float rect[4], const FdhColor colors[4], const float rx[4], const float ry[4], int mode,
                         float factor, float spread, const float shape[2], int fill_mode, FdhColor mid, FdhColor stop, float mid_pos,
                         float aa) {
  std::memset(&r, 0, sizeof r);
  const float w = rect[2], h = rect[3];
  const float qhx = w * 0.5f, qhy = h * 0.5f;
  const bool inset = mode == FDH_SDF_INSET_SHADOW;
  const bool has_shape = shape[0] > 0.0f && shape[1] > 0.0f;
  const float shx = inset ? qhx : (has_shape ? shape[0] : w) * 0.5f;
  const float shy = inset ? qhy : (has_shape ? shape[1] : h) * 0.5f;
  r.p0 = qhx; r.p1 = qhy;
  if (inset) { r.p2 = shape[0]; r.p3 = shape[1]; } else { r.p2 = shx; r.p3 = shy; }
  const bool ellip = rounded_radii_vec(rx, ry, shx, shy, r.r);
  r.f0 = factor;
  r.f1 = fill_mode == 0 ? spread : clampf(mid_pos, 0.01f, 0.99f);
  for (int i = 0; i < 4; i++) r.col[i] = pack_color(colors[i]);
  r.mid = pack_color(mid);
  r.stop = pack_color(stop);
  r.aa = aa;
  r.op_mode = (uint32_t)mode | (ellip ? F_ELLIP : 0u) | ((uint32_t)fill_mode << 9);
  if (r.col[0] == r.col[1] && r.col[1] == r.col[2] && r.col[2] == r.col[3]) r.op_mode |= F_SOLID;
}

// Host-only (no device is touched): the saturated core the submission path would attach to this draw under the identity
// transform.  Lets the CPU test-suite check the derivation against the oracle's pixels.
void saturated_core_of(const float rect[4], const float rx[4], const float ry[4], int mode, float factor, float spread,
                       const float shape[2], float aa, int out[4]) {
  const FdhColor white{255, 255, 255, 255}, zero{0, 0, 0, 0};
  const FdhColor cols[4] = {white, white, white, white};
  DrawRec r;
  fill_sdf_rec(r, rect, cols, rx, ry, mode, factor, spread, shape, 0, zero, zero, 0.5f, aa);
  const float x0 = std::ceil(rect[0]), y0 = std::ceil(rect[1]), x1 = std::ceil(rect[0] + rect[2]), y1 = std::ceil(rect[1] + rect[3]);
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!(x1 > x0 && y1 > y0)) return;
  r.ox = x0; r.oy = y0;
  r.inv_w = 1.0f / (x1 - x0); r.inv_h = 1.0f / (y1 - y0);
  set_saturated_core(r, x1 - x0, y1 - y0);
  out[0] = r.ix0; out[1] = r.iy0; out[2] = r.ix1; out[3] = r.iy1;
}

// drawRoundedRectSdfOpenGl: glcontext.nim:1449-1559
void Context::draw_rounded_rect_sdf(const float rect[4], const FdhColor colors[4], const float rx[4], const float ry[4], int mode,
                                    float factor, float spread, const float shape[2], int fill_mode, FdhColor mid, FdhColor stop,
                                    float mid_pos) {
  { FDH_REC("draw_rounded_rect_sdf").fv(rect, 4).cols(colors).fv(rx, 4).fv(ry, 4).i(mode).f(factor).f(spread).fv(shape, 2).i(fill_mode).col(mid).col(stop).f(mid_pos); }
  if (!frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  if (!(rect[2] > 0.0f) || !(rect[3] > 0.0f)) return;  // (a NaN extent draws nothing)
  if (mode >= FDH_SDF_BEZIER_STROKE_AA) throw Error(FDH_ERR_INVALID, "bezier stroke modes go through drawQuadraticBezierSdf");
  if (culling() && !rect_visible(rect, 0.0f)) { FDH_CULLED(); return; }  // (before the record is built: most of a long table is below the window)
  DrawRec r;
  fill_sdf_rec(r, rect, colors, rx, ry, mode, factor, spread, shape, fill_mode, mid, stop, mid_pos, aa_);
  if (mode == FDH_SDF_BACKDROP_BLUR) r.op_mode |= F_SELF_BACKDROP;  // a bare mode-17 call has no snapshot of its own
  emit_quad(r, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], &fragments_);
}

// fills: figbackend.nim:129-183
static FdhColor lerp_color(FdhColor a, FdhColor b, float t) {
  const float ct = clampf(t, 0.0f, 1.0f), it = 1.0f - ct;
  FdhColor r;
  r.r = (uint8_t)nim_round((float)a.r * it + (float)b.r * ct);
  r.g = (uint8_t)nim_round((float)a.g * it + (float)b.g * ct);
  r.b = (uint8_t)nim_round((float)a.b * it + (float)b.b * ct);
  r.a = (uint8_t)nim_round((float)a.a * it + (float)b.a * ct);
  return r;
}
static float mid_pos01(const FdhFill& f) { return clampf((float)f.mid_pos / 255.0f, 0.01f, 0.99f); }
FdhColor sample_fill(const FdhFill& f, float t) {
  if (f.kind == FDH_FILL_COLOR) return f.start;
  if (f.kind == FDH_FILL_LINEAR2) return lerp_color(f.start, f.stop, t);
  const float ct = clampf(t, 0.0f, 1.0f), mid = mid_pos01(f);
  if (ct <= mid) return lerp_color(f.start, f.mid, ct / mid);
  return lerp_color(f.mid, f.stop, (ct - mid) / (1.0f - mid));
}
void gradient_colors(const FdhFill& f, FdhColor out[4]) {  // vertex order BL,BR,TR,TL
  const int axis = f.kind == FDH_FILL_COLOR ? FDH_AXIS_X : f.axis;
  static const float T[4][4] = {{0, 1, 1, 0}, {1, 1, 0, 0}, {0.5f, 1, 0.5f, 0}, {0, 0.5f, 1, 0.5f}};
  for (int i = 0; i < 4; i++) out[i] = sample_fill(f, T[axis & 3][i]);
}

// drawRoundedRectSdf(fill: BackendFill): glcontext.nim:1581-1617
void Context::draw_rounded_rect_fill(const float rect[4], const FdhFill& fill, const float rx[4], const float ry[4], int mode,
                                     float factor, float spread, const float shape[2]) {
  const FdhColor zero{0, 0, 0, 0};
  if (fill.kind == FDH_FILL_LINEAR3 && (mode == FDH_SDF_CLIP_AA || mode == FDH_SDF_ANNULAR || mode == FDH_SDF_ANNULAR_AA)) {
    const FdhColor cols[4] = {fill.start, fill.start, fill.start, fill.start};
    draw_rounded_rect_sdf(rect, cols, rx, ry, mode, factor, spread, shape, 1 + (fill.axis & 3), fill.mid, fill.stop, mid_pos01(fill));
  } else {
    FdhColor cols[4];
    gradient_colors(fill, cols);
    draw_rounded_rect_sdf(rect, cols, rx, ry, mode, factor, spread, shape, 0, zero, zero, 0.5f);
  }
}

// drawImage / drawUvRect: glcontext.nim:1236-1302, 1350-1367
void Context::draw_image(int64_t key, const float pos[2], const FdhColor colors[4], const float size[2], bool flip_y) {
  { FDH_REC("draw_image").i(key).fv(pos, 2).cols(colors).fv(size, 2).i(flip_y ? 1 : 0); }
  if (!frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  auto it = entries_.find(key);
  if (it == entries_.end()) return;  // "missing image in context": warn + no-op (glcontext.nim:1310-1315)
  const AtlasEntry& e = it->second;
  const float S = (float)atlas_size_;
  const float ex = (float)e.x / S, ey = (float)e.y / S, ew = (float)e.w / S, eh = (float)e.h / S;  // entries = rect / atlasSize
  const bool sized = size[0] > 0.0f && size[1] > 0.0f;
  const float dw = sized ? size[0] : ew * S, dh = sized ? size[1] : eh * S;
  DrawRec r;
  std::memset(&r, 0, sizeof r);
  r.op_mode = FDH_SDF_ATLAS;
  if (flip_y) { r.r[0] = ex; r.r[1] = ey + eh; r.r[2] = ex + ew; r.r[3] = ey; }
  else { r.r[0] = ex; r.r[1] = ey; r.r[2] = ex + ew; r.r[3] = ey + eh; }
  for (int i = 0; i < 4; i++) r.col[i] = pack_color(colors[i]);
  if (r.col[0] == r.col[1] && r.col[1] == r.col[2] && r.col[2] == r.col[3]) r.op_mode |= F_SOLID;
  r.aa = aa_;
  if (subpixel_enabled_) {
    r.op_mode |= F_SUBPIXEL;
    r.aux = std::max(0.0f, std::min(subpixel_shift_, 0.999f));  // activeSubpixelShift glcontext.nim:819-822
  }
  // LOD for the axis-aligned form: rho = max(|du/dx|, |dv/dy|) in level-0 texels per pixel
  const float x0 = pos[0], y0 = pos[1], x1 = pos[0] + dw, y1 = pos[1] + dh;
  bool one_to_one = false;
  float qx0 = 0.0f, qy0 = 0.0f;
  {
    qx0 = std::ceil(mat_.a * x0 + mat_.tx);
    qy0 = std::ceil(mat_.d * y0 + mat_.ty);
    const float qx1 = std::ceil(mat_.a * x1 + mat_.tx), qy1 = std::ceil(mat_.d * y1 + mat_.ty);
    const float rw = std::fabs(qx1 - qx0), rh = std::fabs(qy1 - qy0);
    if (rw > 0.0f && rh > 0.0f) {
      const float rho = std::max(std::fabs(r.r[2] - r.r[0]) * S / rw, std::fabs(r.r[3] - r.r[1]) * S / rh);
      r.aux2 = rho > 0.0f ? std::log2(rho) : 0.0f;
    }
    // texels 1:1 on pixels (a glyph as renderText places it): the quad is as large as the image, upright, unshifted
    one_to_one = !flip_y && qx1 > qx0 && qy1 > qy0 && rw == (float)e.w && rh == (float)e.h && (!subpixel_enabled_ || r.aux == 0.0f) &&
                 mat_.b == 0.0f && mat_.c == 0.0f && std::fabs(qx0) < 1.0e6f && std::fabs(qy0) < 1.0e6f;
  }
  if (!emit_quad(r, x0, y0, x1, y1, &fragments_)) return;
  if (one_to_one && !(recs_.back().op_mode & F_GENERAL)) {
    DrawRec& rr = recs_.back();
    rr.op_mode |= F_TEXEL_1TO1;
    rr.ext = (uint32_t)(int32_t)(e.x - (int)qx0);   // texel x = pixel x + tdx
    rr._pad = (uint32_t)(int32_t)(e.y - (int)qy0);  // texel y = pixel y + tdy
  }
  // atlas.frag:284-295: the source alpha is texel alpha x vertex alpha -- 0 wherever all four taps have alpha 0.  (Level 0 only:
  // a minified image, aux2 > 0, takes its taps from coarser levels.)
  if (!(recs_.back().aux2 > 0.0f)) shrink_to_ink(e, true, 0);
}

// drawMsdfImage / drawMtsdfImage: glcontext.nim:1097-1155, drawUvRectAtlasSdf :1022-1093
void Context::draw_msdf(int64_t key, const float pos[2], FdhColor color, const float size[2], float px_range, float sd_threshold,
                        float stroke_weight, bool mtsdf, bool flip_y) {
  { FDH_REC("draw_msdf").i(key).fv(pos, 2).col(color).fv(size, 2).f(px_range).f(sd_threshold).f(stroke_weight).i(mtsdf ? 1 : 0).i(flip_y ? 1 : 0); }
  if (!frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  auto it = entries_.find(key);
  if (it == entries_.end()) return;
  const AtlasEntry& e = it->second;
  const float S = (float)atlas_size_;
  const float ex = (float)e.x / S, ey = (float)e.y / S, ew = (float)e.w / S, eh = (float)e.h / S;
  const float sw = std::max(0.0f, stroke_weight);
  DrawRec r;
  std::memset(&r, 0, sizeof r);
  r.op_mode = (uint32_t)(mtsdf ? (sw > 0.0f ? FDH_SDF_MTSDF_ANNULAR : FDH_SDF_MTSDF) : (sw > 0.0f ? FDH_SDF_MSDF_ANNULAR : FDH_SDF_MSDF)) | F_SOLID;
  if (flip_y) { r.r[0] = ex; r.r[1] = ey + eh; r.r[2] = ex + ew; r.r[3] = ey; }
  else { r.r[0] = ex; r.r[1] = ey; r.r[2] = ex + ew; r.r[3] = ey + eh; }
  r.p0 = S; r.p1 = sw;
  r.f0 = px_range; r.f1 = sd_threshold;
  for (int i = 0; i < 4; i++) r.col[i] = pack_color(color);
  r.aa = aa_;
  if (!emit_quad(r, pos[0], pos[1], pos[0] + size[0], pos[1] + size[1], &fragments_)) return;
  // atlas.frag:296-318, the fill variants: alpha = clamp(spr (sd - threshold) + 0.5), sd the median of the filtered r, g, b
  // (MTSDF: the filtered alpha) -- exactly 0 wherever sd <= threshold - 0.5 / spr.  Where all four taps have every channel <= t
  // the filtered channels, hence their median, are <= t: the box of texels above a level safely below that bound is all the
  // draw can touch.  (Stroke variants cover a band around the outline whatever sd is beyond it: left alone.)
  if (!(sw > 0.0f) && !(recs_.back().op_mode & F_GENERAL)) {
    const DrawRec& rr = recs_.back();
    const double unit = (double)px_range / (double)S;
    const double fw_u = std::fabs((double)(rr.r[2] - rr.r[0]) * rr.inv_w), fw_v = std::fabs((double)(rr.r[3] - rr.r[1]) * rr.inv_h);
    if (fw_u > 0.0 && fw_v > 0.0) {
      const double spr = std::max(0.5 * (unit / fw_u + unit / fw_v), 1.0);
      const double cut = (double)sd_threshold - 0.5 / spr - 0.008;  // two 8-bit steps below the bound (the kernel's rcp is good to 1e-7)
      if (cut > 0.0 && cut <= 1.0) shrink_to_ink(e, mtsdf, (int)std::floor(cut * 255.0) - 1);  // (NaN parameters: no shrink)
    }
  }
}

// drawQuadraticBezierSdf: glcontext.nim:1619-1741
void Context::draw_quadratic_bezier_sdf(const float rect[4], const FdhFill& fill, const float p0[2], const float p1[2],
                                        const float p2[2], float stroke_weight, int cap) {
  { FDH_REC("draw_quadratic_bezier_sdf").fv(rect, 4).fill(fill).fv(p0, 2).fv(p1, 2).fv(p2, 2).f(stroke_weight).i(cap); }
  if (!frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  if (!(rect[2] > 0.0f) || !(rect[3] > 0.0f) || !(stroke_weight > 0.0f)) return;
  DrawRec r;
  std::memset(&r, 0, sizeof r);
  r.p0 = rect[2] * 0.5f; r.p1 = rect[3] * 0.5f; r.p2 = p0[0]; r.p3 = p0[1];  // params = (quadHalf, p0)
  r.r[0] = p1[0]; r.r[1] = p1[1]; r.r[2] = p2[0]; r.r[3] = p2[1];             // "radii" slot = (p1, p2)
  FdhColor cols[4];
  uint32_t fill_mode = 0;
  if (fill.kind == FDH_FILL_LINEAR3) {
    fill_mode = 1u + (uint32_t)(fill.axis & 3);
    cols[0] = cols[1] = cols[2] = cols[3] = fill.start;
    r.mid = pack_color(fill.mid);
    r.stop = pack_color(fill.stop);
  } else {
    gradient_colors(fill, cols);
  }
  for (int i = 0; i < 4; i++) r.col[i] = pack_color(cols[i]);
  r.f0 = stroke_weight;
  r.f1 = fill_mode == 0 ? 0.0f : clampf(mid_pos01(fill), 0.01f, 0.99f);
  r.aa = aa_;
  const uint32_t mode = cap == FDH_CAP_BUTT ? FDH_SDF_BEZIER_STROKE_BUTT_AA
                                            : (cap == FDH_CAP_SQUARE ? FDH_SDF_BEZIER_STROKE_SQUARE_AA : FDH_SDF_BEZIER_STROKE_AA);
  r.op_mode = mode | (fill_mode << 9);
  if (r.col[0] == r.col[1] && r.col[1] == r.col[2] && r.col[2] == r.col[3]) r.op_mode |= F_SOLID;
  emit_quad(r, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], &fragments_);
}

// The 4x4 white "rect" atlas image drawRect / drawFilledQuad sample (glcontext.nim:966-970, 1411-1415); it takes
// atlas space on first use exactly like the reference's.
static constexpr int64_t kRectImageKey = 0x7265637452454354LL;
const AtlasEntry& Context::rect_entry() {
  auto it = entries_.find(kRectImageKey);
  if (it == entries_.end()) {
    uint8_t white[4 * 4 * 4];
    std::memset(white, 255, sizeof white);
    put_image(kRectImageKey, 4, 4, white, nullptr);
    it = entries_.find(kRectImageKey);
  }
  return it->second;
}
static void white_texel_uv(const AtlasEntry& e, int atlas_size, DrawRec& r) {
  const float S = (float)atlas_size;
  const float ex = (float)e.x / S, ey = (float)e.y / S, ew = (float)e.w / S, eh = (float)e.h / S;
  r.r[0] = r.r[2] = ex + ew / 2.0f;  // uvAt = uvTo = the image centre
  r.r[1] = r.r[3] = ey + eh / 2.0f;
}
// drawFilledQuad: glcontext.nim:963-982 (+ drawQuad :908-961): an arbitrary quad textured with one white texel
void Context::draw_filled_quad(const float verts[8], const FdhColor colors[4]) {
  { FDH_REC("draw_filled_quad").fv(verts, 8).cols(colors); }
  if (!frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  DrawRec r;
  std::memset(&r, 0, sizeof r);
  r.op_mode = FDH_SDF_ATLAS;
  white_texel_uv(rect_entry(), atlas_size_, r);
  for (int i = 0; i < 4; i++) r.col[i] = pack_color(colors[i]);
  if (r.col[0] == r.col[1] && r.col[1] == r.col[2] && r.col[2] == r.col[3]) r.op_mode |= F_SOLID;
  r.aa = aa_;
  const float vx[4] = {verts[0], verts[2], verts[4], verts[6]}, vy[4] = {verts[1], verts[3], verts[5], verts[7]};
  emit_quad_pts(r, vx, vy, &fragments_);
}
// drawRect: glcontext.nim:1410-1426
void Context::draw_rect(const float rect[4], FdhColor color) {
  { FDH_REC("draw_rect").fv(rect, 4).col(color); }
  if (!frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  DrawRec r;
  std::memset(&r, 0, sizeof r);
  r.op_mode = FDH_SDF_ATLAS | F_SOLID;
  white_texel_uv(rect_entry(), atlas_size_, r);
  for (int i = 0; i < 4; i++) r.col[i] = pack_color(color);
  r.aa = aa_;
  emit_quad(r, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], &fragments_);
}

// ------------------------------------------------------------------ masks (glcontext.nim:1873-1949)
void Context::begin_mask(const float rect[4], const float rx[4], const float ry[4]) {
  { FDH_REC("begin_mask").fv(rect, 4).fv(rx, 4).fv(ry, 4); }
  if (!frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  if (mask_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginMask has already been called.");
  mask_begun_ = true;
  mask_depth_++;  // (beyond kMaskDepth levels the compositor's stack spills to a global plane: Context::prepare)
  const FdhColor red{255, 0, 0, 255}, zero{0, 0, 0, 0};
  const FdhColor cols[4] = {red, red, red, red};
  const float shape[2] = {0, 0};
  DrawRec r;
  fill_sdf_rec(r, rect, cols, rx, ry, FDH_SDF_CLIP_AA, 4.0f, 0.0f, shape, 0, zero, zero, 0.5f, aa_);
  r.op_mode |= OP_MASK_PUSH << 12;
  const size_t before = recs_.size();
  if (rect[2] > 0.0f && rect[3] > 0.0f) {
    emit_quad(r, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], nullptr);
  } else {  // drawRoundedRectSdf returns early: the mask plane stays cleared to 0
    r.bx0 = r.by0 = r.bx1 = r.by1 = 0;
    push_rec(r, BBox{0, 0, 0, 0});
  }
  bboxes_[before] = BBox{0, 0, 0, 0};  // grows to the union of the content drawn under it
  open_ops_.push_back((uint32_t)before);
}
void Context::end_mask() {
  { FDH_REC("end_mask"); }
  if (!mask_begun_) throw Error(FDH_ERR_INVALID, "ctx.maskBegun has not been called.");
  mask_begun_ = false;
}
void Context::pop_mask() {
  { FDH_REC("pop_mask"); }
  if (mask_depth_ <= 0 || open_ops_.empty()) throw Error(FDH_ERR_INVALID, "popMask without beginMask");
  const uint32_t push_idx = open_ops_.back();
  open_ops_.pop_back();
  mask_depth_--;
  DrawRec r;
  std::memset(&r, 0, sizeof r);
  r.op_mode = OP_MASK_POP << 12;
  push_rec(r, bboxes_[push_idx]);
}
// makeRectMask glcontext.nim:831-850; beginRectMask :1932-1943
void Context::begin_rect_mask(const float rect[4], const float rx[4], const float ry[4]) {
  { FDH_REC("begin_rect_mask").fv(rect, 4).fv(rx, 4).fv(ry, 4); }
  if (!frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  if (mask_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginRectMask cannot start inside a mask.");
  if (rect_masks_.empty() && rect[2] > 0.0f && rect[3] > 0.0f) {
    const float hx = rect[2] * 0.5f, hy = rect[3] * 0.5f;
    DrawRec r;
    std::memset(&r, 0, sizeof r);
    const bool ellip = rounded_radii_vec(rx, ry, hx, hy, r.r);
    const float det = mat_.a * mat_.d - mat_.b * mat_.c, id = 1.0f / det;
    const float ia = mat_.d * id, ib = -mat_.b * id, ic = -mat_.c * id, idd = mat_.a * id;
    const float itx = -(ia * mat_.tx + ic * mat_.ty), ity = -(ib * mat_.tx + idd * mat_.ty);
    r.ox = ia; r.oy = ic; r.inv_w = itx;   // matX
    r.inv_h = ib; r.f0 = idd; r.f1 = ity;  // matY
    r.p0 = rect[0] + hx; r.p1 = rect[1] + hy; r.p2 = hx; r.p3 = hy;
    r.aa = aa_;
    r.op_mode = (OP_RMASK_BEGIN << 12) | (ellip ? F_ELLIP : 0u);
    open_ops_.push_back((uint32_t)recs_.size());
    push_rec(r, BBox{0, 0, 0, 0});
    rect_masks_.push_back(RectMaskEntry{1});
  } else {
    { const RecPause quiet(rec_on_);  // the fallback's own begin/end are this backend's business, not the caller's
      begin_mask(rect, rx, ry);
      end_mask(); }
    rect_masks_.push_back(RectMaskEntry{2});
  }
}
void Context::pop_rect_mask() {
  { FDH_REC("pop_rect_mask"); }
  if (rect_masks_.empty()) throw Error(FDH_ERR_INVALID, "No rect mask has been pushed.");
  const RectMaskEntry e = rect_masks_.back();
  rect_masks_.pop_back();
  if (e.kind == 2) { const RecPause quiet(rec_on_); pop_mask(); return; }
  const uint32_t begin_idx = open_ops_.back();
  open_ops_.pop_back();
  DrawRec r;
  std::memset(&r, 0, sizeof r);
  r.op_mode = OP_RMASK_END << 12;
  push_rec(r, bboxes_[begin_idx]);
}

// ------------------------------------------------------------------ backdrop blur (glcontext.nim:1743-1841, blur.frag:11-32)
static BlurTaps make_taps(float blur_radius) {
  BlurTaps t;
  std::memset(&t, 0, sizeof t);
  const float radius = clampf(blur_radius, 0.0f, 64.0f);
  const float sigma = std::max(0.5f * radius, 0.5f);
  const float step = std::max(radius / 8.0f, 1.0f);
  float w[17], wsum = 0.0f;
  for (int i = -8; i <= 8; i++) {
    const float x = (float)i * step;
    w[i + 8] = std::exp(-0.5f * (x * x) / (sigma * sigma));
    wsum += w[i + 8];
  }
  const float inv = 1.0f / std::max(wsum, 1e-5f);
  auto add = [&](int off, float c) {
    if (c == 0.0f) return;
    for (int k = 0; k < t.n; k++) if (t.off[k] == off) { t.coef[k] += c; return; }
    t.off[t.n] = off; t.coef[t.n] = c; t.n++;
  };
  for (int i = -8; i <= 8; i++) {
    const float x = (float)i * step;
    const float fl = std::floor(x), a = x - fl;
    add((int)fl, w[i + 8] * (1.0f - a) * inv);
    add((int)fl + 1, w[i + 8] * a * inv);
  }
  for (int k = 0; k < t.n; k++) t.reach = std::max(t.reach, std::abs(t.off[k]));
  for (int k = 0; k < t.n; k++) t.dense[kBlurPad + t.reach + t.off[k]] = t.coef[k];
  return t;
}

void Context::draw_backdrop_blur(const float rect[4], const float rx[4], const float ry[4], float blur_radius) {
  { FDH_REC("draw_backdrop_blur").fv(rect, 4).fv(rx, 4).fv(ry, 4).f(blur_radius); }
  if (!frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  if (!(blur_radius > 0.0f) || !(rect[2] > 0.0f) || !(rect[3] > 0.0f)) return;  // (written so that a NaN draws nothing)
  const FdhColor white{255, 255, 255, 255}, zero{0, 0, 0, 0};
  const FdhColor cols[4] = {white, white, white, white};
  const float shape[2] = {0, 0};
  DrawRec r;
  fill_sdf_rec(r, rect, cols, rx, ry, FDH_SDF_BACKDROP_BLUR, blur_radius, 0.0f, shape, 0, zero, zero, 0.5f, aa_);
  if (blur_radius <= 0.5f) {  // runBackdropSeparableBlur returns early: the snapshot is the live frame
    r.op_mode |= F_SELF_BACKDROP;
    emit_quad(r, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], &fragments_);
    return;
  }
  if (culling() && !rect_visible(rect, 0.0f)) { FDH_CULLED(); return; }  // a blurred backdrop nobody sees: no snapshot, no phase
  // A blurred snapshot is a barrier in painter's order: close the phase, blur, continue in a new phase.
  std::vector<DrawRec> reopen;
  for (auto idx : open_ops_) reopen.push_back(recs_[idx]);
  Phase next;
  next.first = (int)recs_.size();
  next.blur = (int)blurs_.size();
  phases_.push_back(next);
  for (size_t i = 0; i < reopen.size(); i++) {  // re-establish the open clip stack for the new phase
    open_ops_[i] = (uint32_t)recs_.size();
    push_rec(reopen[i], BBox{0, 0, 0, 0});
  }
  const bool fuse = open_ops_.empty();  // no clip state to carry: the V pass can composite the quad itself
  if (!emit_quad(r, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], &fragments_)) throw Error(FDH_ERR_INVALID, "drawBackdropBlur: rect_visible and emit_quad disagree");
  const BBox fb = bboxes_.back();
  BlurJob job;
  job.fuse_draw = -1;
  if (fuse) {
    job.fuse_draw = (int)recs_.size() - 1;
    bboxes_.back() = BBox{0, 0, 0, 0};  // never binned: k_blur_v blends it
  }
  job.radius = blur_radius;
  job.x0 = fb.x0; job.y0 = fb.y0; job.x1 = fb.x1; job.y1 = fb.y1;
  job.taps = make_taps(blur_radius);
  blurs_.push_back(job);
}

// ------------------------------------------------------------------ submission
// end_frame = prepare (this thread) + issue (the context's submit thread).
//   prepare  builds everything the device needs from the recorded frame -- list stride, bin records, bin boxes, the phase table --
//            straight into a pinned staging buffer, and a LaunchJob describing the launches.  It runs on the CALLING thread:
//            the records it reads were written microseconds ago by this very core.  (First version: the whole submission on the
//            other thread.  Recording then took 120 us instead of 40: every record line the caller wrote had last been read by
//            the submit thread's core, and a line costs ~100 ns to pull across the host's core complexes.)
//   issue    launches the upload kernel and the frame's kernels (~20 us of HIP runtime calls) from the submit thread, so the
//            caller is already walking the next frame's tree.  FDH_CREATE_SYNC_SUBMIT contexts run it inline.
void Context::end_frame() {  // glcontext.nim:1982-1989
  { FDH_REC("end_frame"); }
  if (!frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame was not called first.");
  if (mask_depth_ != 0) throw Error(FDH_ERR_INVALID, "Not all masks have been popped.");
  if (!rect_masks_.empty()) throw Error(FDH_ERR_INVALID, "Not all rect masks have been popped.");
  frame_begun_ = false;
  const auto t1 = std::chrono::steady_clock::now();
  host_record_ms_ = std::chrono::duration<float, std::milli>(t1 - t_begin_frame_).count();
  // a list entry carries the draw index in 25 bits beside its path code and flags (k_bin_draws, LE_INDEX)
  if (recs_.size() >= LE_INDEX) throw Error(FDH_ERR_INVALID, "more than 33 554 430 draw records in one frame");
  if (host_only_) return;
  prepare(next_);
  drain();  // the previous frame's launches (normally long issued: they ran while this frame was being recorded)
  std::swap(job_, next_);
  have_frame_ = true;
  if (!worker_.joinable()) { issue(job_); return; }
  {
    std::lock_guard<std::mutex> lk(mu_);
    pending_.store(true, std::memory_order_release);
  }
  cv_job_.notify_one();
}

// Weight fragments of a matrix-pipe blur pass (k_blur_mx, fdh_kernels.hip).  Lane (j, g) of fragment m holds, for the window
// texels 16 m + 8 g + t (t = 0..7) of a 32-output block, the tap each meets at output j: k = texel - delta - j, weight
// dense[k] * 2^10 when 0 <= k <= 2 reach, else 0 -- split into two halves hi + lo (hi = RNE(w), lo = RNE(w - hi): 22
// significant bits, every product with an 8-bit texel is exact in f32).
static uint16_t half_bits_rne(float f) {  // |f| < 65504
  uint32_t u;
  std::memcpy(&u, &f, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  const float a = std::fabs(f);
  if (a == 0.0f) return sign;
  std::memcpy(&u, &a, 4);
  const int e = (int)(u >> 23) - 127;
  if (e < -14) return sign | (uint16_t)std::nearbyint(a * 16777216.0f);  // subnormal half: units of 2^-24 (1024 = the smallest normal)
  const uint32_t mant = u & 0x7fffffu, m = mant >> 13, rem = mant & 0x1fffu;
  uint32_t h = ((uint32_t)(e + 15) << 10) | m;
  if (rem > 0x1000u || (rem == 0x1000u && (m & 1u))) h++;  // round to nearest even; a carry moves into the exponent
  return sign | (uint16_t)h;
}
static float half_value(uint16_t h) {
  const int e = (h >> 10) & 31, m = h & 1023;
  const float v = e == 0 ? std::ldexp((float)m, -24) : std::ldexp((float)(m + 1024), e - 25);
  return (h & 0x8000u) ? -v : v;
}
static void build_mx_weights(const BlurTaps& t, bool vertical, uint8_t* out) {
  const int nk = mx_nk(t.reach, vertical), delta = mx_delta(t.reach, vertical);
  uint16_t* o = reinterpret_cast<uint16_t*>(out);
  for (int m = 0; m < nk; m++)
    for (int lane = 0; lane < 64; lane++) {
      const int j = lane & 31, g = lane >> 5;
      for (int e = 0; e < 8; e++) {
        const int k = 16 * m + 8 * g + e - delta - j;
        const float w = (k >= 0 && k <= 2 * t.reach) ? t.dense[kBlurPad + k] * 1024.0f : 0.0f;
        const uint16_t hi = half_bits_rne(w), lo = half_bits_rne(w - half_value(hi));
        o[(((size_t)(2 * m) * 64 + lane) * 8) + e] = hi;
        o[(((size_t)(2 * m + 1) * 64 + lane) * 8) + e] = lo;
      }
    }
}

void blur_weight_fragments(float blur_radius, bool vertical, float* dense, uint16_t* frag_bits, int* reach, int* k_steps) {
  const BlurTaps t = make_taps(blur_radius);
  for (int k = 0; k <= 2 * t.reach; k++) dense[k] = t.dense[kBlurPad + k];
  *reach = t.reach;
  *k_steps = mx_nk(t.reach, vertical);
  if (*k_steps > kMxMaxNK) throw Error(FDH_ERR_INVALID, "blur_weight_fragments: filter too wide");
  build_mx_weights(t, vertical, reinterpret_cast<uint8_t*>(frag_bits));
}

// a device buffer the frame in submission may still use: wait for it before the block moves
template <typename Buf> void Context::reserve_quiet(Buf& buf, size_t n) {
  if (n <= buf.cap) return;
  drain();
  FDH_HIP(hipStreamSynchronize(stream_));
  buf.reserve(n);
}

void Context::prepare(LaunchJob& J) {
  const auto t_s0 = std::chrono::steady_clock::now();
  FDH_HIP(hipSetDevice(device_));
  const size_t n = recs_.size();
  J.W = W_; J.H = H_; J.clear = clear_; J.clear_rgba8 = clear_rgba8_;
  J.rec_y0 = culling() ? cull_y0_ : 0; J.rec_y1 = culling() ? cull_y1_ : H_;
  J.phases = phases_;  // (copies: the recording side keeps its own for fdh_debug_record_digest)
  J.blurs = blurs_;
  J.n_recs = (int)n;
  int& bins_x_ = J.bins_x; int& bins_y_ = J.bins_y; int& list_stride_ = J.list_stride; int& binbox_shift_ = J.binbox_shift; int& big_blur_ = J.big_blur;
  std::vector<const uint4*>&mx_w_h_ = J.mx_w_h, &mx_w_v_ = J.mx_w_v;
  LaunchJob::View& dv_ = J.dv;
  bins_x_ = (W_ + kBin - 1) / kBin;
  bins_y_ = (H_ + kBin - 1) / kBin;
  const int nb = bins_x_ * bins_y_;
  // List stride = the largest number of draws any bin of any phase can receive, counted exactly with a 2-D difference
  // array (O(draws + bins) per phase).  Sizing the lists for "every draw of the phase in every bin" cost 163 MB for
  // the 10 001-draw glyph frame; the exact bound is 2040 bins x a few dozen entries.
  int max_count = 1;
  std::vector<int>& diff = diff_scratch_;
  const int dw = bins_x_ + 1;
  diff.resize((size_t)dw * (bins_y_ + 1));
  constexpr int kBinShift = 6;
  static_assert((1 << kBinShift) == kBin, "bins are 64 px");
  int deepest_clip = 0;  // counted from the records (a retained scene splices cached records in: no begin_mask call sees them)
  int64_t frag_mode[4] = {0, 0, 0, 0}, frag_ellip = 0, frag_other = 0;  // phase 0, by SdfMode 3 / 7 / 9 / 12 (SURVEY.md 8d flop table)
  for (size_t pi = 0; pi < J.phases.size(); pi++) {
    Phase& p = J.phases[pi];
    // one pass over the phase's draws: union of the bounds, which compositor build the phase needs, fragment counts by mode
    BBox u{0, 0, 0, 0};
    p.has_slow = false;
    p.has_atlas = false;
    p.has_masks = false;
    int depth = 0;  // clip nesting inside the phase (open pushes are re-emitted at a phase's start)
    for (int i = p.first; i < p.first + p.count; i++) {
      const BBox& b = bboxes_[i];
      bbox_union(u, b);
      // mirrors the `fast` predicate of k_composite_tiles
      const uint32_t om = recs_[i].op_mode, op = (om >> 12) & 15u, mode = om & 255u;
      const bool atlas_mode = mode == 0u || (mode >= 13u && mode <= 16u);
      if (op != OP_DRAW) p.has_masks = true;
      if (op == OP_MASK_PUSH) { depth++; deepest_clip = std::max(deepest_clip, depth); } else if (op == OP_MASK_POP) depth--;
      // mirrors the path selection of k_composite_tiles: 4-wide atlas path for axis-aligned atlas quads sampled from level 0
      const bool atlas4 = atlas_mode && !(om & F_GENERAL) && op == OP_DRAW && !(mode == 0u && recs_[i].aux2 > 0.0f && n_levels_ >= 2);
      if (atlas4) p.has_atlas = true;
      // (a rect mask under a rotated transform -- matY.x != 0 -- is set up one pixel slot at a time; an upright one runs 4-wide)
      else if ((op == OP_RMASK_BEGIN && recs_[i].inv_h != 0.0f) || ((op == OP_DRAW || op == OP_MASK_PUSH) && ((om & F_GENERAL) || atlas_mode || mode >= 18u))) p.has_slow = true;
      if (pi == 0 && op == OP_DRAW && !bbox_empty(b)) {
        const int64_t area = (int64_t)(b.x1 - b.x0) * (b.y1 - b.y0);
        if (mode == 3u) frag_mode[0] += area; else if (mode == 7u) frag_mode[1] += area; else if (mode == 9u) frag_mode[2] += area;
        else if (mode == 12u) frag_mode[3] += area; else frag_other += area;
        if (om & F_ELLIP) frag_ellip += area;
      }
    }
    p.bin_x0 = u.x0 >> kBinShift; p.bin_y0 = u.y0 >> kBinShift;
    p.bin_x1 = bbox_empty(u) ? p.bin_x0 : (u.x1 + kBin - 1) >> kBinShift;
    p.bin_y1 = bbox_empty(u) ? p.bin_y0 : (u.y1 + kBin - 1) >> kBinShift;
    if (bbox_empty(u)) continue;
    // the count, over the bins the phase reaches only (the phases behind a blur are a handful of draws on a few bins)
    const int cx0 = p.bin_x0, cy0 = p.bin_y0, cx1 = p.bin_x1, cy1 = p.bin_y1;  // cells [cx0, cx1] x [cy0, cy1] of the difference array
    for (int y = cy0; y <= cy1; y++) std::fill(diff.begin() + (size_t)y * dw + cx0, diff.begin() + (size_t)y * dw + cx1 + 1, 0);
    for (int i = p.first; i < p.first + p.count; i++) {
      const BBox& b = bboxes_[i];
      if (bbox_empty(b)) continue;
      const int bx0 = b.x0 >> kBinShift, by0 = b.y0 >> kBinShift, bx1 = ((b.x1 - 1) >> kBinShift) + 1, by1 = ((b.y1 - 1) >> kBinShift) + 1;  // [bx0,bx1) x [by0,by1)
      diff[(size_t)by0 * dw + bx0]++; diff[(size_t)by0 * dw + bx1]--; diff[(size_t)by1 * dw + bx0]--; diff[(size_t)by1 * dw + bx1]++;
    }
    for (int y = cy0; y < cy1; y++) {
      int run = 0;
      int* row = diff.data() + (size_t)y * dw;
      const int* above = y > cy0 ? row - dw : nullptr;
      for (int x = cx0; x < cx1; x++) {
        run += row[x];
        const int cell = run + (above ? above[x] : 0);  // column prefix over the row prefixes
        row[x] = cell;
        max_count = std::max(max_count, cell);
      }
    }
  }
  list_stride_ = (max_count + 7) & ~7;
  {
    reserve_quiet(d_lists_, (size_t)J.phases.size() * nb * list_stride_);
    reserve_quiet(d_counts_, (size_t)J.phases.size() * nb);
    J.lists = d_lists_.ptr; J.counts = d_counts_.ptr;
    // clip nesting beyond the LDS stack (kMaskDepth levels): one global plane per extra level, 256 bytes per strip
    J.mask_spill = nullptr;
    J.spill_stride = (size_t)nb * 16 * 64;
    if (deepest_clip > kMaskDepth) {
      const size_t levels = (size_t)(deepest_clip - kMaskDepth);
      if (levels * J.spill_stride * sizeof(uint32_t) > ((size_t)2 << 30))
        throw Error(FDH_ERR_UNSUPPORTED, "clip masks nested too deep for this frame size (the spill plane would exceed 2 GiB)");
      reserve_quiet(d_mask_spill_, levels * J.spill_stride);
      J.mask_spill = d_mask_spill_.ptr;
    }
    std::vector<int> pf(J.phases.size() + 1);
    for (size_t i = 0; i < J.phases.size(); i++) pf[i] = J.phases[i].first;
    pf[J.phases.size()] = (int)n;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t b_recs = n * sizeof(DrawRec), b_ext = exts_.size() * sizeof(QuadExt), b_bb = n * sizeof(BinRec), b_pf = pf.size() * sizeof(int);
    const size_t b_box = ((n + 3) & ~(size_t)3) * sizeof(uint32_t);
    const size_t n_chunks = (n + 255) / 256, b_chunk = std::max<size_t>(n_chunks, 1) * sizeof(uint32_t);
    const size_t o_recs = 0, o_ext = up(o_recs + b_recs), o_bb = up(o_ext + b_ext), o_box = up(o_bb + b_bb), o_chunk = up(o_box + b_box),
                 o_pf = up(o_chunk + b_chunk);
    // weight fragments of the matrix-pipe blur passes, two tables (H, V) per blur job
    std::vector<size_t> o_mxh(J.blurs.size(), 0), o_mxv(J.blurs.size(), 0);
    size_t total = up(o_pf + b_pf);
    for (size_t i = 0; i < J.blurs.size(); i++) {
      const int nkh = mx_nk(J.blurs[i].taps.reach, false), nkv = mx_nk(J.blurs[i].taps.reach, true);
      if (nkh > kMxMaxNK || nkv > kMxMaxNK) continue;
      o_mxh[i] = total; total = up(total + mx_table_bytes(nkh));
      o_mxv[i] = total; total = up(total + mx_table_bytes(nkv));
    }
    reserve_quiet(d_frame_, total);
    dv_.recs = reinterpret_cast<DrawRec*>(d_frame_.ptr + o_recs);
    dv_.exts = reinterpret_cast<QuadExt*>(d_frame_.ptr + o_ext);
    dv_.binrecs = reinterpret_cast<BinRec*>(d_frame_.ptr + o_bb);
    dv_.phase_first = reinterpret_cast<int*>(d_frame_.ptr + o_pf);
    dv_.binbox = reinterpret_cast<uint32_t*>(d_frame_.ptr + o_box);
    dv_.chunkbox = reinterpret_cast<uint32_t*>(d_frame_.ptr + o_chunk);
    mx_w_h_.assign(J.blurs.size(), nullptr);
    mx_w_v_.assign(J.blurs.size(), nullptr);
    for (size_t i = 0; i < J.blurs.size(); i++)
      if (o_mxh[i]) { mx_w_h_[i] = reinterpret_cast<const uint4*>(d_frame_.ptr + o_mxh[i]); mx_w_v_[i] = reinterpret_cast<const uint4*>(d_frame_.ptr + o_mxv[i]); }
    // A blur node that covers the whole frame, composited by its own vertical pass (no clip open), in a frame that starts from
    // the clear colour: both passes as ONE kernel, out of place (k_blur_fx) -- launch_frame alternates between fb_ and alt_.
    // Which route is a matter of speed only -- the two give the same pixels bit for bit (tests/test_hip_parity.py).  The fused
    // kernel moves half the bytes and shortens a frame rendered ALONE (4K bench frame: both passes 38.5 -> 34 us), but it
    // re-filters 40 % more rows horizontally (segment halos) and holds 22 KB of LDS and 249 VGPRs per wave: with other contexts'
    // frames in flight on the GPU, where total work is what counts, the two-pass route is 3 % faster.  So: fused when no other
    // context of this process has submitted a frame in the last millisecond (FDH_BLUR_FUSED=1 always, =0 never).
    static const int fx_env = [] { const char* e = std::getenv("FDH_BLUR_FUSED"); return e ? (std::atoi(e) != 0 ? 1 : 0) : -1; }();
    const int route = blur_route_ >= 0 ? blur_route_ : fx_env;
    bool fx_on = route != 0;
    {
      const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
      if (route < 0)
        for (int k = 0; k < kSubmitSlots; k++)
          if (k != submit_slot_ && now - g_last_submit_ns[k].load(std::memory_order_relaxed) < 1000000) { fx_on = false; break; }
      g_last_submit_ns[submit_slot_].store(now, std::memory_order_relaxed);
    }
    J.blur_fused.assign(J.blurs.size(), 0);
    J.n_fused = 0;
    for (size_t i = 0; i < J.blurs.size(); i++) {
      const BlurJob& j = J.blurs[i];
      // (frames under 0.4 Mpx keep the two small-region passes, like every region of that size: launch_blur_h)
      static const bool any_size = [] { const char* e = std::getenv("FDH_FORCE_BLUR_PATH"); return e && std::atoi(e) == 3; }();
      if (fx_on && clear_ && j.fuse_draw >= 0 && mx_w_h_[i] && j.x0 == 0 && j.y0 == 0 && j.x1 == W_ && j.y1 == H_ && blur_fused_supported(j.taps.reach, W_, W_) &&
          (any_size || (long long)W_ * H_ >= 384 * 1024)) {
        J.blur_fused[i] = 1;
        J.n_fused++;
      }
    }
    if (J.n_fused > 0 && !alt_) {
      FDH_HIP(hipMalloc((void**)&alt_, (size_t)W_ * H_ * 4));
      FDH_HIP(hipMemsetAsync(alt_, 0, (size_t)W_ * H_ * 4, stream_));
    }
    const int slot = staging_i_;
    staging_i_ = (staging_i_ + 1) % kStaging;
    // its copy of kStaging frames ago (that frame's issue was waited for by the end_frame after it: the event is recorded)
    if (staging_busy_[slot]) FDH_HIP(hipEventSynchronize(staging_ev_[slot]));
    staging_[slot].reserve(total);
    J.staging_slot = slot;
    uint8_t* s = staging_[slot].ptr;
    if (b_recs) std::memcpy(s + o_recs, recs_.data(), b_recs);
    {  // The device's copy of a one-colour upright SDF draw carries the colour once more as three floats, c / 255, in the slots of
       // the three redundant vertex colours: the compositor's uniform-blend and packed edge paths (the only readers: LE_PLAIN and
       // the path codes are given to exactly these records below) take them as they are instead of converting and scaling three
       // bytes per strip.  The same IEEE product the kernels formed (one multiply by the float 1 / 255): bit-identical frames.
      DrawRec* dr = reinterpret_cast<DrawRec*>(s + o_recs);
      const float inv255 = 1.0f / 255.0f;
      for (size_t i = 0; i < n; i++) {
        const uint32_t om = dr[i].op_mode, mode = om & 255u;
        const bool atlas_mode = mode == 0u || (mode >= 13u && mode <= 16u);
        if ((om & F_GENERAL) || atlas_mode || mode >= 18u || ((om >> 12) & 15u) != OP_DRAW || !(om & F_SOLID)) continue;
        const uint32_t c = dr[i].col[0];
        const float u[3] = {(float)(c & 255u) * inv255, (float)((c >> 8) & 255u) * inv255, (float)((c >> 16) & 255u) * inv255};
        std::memcpy(&dr[i].col[1], u, sizeof u);
      }
    }
    if (b_ext) std::memcpy(s + o_ext, exts_.data(), b_ext);
    {  // what the bin kernel reads of a draw: bounds, saturated core, and the bin-independent part of its list entries
      BinRec* br = reinterpret_cast<BinRec*>(s + o_bb);
      for (size_t i = 0; i < n; i++) {
        const DrawRec& r = recs_[i];
        const uint32_t om = r.op_mode, op = (om >> 12) & 15u, mode = om & 255u, fill_mode = (om >> 9) & 7u;
        const bool atlas_mode = mode == 0u || (mode >= 13u && mode <= 16u);
        const bool sdf = !(om & F_GENERAL) && !atlas_mode && mode < 18u && (op == OP_DRAW || op == OP_MASK_PUSH);
        const uint32_t ell = (om & F_ELLIP) ? 4u : 0u;
        uint32_t flags = 0;
        if (sdf) {
          flags |= BR_HAS_CORE;
          if (op == OP_DRAW && (mode == 9u || mode == 11u || mode == 12u)) {
            flags |= BR_CORE_REMOVED;  // the stroke's interior, or so deep inside an inner shadow that no 8-bit channel moves
            if ((om & F_SOLID) && fill_mode == 0u && mode != 11u) flags |= ((mode == 9u ? 3u : 4u) + ell) << LE_PATH_SHIFT;
          } else {
            if (op == OP_DRAW && (om & F_SOLID) && fill_mode == 0u && mode != 17u) {
              flags |= LE_PLAIN;
              const uint32_t code = mode == 3u ? 1u : mode == 7u ? 2u : 0u;
              if (code) flags |= (code + ell) << LE_PATH_SHIFT;
            }
            if (op == OP_DRAW && mode == 3u) {
              uint32_t a = r.col[0] & r.col[1] & r.col[2] & r.col[3];
              if (fill_mode != 0u) a &= r.mid & r.stop;
              if ((a >> 24) == 255u) flags |= LE_OPAQUE;
            }
          }
        }
        // the next draw shares this one's distance field (LE_SHARE): both one-colour fill / stroke / inner shadow (path codes
        // 1, 3, 4 and their elliptical twins) over the same quad, radii and AA factor, the same shape half extents
        if (((flags >> LE_PATH_SHIFT) & 15u) != 0u && mode != 7u && i + 1 < n) {
          const DrawRec& b = recs_[i + 1];
          const uint32_t omb = b.op_mode, modeb = omb & 255u;
          const bool simple_b = ((omb >> 12) & 15u) == OP_DRAW && !(omb & F_GENERAL) && (omb & F_SOLID) && ((omb >> 9) & 7u) == 0u &&
                                (modeb == 3u || modeb == 9u || modeb == 12u) && ((omb ^ om) & F_ELLIP) == 0u;
          bool same_phase = false;
          for (const Phase& ph : J.phases) if ((int)i >= ph.first && (int)i + 1 < ph.first + ph.count) same_phase = true;
          if (simple_b && same_phase && std::memcmp(&r.ox, &b.ox, 6 * sizeof(float)) == 0 && std::memcmp(r.r, b.r, sizeof r.r) == 0 &&
              std::memcmp(&r.bx0, &b.bx0, 4 * sizeof(int16_t)) == 0 && r.aa == b.aa) {
            const float sax = mode == 9u ? r.p0 : r.p2, say = mode == 9u ? r.p1 : r.p3, sbx = modeb == 9u ? b.p0 : b.p2, sby = modeb == 9u ? b.p1 : b.p3;
            if (sax == sbx && say == sby) flags |= LE_SHARE;
          }
        }
        br[i] = BinRec{bboxes_[i], r.ix0, r.iy0, r.ix1, r.iy1, flags, 0u};
      }
    }
    {  // bin boxes (what k_bin_draws scans): 7-bit inclusive bounds in bin units, upper bounds complemented
      uint32_t* bxp = reinterpret_cast<uint32_t*>(s + o_box);
      binbox_shift_ = (bins_x_ > 128 || bins_y_ > 128) ? 1 : 0;
      const int ush = kBinShift + binbox_shift_;  // (bounds are clipped to the frame: non-negative, a shift divides)
      for (size_t i = 0; i < b_box / sizeof(uint32_t); i++) {
        uint32_t v = 0x7f7f7f7fu;  // x0 = y0 = 127, x1 = y1 = 0: never hits
        if (i < n && !bbox_empty(bboxes_[i])) {
          const BBox& b = bboxes_[i];
          v = (uint32_t)(b.x0 >> ush) | ((uint32_t)(b.y0 >> ush) << 8) | ((127u - (uint32_t)((b.x1 - 1) >> ush)) << 16) |
              ((127u - (uint32_t)((b.y1 - 1) >> ush)) << 24);
        }
        bxp[i] = v;
      }
      // second level: the union box of every 256 draws = byte-wise min (x0, y0 min; 127 - x1, 127 - y1 min)
      uint32_t* cb = reinterpret_cast<uint32_t*>(s + o_chunk);
      for (size_t c = 0; c < std::max<size_t>(n_chunks, 1); c++) {
        uint32_t m = 0x7f7f7f7fu;
        for (size_t i = c * 256; i < std::min((c + 1) * 256, n); i++) {
          const uint32_t v = bxp[i];
          uint32_t o = 0;
          for (int sh = 0; sh < 32; sh += 8) o |= std::min((m >> sh) & 255u, (v >> sh) & 255u) << sh;
          m = o;
        }
        cb[c] = m;
      }
    }
    std::memcpy(s + o_pf, pf.data(), b_pf);
    // The weight tables depend on the filters alone and sit behind everything else in the block: when the device block already
    // holds these very tables at these very offsets (an animation blurs with the same radii frame after frame) they are neither
    // staged nor uploaded again -- 40 of the bench frame's 130 KB.
    std::vector<size_t> layout{total, o_recs, o_ext, o_bb, o_box, o_chunk, o_pf};
    for (size_t i = 0; i < J.blurs.size(); i++) { layout.push_back(o_mxh[i]); layout.push_back(o_mxv[i]); }
    std::vector<float> tables_sig;
    for (size_t i = 0; i < J.blurs.size(); i++)
      if (o_mxh[i]) { const BlurTaps& t = J.blurs[i].taps; tables_sig.push_back((float)t.reach); tables_sig.insert(tables_sig.end(), t.dense + kBlurPad, t.dense + kBlurPad + 2 * t.reach + 1); }
    const size_t o_tables = up(o_pf + b_pf);
    // (the diff route of retained scenes compares against a host shadow of the WHOLE block: without a valid shadow the tables
    // are staged once more so that one can be taken)
    if (!rec_diff_upload_) shadow_dev_ = nullptr;
    const bool shadow_ok = rec_diff_upload_ && shadow_dev_ == d_frame_.ptr && shadow_layout_ == layout && shadow_.size() == total;
    const bool tables_resident = tables_dev_ == d_frame_.ptr && tables_layout_ == layout && tables_sig_ == tables_sig && (!rec_diff_upload_ || shadow_ok);
    for (size_t i = 0; i < J.blurs.size() && !tables_resident; i++)
      if (o_mxh[i]) {
        // the fragments depend on the filter alone: an animation blurs with the same radii frame after frame, and building
        // the four tables of the bench frame took 35 of the 54 us this function spent before its first launch
        const BlurTaps& t = J.blurs[i].taps;
        const size_t bh = mx_table_bytes(mx_nk(t.reach, false)), bv = mx_table_bytes(mx_nk(t.reach, true));
        const MxTables* hit = nullptr;
        for (const MxTables& c : mx_cache_)
          if (c.reach == t.reach && std::memcmp(c.dense.data(), t.dense + kBlurPad, sizeof(float) * (2 * t.reach + 1)) == 0) { hit = &c; break; }
        if (!hit) {
          if (mx_cache_.size() >= 8) mx_cache_.erase(mx_cache_.begin());
          MxTables c;
          c.reach = t.reach;
          c.dense.assign(t.dense + kBlurPad, t.dense + kBlurPad + 2 * t.reach + 1);
          c.h.resize(bh); c.v.resize(bv);
          build_mx_weights(t, false, c.h.data());
          build_mx_weights(t, true, c.v.data());
          mx_cache_.push_back(std::move(c));
          hit = &mx_cache_.back();
        }
        std::memcpy(s + o_mxh[i], hit->h.data(), bh);
        std::memcpy(s + o_mxv[i], hit->v.data(), bv);
      }
    void* s_dev = nullptr;
    FDH_HIP(hipHostGetDevicePointer(&s_dev, s, 0));
    // Only what differs from the block the device already holds travels: after an edit of a retained scene (or between two
    // frames of an animation) that is a few hundred bytes of records, bounds and bin boxes out of ~110 KB.  The comparison runs
    // against a host shadow of the device block in 256-byte chunks; up to kUploadRuns runs go out in ONE launch.
    bool patched = false;
    J.s_dev = s_dev; J.d_dst = d_frame_.ptr; J.runs = UploadRuns{};
    // (a frame recorded from scratch differs from its predecessor nearly everywhere: comparing 110 KB to find that out, and
    // keeping the shadow current, cost 12 us per frame -- only frames of a retained scene take the diff route)
    const size_t staged = tables_resident ? o_tables : total;  // bytes of the staging buffer that hold this frame
    if (shadow_ok) {
      UploadRuns R{};
      size_t dirty = 0;
      bool fits = true;
      for (size_t at = 0; at < staged && fits; at += 256) {
        const size_t len = std::min<size_t>(256, staged - at);
        if (std::memcmp(shadow_.data() + at, s + at, len) == 0) continue;
        dirty += len;
        if (R.n > 0 && (size_t)(R.off16[R.n - 1] + R.len16[R.n - 1]) * 16 == at) R.len16[R.n - 1] += (uint32_t)((len + 15) / 16);
        else if (R.n < (uint32_t)kUploadRuns) { R.off16[R.n] = (uint32_t)(at / 16); R.len16[R.n] = (uint32_t)((len + 15) / 16); R.n++; }
        else fits = false;
      }
      if (fits && dirty * 2 < total) {
        J.runs = R;
        for (uint32_t r = 0; r < R.n; r++) std::memcpy(shadow_.data() + (size_t)R.off16[r] * 16, s + (size_t)R.off16[r] * 16, std::min((size_t)R.len16[r] * 16, staged - (size_t)R.off16[r] * 16));
        uploaded_bytes_ = (int64_t)dirty;
        patched = true;
      }
    }
    J.patched = patched;
    if (!patched) {
      J.upload_bytes = staged;  // whole 16-byte groups: both sides are padded to 256 B
      if (rec_diff_upload_) {
        if (tables_resident) std::memcpy(shadow_.data(), s, staged);  // (shadow_ok: the tables behind are what they were)
        else shadow_.assign(s, s + total);
        shadow_layout_ = layout;
        shadow_dev_ = d_frame_.ptr;
      }
      uploaded_bytes_ = (int64_t)J.upload_bytes;
    }
    tables_dev_ = d_frame_.ptr; tables_layout_ = layout; tables_sig_.swap(tables_sig);
  }
  // algorithmic bytes of this frame: final store + per blur (pre-blur store is the store above for
  // a full-frame node; H read + H write + V read + V write + composite read) + records once
  int64_t bytes = 4LL * W_ * H_ + (int64_t)n * (int64_t)sizeof(DrawRec), bytes_blur = 0, bytes_fused = 0, bytes_saved = 0;
  // a cleared opaque surface stays opaque under SRC_ALPHA / ONE_MINUS_SRC_ALPHA blending (a' = sa + da (1 - sa), da = 1): a
  // fused vertical pass then replaces pixels under full coverage without reading them
  const bool surface_opaque = clear_ && (clear_rgba8_ >> 24) == 255u;
  big_blur_ = -1;
  int64_t big_area = 0;
  stats_.bytes_blur_big_h = stats_.bytes_blur_big_v = 0;
  for (size_t bi = 0; bi < J.blurs.size(); bi++) {
    const BlurJob& j = J.blurs[bi];
    const int ylo = std::max(0, j.y0 - j.taps.reach), yhi = std::min(H_, j.y1 + j.taps.reach);
    const int64_t a_h = (int64_t)(j.x1 - j.x0) * (yhi - ylo), a_v = (int64_t)(j.x1 - j.x0) * (j.y1 - j.y0);
    int64_t b_h = 4 * a_h + 4 * a_h, b_v = 4 * a_h + 4 * a_v;  // H read + H write; V read + V write
    // the consuming composite: fused into the V pass it reads the live surface there (where it has to blend); otherwise a
    // composite launch reads the blurred snapshot
    if (j.fuse_draw >= 0) { if (!surface_opaque) b_v += 4 * a_v; } else bytes += 4 * a_v;
    bytes_blur += b_h + b_v;
    if (bi < J.blur_fused.size() && J.blur_fused[bi]) {  // one kernel: the region read once (+ the surface under a translucent composite), written once
      const int64_t b_fx = 4 * a_v + 4 * a_v;
      bytes_fused += b_fx;
      bytes_saved += b_h + b_v - b_fx;
    }
    if (a_v > big_area) { big_area = a_v; big_blur_ = (int)bi; stats_.bytes_blur_big_h = b_h; stats_.bytes_blur_big_v = b_v; }
  }
  bytes += bytes_blur;
  stats_.bytes_blur = bytes_blur;
  stats_.bytes_composite_main = 4LL * W_ * H_ * (clear_ ? 1 : 2) + (int64_t)J.phases[0].count * (int64_t)sizeof(DrawRec);
  // algorithmic flops of the phase-0 composite launch, SURVEY.md 8(d): per fragment ClipAA 25, DropShadow 35 + exp, InsetShadow
  // 70 + exp, AnnularAA 28 (other modes priced as ClipAA), elliptical corners + 30, blend + re-quantise + 16
  for (int k = 0; k < 4; k++) stats_.fragments_main_by_mode[k] = frag_mode[k];
  stats_.fragments_main_elliptical = frag_ellip;
  stats_.fragments_main_other = frag_other;
  stats_.flops_composite_main = frag_mode[0] * 25 + frag_mode[1] * 36 + frag_mode[2] * 71 + frag_mode[3] * 28 + frag_other * 25 + frag_ellip * 30 +
                                (frag_mode[0] + frag_mode[1] + frag_mode[2] + frag_mode[3] + frag_other) * 16;
  stats_.n_draws = (int32_t)n;
  stats_.n_phases = (int32_t)J.phases.size();
  stats_.n_blurs = (int32_t)J.blurs.size();
  stats_.n_bins = nb;
  stats_.bytes_algorithmic = bytes;
  stats_.bytes_blur_fused = bytes_fused;
  stats_.bytes_frame_implementation = bytes - bytes_saved;
  stats_.fragments = fragments_;
  const auto t_l0 = std::chrono::steady_clock::now();
  stats_.ms_host_record = host_record_ms_;
  stats_.ms_host_upload = std::chrono::duration<float, std::milli>(t_l0 - t_s0).count();
}

// The launches of one prepared frame: the upload (a kernel on the render stream reading the pinned staging buffer), then
// binning, blur passes and compositing.  Submit thread (or the caller's, FDH_CREATE_SYNC_SUBMIT).
void Context::issue(LaunchJob& J) {
  const auto t_l0 = std::chrono::steady_clock::now();
  FDH_HIP(hipSetDevice(device_));
  if (J.patched) launch_upload_runs(stream_, J.d_dst, J.s_dev, J.runs);
  else launch_upload(stream_, J.d_dst, J.s_dev, J.upload_bytes);
  if (J.staging_slot >= 0) {
    FDH_HIP(hipEventRecord(staging_ev_[J.staging_slot], stream_));
    staging_busy_[J.staging_slot] = true;
  }
  launch_frame(J, false);
  launch_ms_.store(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_l0).count(), std::memory_order_relaxed);
}


void Context::launch_frame(const LaunchJob& J, bool profile) {
  const int bins_x_ = J.bins_x, bins_y_ = J.bins_y, list_stride_ = J.list_stride, binbox_shift_ = J.binbox_shift, big_blur_ = J.big_blur;
  const std::vector<const uint4*>&mx_w_h_ = J.mx_w_h, &mx_w_v_ = J.mx_w_v;
  const LaunchJob::View& dv_ = J.dv;
  const int nb = bins_x_ * bins_y_;
  const int np = (int)J.phases.size();
  // rows each phase has to produce: the stripe, widened by the vertical reach of every later blur
  int s0 = 0, s1 = J.H;
  if (stripe_y1_ > stripe_y0_) { s0 = std::max(0, stripe_y0_); s1 = std::min(J.H, stripe_y1_); }
  std::vector<int> lo(np), hi(np);
  {
    int l = s0, h = s1;
    for (int p = np - 1; p >= 0; p--) {
      lo[p] = l; hi[p] = h;
      if (J.phases[p].blur >= 0) {
        const int reach = J.blurs[J.phases[p].blur].taps.reach;
        l = std::max(0, l - reach);
        h = std::min(J.H, h + reach);
      }
    }
    // the records were culled to rows [rec_y0, rec_y1) when they were made (fdh_set_cull): every row a phase produces must lie inside
    if (s1 > s0 && (l < J.rec_y0 || h > J.rec_y1))
      throw Error(FDH_ERR_INVALID, "the resident draw records were culled to another row stripe: render the frame again after fdh_set_stripe (or fdh_set_cull(0))");
  }
  // profile mode: every launch stamps its own pair of events (set_launch_events: the kernel's execution time, no gaps)
  auto span_begin = [&](int kind) { if (profile) { Span sp{kind, next_event(), next_event()}; set_launch_events(sp.a, sp.b); spans_.push_back(sp); } };
  auto span_end = [&]() { if (profile) { if (!launch_events_used()) spans_.pop_back(); set_launch_events(nullptr, nullptr); } };
  span_begin(0);
  BinParams B;
  B.binrec = dv_.binrecs; B.binbox = dv_.binbox; B.chunkbox = dv_.chunkbox; B.n_draws = J.n_recs; B.binbox_shift = binbox_shift_; B.lists = J.lists; B.counts = J.counts; B.phase_first = dv_.phase_first; B.draws = dv_.recs;
  B.n_phases = np; B.bins_x = bins_x_; B.bins_y = bins_y_; B.stride = list_stride_;
  launch_bin(stream_, B);
  span_end();
  // Phase 0's full-grid composite takes its bins longest-list first, in the order its predecessor sorted (an extra
  // wavefront of that launch); it sorts this frame's counts for its successor.  Any permutation is a correct schedule.
  const int order_key = bins_x_ * 65536 + bins_y_;  // entries are (row << 16 | column) of THIS grid
  if (order_valid_ && order_nb_ != order_key) order_valid_ = false;  // frame size changed
  const bool sorting = J.clear && np > 0 && nb <= 8192 && J.phases[0].count > 0;
  const int* order_now = (sorting && order_valid_) ? d_order_[order_read_].ptr : nullptr;
  int* order_next = nullptr;
  if (sorting) {
    const int wr = order_valid_ ? 1 - order_read_ : order_read_;
    d_order_[wr].reserve(nb);
    order_next = d_order_[wr].ptr;
    order_read_ = wr;
    order_nb_ = order_key;
    order_valid_ = true;
  }
  // The surface that holds the live image.  A fused full-frame blur renders out of place and flips it; a frame that flips an odd
  // number of times ends in alt_, and the two pointers trade places: the frame surface IS the one the frame ended in
  // (fdh_frame_device_ptr is asked again after every frame).  Phase 0 always starts in fb_, i.e. on the surface the previous
  // frame's last pass WROTE: starting in the other one -- the surface that pass had only read -- cost the phase-0 launch 2 us
  // (32.1 against 30.2: its 33 MB of stores then land on lines other XCDs' L2s hold clean copies of).
  uint32_t* cur = fb_;
  for (int p = 0; p < np; p++) {
    const Phase& ph = J.phases[p];
    if (ph.blur >= 0) {
      const BlurJob& j = J.blurs[ph.blur];
      // V output: footprint rows this phase must produce; H output: those rows widened by the tap reach
      const int vy0 = std::max(j.y0, lo[p]), vy1 = std::min(j.y1, hi[p]);
      if (vy1 > vy0 && j.x1 > j.x0) {
        BlurParams bp;
        bp.W = J.W; bp.H = J.H; bp.pitch = J.W;
        bp.taps = j.taps;
        bp.fuse_draw = -1;
        bp.mx_w = (size_t)ph.blur < mx_w_h_.size() ? mx_w_h_[ph.blur] : nullptr;
        bool done = false;
        if ((size_t)ph.blur < J.blur_fused.size() && J.blur_fused[ph.blur]) {
          uint32_t* other = cur == fb_ ? alt_ : fb_;
          bp.src = cur; bp.dst = other;
          bp.fuse_draw = j.fuse_draw;
          bp.x0 = j.x0; bp.x1 = j.x1; bp.y0 = vy0; bp.y1 = vy1;
          span_begin(7);
          done = launch_blur_fused(stream_, bp, mx_w_v_[ph.blur], dv_.recs, dv_.exts);
          span_end();
          if (!done) throw Error(FDH_ERR_HIP, "fused blur: no kernel for this filter width (blur_fused_supported out of step with the launcher)");
          cur = other;
        }
        if (!done) {
        bp.fuse_draw = -1;
        bp.src = cur; bp.dst = blur_tmp_;
        bp.x0 = j.x0; bp.x1 = j.x1; bp.y0 = std::max(0, vy0 - j.taps.reach); bp.y1 = std::min(J.H, vy1 + j.taps.reach);
        span_begin(ph.blur == big_blur_ ? 5 : 3);
        launch_blur_h(stream_, bp);
        span_end();
        bp.src = blur_tmp_; bp.dst = j.fuse_draw >= 0 ? cur : backdrop_;
        bp.mx_w = (size_t)ph.blur < mx_w_v_.size() ? mx_w_v_[ph.blur] : nullptr;
        bp.fuse_draw = j.fuse_draw;
        bp.y0 = vy0; bp.y1 = vy1;
        span_begin(ph.blur == big_blur_ ? 6 : 4);
        launch_blur_v(stream_, bp, dv_.recs, dv_.exts);
        span_end();
        }
      }
    }
    CompositeParams C;
    C.lists = J.lists + (size_t)p * nb * list_stride_;
    C.counts = J.counts + (size_t)p * nb;
    C.backdrop = backdrop_;
    C.fb = cur;
    for (int l = 0; l < kMaxMips; l++) C.atlas.level[l] = atlas_levels_[l];
    C.atlas.size = atlas_size_; C.atlas.n_levels = n_levels_;
    C.W = J.W; C.H = J.H; C.pitch = J.W;
    C.bins_x = bins_x_; C.stride = list_stride_;
    const bool full = (p == 0 && J.clear);
    C.bin_x0 = full ? 0 : ph.bin_x0; C.bin_y0 = full ? 0 : ph.bin_y0;
    C.bin_nx = full ? bins_x_ : ph.bin_x1 - ph.bin_x0; C.bin_ny = full ? bins_y_ : ph.bin_y1 - ph.bin_y0;
    C.row_lo = lo[p]; C.row_hi = hi[p];
    C.load_fb = full ? 0 : 1;
    C.clear_rgba8 = J.clear_rgba8;
    C.n_wg = 0;
    C.order = full ? order_now : nullptr;
    C.order_next = full ? order_next : nullptr;
    C.has_slow = ph.has_slow ? 1 : 0;
    C.has_atlas = ph.has_atlas ? 1 : 0;
    C.has_masks = ph.has_masks ? 1 : 0;
    C.mask_spill = J.mask_spill; C.spill_stride = J.spill_stride;
    span_begin(p == 0 ? 1 : 2);
    launch_composite(stream_, dv_.recs, dv_.exts, C);
    span_end();
    static const bool snap = [] { const char* e = std::getenv("FDH_DEBUG_SNAP"); return e && std::atoi(e) != 0; }();
    if (snap && p == 0) {  // diagnostic only (tools/race_probe.py): what the first blur pass is about to read
      if (!dbg_snap_) FDH_HIP(hipMalloc((void**)&dbg_snap_, (size_t)J.W * J.H * 4));
      FDH_HIP(hipMemcpyAsync(dbg_snap_, cur, (size_t)J.W * J.H * 4, hipMemcpyDeviceToDevice, stream_));
    }
  }
  if (cur != fb_) std::swap(fb_, alt_);  // the frame ended in the other surface: it is the frame surface now
  FDH_HIP(hipGetLastError());
}

void Context::replay(int times) {
  need_device("replay");
  drain();
  if (!have_frame_) throw Error(FDH_ERR_INVALID, "replay: no frame has been submitted");
  FDH_HIP(hipSetDevice(device_));
  if (times <= 0) return;
  FDH_HIP(hipEventRecord(ev_[0], stream_));
  for (int i = 0; i < times; i++) launch_frame(job_, false);
  FDH_HIP(hipEventRecord(ev_[1], stream_));
  FDH_HIP(hipEventSynchronize(ev_[1]));
  float ms = 0.0f;
  FDH_HIP(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
  stats_.ms_total = ms / (float)times;
}

// enqueue only: several contexts (own streams, own surfaces) can then have frames in flight on one GPU at once
void Context::replay_async(int times) {
  need_device("replay_async");
  drain();
  if (!have_frame_) throw Error(FDH_ERR_INVALID, "replay: no frame has been submitted");
  FDH_HIP(hipSetDevice(device_));
  for (int i = 0; i < times; i++) launch_frame(job_, false);
}

// `times` frames back to back with one event between consecutive frames: ms_out[i] = duration of frame i on the stream
void Context::replay_timed(int times, float* ms_out) {
  need_device("replay_timed");
  drain();
  if (!have_frame_) throw Error(FDH_ERR_INVALID, "replay: no frame has been submitted");
  if (times <= 0 || !ms_out) return;
  FDH_HIP(hipSetDevice(device_));
  ev_used_ = 0;
  std::vector<hipEvent_t> marks;
  marks.push_back(next_event());
  FDH_HIP(hipEventRecord(marks.back(), stream_));
  for (int i = 0; i < times; i++) {
    launch_frame(job_, false);
    marks.push_back(next_event());
    FDH_HIP(hipEventRecord(marks.back(), stream_));
  }
  FDH_HIP(hipEventSynchronize(marks.back()));
  for (int i = 0; i < times; i++) FDH_HIP(hipEventElapsedTime(&ms_out[i], marks[i], marks[i + 1]));
}

hipEvent_t Context::next_event() {
  if (ev_used_ == ev_pool_.size()) {
    hipEvent_t e;
    FDH_HIP(hipEventCreate(&e));
    ev_pool_.push_back(e);
  }
  return ev_pool_[ev_used_++];
}

// Per-kernel timing: events bracket every launch, so this is kept apart from replay()'s batch timing.
void Context::profile(int times) {
  need_device("profile");
  drain();
  if (!have_frame_) throw Error(FDH_ERR_INVALID, "profile: no frame has been submitted");
  FDH_HIP(hipSetDevice(device_));
  if (times <= 0) return;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < times; i++) {
    ev_used_ = 0;
    spans_.clear();
    launch_frame(job_, true);
    FDH_HIP(hipStreamSynchronize(stream_));
    for (auto& sp : spans_) {
      float t = 0.0f;
      if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) acc[sp.kind] += t;  // (a span whose launch had nothing to do never stamped its events)
      else (void)hipGetLastError();
    }
  }
  stats_.ms_bin = (float)(acc[0] / times);
  stats_.ms_composite_main = (float)(acc[1] / times);
  stats_.ms_composite = (float)((acc[1] + acc[2]) / times);
  stats_.ms_blur_h = (float)((acc[3] + acc[5]) / times);
  stats_.ms_blur_v = (float)((acc[4] + acc[6]) / times);
  stats_.ms_blur_big_h = (float)(acc[5] / times);
  stats_.ms_blur_big_v = (float)(acc[6] / times);
  stats_.ms_blur_fused = (float)(acc[7] / times);
}

// ------------------------------------------------------------------ readback (glcontext.nim:2094-2135)
void Context::read_pixels(int x, int y, int w, int h, uint8_t* out) {
  need_device("read_pixels");
  drain();
  if (!fb_) throw Error(FDH_ERR_INVALID, "readPixels before the first frame");
  FDH_HIP(hipSetDevice(device_));
  if (w <= 0 || h <= 0) { x = 0; y = 0; w = W_; h = H_; }
  if (x < 0 || y < 0 || x + w > W_ || y + h > H_) throw Error(FDH_ERR_INVALID, "readPixels: rectangle outside the frame");
  FDH_HIP(hipStreamSynchronize(stream_));
  FDH_HIP(hipMemcpy2D(out, (size_t)w * 4, fb_ + (size_t)y * W_ + x, (size_t)W_ * 4, (size_t)w * 4, h, hipMemcpyDeviceToHost));
}
void Context::debug_read_surface(int which, uint8_t* out) {
  need_device("debug_read_surface");
  drain();
  const uint32_t* src = which == 0 ? fb_ : which == 1 ? blur_tmp_ : which == 2 ? backdrop_ : which == 3 ? dbg_snap_ : nullptr;
  if (!src) throw Error(FDH_ERR_INVALID, "debug_read_surface: no such surface (or no frame yet)");
  FDH_HIP(hipSetDevice(device_));
  FDH_HIP(hipStreamSynchronize(stream_));
  FDH_HIP(hipMemcpy(out, src, (size_t)W_ * H_ * 4, hipMemcpyDeviceToHost));
}
uint64_t Context::record_digest() {
  drain();
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(p); for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
  const uint64_t n = recs_.size();
  mix(&n, sizeof n);
  for (const DrawRec& r : recs_) mix(&r, sizeof r);
  for (const BBox& b : bboxes_) mix(&b, sizeof b);
  for (const QuadExt& q : exts_) mix(&q, sizeof q);
  for (const Phase& ph : phases_) { mix(&ph.first, sizeof ph.first); mix(&ph.count, sizeof ph.count); mix(&ph.blur, sizeof ph.blur); }
  return h;
}
void Context::frame_device_ptr(void** p, int* w, int* h, int64_t* pitch_bytes) {
  need_device("frame_device_ptr");
  drain();
  if (!fb_) throw Error(FDH_ERR_INVALID, "no frame surface yet");
  *p = fb_; *w = W_; *h = H_; *pitch_bytes = (int64_t)W_ * 4;
}

}  // namespace fdh
