// fdh_context.cpp -- host side: BackendContext calls -> draw records -> GPU submission.
//
// What glcontext.nim does with ten vertex streams and a batch flush, this does with one 128-byte record
// per call.  The record carries exactly what the reference's vertex attributes carry (ceil-snapped quad,
// un-snapped half extents, packed radii, mode word, factors, colours) so the kernels can restate the
// fragment shaders.  Clip masks become push/pop records evaluated analytically per pixel, backdrop
// blurs split the list into phases (a blur is a global barrier in painter's order, glcontext.nim:1788-1841).
#include "fdh_context.h"
#include "fdh_host.h"
#include "fdh_walkpool.h"

#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace fdh {

void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw Error(FDH_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}


// when each device context of the process last submitted a frame (steady-clock ns; 0 = never): Context::prepare asks whether
// frames of OTHER contexts are in flight

void poison_fresh(void* p, size_t bytes) {
#if defined(FDH_POISON)  // fault-hunting builds only (make variant DEFS=-DFDH_POISON=0xA5)
  if (!p || !bytes) return;
  (void)hipMemset(p, FDH_POISON, bytes);
  (void)hipDeviceSynchronize();
#else
  (void)p; (void)bytes;
#endif
}

// ------------------------------------------------------------------ lifetime
// Staging in device memory (HostVec::vram): when the device exposes all of its memory to the host (large BAR: every MI355X box of
// the pool) the recording threads write the frame's records straight into HBM and the gather kernel reads them locally.
// FDH_VRAM_STAGING=0 keeps pinned host memory (the path for devices without a large BAR), =1 forces device memory.
// Decided PER DEVICE (fdh_create takes an ordinal: one process may hold contexts on several GPUs), once, on first use.
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_vram_mu;
int g_vram_probe[kMaxDevices];  // 0 not probed yet, 1 staging in device memory, -1 pinned host memory
// the store of released blocks: by device, then by log2 of the size class
std::vector<void*> g_vram_free[kMaxDevices][48];
int g_vram_contexts[kMaxDevices];  // device contexts alive per device (the store of a device is trimmed when its last one goes)
int vram_class(size_t bytes) { int k = 12; while (((size_t)1 << k) < bytes) k++; return k; }
// the calling thread's current device for the scope (pool threads and callers with contexts on several devices allocate here)
struct DeviceScope {
  int prev = -1, dev;
  explicit DeviceScope(int d) : dev(d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); }
  ~DeviceScope() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};
bool vram_probe(int dev) {
  if (const char* e = std::getenv("FDH_VRAM_STAGING")) return std::atoi(e) != 0;
  DeviceScope scope(dev);
  int large = 0;
  if (hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, dev) != hipSuccess || !large) return false;
  // ... and a round trip to make sure: the CPU stores a pattern into such a block, a device-to-host copy must bring it back
  uint32_t* d = nullptr;
  if (hipExtMallocWithFlags((void**)&d, 4096, hipDeviceMallocUncached) != hipSuccess || !d) return false;
  bool ok = true;
  for (uint32_t i = 0; i < 1024; i++) d[i] = 0x9e3779b9u * (i + 1);
  store_fence();
  uint32_t back[1024];
  if (hipMemcpy(back, d, sizeof back, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
  for (uint32_t i = 0; ok && i < 1024; i++) ok = back[i] == 0x9e3779b9u * (i + 1);
  (void)hipFree(d);
  return ok;
}
}  // namespace
bool vram_staging(int device) {
  if (device < 0 || device >= kMaxDevices) return false;
  std::lock_guard<std::mutex> lk(g_vram_mu);
  if (g_vram_probe[device] == 0) g_vram_probe[device] = vram_probe(device) ? 1 : -1;
  return g_vram_probe[device] > 0;
}
void* vram_block_acquire(int device, size_t bytes, size_t* size_class) {
  const int k = vram_class(bytes);
  *size_class = (size_t)1 << k;
  if (device < 0 || device >= kMaxDevices) throw Error(FDH_ERR_NO_DEVICE, "staging block asked for a device ordinal out of range");
  {
    std::lock_guard<std::mutex> lk(g_vram_mu);
    auto& fl = g_vram_free[device][k];
    if (!fl.empty()) { void* p = fl.back(); fl.pop_back(); return p; }
  }
  DeviceScope scope(device);  // (a walk-pool thread's current device is whatever it was created with)
  void* p = nullptr;
  FDH_HIP(hipExtMallocWithFlags(&p, (size_t)1 << k, hipDeviceMallocUncached));
  return p;
}
void vram_block_release(int device, void* p, size_t size_class) {
  if (!p) return;
  // (fault hunting: FDH_VRAM_STORE=0 gives blocks back to the driver, as until the end of round 4; =2 neither frees nor reuses them)
  static const int mode = [] { const char* e = std::getenv("FDH_VRAM_STORE"); return e ? std::atoi(e) : 1; }();
  if (mode == 2) return;
  if (mode == 0 || !size_class || device < 0 || device >= kMaxDevices) { (void)hipFree(p); return; }
  std::lock_guard<std::mutex> lk(g_vram_mu);
  g_vram_free[device][vram_class(size_class)].push_back(p);
}
size_t vram_store_bytes(int device) {
  if (device < 0 || device >= kMaxDevices) return 0;
  std::lock_guard<std::mutex> lk(g_vram_mu);
  size_t b = 0;
  for (int k = 0; k < 48; k++) b += g_vram_free[device][k].size() << k;
  return b;
}
int vram_contexts_alive(int device) {
  if (device < 0 || device >= kMaxDevices) return 0;
  std::lock_guard<std::mutex> lk(g_vram_mu);
  return g_vram_contexts[device];
}
void vram_context_born(int device) {
  if (device < 0 || device >= kMaxDevices) return;
  std::lock_guard<std::mutex> lk(g_vram_mu);
  g_vram_contexts[device]++;
}
// The last device context of a device is gone: its store keeps at most kVramStoreKeep bytes (largest blocks go first -- one huge
// frame must not pin its HBM for the life of the process).  Called with the device idle (the context has synchronised its stream).
// The blocks are unmapped UNDER g_vram_mu, and a context counts itself in (vram_context_born) before it creates its stream: freeing
// uncached BAR-visible blocks while another context renders is the trigger of round 4's stale-line faults, so
// while a trim runs no context of the device exists and none can come to exist -- a constructor on another thread waits at the lock.
void vram_context_gone(int device) {
  if (device < 0 || device >= kMaxDevices) return;
  std::lock_guard<std::mutex> lk(g_vram_mu);
  if (--g_vram_contexts[device] > 0) return;
  std::vector<void*> drop;
  size_t held = 0;
  for (int k = 0; k < 48; k++) held += g_vram_free[device][k].size() << k;
  for (int k = 47; k >= 12 && held > kVramStoreKeep; k--)
    while (!g_vram_free[device][k].empty() && held > kVramStoreKeep) { drop.push_back(g_vram_free[device][k].back()); g_vram_free[device][k].pop_back(); held -= (size_t)1 << k; }
  if (drop.empty()) return;
  DeviceScope scope(device);
  (void)hipDeviceSynchronize();
  for (void* p : drop) (void)hipFree(p);
}

Context::Context(int atlas_size, float pixel_scale, int device, uint32_t flags) : Recorder(this, true), device_(device), flags_(flags), pixel_scale_(pixel_scale) {
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
  // host writes into device memory through the BAR pass the GPU's host data path, which may hold them: its flush register (mapped
  // for exactly this: HSA_AMD_AGENT_INFO_HDP_FLUSH) is written before the launches that read the staging mirrors (Context::prepare)
  hdp_flush_reg_ = prop.hdpMemFlushCntl;
  // counted in BEFORE the first piece of device state: from here on no trim of the staging store can run (vram_context_gone); a
  // constructor that fails below counts itself out again (the destructor of a half-built object never runs)
  vram_context_born(device_);
  try {
    FDH_HIP(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
    stream_ = own_stream_;
    for (auto& e : ev_) FDH_HIP(hipEventCreate(&e));
    for (auto& e : staging_ev_) FDH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    initial_atlas_size_ = atlas_size > 0 ? atlas_size : 1024;  // newContext default, glcontext.nim:255-261
    alloc_atlas(initial_atlas_size_);
    static const bool env_sync = [] { const char* e = std::getenv("FDH_SYNC_SUBMIT"); return e && std::atoi(e) != 0; }();
    if (!(flags & FDH_CREATE_SYNC_SUBMIT) && !env_sync) worker_ = std::thread([this] { worker_main(); });
  } catch (...) {
    vram_context_gone(device_);
    throw;
  }
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
  for (auto& set : lanes_) set.clear();  // (pinned arrays: freed while the device is still this thread's)
  for (auto& m : misc_) m.release();
  for (auto& e : staging_ev_) if (e) (void)hipEventDestroy(e);
  if (seq_host_) (void)hipHostFree((void*)seq_host_);
  if (deep_host_) (void)hipHostFree((void*)deep_host_);
  if (own_stream_) (void)hipStreamDestroy(own_stream_);
  for (auto& l : merge_lane_) l.reset();  // (their staging blocks go to the store before the store is looked at)
  vram_context_gone(device_);
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
    // what that frame was to upload may not have arrived: nothing is taken for resident in the device block any more
    tables_dev_ = nullptr; shadow_dev_ = nullptr; have_frame_ = false;
    std::rethrow_exception(e);
  }
}

void Context::wait_staging(int slot) {
  if (staging_busy_[slot] == 2) {
    // the word is pinned host memory the device writes: a load of it is an uncached read (~100 ns); spin, then yield, and after
    // a few milliseconds stop trusting it and wait for the stream (the frame is then done for certain)
    const uint32_t want = staging_seq_[slot];
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; (int32_t)(*seq_host_ - want) < 0; spins++) {
      if (spins < 2000) { cpu_relax(); continue; }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) { FDH_HIP(hipStreamSynchronize(stream_)); break; }
      std::this_thread::yield();
    }
  } else if (staging_busy_[slot] == 1) {
    FDH_HIP(hipEventSynchronize(staging_ev_[slot]));
  }
  staging_busy_[slot] = 0;
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
  if (flags & ~(uint32_t)(FDH_GLYPH_LCD_FILTER | FDH_GLYPH_LCD_CONTEXT)) throw Error(FDH_ERR_INVALID, "put_glyph_image: unknown flag");
  if (flags & FDH_GLYPH_LCD_CONTEXT) flags = text_lcd_filtering_ ? FDH_GLYPH_LCD_FILTER : 0u;  // as setTextLcdFilteringEnabled said
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
  if (flags & ~(uint32_t)(FDH_GLYPH_LCD_FILTER | FDH_GLYPH_LCD_CONTEXT)) throw Error(FDH_ERR_INVALID, "put_glyph_outline: unknown flag");
  if (flags & FDH_GLYPH_LCD_CONTEXT) flags = text_lcd_filtering_ ? FDH_GLYPH_LCD_FILTER : 0u;
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
  poison_fresh(backdrop_, n * 4); poison_fresh(blur_tmp_, n * 4); poison_fresh(fb_, n * 4);
#if defined(FDH_POISON)  // (fault-hunting builds: the surface starts as 0x11 bytes instead of zeros -- a zero pixel in a frame is then nobody's of ours)
  FDH_HIP(hipMemsetAsync(fb_, 0x11, n * 4, stream_));
#else
  FDH_HIP(hipMemsetAsync(fb_, 0, n * 4, stream_));
#endif
  surf_w_ = W_;
  surf_h_ = H_;
}

// Weight fragments of a matrix-pipe blur pass (k_blur_mx, k_blur_mx.hip).  Lane (j, g) of fragment m holds, for the window
// texels 16 m + 8 g + t (t = 0..7) of a 32-output block, the tap each meets at output j: k = texel - delta - j, weight
// q[k] (the tap at scale 2^10 as one f16: quantise_taps_f16 below) when 0 <= k <= 2 reach, else 0.  Every product with an 8-bit
// texel is exact in f32.  (Rounds 2 - 4 carried a second half, lo = RNE(w - hi), 22 significant bits: its slot in the layout remains, zero.)
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
// Round 5: the taps as ONE f16 each at scale 2^10 (the kernels multiply once per operand and k-step: FDH_MX_LO in fdh_types.h).
// Rounded from the CENTRE tap outwards, the rounding error carried to the next tap out (the filter is symmetric: each side takes half
// of the centre's error): a tap's error is made good by its neighbour, and what is left at the end falls on the outermost taps, whose
// f16 steps are thousands of times finer than the centre's -- the sum of the weights is kept to ~1e-7 (a flat region keeps its value)
// and the error is a fine alternating pattern that smooth content cancels.  (Measured, numpy, two passes with RGBA8 between them, against
// the exact taps: 0.00 - 0.23 % of a UI-like image's texels move, by one LSB; 0.4 - 1.0 % of white noise's.  Rounding every tap on its
// own moves 0.3 - 1.0 % / 2 - 8 %; carrying the error from the outside in and letting the centre tap keep the sum, 0.2 - 0.7 %: the
// centre tap's step is the coarsest of all.)  q[k], k = 0 .. 2 reach, in units of 2^-10.
static void quantise_taps_f16(const BlurTaps& t, float* q) {
  const int r = t.reach;
  const double centre = (double)t.dense[kBlurPad + r] * 1024.0;
  q[r] = half_value(half_bits_rne((float)centre));
  double carry = 0.5 * (centre - (double)q[r]);
  for (int k = r - 1; k >= 0; k--) {
    if (t.dense[kBlurPad + k] == 0.0f) { q[k] = q[2 * r - k] = 0.0f; continue; }  // (a texel the merged FIR does not read stays unread: the error waits for the next tap)
    const double want = std::max((double)t.dense[kBlurPad + k] * 1024.0 + carry, 0.0);
    const float v = half_value(half_bits_rne((float)want));
    carry = want - (double)v;
    q[k] = q[2 * r - k] = v;
  }
}
static void build_mx_weights(const BlurTaps& t, bool vertical, uint8_t* out) {
  const int nk = mx_nk(t.reach, vertical), delta = mx_delta(t.reach, vertical);
  uint16_t* o = reinterpret_cast<uint16_t*>(out);
  float q[2 * kMaxBlurReach + 1];
  quantise_taps_f16(t, q);
  for (int m = 0; m < nk; m++)
    for (int lane = 0; lane < 64; lane++) {
      const int j = lane & 31, g = lane >> 5;
      for (int e = 0; e < 8; e++) {
        // which window texel element e of lane group g stands for (mx_krow, fdh_types.h): the natural order for the horizontal pass;
        // for the vertical one the order in which a 32 x 32 accumulator tile holds its rows, so that the fused kernel's horizontal
        // product feeds the vertical one from registers (k_blur_fx) -- any order serves as long as both operands use the same
        const int k = 16 * m + mx_krow(g, e, vertical) - delta - j;
#if FDH_MX_LO  // (variant builds: the 22-bit weights of rounds 2 - 4, hi = RNE(w), lo = RNE(w - hi); the kernels then multiply twice)
        const float w = (k >= 0 && k <= 2 * t.reach) ? t.dense[kBlurPad + k] * 1024.0f : 0.0f;
        const uint16_t hi = half_bits_rne(w), lo = half_bits_rne(w - half_value(hi));
#else
        const float w = (k >= 0 && k <= 2 * t.reach) ? q[k] : 0.0f;
        const uint16_t hi = half_bits_rne(w), lo = 0;  // (the fragment layout keeps the second half's slot: zeros)
#endif
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
  const size_t n = n_total_, n_ext = n_ext_total_;
  J.W = W_; J.H = H_; J.clear = clear_; J.clear_rgba8 = clear_rgba8_;
  J.rec_y0 = culling() ? cull_y0_ : 0; J.rec_y1 = culling() ? cull_y1_ : H_;
  J.latency_routes = latency_routes_;
  J.phases = phases_;  // (copies: the recording side keeps its own for fdh_debug_record_digest)
  J.blurs = blurs_;
  J.n_recs = (int)n;
  J.bins_x = (W_ + kBin - 1) / kBin;
  J.bins_y = (H_ + kBin - 1) / kBin;
  J.binbox_shift = binbox_shift_;
  const int nb = J.bins_x * J.bins_y;
  // List stride = the largest number of draws any bin of any phase can receive: counted while the frame was recorded
  // (Lane::count_add / count_close per phase; lanes of pool threads add their own maxima: an upper bound)
  J.list_stride = (stride_max_ + 7) & ~7;
  reserve_quiet(d_lists_, (size_t)J.phases.size() * nb * J.list_stride);
  reserve_quiet(d_counts_, (size_t)J.phases.size() * nb);
  J.lists = d_lists_.ptr; J.counts = d_counts_.ptr;
  // clip nesting beyond the LDS stack (kMaskDepth levels): one global plane per extra level, 256 bytes per strip
  J.mask_spill = nullptr;
  J.spill_stride = (size_t)nb * 16 * 64;
  if (deepest_clip_ > kMaskDepth) {
    const size_t levels = (size_t)(deepest_clip_ - kMaskDepth);
    if (levels * J.spill_stride * sizeof(uint32_t) > ((size_t)2 << 30))
      throw Error(FDH_ERR_UNSUPPORTED, "clip masks nested too deep for this frame size (the spill plane would exceed 2 GiB)");
    reserve_quiet(d_mask_spill_, levels * J.spill_stride);
    J.mask_spill = d_mask_spill_.ptr;
  }
  // ---- layout of the frame block: records | extensions | bin records | bin boxes | chunk boxes | phase table | blur tables
  std::vector<int> pf(J.phases.size() + 1);
  for (size_t i = 0; i < J.phases.size(); i++) pf[i] = J.phases[i].first;
  pf[J.phases.size()] = (int)n;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t b_recs = n * sizeof(DrawRec), b_ext = n_ext * sizeof(QuadExt), b_bb = n * sizeof(BinRec), b_pf = pf.size() * sizeof(int);
  const size_t b_box = ((n + 3) & ~(size_t)3) * sizeof(uint32_t);
  const size_t n_chunks = std::max<size_t>((n + 255) / 256, 1), b_chunk = n_chunks * sizeof(uint32_t);
  const size_t o_recs = 0, o_ext = up(o_recs + b_recs), o_bb = up(o_ext + b_ext), o_box = up(o_bb + b_bb), o_chunk = up(o_box + b_box),
               o_pf = up(o_chunk + b_chunk);
  // weight fragments of the matrix-pipe blur passes, two tables (H, V) per blur job
  std::vector<size_t> o_mxh(J.blurs.size(), 0), o_mxv(J.blurs.size(), 0);
  const size_t o_tables = up(o_pf + b_pf);
  size_t total = o_tables;
  for (size_t i = 0; i < J.blurs.size(); i++) {
    const int nkh = mx_nk(J.blurs[i].taps.reach, false), nkv = mx_nk(J.blurs[i].taps.reach, true);
    if (nkh > kMxMaxNK || nkv > kMxMaxNK) continue;
    o_mxh[i] = total; total = up(total + mx_table_bytes(nkh));
    o_mxv[i] = total; total = up(total + mx_table_bytes(nkv));
  }
  if (total >= ((size_t)1 << 32)) throw Error(FDH_ERR_UNSUPPORTED, "frame block beyond 4 GiB");
  reserve_quiet(d_frame_, total);
  LaunchJob::View& dv = J.dv;
  dv.recs = reinterpret_cast<DrawRec*>(d_frame_.ptr + o_recs);
  dv.exts = reinterpret_cast<QuadExt*>(d_frame_.ptr + o_ext);
  dv.binrecs = reinterpret_cast<BinRec*>(d_frame_.ptr + o_bb);
  dv.phase_first = reinterpret_cast<int*>(d_frame_.ptr + o_pf);
  dv.binbox = reinterpret_cast<uint32_t*>(d_frame_.ptr + o_box);
  dv.chunkbox = reinterpret_cast<uint32_t*>(d_frame_.ptr + o_chunk);
  J.mx_w_h.assign(J.blurs.size(), nullptr);
  J.mx_w_v.assign(J.blurs.size(), nullptr);
  for (size_t i = 0; i < J.blurs.size(); i++)
    if (o_mxh[i]) { J.mx_w_h[i] = reinterpret_cast<const uint4*>(d_frame_.ptr + o_mxh[i]); J.mx_w_v[i] = reinterpret_cast<const uint4*>(d_frame_.ptr + o_mxv[i]); }
  // A blur node that covers the whole frame, composited by its own vertical pass (no clip open), in a frame that starts from
  // the clear colour: both passes as ONE kernel, out of place (k_blur_fx) -- launch_frame alternates between fb_ and alt_.
  // Which route is a matter of speed only -- the two give the same pixels bit for bit (tests/test_hip_parity.py): Context::pick_routes.
  const bool fx_on = latency_routes_;  // (decided when the frame began: Context::pick_routes)
  J.blur_fused.assign(J.blurs.size(), 0);
  J.n_fused = 0;
  for (size_t i = 0; i < J.blurs.size(); i++) {
    const BlurJob& j = J.blurs[i];
    // (frames under 0.4 Mpx keep the two small-region passes, like every region of that size: launch_blur_h)
    static const bool any_size = [] { const char* e = std::getenv("FDH_FORCE_BLUR_PATH"); return e && std::atoi(e) == 3; }();
    if (fx_on && clear_ && j.fuse_draw >= 0 && J.mx_w_h[i] && j.x0 == 0 && j.y0 == 0 && j.x1 == W_ && j.y1 == H_ && blur_fused_supported(j.taps.reach, W_, W_) &&
        (any_size || (long long)W_ * H_ >= 384 * 1024)) {
      J.blur_fused[i] = 1;
      J.n_fused++;
    }
  }
  if (J.n_fused > 0 && !alt_) {
    FDH_HIP(hipMalloc((void**)&alt_, (size_t)W_ * H_ * 4));
    poison_fresh(alt_, (size_t)W_ * H_ * 4);
    FDH_HIP(hipMemsetAsync(alt_, 0, (size_t)W_ * H_ * 4, stream_));
  }
  const int slot = staging_i_;
  J.staging_slot = slot;
  J.d_dst = d_frame_.ptr;
  // CLEAR FOLDING.  A frame that is cleared and whose first draw is one colour at full coverage over the whole frame -- a window's
  // background rectangle, the first node of nearly every UI tree (the bench scene's is translucent white over the clear colour)
  // -- starts, in effect, from another clear colour: blend(clear, colour), the very arithmetic the compositor's uniform-blend path
  // applies to every strip (blend_pre: F = rint(fma(F, 1 - sa, c * 255 sa)) per channel, IEEE single, no approximations), computed
  // once here.  The draw's bin record goes to the device with empty bounds (it is never binned); the lanes keep what was recorded
  // (fdh_debug_record_digest).  Bench frame: 32 640 uniform blends fewer, 8 % of the phase-0 launch's VALU instructions.
  // (more pieces than the upload's kernel-argument table holds: they are copied together first -- BEFORE the fold below empties the
  // first record's bounds in its lane: the copy would carry the emptied box, and the restore at the end of this function would
  // reach the original lane only, leaving fdh_debug_record_digest an empty box for record 0 of such a frame -- ADVICE r4)
  if (pieces_.size() * 3 + 2 > (size_t)kMaxUploadRuns) consolidate_pieces();
  bool folded = false;
  BBox folded_box{0, 0, 0, 0};
  BinRec* folded_br = nullptr;
  static const bool fold_on = [] { const char* e = std::getenv("FDH_FOLD_CLEAR"); return !e || std::atoi(e) != 0; }();
  if (fold_on && clear_ && !pieces_.empty() && J.phases[0].count > 0) {
    const Piece& p0 = pieces_[0];
    Lane& L = lane(p0.lane);
    BinRec& br = L.bins[p0.first];
    const DrawRec& r = L.recs[p0.first];
    const uint32_t om = r.op_mode;
    if (((om >> 12) & 15u) == OP_DRAW && (br.flags & LE_PLAIN) && !(br.flags & BR_CORE_REMOVED) && (br.flags & BR_HAS_CORE) && br.ix0 <= 0 && br.iy0 <= 0 &&
        br.ix1 >= W_ && br.iy1 >= H_ && br.box.x0 <= 0 && br.box.y0 <= 0 && br.box.x1 >= W_ && br.box.y1 >= H_) {
      const float inv255 = 1.0f / 255.0f;
      const uint32_t c = r.col[0];
      float cf[3];
      std::memcpy(cf, &r.col[1], sizeof cf);  // (device form: r, g, b / 255 as floats -- Recorder::commit_bins)
      const float sa = (float)(c >> 24) * inv255, A = 255.0f * sa, ia = 1.0f - sa;
      const float src[4] = {cf[0] * A, cf[1] * A, cf[2] * A, A};
      uint32_t out = 0;
      for (int k = 0; k < 4; k++) {
        const float F = (float)((J.clear_rgba8 >> (8 * k)) & 255u);
        const float v = std::nearbyintf(std::fmaf(F, ia, src[k]));  // (round to nearest even, like v_rndne_f32)
        out |= (uint32_t)std::min(std::max((int)v, 0), 255) << (8 * k);
      }
      J.clear_rgba8 = out;
      folded = true;
      folded_br = &br;
      folded_box = br.box;
      br.box = BBox{0, 0, 0, 0};
      if (p0.lane > 0) L.publish_bytes(1, (size_t)p0.first * sizeof(BinRec), sizeof(BinRec));  // (a pool thread published the piece already)
    }
  }
  stats_.clear_folded = folded ? 1.0f : 0.0f;
  // ---- the slot's small print: chunk boxes, phase table, then the blur weight tables (when the device block does not hold them already)
  std::vector<size_t> layout{total, o_recs, o_ext, o_bb, o_box, o_chunk, o_pf, n, n_ext};
  for (size_t i = 0; i < J.blurs.size(); i++) { layout.push_back(o_mxh[i]); layout.push_back(o_mxv[i]); }
  std::vector<float> tables_sig;
  for (size_t i = 0; i < J.blurs.size(); i++)
    if (o_mxh[i]) { const BlurTaps& t = J.blurs[i].taps; tables_sig.push_back((float)t.reach); tables_sig.insert(tables_sig.end(), t.dense + kBlurPad, t.dense + kBlurPad + 2 * t.reach + 1); }
  // The weight tables depend on the filters alone and sit behind everything else in the block: when the device block already
  // holds these very tables at these very offsets (an animation blurs with the same radii frame after frame) they are neither
  // staged nor uploaded again -- 40 of the bench frame's 130 KB.
  if (!rec_diff_upload_) shadow_dev_ = nullptr;
  const bool shadow_ok = rec_diff_upload_ && shadow_dev_ == d_frame_.ptr && shadow_layout_ == layout && shadow_.size() == total;
  const bool tables_resident = tables_dev_ == d_frame_.ptr && tables_layout_ == layout && tables_sig_ == tables_sig && (!rec_diff_upload_ || shadow_ok);
  // (built in ordinary memory -- the retained path compares and keeps it -- and copied to the slot's pinned buffer in one go)
  std::vector<uint8_t>& misc = misc_host_;
  const size_t o_misc = o_chunk;  // the block from the chunk boxes on
  misc.assign((tables_resident ? o_tables : total) - o_misc, 0);
  std::memcpy(misc.data() + (o_pf - o_misc), pf.data(), b_pf);
  {  // chunk boxes: the union box of every 256 consecutive draws = byte-wise min of their bin boxes (x0, y0 min; 127 - x1, 127 - y1 min)
    uint32_t* cb = reinterpret_cast<uint32_t*>(misc.data());
    for (size_t c = 0; c < n_chunks; c++) cb[c] = 0x7f7f7f7fu;
    size_t g = 0;
    for (const Piece& p : pieces_) {
      const uint32_t* bx = lane(p.lane).boxes.p + p.first;
      for (uint32_t i = 0; i < p.n; i++, g++) {
        const uint32_t m = cb[g >> 8], v = bx[i];
        uint32_t o = 0;
        for (int sh = 0; sh < 32; sh += 8) o |= std::min((m >> sh) & 255u, (v >> sh) & 255u) << sh;
        cb[g >> 8] = o;
      }
    }
  }
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
      std::memcpy(misc.data() + (o_mxh[i] - o_misc), hit->h.data(), bh);
      std::memcpy(misc.data() + (o_mxv[i] - o_misc), hit->v.data(), bv);
    }
  HostVec<uint8_t>& up_misc = misc_[slot];
  up_misc.pinned = true;
  up_misc.vram = vram_staging(device_);
  up_misc.dev = device_;
  up_misc.n = 0;
  up_misc.reserve(misc.size());
  // ---- the runs k_upload_frame gathers.  Every piece brings three: its records (extension indices re-based on the way), its bin
  // records, its extensions; then the phase table (+ tables).  A frame recorded by one thread is one piece.
  // device views of the lanes' mirrors (taken by whoever allocated them), index lane + 1 (slot 0: the consolidated lane)
  const size_t n_lanes = lanes_[(size_t)slot].size() + 1;
  std::vector<const uint8_t*> d_recs(n_lanes, nullptr), d_bins(n_lanes, nullptr), d_exts(n_lanes, nullptr);
  auto views = [&](const Piece& p) {  // (after the piece's lane was published: the mirrors are where they will stay)
    const size_t l = (size_t)(p.lane + 1);
    const Lane& Ln = lane(p.lane);
    d_recs[l] = Ln.d_recs; d_bins[l] = Ln.d_bins; d_exts[l] = Ln.d_exts;
  };
  if (up_misc.p != misc_dev_host_[slot]) { misc_dev_[slot] = up_misc.device_view(); misc_dev_host_[slot] = up_misc.p; }
  const uint8_t* d_misc = misc_dev_[slot];
  J.runs.clear();
  auto add_run = [&](std::vector<UploadRun>& to, const uint8_t* src, size_t dst_off, size_t bytes, uint32_t ext_add, uint32_t kind) {
    if (!bytes) return;
    to.push_back(UploadRun{src, (uint32_t)dst_off, (uint32_t)bytes, ext_add, kind});
  };
  int64_t link_bytes = 0;
  bool patched = false;
  // Retained scenes: only what differs from the block the device already holds travels -- after an edit (or between two frames
  // of an animation) that is a few hundred bytes of records and bin records out of ~110 KB.  The comparison runs against a
  // host shadow of the device block in 256-byte chunks.  (A frame recorded from scratch differs from its predecessor nearly
  // everywhere: comparing 110 KB to find that out, and keeping the shadow current, cost 12 us per frame -- only frames of a
  // retained scene take the diff route.)
  if (shadow_ok && pieces_.size() <= 1 && (pieces_.empty() || pieces_[0].lane == 0)) {
    const Lane& L = lane(0);
    const Piece p0 = pieces_.empty() ? Piece{} : pieces_[0];
    std::vector<UploadRun> runs;
    std::vector<const uint8_t*> from;  // the host bytes behind each run (the shadow is brought up to date from them)
    size_t dirty = 0;
    bool fits = true;
    auto diff = [&](const uint8_t* host, const uint8_t* dev_src, size_t off, size_t bytes, uint32_t kind) {
      for (size_t at = 0; at < bytes && fits; at += 256) {
        const size_t len = std::min<size_t>(256, bytes - at);
        if (std::memcmp(shadow_.data() + off + at, host + at, len) == 0) continue;
        dirty += len;
        if (!runs.empty() && runs.back().kind == kind && (size_t)runs.back().dst_off + runs.back().bytes == off + at) runs.back().bytes += (uint32_t)len;
        else if (runs.size() + 4 < (size_t)kMaxUploadRuns) { runs.push_back(UploadRun{dev_src + at, (uint32_t)(off + at), (uint32_t)len, 0u, kind}); from.push_back(host + at); }
        else fits = false;
      }
    };
    // (one piece of lane 0 starting at extension 0: the records' extension indices are the frame's already)
    if (p0.ext_first == 0) {
      Lane& Lw = lane(0);
      Lw.publish(0, 0, 0, 0);  // (the mirrors exist and fit the lane: their addresses are final)
      views(p0);
      diff(reinterpret_cast<const uint8_t*>(L.recs.p + p0.first), d_recs[1] + (size_t)p0.first * sizeof(DrawRec), o_recs, b_recs, 0u);
      diff(reinterpret_cast<const uint8_t*>(L.exts.p), d_exts[1], o_ext, b_ext, 0u);
      diff(reinterpret_cast<const uint8_t*>(L.bins.p + p0.first), d_bins[1] + (size_t)p0.first * sizeof(BinRec), o_bb, b_bb, 1u);
      diff(misc.data(), d_misc, o_misc, o_tables - o_misc, 0u);
      if (fits && dirty * 2 < total) {
        for (size_t k = 0; k < runs.size(); k++) {
          std::memcpy(shadow_.data() + runs[k].dst_off, from[k], runs[k].bytes);
          // what travels is published now: the dirty chunks alone
          const size_t off = runs[k].dst_off;
          if (off >= o_misc) std::memcpy(up_misc.p + (off - o_misc), misc.data() + (off - o_misc), runs[k].bytes);
          else if (off >= o_bb) Lw.publish_bytes(1, (size_t)p0.first * sizeof(BinRec) + (off - o_bb), runs[k].bytes);
          else if (off >= o_ext && b_ext) Lw.publish_bytes(2, off - o_ext, runs[k].bytes);
          else Lw.publish_bytes(0, (size_t)p0.first * sizeof(DrawRec) + (off - o_recs), runs[k].bytes);
        }
        J.runs = runs;
        link_bytes = (int64_t)dirty;
        patched = true;
      }
    }
  }
  if (!patched) {
    uint32_t at_rec = 0, at_ext = 0;
    std::memcpy(up_misc.p, misc.data(), misc.size());
    for (const Piece& p : pieces_) {
      // pieces the calling thread recorded are published here (clips open around a sibling group took the group's bounds after
      // their records were made); a pool thread published its pieces when it finished them
      if (p.lane <= 0) { HostTimer t(host_ns_[5]); lane(p.lane).publish(p.first, p.n, p.ext_first, p.n_ext); }
      views(p);
      const size_t l = (size_t)(p.lane + 1);
      add_run(J.runs, d_recs[l] + (size_t)p.first * sizeof(DrawRec), o_recs + (size_t)at_rec * sizeof(DrawRec), (size_t)p.n * sizeof(DrawRec), at_ext - p.ext_first, 2u);
      add_run(J.runs, d_bins[l] + (size_t)p.first * sizeof(BinRec), o_bb + (size_t)at_rec * sizeof(BinRec), (size_t)p.n * sizeof(BinRec), 0u, 1u);
      if (p.n_ext) add_run(J.runs, d_exts[l] + (size_t)p.ext_first * sizeof(QuadExt), o_ext + (size_t)at_ext * sizeof(QuadExt), (size_t)p.n_ext * sizeof(QuadExt), 0u, 0u);
      at_rec += p.n; at_ext += p.n_ext;
    }
    add_run(J.runs, d_misc, o_misc, misc.size(), 0u, 0u);
    for (const UploadRun& r : J.runs) link_bytes += r.bytes;
    if (rec_diff_upload_) {  // take the shadow this frame's successors are compared with
      if (shadow_.size() != total) shadow_.assign(total, 0);
      else if (!tables_resident) std::fill(shadow_.begin(), shadow_.end(), 0);
      size_t ar = 0, ae = 0;
      for (const Piece& p : pieces_) {
        const Lane& L = lane(p.lane);
        std::memcpy(shadow_.data() + o_recs + ar * sizeof(DrawRec), L.recs.p + p.first, (size_t)p.n * sizeof(DrawRec));
        if (p.ext_first != ae)  // (the device's copy holds frame-relative extension indices)
          for (size_t i = 0; i < p.n; i++) { DrawRec* r = reinterpret_cast<DrawRec*>(shadow_.data() + o_recs) + ar + i; if (r->op_mode & F_GENERAL) r->ext += (uint32_t)ae - p.ext_first; }
        std::memcpy(shadow_.data() + o_bb + ar * sizeof(BinRec), L.bins.p + p.first, (size_t)p.n * sizeof(BinRec));
        if (p.n_ext) std::memcpy(shadow_.data() + o_ext + ae * sizeof(QuadExt), L.exts.p + p.ext_first, (size_t)p.n_ext * sizeof(QuadExt));
        ar += p.n; ae += p.n_ext;
      }
      std::memcpy(shadow_.data() + o_misc, misc.data(), misc.size());
      shadow_layout_ = layout;
      shadow_dev_ = d_frame_.ptr;
    }
  }
  uploaded_bytes_ = link_bytes;
  J.table = UploadTable{};
  J.table.n_draws = (uint32_t)n; J.table.binbox_shift = (uint32_t)J.binbox_shift;
  J.table.bins_off = (uint32_t)o_bb; J.table.box_off = (uint32_t)o_box;
  tables_dev_ = d_frame_.ptr; tables_layout_ = layout; tables_sig_.swap(tables_sig);
  // algorithmic bytes of this frame: final store + per blur (pre-blur store is the store above for
  // a full-frame node; H read + H write + V read + V write + composite read) + records once
  int64_t bytes = 4LL * W_ * H_ + (int64_t)n * (int64_t)sizeof(DrawRec), bytes_blur = 0, bytes_fused = 0, bytes_saved = 0;
  // a cleared opaque surface stays opaque under SRC_ALPHA / ONE_MINUS_SRC_ALPHA blending (a' = sa + da (1 - sa), da = 1): a
  // fused vertical pass then replaces pixels under full coverage without reading them
  const bool surface_opaque = clear_ && (clear_rgba8_ >> 24) == 255u;
  J.big_blur = -1;
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
    if (a_v > big_area) { big_area = a_v; J.big_blur = (int)bi; stats_.bytes_blur_big_h = b_h; stats_.bytes_blur_big_v = b_v; }
  }
  bytes += bytes_blur;
  stats_.bytes_blur = bytes_blur;
  stats_.bytes_composite_main = 4LL * W_ * H_ * (clear_ ? 1 : 2) + (int64_t)J.phases[0].count * (int64_t)sizeof(DrawRec);
  // algorithmic flops of the phase-0 composite launch, SURVEY.md 8(d): per fragment ClipAA 25, DropShadow 35 + exp, InsetShadow
  // 70 + exp, AnnularAA 28 (other modes priced as ClipAA), elliptical corners + 30, blend + re-quantise + 16
  for (int k = 0; k < 4; k++) stats_.fragments_main_by_mode[k] = frag_mode_[k];
  stats_.fragments_main_elliptical = frag_ellip_;
  stats_.fragments_main_other = frag_other_;
  stats_.flops_composite_main = frag_mode_[0] * 25 + frag_mode_[1] * 36 + frag_mode_[2] * 71 + frag_mode_[3] * 28 + frag_other_ * 25 + frag_ellip_ * 30 +
                                (frag_mode_[0] + frag_mode_[1] + frag_mode_[2] + frag_mode_[3] + frag_other_) * 16;
  stats_.n_draws = (int32_t)n;
  stats_.n_phases = (int32_t)J.phases.size();
  stats_.n_blurs = (int32_t)J.blurs.size();
  stats_.n_bins = nb;
  stats_.bytes_algorithmic = bytes;
  stats_.bytes_blur_fused = bytes_fused;
  stats_.bytes_frame_implementation = bytes - bytes_saved;
  stats_.fragments = fragments_;
  if (folded) folded_br->box = folded_box;  // (the recorded frame stays what the calls produced)
  if (vram_staging(device_)) {
    store_fence();  // (what this thread wrote into device memory is on its way before the launches are)
    // ... and what it and the pool's threads wrote is pushed out of the host data path: without this, frames of fresh contexts on
    // several host threads came out wrong -- or faulted -- in ~5 % of tools/thread_churn.py runs (end of round 4)
    // (a device register every context of the device writes 1 to, from whichever host thread renders it: an atomic store, so that the
    // language knows too)
    if (hdp_flush_reg_) { __atomic_store_n(hdp_flush_reg_, 1u, __ATOMIC_RELAXED); store_fence(); }
  }
  const auto t_l0 = std::chrono::steady_clock::now();
  stats_.ms_host_record = host_record_ms_;
  stats_.ms_host_upload = std::chrono::duration<float, std::milli>(t_l0 - t_s0).count();
}

// Which blur routes a frame takes is a matter of speed only (same pixels either way): the one-kernel routes -- k_blur_fx for a node
// that covers the frame, k_blur_small for a small one: fewer dependent launches, half the bytes -- or the two passes as two
// kernels.  Rounds 3 and early 4 chose per frame: one-kernel routes for a frame rendered alone, two-pass routes when another
// context of the process had submitted a frame within the last millisecond, where they measured 3 - 4 % faster (135 against 141
// Gpixel/s with four contexts).  With the last bubbles out of the launch chain (no event behind the upload, bin workgroups per
// phase box) that has turned: one-kernel routes 150.0 - 150.6 Gpixel/s against 145.1 - 147.3 with four contexts (tools/ab_routes.sh,
// three alternations on one box), and 84 against 94 us one frame at a time.  So: the one-kernel routes, always
// (fdh_set_blur_route / FDH_BLUR_FUSED = 0: the two-pass routes).
void Context::pick_routes() {
  static const int fx_env = [] { const char* e = std::getenv("FDH_BLUR_FUSED"); return e ? (std::atoi(e) != 0 ? 1 : 0) : -1; }();
  const int route = blur_route_ >= 0 ? blur_route_ : fx_env;
  latency_routes_ = route != 0;
}

// More pieces than the upload's run table holds (a frame with many parallel sibling groups): they are copied together into one
// spare lane, in order, extension indices re-based -- the frame becomes one piece again.
void Context::consolidate_pieces() {
  std::unique_ptr<Lane>& slot = merge_lane_[(size_t)staging_i_];
  if (!slot) { slot.reset(new Lane()); slot->set_pinned(!host_only_, device_); }
  Lane& S = *slot;
  S.clear();
  S.recs.reserve(n_total_); S.bins.reserve(n_total_); S.exts.reserve(n_ext_total_);
  for (const Piece& p : pieces_) {
    const Lane& L = lane(p.lane);
    const size_t r0 = S.recs.n, e0 = S.exts.n;
    S.recs.append(L.recs.p + p.first, p.n);
    S.bins.append(L.bins.p + p.first, p.n);
    S.exts.append(L.exts.p + p.ext_first, p.n_ext);
    S.boxes.append(L.boxes.p + p.first, p.n);
    for (size_t i = r0; i < S.recs.n; i++) if (S.recs[i].op_mode & F_GENERAL) S.recs[i].ext += (uint32_t)e0 - p.ext_first;
  }
  Piece all;
  all.lane = -1; all.first = 0; all.n = n_total_; all.ext_first = 0; all.n_ext = n_ext_total_;
  pieces_.assign(1, all);
}

void Context::pool_slots(int slots) {
  while ((int)pool_recs_.size() < slots) pool_recs_.emplace_back(new Recorder(this, false));
  for (int s = 0; s < slots; s++) {
    Lane& Ln = ensure_lane(s + 1);
    if (Ln.stamp != frame_no_) {  // first use in this frame
      Ln.clear();
      Ln.count_begin((W_ + kBin - 1) / kBin, (H_ + kBin - 1) / kBin);
      Ln.stamp = frame_no_;
    }
  }
}
int Context::walk_threads() const { return walk_threads_ >= 0 ? walk_threads_ : WalkPool::default_helpers(); }

// The launches of one prepared frame: the upload (a kernel on the render stream gathering the recorded pieces out of pinned host
// memory), then binning, blur passes and compositing.  Submit thread (or the caller's, FDH_CREATE_SYNC_SUBMIT).
// A frame every phase of which holds at most 64 draws, none of them a rotated quad or a curve (their entries need the bin kernel's
// per-strip tests): no bin launch, the compositor's waves make their entries themselves (k_composite_tiles, "direct").  FDH_DIRECT=0: never.
static bool direct_frame(const LaunchJob& J) {
  // (FDH_FORCE_KERNEL_PATHS=3 / 19 / 8, the test hook that puts a frame on the builds with the slot path / the rotated-quad path: those
  // have no direct form)
  static const bool on = [] {
    const char* e = std::getenv("FDH_DIRECT");
    const char* f = std::getenv("FDH_FORCE_KERNEL_PATHS");
    const int forced = f ? std::atoi(f) : 0;
    return (!e || std::atoi(e) != 0) && forced != 3 && forced != 8 && forced != 19;
  }();
  if (!on || J.phases.empty()) return false;
  for (const Phase& ph : J.phases)
    if (ph.count > 64 || ph.has_rot || ph.has_slow) return false;
  return true;
}

void Context::issue(LaunchJob& J) {
  const auto t_l0 = std::chrono::steady_clock::now();
  FDH_HIP(hipSetDevice(device_));
  {
    UploadTable& T = J.table;
    T.n_runs = 0; T.copy_units = 0;
    for (const UploadRun& r : J.runs) {
      if (T.n_runs >= (uint32_t)kMaxUploadRuns) throw Error(FDH_ERR_UNSUPPORTED, "upload: run table overflow");
      T.run[T.n_runs] = r;
      T.unit_first[T.n_runs] = T.copy_units;
      T.copy_units += (r.bytes + 1023u) / 1024u;
      T.n_runs++;
    }
    launch_upload_frame(stream_, J.d_dst, T);
  }
  // WHO RELEASES THE STAGING SET.  The calling thread may write into a set of lanes again once the upload that read it has run.
  // An event recorded behind the upload said so until round 4 -- and cost the GPU 5.7 - 5.9 us of idle time on every frame: the
  // bin launch started that long after the upload had ended, whether the event was a packet of its own (hipEventRecord) or rode on
  // the upload's dispatch (hipExtLaunchKernelGGL's stop event: 4.6 us), and not at all without one
  // (tools/trace_gaps.sh).  Now the BIN launch says it: its first wave stores the frame's sequence number to a word of pinned host
  // memory (k_bin_draws) -- it has started, so the upload in front of it is done -- and begin_frame compares that word.
  // (A frame without a bin launch -- no phase -- keeps the event.)
  uint32_t seq = 0;
  if (J.staging_slot >= 0) {
    const bool binned = !J.phases.empty() && J.bins_x * J.bins_y > 0 && !direct_frame(J);
    if (binned && !seq_host_) {
      FDH_HIP(hipHostMalloc((void**)&seq_host_, 64, hipHostMallocDefault));
      *seq_host_ = 0;
    }
    if (binned) {
      seq = ++upload_seq_;
      if (seq == 0) seq = ++upload_seq_;  // (0 = no store)
      staging_seq_[J.staging_slot] = seq;
      staging_busy_[J.staging_slot] = 2;
    } else {
      FDH_HIP(hipEventRecord(staging_ev_[J.staging_slot], stream_));
      staging_busy_[J.staging_slot] = 1;
    }
  }
  launch_frame(J, false, seq);
  launch_ms_.store(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_l0).count(), std::memory_order_relaxed);
}


void Context::launch_frame(const LaunchJob& J, bool profile, uint32_t upload_seq) {
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
  const bool direct = direct_frame(J);
  span_begin(0);
  BinParams B;
  B.binrec = dv_.binrecs; B.binbox = dv_.binbox; B.chunkbox = dv_.chunkbox; B.n_draws = J.n_recs; B.binbox_shift = binbox_shift_; B.lists = J.lists; B.counts = J.counts; B.phase_first = dv_.phase_first; B.draws = dv_.recs; B.exts = dv_.exts;
  B.refine = 0;
  for (const Phase& ph : J.phases) if (ph.has_rot || ph.has_slow) B.refine = 1;
  B.n_phases = np; B.bins_x = bins_x_; B.bins_y = bins_y_; B.stride = list_stride_;
  if (upload_seq) { B.seq_out = const_cast<uint32_t*>(seq_host_); B.seq = upload_seq; }
  static const bool sub_on = [] { const char* e = std::getenv("FDH_BIN_SUBGRIDS"); return !e || std::atoi(e) != 0; }();
  if (sub_on && np >= 1 && np <= BinParams::kBinSubs) {  // later phases: the bins their compositor launch reads (the same Phase::bin_* box)
    int at = 0;
    for (int p = 0; p < np; p++) {
      const Phase& ph = J.phases[p];
      const bool whole = p == 0;
      const int x0 = whole ? 0 : std::max(0, ph.bin_x0), y0 = whole ? 0 : std::max(0, ph.bin_y0);
      const int x1 = whole ? bins_x_ : std::min(bins_x_, ph.bin_x1), y1 = whole ? bins_y_ : std::min(bins_y_, ph.bin_y1);
      const int nx = std::max(0, x1 - x0), ny = std::max(0, y1 - y0);
      B.sub_first[p] = at; B.sub_x0[p] = x0; B.sub_y0[p] = y0; B.sub_nx[p] = std::max(1, nx);
      at += nx * ny;
    }
    B.sub_first[np] = at;
    B.sub_n = np;
  }
  if (!direct) launch_bin(stream_, B);
  span_end();
  // Phase 0's full-grid composite takes its bins longest-list first, in the order its predecessor sorted (an extra
  // wavefront of that launch); it sorts this frame's counts for its successor.  Any permutation is a correct schedule.
  const int order_key = bins_x_ * 65536 + bins_y_;  // entries are (row << 16 | column) of THIS grid
  if (order_valid_ && order_nb_ != order_key) order_valid_ = false;  // frame size changed
  const bool sorting = J.clear && np > 0 && nb <= 8192 && J.phases[0].count > 0 && !direct;  // (a direct frame has no counts to sort by)
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
  // Quarter strips for the frame's longest lists (k_composite_tiles, round 6): the sorting waves of earlier full-frame launches left, per
  // class of bins, how many hold at least deep_min draws; that many leading positions of the order (x 8 classes) get four waves per
  // strip.  Whatever value is there serves -- a frame or two stale, 0 before the first launch has run: any count is a correct schedule.
  // The threshold goes by who else renders: a deep strip holds four wave slots and its waves mostly wait, which is what a frame that has
  // the GPU to itself wants (the launch is its longest strips' chain) and what frames of OTHER contexts pay for.  Bench tree through
  // fdh_render_frame (profiles/r06_deep_in_flight.txt): one context, 1080p, 49.7 us per frame without deep strips, 46.6 with bins of >= 24
  // draws, 46.4 with >= 40; four contexts in flight 30.3 / 31.1 / 28.9 us -- so 24 for a device's only context, 40 beside others.
  static const int deep_env = [] { const char* e = std::getenv("FDH_DEEP_MIN"); return e ? std::atoi(e) : -1; }();
  const int deep_min = deep_env >= 0 ? deep_env : (vram_contexts_alive(device_) > 1 ? kDeepMinInFlight : kDeepMinDefault);
  int deep_k8 = 0;
  if (sorting && deep_min > 0) {
    if (!deep_host_) {
      FDH_HIP(hipHostMalloc((void**)&deep_host_, 64, hipHostMallocDefault));
      for (int c = 0; c < 8; c++) deep_host_[c] = 0;
    }
    if (order_now) {
      uint32_t most = 0;
      for (int c = 0; c < 8; c++) most = std::max(most, (uint32_t)deep_host_[c]);
      deep_k8 = 8 * (int)std::min<uint32_t>(most, 1024u);
    }
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
        bp.node_pixels = (long long)(j.x1 - j.x0) * (j.y1 - j.y0);
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
        if (!done && j.fuse_draw < 0 && J.latency_routes && blur_one_kernel_ok(j.x1 - j.x0, j.y1 - j.y0, j.taps.reach)) {
          // a small region: one kernel, source window -> LDS -> horizontal -> LDS -> vertical -> the backdrop surface
          bp.fuse_draw = -1;
          bp.src = cur; bp.dst = backdrop_;
          bp.x0 = j.x0; bp.x1 = j.x1; bp.y0 = vy0; bp.y1 = vy1;
          span_begin(ph.blur == big_blur_ ? 6 : 4);
          launch_blur_small(stream_, bp);
          span_end();
          done = true;
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
    C.direct = direct ? 1 : 0; C.direct_first = ph.first; C.direct_n = ph.count; C.binrec = dv_.binrecs;
    C.order = full ? order_now : nullptr;
    C.order_next = full ? order_next : nullptr;
    C.deep_k8 = full ? deep_k8 : 0;
    if (full) stats_.deep_bins = (float)((ph.has_slow || ph.has_rot || ph.has_atlas || ph.has_masks || !order_now || !order_next) ? 0 : std::min(deep_k8, 8 * ((nb + 7) / 8)));
    C.deep_min = deep_min;
    static const int deep_strip_min = [] { const char* e = std::getenv("FDH_DEEP_STRIP_MIN"); return e ? std::atoi(e) : kDeepStripMinDefault; }();
    C.deep_strip_min = deep_strip_min;
    C.deep_out = (full && order_next && deep_min > 0) ? const_cast<uint32_t*>(deep_host_) : nullptr;
    C.has_slow = ph.has_slow ? 1 : 0;
    C.has_slow_atlas = ph.has_slow_atlas ? 1 : 0;
    C.has_rot = ph.has_rot ? 1 : 0;
    C.has_atlas = ph.has_atlas ? 1 : 0;
    C.has_masks = ph.has_masks ? 1 : 0;
    C.mask_spill = J.mask_spill; C.spill_stride = J.spill_stride;
    span_begin(p == 0 ? 1 : 2);
    launch_composite(stream_, dv_.recs, dv_.exts, C);
    span_end();
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
void Context::frame_device_ptr(void** p, int* w, int* h, int64_t* pitch_bytes) {
  need_device("frame_device_ptr");
  drain();
  if (!fb_) throw Error(FDH_ERR_INVALID, "no frame surface yet");
  *p = fb_; *w = W_; *h = H_; *pitch_bytes = (int64_t)W_ * 4;
}

}  // namespace fdh
