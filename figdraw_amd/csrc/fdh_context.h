// fdh_context.h -- host side of libfigdraw_hip.so: the BackendContext-shaped state machine that turns
// backend calls into draw records (what glcontext.nim does into vertex streams) and submits them.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/figdraw_hip.h"
#include "fdh_kernels.h"

namespace fdh {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void hip_check(hipError_t e, const char* what);
#define FDH_HIP(x) ::fdh::hip_check((x), #x)

// 2D affine part of the vmath Mat4 stack: [a c tx; b d ty]
struct Aff {
  float a = 1, b = 0, c = 0, d = 1, tx = 0, ty = 0;
};

// Fault-hunting builds (-DFDH_POISON=<byte>): every fresh device allocation is filled with that byte before its first use -- a kernel that
// reads memory nothing has written yet then reads the same garbage every time, not what the allocation's previous owner left.
void poison_fresh(void* p, size_t bytes);  // fdh_context.cpp
template <typename T>
struct DeviceBuf {
  T* ptr = nullptr;
  size_t cap = 0;
  void reserve(size_t n) {
    if (n <= cap) return;
    size_t want = cap ? cap : 256;
    while (want < n) want *= 2;
    T* fresh = nullptr;  // allocate first: a failing hipMalloc (FDH_HIP throws) must leave ptr / cap describing a live block
    FDH_HIP(hipMalloc((void**)&fresh, want * sizeof(T)));
    poison_fresh(fresh, want * sizeof(T));
    if (ptr) (void)hipFree(ptr);
    ptr = fresh;
    cap = want;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
};

template <typename T>
struct PinnedBuf {
  T* ptr = nullptr;
  size_t cap = 0;
  void reserve(size_t n) {
    if (n <= cap) return;
    size_t want = cap ? cap : 256;
    while (want < n) want *= 2;
    T* fresh = nullptr;
    FDH_HIP(hipHostMalloc((void**)&fresh, want * sizeof(T), hipHostMallocDefault));
    if (ptr) (void)hipHostFree(ptr);
    ptr = fresh;
    cap = want;
  }
  void release() {
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
};

// A growable array of plain data, in ordinary memory or -- `pinned` -- in host memory the GPU fetches over PCIe (hipHostMalloc).
// Pinned memory is written once, front to back, and never read by the CPU: on this platform the CPU's loads from it are not
// served from its caches (a record read back from a pinned lane cost ~100 ns; round 4 measured 82 us for a 700-record frame that
// took 4 us from ordinary memory).  Growth copies what the array holds (doubling: rare once a context has seen its scene).
//
// `vram` (with `pinned`): the same role in DEVICE memory the host writes through the PCIe BAR.  On the MI355X boxes all of HBM is
// host-addressable (large BAR): the CPU's stores into a hipExtMallocWithFlags(hipDeviceMallocUncached) block are write-combined
// posted writes (measured: 128 KB in 3.2 us = 41 GB/s, tools/microbench/bar_write.hip) and the GPU reads what they wrote from its
// own memory instead of fetching it over the link with a round trip per lane: k_upload_frame 7.1 -> ~3 us.  Write-only for the
// CPU -- a load from it crosses the link uncached -- so growth does NOT carry contents over (callers set n = 0 first, as the pinned
// mirrors always did), and a store fence (store_fence()) precedes the hand-over to whoever launches the kernels.
bool vram_staging(int device);  // large-BAR device and FDH_VRAM_STAGING != 0, decided per device ordinal (fdh_context.cpp)
// Device blocks for staging come from, and go back to, a process-wide store by size class (powers of two from 4 KB): they are
// never handed back to the driver while the process lives.  A block the driver recycles may be memory another allocation's
// kernels wrote through the L2s; the host's stores reach memory BESIDE those caches, and a line written back later lands on top
// of them (tools/thread_churn.py: fresh contexts on four host threads, wrong first frames or a fault in ~5 % of runs).  Staging
// blocks are only ever written by the host and read uncached by the device, so recycled among themselves they carry no such lines.
// The store is keyed by DEVICE ORDINAL: a block allocated on GPU 0 never reaches a context on GPU 1.
void* vram_block_acquire(int device, size_t bytes, size_t* size_class);  // throws Error on failure
void vram_block_release(int device, void* p, size_t size_class);
size_t vram_store_bytes(int device);   // bytes the store of a device holds (released blocks)
int vram_contexts_alive(int device);   // device contexts alive on `device` (the deep strips' threshold goes by it: Context::launch_frame)
void vram_context_born(int device);    // a device context exists on `device` ...
void vram_context_gone(int device);    // ... and is gone: with the last one, the device's store is trimmed to kVramStoreKeep
constexpr size_t kVramStoreKeep = (size_t)16 << 20;
inline void store_fence() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_sfence();
#else
  __sync_synchronize();
#endif
}
template <typename T>
struct HostVec {
  T* p = nullptr;
  size_t n = 0, cap = 0;
  bool pinned = false, vram = false;
  int dev = 0;  // device ordinal of a vram block
  HostVec() = default;
  HostVec(const HostVec&) = delete;
  HostVec& operator=(const HostVec&) = delete;
  ~HostVec() { release(); }
  void free_block(T* q) { if (!q) return; if (pinned && vram) vram_block_release(dev, q, vram_bytes_); else if (pinned) (void)hipHostFree(q); else std::free(q); }
  size_t vram_bytes_ = 0;  // size class of the device block p (vram_block_acquire)
  void release() {
    free_block(p);
    p = nullptr; n = cap = 0;
  }
  void reserve(size_t want) {
    if (want <= cap) return;
    size_t c = cap ? cap : 256;
    while (c < want) c *= 2;
    T* fresh = nullptr;
    size_t fresh_bytes = 0;
    if (pinned && vram) fresh = static_cast<T*>(vram_block_acquire(dev, c * sizeof(T), &fresh_bytes));
    else if (pinned) FDH_HIP(hipHostMalloc((void**)&fresh, c * sizeof(T), hipHostMallocDefault));
    else if (!(fresh = static_cast<T*>(std::aligned_alloc(64, (c * sizeof(T) + 63) & ~(size_t)63)))) throw std::bad_alloc();
    if (n && !(pinned && vram)) std::memcpy(static_cast<void*>(fresh), static_cast<const void*>(p), n * sizeof(T));
    free_block(p);
    p = fresh; cap = c; vram_bytes_ = fresh_bytes;
  }
  // the block as the GPU addresses it
  const uint8_t* device_view() const {
    if (!p) return nullptr;
    if (pinned && vram) return reinterpret_cast<const uint8_t*>(p);
    void* d = nullptr;
    FDH_HIP(hipHostGetDevicePointer(&d, const_cast<void*>(static_cast<const void*>(p)), 0));
    return static_cast<const uint8_t*>(d);
  }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* slot() { if (n == cap) reserve(n + 1); return p + n; }  // the next element's place (not yet counted)
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T& back() { return p[n - 1]; }
  void clear() { n = 0; }
  void append(const T* src, size_t k) { if (!k) return; reserve(n + k); std::memcpy(static_cast<void*>(p + n), static_cast<const void*>(src), k * sizeof(T)); n += k; }
};

// An atlas entry and, when its level-0 texels were seen on the host (fdh_put_image), the bounds of what is IN it: for eight
// levels t = 0, 16, .. 112 the box (entry-relative texels, x1 / y1 exclusive) of texels whose alpha, and whose largest colour
// channel, exceeds t.  A draw whose coverage is exactly 0 wherever the sampled value is <= t (a glyph image: alpha 0; an MSDF
// image: distance below threshold - 0.5 / screen range) shrinks its pixel bounds to the image of that box: the strips outside
// would blend with alpha 0, which leaves every texel as it is (Recorder::shrink_to_ink).
constexpr int kInkLevels = 8;
struct InkBox { int16_t x0, y0, x1, y1; };
struct AtlasEntry {
  int x, y, w, h;
  bool has_ink = false;
  InkBox ink_a[kInkLevels], ink_rgb[kInkLevels];
};

struct Phase {
  int first = 0, count = 0;
  int blur = -1;  // index into blurs: executed before this phase's composite
  int bin_x0 = 0, bin_y0 = 0, bin_x1 = 0, bin_y1 = 0;  // bins touched by the phase's draws
  bool has_masks = false; // clip / rect-mask ops present
  bool has_atlas = false; // axis-aligned atlas quads at >= 1:1 present (k_composite_tiles<2> unless has_slow)
  bool has_slow = false;  // some draw needs k_composite_tiles<true> (atlas / rotated quad / bezier / rect-mask setup)
  bool has_slow_atlas = false;  // ... and one of them is an atlas quad off the 4-wide path (rotated, or minified over mip levels): the 168-register form of that build
  bool has_rot = false;   // rotated / skewed SDF quads whose edge functions fit 32 bits (F_EDGE32): the 4-wide path of builds <8> and <3>
};
struct BlurJob {
  float radius;
  int x0, y0, x1, y1;  // footprint: the mode-17 quad's pixel bounds
  int fuse_draw;       // record index of the consuming mode-17 quad when k_blur_v composites it, else -1
  BlurTaps taps;
};

// ------------------------------------------------------------------ the recorded frame
// A frame's draw records are produced in their FINAL form while the calls arrive -- the 128-byte DrawRec the compositor reads,
// the 24-byte BinRec the bin kernel reads (pixel bounds, saturated core, list-entry flags), the quad extensions: nothing is
// built a second time at submit.  A LANE is what one thread records: lane 0 belongs to the thread that calls the context,
// lanes 1.. to the walk pool's threads (fdh_frontend.cpp: large sibling groups of the scene tree are decomposed in parallel).
// The frame in painter's order is a list of PIECES, each a run of consecutive records of one lane.  A finished piece is
// PUBLISHED -- copied, by the thread that recorded it, into the lane's pinned mirror arrays -- and the upload kernel gathers the
// published pieces into the dense device arrays (k_upload_frame).
struct Lane {
  HostVec<DrawRec> recs;
  HostVec<BinRec> bins;   // bins[i].box IS the bounds of record i (clip pushes: the union of their content, final at the pop)
  HostVec<QuadExt> exts;  // DrawRec::ext of an F_GENERAL record indexes THIS array; the upload re-bases it
  HostVec<uint32_t> boxes;  // the records' 4-byte bin boxes (what k_bin_draws scans; the device derives its own from the BinRecs):
                            // kept here for the chunk boxes -- the union box of every 256 draws -- which prepare builds over the pieces
  HostVec<DrawRec> up_recs;  // pinned mirrors (device contexts): what the GPU reads; element i = element i of the array above
  HostVec<BinRec> up_bins;
  HostVec<QuadExt> up_exts;
  bool device = false;
  const uint8_t *d_recs = nullptr, *d_bins = nullptr, *d_exts = nullptr;  // the mirrors as the device sees them (taken when a mirror is allocated)
  size_t pub_recs = 0, pub_exts = 0;  // elements below these may have been published this frame (kept across a mirror's growth)
  // List stride (the largest number of list entries any bin of any phase can receive: it sizes the bin lists): a 2-D difference
  // array over the bin grid, four updates per record when its bounds are final, evaluated per phase (count_close).
  std::vector<int> diff;
  int dw = 0, dh = 0;
  int tx0 = 0, ty0 = 0, tx1 = 0, ty1 = 0;
  bool touched = false;
  uint64_t stamp = 0;  // the frame a pool thread's lane was last cleared for
  void set_pinned(bool on, int dev) {
    device = on;
    up_recs.pinned = up_bins.pinned = up_exts.pinned = on;
    up_recs.vram = up_bins.vram = up_exts.vram = on && vram_staging(dev);
    up_recs.dev = up_bins.dev = up_exts.dev = dev;
  }
  void clear() { recs.clear(); bins.clear(); exts.clear(); boxes.clear(); pub_recs = pub_exts = 0; }
  void publish(uint32_t first, uint32_t n, uint32_t ext_first, uint32_t n_ext);  // records / extensions are final: copy them to the mirrors
  void publish_bytes(int array, size_t at, size_t len);                          // ... a byte range of one array (0 recs, 1 bins, 2 exts)
  void count_begin(int bins_x, int bins_y);
  void count_add(const BBox& b);
  int count_close();  // the largest count of any bin since the last close; leaves the array zeroed
};

// what a run of records adds to its phase (kept per parallel chunk by the walk pool's threads, merged by the calling thread)
struct PhaseSum {
  BBox u{0, 0, 0, 0};  // union of the records' final bounds
  bool has_masks = false, has_atlas = false, has_slow = false, has_rot = false, has_slow_atlas = false;
  int deepest = 0;     // deepest clip nesting reached, relative to the run's start
  int64_t frag_mode[4] = {0, 0, 0, 0}, frag_ellip = 0, frag_other = 0;  // covered fragments by SdfMode 3 / 7 / 9 / 12
};
struct Piece {
  int lane = 0;                     // (-1: the lane Context::consolidate_pieces copies a frame of too many pieces into)
  uint32_t first = 0, n = 0;        // records [first, first + n) of the lane
  uint32_t ext_first = 0, n_ext = 0;  // their quad extensions
};

struct RectMaskEntry { int kind; };  // 1 = fast analytic, 2 = real mask (glcontext.nim:36-44)
class Context;
struct SerialOnly {};  // thrown by a pool thread's recorder at a call only the calling thread can serve (a blur node, the lazily
                       // created 4x4 "rect" atlas image): the calling thread then walks that sibling group itself

// The BackendContext state machine (glcontext.nim): transform stack, clip / rect-mask stacks, and the draw calls, each turning
// into records of ONE lane.  Context derives from it (lane 0: the C entry points and the serial walk); the walk pool's threads
// each own one more (fdh_frontend.cpp).
class Recorder {
 public:
  Recorder(Context* cx, bool is_main) : cx_(cx), is_main_(is_main) {}
  void save_transform();
  void restore_transform();
  void translate(float x, float y);
  void rotate(float a);
  void scale(float sx, float sy);
  void apply_transform(const float m[16]);
  bool transform_mirrors_y() const;
  void set_aa(float aa);
  float aa() const { return aa_; }
  void draw_rounded_rect_sdf(const float rect[4], const FdhColor colors[4], const float rx[4], const float ry[4], int mode,
                             float factor, float spread, const float shape[2], int fill_mode, FdhColor mid, FdhColor stop,
                             float mid_pos);
  void draw_rounded_rect_fill(const float rect[4], const FdhFill& fill, const float rx[4], const float ry[4], int mode,
                              float factor, float spread, const float shape[2]);
  void draw_image(int64_t key, const float pos[2], const FdhColor colors[4], const float size[2], bool flip_y);
  void draw_image_adj(int64_t key, const float pos[2], FdhColor color, const float size[2]);
  void draw_msdf(int64_t key, const float pos[2], FdhColor color, const float size[2], float px_range, float sd_threshold,
                 float stroke_weight, bool mtsdf, bool flip_y);
  void draw_quadratic_bezier_sdf(const float rect[4], const FdhFill& fill, const float p0[2], const float p1[2], const float p2[2],
                                 float stroke_weight, int cap);
  void draw_filled_quad(const float verts[8], const FdhColor colors[4]);
  void draw_rect(const float rect[4], FdhColor color);
  void draw_backdrop_blur(const float rect[4], const float rx[4], const float ry[4], float blur_radius);
  void begin_mask(const float rect[4], const float rx[4], const float ry[4]);
  void end_mask();
  void pop_mask();
  void begin_rect_mask(const float rect[4], const float rx[4], const float ry[4]);
  void pop_rect_mask();
  void set_subpixel_shift(float s);
  bool subpixel_enabled() const;
  bool subpixel_variants() const;
  bool culling() const;
  // would a quad over `rect` (pre-transform units), grown by `pad` pixels on every side, reach a pixel the frame will produce?
  bool rect_visible(const float rect[4], float pad) const;

  // ---- state (the calling thread's recorder is reset by begin_frame; a pool thread's by adopt())
  Context* const cx_;
  const bool is_main_;
  Lane* lane_ = nullptr;
  Aff mat_;
  std::vector<Aff> mats_;
  float aa_ = 1.2f;
  float subpixel_shift_ = 0.0f;
  bool mask_begun_ = false;
  int mask_depth_ = 0;
  std::vector<RectMaskEntry> rect_masks_;
  int outer_rect_masks_ = 0;        // pool threads: rect masks open around the sibling group (begin_rect_mask nests differently inside one)
  std::vector<uint32_t> open_ops_;  // lane indices of the open MASK_PUSH / RMASK_BEGIN records (their bounds grow with their content)
  bool outer_open_ = false;         // pool threads: clips are open around the sibling group; their bounds take ...
  BBox outer_union_{0, 0, 0, 0};    // ... the union of everything emitted here (merged by the calling thread)
  int depth_now_ = 0;               // clip nesting inside the current phase (open pushes are re-emitted at a phase's start)
  PhaseSum sum_;                    // of the records committed since the last take_sum()
  int64_t fragments_ = 0, culled_draws_ = 0;
  int phase_floor_ = 0;             // lane index of the first record of the current phase in this lane (LE_SHARE never crosses it)

 protected:
  DrawRec& next_rec();  // the lane's next record slot, zeroed (counted by emit_* when the draw survives culling)
  bool emit_quad(DrawRec& r, float x0, float y0, float x1, float y1, bool count_fragments);  // false: culled, nothing was recorded
  bool emit_quad_pts(DrawRec& r, const float vx[4], const float vy[4], bool count_fragments);
  struct QuadPx { float px[4], py[4]; BBox b; bool finite; };
  void quad_corners(const float vx[4], const float vy[4], QuadPx& q) const;
  bool emit_corners(DrawRec& r, const QuadPx& q, bool count_fragments);
  void push_rec(BBox b);  // count the slot next_rec() handed out
  void commit_bins(uint32_t idx);  // the record's bounds are final: list-entry flags, list-stride count, phase summary
  void link_share(uint32_t idx);   // LE_SHARE on idx - 1 when record idx is drawn over the same quad with the same shape
  bool bbox_visible(const BBox& b) const;
  const AtlasEntry& rect_entry();
  void shrink_to_ink(const AtlasEntry& e, bool use_alpha, int level_t);
  friend class Context;
};

// Retained scene (fdh_scene_*): the library-side half of the reference's RenderFragments (renderfragments.nim:426-544) --
// a deep copy of the node tree plus, per root, the draw records its decomposition produced.  A frame re-decomposes only the
// roots an update touched; every other root's records are spliced back from the cache.
struct RetainedRoot {
  std::vector<DrawRec> recs;     // in device form (Recorder::push_rec)
  std::vector<BinRec> bins;      // bounds, cores, list-entry flags (the last record's LE_SHARE is decided again at every splice)
  std::vector<QuadExt> exts;     // of this root's records, DrawRec::ext relative to exts.front()
  PhaseSum sum;
  int64_t fragments = 0;
  bool cacheable = false;        // no blur node inside (those split the frame into phases: re-walked every frame)
  bool dirty = true;
  uint64_t atlas_epoch = 0;      // image draws carry atlas positions: stale after the atlas was rebuilt
  int cull_y0 = 0, cull_y1 = 0;  // the rows the records were culled to (Context::begin_frame)
};
struct RetainedLayer {
  int32_t zlevel = 0;
  std::vector<FdhFig> nodes;
  std::vector<int32_t> roots;
  std::vector<RetainedRoot> cache;  // parallel to `roots`
};
struct RetainedScene {
  bool valid = false;
  float fw = 0, fh = 0, rgba[4] = {1, 1, 1, 1};
  bool clear = true;
  float ui_scale = 1.0f, aa = 0.0f;
  bool subpixel = false, variants = false;  // the text front-end settings the cached records were made under
  uint32_t table_epoch = 0, table_epoch_seen = 0;  // bumped when a glyph-variant table first appears (rebase_side)
  std::vector<RetainedLayer> layers;
  std::vector<FdhGlyph> glyphs;
  std::vector<int64_t> variant_ids;  // [glyphs][FDH_GLYPH_VARIANT_STEPS] or empty
  std::vector<FdhDrawOp> ops;
  std::vector<float> controls;
  std::vector<FdhTextRect> text_rects;
  int64_t roots_walked = 0, roots_reused = 0;  // of the last fdh_scene_render
};

// What the launch side needs of one frame: filled by Context::prepare on the calling thread (which also fills the run table
// the upload kernel works through), consumed by Context::issue / launch_frame on the context's submit thread, and kept
// for fdh_replay / fdh_profile.
struct LaunchJob {
  struct View { DrawRec* recs = nullptr; QuadExt* exts = nullptr; BinRec* binrecs = nullptr; int* phase_first = nullptr; uint32_t* binbox = nullptr; uint32_t* chunkbox = nullptr; };
  int W = 0, H = 0;
  int rec_y0 = 0, rec_y1 = 0;  // rows the records were culled to (the frame, or a stripe + blur reach): a replay must stay inside
  bool clear = true;
  bool latency_routes = true;  // the frame was recorded for the one-kernel blur routes (Context::pick_routes)
  uint32_t clear_rgba8 = 0xFFFFFFFFu;
  std::vector<Phase> phases;
  std::vector<BlurJob> blurs;
  std::vector<const uint4*> mx_w_h, mx_w_v;  // per blur job: weight fragments of the matrix-pipe passes (in the frame block), or null
  std::vector<char> blur_fused;              // per blur job: both passes run as ONE out-of-place kernel (full-frame nodes)
  int n_fused = 0;
  int n_recs = 0;
  View dv;                 // typed views into the device frame block
  uint32_t* mask_spill = nullptr;  // clip levels beyond kMaskDepth, [level][strip][lane]; spill_stride dwords per level
  size_t spill_stride = 0;
  uint2* lists = nullptr;  // bin lists / counts (device)
  uint32_t* counts = nullptr;
  int bins_x = 0, bins_y = 0, list_stride = 0, binbox_shift = 0;
  int big_blur = -1;       // index of the frame's largest blur job (its passes are timed on their own)
  // the upload: the runs k_upload_frame gathers (records, bin records, extensions of every piece; phase table; blur tables)
  std::vector<UploadRun> runs;
  UploadTable table;       // (filled from `runs` when the frame is issued)
  void* d_dst = nullptr;
  int staging_slot = -1;
};

class Context : public Recorder {
 public:
  Context(int atlas_size, float pixel_scale, int device, uint32_t flags);
  ~Context();

  // BackendContext surface: the draw calls are Recorder's
  void begin_frame(int w, int h, bool clear, const float rgba[4]);
  void end_frame();
  float pixel_scale() const { return pixel_scale_; }
  void set_subpixel_enabled(bool e) { subpixel_enabled_ = e; }
  void set_subpixel_variants(bool e) { subpixel_variants_ = e; }
  void set_text_lcd_filtering(bool e) { text_lcd_filtering_ = e; }
  bool text_lcd_filtering() const { return text_lcd_filtering_; }
  void comm_info(int* rank, int* world) const { *rank = comm_ ? comm_rank_ : 0; *world = comm_ ? comm_world_ : 1; }

  // atlas
  void put_image(int64_t key, int w, int h, const uint8_t* rgba, int out_rect[4]);
  void put_glyph_image(int64_t key, int w, int h, const uint8_t* rgba, uint32_t flags, int out_rect[4]);
  void put_glyph_outline(int64_t key, int w, int h, const float* segs, int n, uint32_t flags, int out_rect[4]);
  void update_image(int64_t key, int w, int h, const uint8_t* rgba);
  void put_mips(int64_t key, int n, const int* ws, const int* hs, const uint8_t* const* premul_rgba, int out_rect[4]);
  void put_flippy(int64_t key, const uint8_t* data, size_t n, int out_rect[4]);
  void remove_image(int64_t key) { entries_.erase(key); atlas_epoch_++; }
  bool has_image(int64_t key) const { return entries_.count(key) != 0; }
  void reset_atlas(int minimum_size);
  int atlas_size() const { return atlas_size_; }
  int64_t atlas_packed_area() const;

  // readback / interop
  void read_pixels(int x, int y, int w, int h, uint8_t* out);
  void frame_device_ptr(void** p, int* w, int* h, int64_t* pitch_bytes);
  void debug_read_surface(int which, uint8_t* out);
  // call recorder (fdh_record_begin / fdh_record_json): the backend-level calls the scene front-end makes, as JSON
  void record_begin();
  const char* record_json();
  void sync();
  void set_stream(void* s);

  // scene front-end (fdh_frontend.cpp)
  void set_ui_scale(float s) { ui_scale_ = s; }
  float ui_scale() const { return ui_scale_; }
  void render_frame(const FdhScene* scene, float fw, float fh, bool clear, const float rgba[4]);
  // retained scenes (fdh_frontend.cpp)
  void scene_retain(const FdhScene* scene, float fw, float fh, bool clear, const float rgba[4]);
  void scene_update_nodes(int layer, int first, int count, const FdhFig* nodes, const FdhScene* side);
  void scene_replace_root(int layer, int slot, const FdhFig* subtree, int n, const FdhScene* side, bool insert);
  void scene_render();
  void scene_stats(int64_t* walked, int64_t* reused) const { *walked = retained_.roots_walked; *reused = retained_.roots_reused; }
  int64_t uploaded_bytes() { drain(); return uploaded_bytes_; }
  void debug_verify_upload(uint32_t out[24]);
  void debug_bin_digest(uint64_t out[8]);        // fdh_debug_bin_digest (fdh_record.cpp)  // fdh_debug_verify_upload (fdh_record.cpp)
  uint64_t record_digest();  // FNV-1a over the last frame's draw records (diagnostic: works on record-only contexts)

  // multi-GPU: the gather over RCCL (fdh_comm.cpp)
  void comm_init(const uint8_t id[FDH_COMM_ID_BYTES], int rank, int world);
  void comm_share(Context* owner);
  void comm_destroy();
  void gather_stripes(int dst_rank, void* dst_image);
  void gather_frames(int dst_rank, void* const* dst_images);
  int comm_world() const { return comm_ ? comm_world_ : 1; }

  // multi-GPU / measurement
  void set_stripe(int y0, int y1) { drain(); stripe_y0_ = y0; stripe_y1_ = y1; }
  // Culling (fdh_set_cull): draws whose pixel bounds miss the frame -- or, under fdh_set_stripe, the stripe's rows widened by the
  // reach of the scene's blur nodes -- are not recorded, and the scene front-end skips the content of a clipping node whose mask
  // lies outside.  0 off, 1 on (default; off while the call recorder runs, so that recorded streams stay the reference's), 2 on
  // even while recording (tests).  The pixels are the same either way.
  void set_cull(int mode) { cull_mode_ = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
  int64_t culled_draws() const { return culled_total_; }
  // The scene front-end decomposes large sibling groups on `n` pool threads beside the calling one (0: serial; default:
  // FDH_WALK_THREADS or a few, by the host's core count).  Same records either way (fdh_debug_record_digest).
  void set_walk_threads(int n) { walk_threads_ = n < 0 ? -1 : (n > 64 ? 64 : n); }
  int walk_threads() const;
  int64_t parallel_groups() const { return parallel_groups_; }
  void set_blur_route(int route) { blur_route_ = route < 0 ? -1 : (route ? 1 : 0); }
  void replay(int times);
  void replay_timed(int times, float* ms_out);
  void replay_async(int times);
  void profile(int times);
  void frame_stats(FdhFrameStats* out) { drain(); *out = stats_; out->ms_host_launch = launch_ms_.load(std::memory_order_relaxed); }
  // Submission is asynchronous: end_frame hands the recorded frame to the context's submit thread (the upload, the kernel
  // launches) and returns.  flush() returns once everything submitted so far has been ENQUEUED on the
  // stream (a consumer that orders its own work after the frame on the same stream calls it first); sync() also waits for the GPU.
  void flush() { drain(); }

 private:
  friend class Recorder;
  friend struct ParallelWalk;
  void alloc_atlas(int size);
  void upload_atlas_rect(int level, int x, int y, int w, int h, const uint8_t* rgba);
  void put_levels(int x, int y, int w, int h, const uint8_t* rgba);
  void find_empty_rect(int w, int h, int* ox, int* oy);
  void prepare(LaunchJob& J);  // calling thread: recorded frame -> run table + launch description
  void issue(LaunchJob& J);    // submit thread: upload + kernel launches
  void launch_frame(const LaunchJob& J, bool profile, uint32_t upload_seq = 0);  // upload_seq: the bin launch reports it to the host
  template <typename Buf> void reserve_quiet(Buf& buf, size_t n);
  void drain();        // wait until the submit thread is idle; rethrows what its last job threw
  void worker_main();
  hipEvent_t next_event();
  void ensure_surfaces();
  void need_device(const char* what) const;
  // pieces / phases (calling thread)
  void split_phase(int blur);               // a blurred snapshot is a barrier in painter's order
  void open_piece();                        // lane 0's records from here on form a new piece
  void close_piece();
  void add_piece(const Piece& p, const PhaseSum& s, const BBox& outer_union, int64_t fragments, int64_t culled);  // a pool thread's chunk, in order
  void add_sum(const PhaseSum& s, int depth_base);
  void close_phase();                       // the phase ends here: summary, bins reached, its share of the list stride
  uint32_t global_index(uint32_t lane0_index) const { return lane0_index + g0_delta_; }
  uint32_t global_count() const;            // records of the frame so far
  Lane& lane(int i) { return i < 0 ? *merge_lane_[(size_t)staging_i_] : *lanes_[(size_t)staging_i_][(size_t)i]; }  // (-1: consolidate_pieces' lane)
  Lane& ensure_lane(int i);
  void splice_cached(const RetainedRoot& C);
  void consolidate_pieces();
  void pick_routes();                       // begin_frame: the one-kernel blur routes (a frame alone) or the two-pass ones (frames in flight)
  void pool_slots(int slots);               // lanes 1 .. slots and their recorders, ready for a sibling group

  int device_ = 0;
  uint32_t flags_ = 0;
  int blur_route_ = -1;   // fdh_set_blur_route: -1 per-frame decision, 0 two passes, 1 fused
  bool latency_routes_ = true;  // this frame takes the one-kernel blur routes (pick_routes)
  std::shared_ptr<void> comm_;  // shared communicator object (fdh_comm.cpp), shared with the contexts that borrowed it: destroyed with its last holder
  int comm_rank_ = 0, comm_world_ = 1;
  hipStream_t own_stream_ = nullptr, stream_ = nullptr;
  volatile unsigned int* hdp_flush_reg_ = nullptr;  // the device's HDP_MEM_COHERENCY_FLUSH_CNTL register, mapped by the runtime (or null)
  hipEvent_t ev_[2] = {};
  std::vector<hipEvent_t> ev_pool_;
  size_t ev_used_ = 0;
  struct Span { int kind; hipEvent_t a, b; };  // kind: 0 bin, 1 composite main, 2 composite later, 3 blur h, 4 blur v, 5 / 6 largest node's h / v, 7 fused h + v
  std::vector<Span> spans_;

  // frame state
  int W_ = 0, H_ = 0;
  bool frame_begun_ = false;
  bool clear_ = true;
  uint32_t clear_rgba8_ = 0xFFFFFFFFu;
  float pixel_scale_ = 1.0f, ui_scale_ = 1.0f;
  float ctx_aa_ = 1.2f;   // (Recorder::aa_ of lane 0 is the live value; kept across frames)
  bool subpixel_enabled_ = false, subpixel_variants_ = false, text_lcd_filtering_ = false;
  int stripe_y0_ = 0, stripe_y1_ = 0;
  int cull_mode_ = 1;
  int binbox_shift_ = 0;           // bin boxes in 64 << shift px units (frames of more than 128 bins along an axis)
  int cull_y0_ = 0, cull_y1_ = 0;  // rows a draw must reach to be recorded (begin_frame: the frame, or the stripe + blur reach)
  int pending_reach_ = -1;         // render_frame / scene_render: summed vertical reach of the scene's blur nodes (-1: unknown)
  int64_t culled_total_ = 0;       // of the last recorded frame
  int walk_threads_ = -1;
  int64_t parallel_groups_ = 0;    // sibling groups of the last frame that were decomposed on the pool
  // Where the last frames cut their forked sibling groups into chunks, and what each chunk cost (records made + a share per item
  // visited): a group that comes again -- same first item, same length -- is cut where that cost says the work is, not into equal
  // item counts (a viewport's visible rows are a fifth of its cells).  Boundaries change nothing a frame records.
  struct GroupCuts { int first_item = -1, n = 0; std::vector<int> cut; std::vector<float> cost; uint64_t used = 0; };
  std::vector<GroupCuts> group_cuts_;

  // submit thread (device contexts, unless FDH_CREATE_SYNC_SUBMIT): one job in flight at most.  The flag both sides poll
  // sits on a cache line of its own, and so do the submission side's state and the recording side's: an idle submit thread
  // polling `pending_` next to the vector headers the caller bumps with every draw call tripled the cost of recording.
  std::thread worker_;
  alignas(128) std::atomic<bool> pending_{false};
  alignas(128) std::mutex mu_;
  std::condition_variable cv_job_, cv_done_;
  bool quit_ = false;
  std::exception_ptr worker_error_;
  alignas(128) LaunchJob job_;  // the frame being (or last) issued; what fdh_replay / fdh_profile launch again
  std::atomic<float> launch_ms_{0.0f};
  alignas(128) LaunchJob next_; // the frame being prepared (calling thread)
  const void* tables_dev_ = nullptr;  // the device block whose tail holds the blur weight tables described by ...
  std::vector<size_t> tables_layout_;
  std::vector<float> tables_sig_;

  // recorded frame (calling thread)
  alignas(128) bool rec_diff_upload_ = false;
  std::vector<Piece> pieces_;      // the frame in painter's order
  bool piece_open_ = false;        // the last piece is lane 0's and still growing
  uint32_t n_total_ = 0, n_ext_total_ = 0;  // records / extensions in closed pieces
  uint32_t g0_delta_ = 0;          // global index of a record of lane 0's open piece = its lane index + this
  std::vector<Phase> phases_;
  std::vector<BlurJob> blurs_;
  BBox phase_u_{0, 0, 0, 0};       // union of the current phase's record bounds
  int stride_max_ = 1;             // list stride so far: max over the closed phases
  int phase_extra_ = 0;            // the current phase: bound contributed by pool threads' lanes (sum of their own maxima)
  int deepest_clip_ = 0;
  int64_t frag_mode_[4] = {0, 0, 0, 0}, frag_ellip_ = 0, frag_other_ = 0;  // phase 0
  bool have_frame_ = false;

  // device state (submission side)
  alignas(128) uint32_t* fb_ = nullptr;
  uint32_t *backdrop_ = nullptr, *blur_tmp_ = nullptr;
  uint32_t* alt_ = nullptr;  // second frame surface: a fused full-frame blur renders out of place, phases alternate between fb_ and this
  uint32_t* dbg_snap_ = nullptr;
  void glyph_to_atlas(uint32_t* cur, uint32_t* nxt, int w, int h, int x, int y, uint32_t flags);
  DeviceBuf<float> glyph_lines_, glyph_acc_;
  DeviceBuf<uint32_t> glyph_a_, glyph_b_;  // put_glyph_image: the raster and its filtered / minified successors
  RetainedScene retained_;
  uint64_t atlas_epoch_ = 1;
  void rebase_side(FdhFig* nodes, int n, const FdhScene* side);
  void compact_side();
  bool host_only_ = false;  // FDH_CREATE_RECORD_ONLY
  bool rec_on_ = false, rec_first_ = true, rec_mark_first_ = true;
  size_t rec_mark_ = 0;
  std::string rec_;
  int surf_w_ = 0, surf_h_ = 0;
  // records, quad extensions, bin records and phase offsets of a frame live in ONE device block; the typed views point into it
  DeviceBuf<uint8_t> d_frame_;
  DeviceBuf<uint32_t> d_mask_spill_;  // clip-stack levels beyond kMaskDepth (Context::prepare sizes it)
  DeviceBuf<uint2> d_lists_;
  DeviceBuf<uint32_t> d_counts_;
  DeviceBuf<int> d_order_[2];  // phase 0's bins, longest list first: read by this frame's launch / written for the next
  int order_read_ = 0, order_nb_ = 0;
  bool order_valid_ = false;
  // kStaging sets of lanes in rotation, each released when the upload that reads it has run (the bin launch behind it says so
  // through *seq_host_, Context::issue; an event where a frame has no bin launch): the host records
  // frames N + 1 .. while frame N's upload has not run yet (one set forced a stream sync per frame)
  struct MxTables { int reach; std::vector<float> dense; std::vector<uint8_t> h, v; };  // k_blur_mx weight fragments of one filter
  std::vector<MxTables> mx_cache_;
  static constexpr int kStaging = 4;
  std::vector<std::unique_ptr<Lane>> lanes_[kStaging];
  std::unique_ptr<Lane> merge_lane_[kStaging];
  std::vector<std::unique_ptr<Recorder>> pool_recs_;  // the pool threads' recorders (slot s records into lane s + 1)
  uint64_t frame_no_ = 0;
  HostVec<uint8_t> misc_[kStaging];   // per slot (pinned): phase table, blur weight tables
  std::vector<uint8_t> misc_host_;    // ... as prepare builds them
  const uint8_t* misc_dev_[kStaging] = {};   // the pinned buffers' device views (hipHostGetDevicePointer costs a third of a microsecond)
  const uint8_t* misc_dev_host_[kStaging] = {};
  // retained scenes: host copy of what the device's frame block holds (prepare: upload only what differs)
  std::vector<uint8_t> shadow_;
  std::vector<size_t> shadow_layout_; // the offsets that block was laid out with
  const void* shadow_dev_ = nullptr;  // ... and where it lives
  int64_t uploaded_bytes_ = 0;        // by the last submit
  hipEvent_t staging_ev_[kStaging] = {};
  char staging_busy_[kStaging] = {};     // 0: free; 1: until staging_ev_ fires; 2: until *seq_host_ reaches staging_seq_ (Context::issue)
  uint32_t staging_seq_[kStaging] = {};
  volatile uint32_t* deep_host_ = nullptr;  // pinned, 8 words: bins per class with lists of at least deep_min draws, written by the compositor's sorting waves
  volatile uint32_t* seq_host_ = nullptr;  // pinned: the sequence number of the last frame whose bin launch has started (k_bin_draws)
  uint32_t upload_seq_ = 0;
  void wait_staging(int slot);             // calling thread: until the set's last upload has run
  int staging_i_ = 0;

  // atlas
  int atlas_size_ = 0, initial_atlas_size_ = 0, atlas_margin_ = 4, n_levels_ = 0;
  uint32_t* atlas_levels_[kMaxMips] = {};
  std::vector<uint16_t> heights_;
  std::unordered_map<int64_t, AtlasEntry> entries_;

  FdhFrameStats stats_ = {};
 public:
  // where the calling thread's time went in the last frame, ns (fdh_debug_host_times): 0 begin_frame, 1 of it: waiting for the
  // lane set's previous upload, 2 walk / calls (begin_frame's end .. end_frame), 3 end_frame before prepare, 4 prepare, 5 of it:
  // publishing, 6 waiting for the submit thread, 7 sibling groups on the pool (of 2), 8 of it: the pool's run, 9 of it: merging
  int64_t host_ns_[12] = {};
  struct HostTimer {
    int64_t& acc; std::chrono::steady_clock::time_point t0;
    explicit HostTimer(int64_t& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
  };
 private:
  std::chrono::steady_clock::time_point t_begin_frame_, t_walk_begin_;
  float host_record_ms_ = 0.0f;
};

void record_host_form(DrawRec& r);  // undo the device form of a committed record's colours (fdh_record.cpp)
void stripe_rows(int height, int world, int rank, int* y0, int* y1);
void comm_unique_id(uint8_t out[FDH_COMM_ID_BYTES]);
void blur_weight_fragments(float blur_radius, bool vertical, float* dense, uint16_t* frag_bits, int* reach, int* k_steps);
void saturated_core_of(const float rect[4], const float rx[4], const float ry[4], int mode, float factor, float spread,
                       const float shape[2], float aa, int out[4]);

}  // namespace fdh
