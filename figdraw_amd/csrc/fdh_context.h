// fdh_context.h -- host side of libfigdraw_hip.so: the BackendContext-shaped state machine that turns
// backend calls into draw records (what glcontext.nim does into vertex streams) and submits them.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
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

// An atlas entry and, when its level-0 texels were seen on the host (fdh_put_image), the bounds of what is IN it: for eight
// levels t = 0, 16, .. 112 the box (entry-relative texels, x1 / y1 exclusive) of texels whose alpha, and whose largest colour
// channel, exceeds t.  A draw whose coverage is exactly 0 wherever the sampled value is <= t (a glyph image: alpha 0; an MSDF
// image: distance below threshold - 0.5 / screen range) shrinks its pixel bounds to the image of that box: the strips outside
// would blend with alpha 0, which leaves every texel as it is (Context::shrink_to_ink).
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
};
struct BlurJob {
  float radius;
  int x0, y0, x1, y1;  // footprint: the mode-17 quad's pixel bounds
  int fuse_draw;       // record index of the consuming mode-17 quad when k_blur_v composites it, else -1
  BlurTaps taps;
};

// Retained scene (fdh_scene_*): the library-side half of the reference's RenderFragments (renderfragments.nim:426-544) --
// a deep copy of the node tree plus, per root, the draw records its decomposition produced.  A frame re-decomposes only the
// roots an update touched; every other root's records are spliced back from the cache.
struct RetainedRoot {
  std::vector<DrawRec> recs;
  std::vector<BBox> bboxes;
  std::vector<QuadExt> exts;     // of this root's records, DrawRec::ext relative to exts.front()
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

// What the launch side needs of one frame: filled by Context::prepare on the calling thread (which also fills the pinned
// staging buffer the upload kernel reads), consumed by Context::issue / launch_frame on the context's submit thread, and kept
// for fdh_replay / fdh_profile.
struct LaunchJob {
  struct View { DrawRec* recs = nullptr; QuadExt* exts = nullptr; BinRec* binrecs = nullptr; int* phase_first = nullptr; uint32_t* binbox = nullptr; uint32_t* chunkbox = nullptr; };
  int W = 0, H = 0;
  int rec_y0 = 0, rec_y1 = 0;  // rows the records were culled to (the frame, or a stripe + blur reach): a replay must stay inside
  bool clear = true;
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
  // the upload: `upload_bytes` from the head of the staging buffer, or (a retained scene's edit) a few runs of it
  const void* s_dev = nullptr;
  void* d_dst = nullptr;
  size_t upload_bytes = 0;
  bool patched = false;
  UploadRuns runs{};
  int staging_slot = -1;
};

struct RectMaskEntry { int kind; };  // 1 = fast analytic, 2 = real mask (glcontext.nim:36-44)

class Context {
 public:
  Context(int atlas_size, float pixel_scale, int device, uint32_t flags);
  ~Context();

  // BackendContext surface
  void begin_frame(int w, int h, bool clear, const float rgba[4]);
  void end_frame();
  void save_transform();
  void restore_transform();
  void translate(float x, float y);
  void rotate(float a);
  void scale(float sx, float sy);
  void apply_transform(const float m[16]);
  bool transform_mirrors_y() const;
  void set_aa(float aa);
  float aa() const { return aa_; }
  float pixel_scale() const { return pixel_scale_; }
  void draw_rounded_rect_sdf(const float rect[4], const FdhColor colors[4], const float rx[4], const float ry[4], int mode,
                             float factor, float spread, const float shape[2], int fill_mode, FdhColor mid, FdhColor stop,
                             float mid_pos);
  void draw_rounded_rect_fill(const float rect[4], const FdhFill& fill, const float rx[4], const float ry[4], int mode,
                              float factor, float spread, const float shape[2]);
  void draw_image(int64_t key, const float pos[2], const FdhColor colors[4], const float size[2], bool flip_y);
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
  void set_subpixel_enabled(bool e) { subpixel_enabled_ = e; }
  bool subpixel_enabled() const { return subpixel_enabled_; }
  void set_subpixel_variants(bool e) { subpixel_variants_ = e; }
  bool subpixel_variants() const { return subpixel_variants_; }
  void set_subpixel_shift(float s);

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
  uint64_t record_digest();  // FNV-1a over the last frame's draw records (diagnostic: works on record-only contexts)

  // multi-GPU: the gather over RCCL (fdh_comm.cpp)
  void comm_init(const uint8_t id[FDH_COMM_ID_BYTES], int rank, int world);
  void comm_share(Context* owner);
  void comm_destroy();
  void gather_stripes(int dst_rank, void* dst_image);
  void gather_frames(int dst_rank, void* const* dst_images);

  // multi-GPU / measurement
  void set_stripe(int y0, int y1) { drain(); stripe_y0_ = y0; stripe_y1_ = y1; }
  // Culling (fdh_set_cull): draws whose pixel bounds miss the frame -- or, under fdh_set_stripe, the stripe's rows widened by the
  // reach of the scene's blur nodes -- are not recorded, and the scene front-end skips the content of a clipping node whose mask
  // lies outside.  0 off, 1 on (default; off while the call recorder runs, so that recorded streams stay the reference's), 2 on
  // even while recording (tests).  The pixels are the same either way.
  void set_cull(int mode) { cull_mode_ = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
  bool culling() const { return cull_mode_ == 2 || (cull_mode_ == 1 && !rec_on_); }
  // would a quad over `rect` (pre-transform units), grown by `pad` pixels on every side, reach a pixel the frame will produce?
  bool rect_visible(const float rect[4], float pad) const;
  int64_t culled_draws() const { return culled_draws_; }
  void set_blur_route(int route) { blur_route_ = route < 0 ? -1 : (route ? 1 : 0); }
  void replay(int times);
  void replay_timed(int times, float* ms_out);
  void replay_async(int times);
  void profile(int times);
  void frame_stats(FdhFrameStats* out) { drain(); *out = stats_; out->ms_host_launch = launch_ms_.load(std::memory_order_relaxed); }
  // Submission is asynchronous: end_frame hands the recorded frame to the context's submit thread (upload preparation, the
  // upload, the kernel launches) and returns.  flush() returns once everything submitted so far has been ENQUEUED on the
  // stream (a consumer that orders its own work after the frame on the same stream calls it first); sync() also waits for the GPU.
  void flush() { drain(); }

 private:
  void push_rec(const DrawRec& r, const BBox& b);
  bool emit_quad(DrawRec& r, float x0, float y0, float x1, float y1, int64_t* fragments);  // false: culled, nothing was recorded
  bool emit_quad_pts(DrawRec& r, const float vx[4], const float vy[4], int64_t* fragments);
  bool bbox_visible(const BBox& b) const { return b.x1 > b.x0 && std::min<int>(b.y1, cull_y1_) > std::max<int>(b.y0, cull_y0_); }
  const AtlasEntry& rect_entry();
  void shrink_to_ink(const AtlasEntry& e, bool use_alpha, int level_t);
  void upload_atlas_rect(int level, int x, int y, int w, int h, const uint8_t* rgba);
  void put_levels(int x, int y, int w, int h, const uint8_t* rgba);
  void alloc_atlas(int size);
  void find_empty_rect(int w, int h, int* ox, int* oy);
  void prepare(LaunchJob& J);  // calling thread: recorded frame -> staging buffer + launch description
  void issue(LaunchJob& J);    // submit thread: upload + kernel launches
  void launch_frame(const LaunchJob& J, bool profile);
  template <typename Buf> void reserve_quiet(Buf& buf, size_t n);
  void drain();        // wait until the submit thread is idle; rethrows what its last job threw
  void worker_main();
  hipEvent_t next_event();
  void ensure_surfaces();
  void need_device(const char* what) const;

  int device_ = 0;
  uint32_t flags_ = 0;
  int blur_route_ = -1;   // fdh_set_blur_route: -1 per-frame decision, 0 two passes, 1 fused
  int submit_slot_ = 0;   // this context's entry in the process-wide table of last submissions (Context::prepare)
  std::shared_ptr<void> comm_;  // ncclComm_t (fdh_comm_init), shared with the contexts that borrowed it: destroyed with its last holder
  int comm_rank_ = 0, comm_world_ = 1;
  hipStream_t own_stream_ = nullptr, stream_ = nullptr;
  hipEvent_t ev_[2] = {};
  std::vector<hipEvent_t> ev_pool_;
  size_t ev_used_ = 0;
  struct Span { int kind; hipEvent_t a, b; };  // kind: 0 bin, 1 composite main, 2 composite later, 3 blur h, 4 blur v, 5 / 6 largest node's h / v, 7 fused h + v
  std::vector<Span> spans_;

  // frame state
  int W_ = 0, H_ = 0;
  bool frame_begun_ = false, mask_begun_ = false;
  bool clear_ = true;
  uint32_t clear_rgba8_ = 0xFFFFFFFFu;
  int mask_depth_ = 0;
  std::vector<RectMaskEntry> rect_masks_;
  std::vector<uint32_t> open_ops_;  // indices of currently open MASK_PUSH / RMASK_BEGIN records (re-emitted after a blur)
  Aff mat_;
  std::vector<Aff> mats_;
  float aa_ = 1.2f, pixel_scale_ = 1.0f, ui_scale_ = 1.0f;
  bool subpixel_enabled_ = false, subpixel_variants_ = false;
  float subpixel_shift_ = 0.0f;
  int stripe_y0_ = 0, stripe_y1_ = 0;
  int cull_mode_ = 1;
  int cull_y0_ = 0, cull_y1_ = 0;  // rows a draw must reach to be recorded (begin_frame: the frame, or the stripe + blur reach)
  int pending_reach_ = -1;         // render_frame / scene_render: summed vertical reach of the scene's blur nodes (-1: unknown)
  int64_t culled_draws_ = 0;       // of the frame being recorded / last recorded

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

  // recorded frame
  alignas(128) bool rec_diff_upload_ = false;
  std::vector<DrawRec> recs_;
  std::vector<BBox> bboxes_;
  std::vector<QuadExt> exts_;
  std::vector<Phase> phases_;
  std::vector<BlurJob> blurs_;
  int64_t fragments_ = 0;
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
  std::string rec_;  // FDH_DEBUG_SNAP=1 (diagnostic): the surface as phase 0 left it, copied in-stream
  int surf_w_ = 0, surf_h_ = 0;
  // records, quad extensions, bounding boxes and phase offsets of a frame live in ONE device block and arrive with ONE
  // copy (four small hipMemcpyAsync calls cost the host ~100 us per frame); the typed views point into it
  DeviceBuf<uint8_t> d_frame_;
  DeviceBuf<uint32_t> d_mask_spill_;  // clip-stack levels beyond kMaskDepth (Context::prepare sizes it)
  DeviceBuf<uint2> d_lists_;
  DeviceBuf<uint32_t> d_counts_;
  DeviceBuf<int> d_order_[2];  // phase 0's bins, longest list first: read by this frame's launch / written for the next
  int order_read_ = 0, order_nb_ = 0;
  bool order_valid_ = false;
  // three pinned staging buffers in rotation, each guarded by an event recorded after its copies: the host builds
  // frame N+1 and N+2 while frame N still runs (a single buffer forced a stream sync per frame)
  struct MxTables { int reach; std::vector<float> dense; std::vector<uint8_t> h, v; };  // k_blur_mx weight fragments of one filter
  std::vector<MxTables> mx_cache_;
  static constexpr int kStaging = 3;
  PinnedBuf<uint8_t> staging_[kStaging];
  std::vector<uint8_t> shadow_;       // host copy of what the device's frame block holds (submit: upload only what differs)
  std::vector<size_t> shadow_layout_; // the offsets that block was laid out with
  const void* shadow_dev_ = nullptr;  // ... and where it lives
  int64_t uploaded_bytes_ = 0;        // by the last submit
  hipEvent_t staging_ev_[kStaging] = {};
  bool staging_busy_[kStaging] = {};
  int staging_i_ = 0;
  std::vector<int> diff_scratch_;

  // atlas
  int atlas_size_ = 0, initial_atlas_size_ = 0, atlas_margin_ = 4, n_levels_ = 0;
  uint32_t* atlas_levels_[kMaxMips] = {};
  std::vector<uint16_t> heights_;
  std::unordered_map<int64_t, AtlasEntry> entries_;

  FdhFrameStats stats_ = {};
  std::chrono::steady_clock::time_point t_begin_frame_;
  float host_record_ms_ = 0.0f;
};

void stripe_rows(int height, int world, int rank, int* y0, int* y1);
void comm_unique_id(uint8_t out[FDH_COMM_ID_BYTES]);
void blur_weight_fragments(float blur_radius, bool vertical, float* dense, uint16_t* frag_bits, int* reach, int* k_steps);
void saturated_core_of(const float rect[4], const float rx[4], const float ry[4], int mode, float factor, float spread,
                       const float shape[2], float aa, int out[4]);

}  // namespace fdh
