// fdh_kernels.h -- kernel parameter blocks and launchers shared by the kernel units (k_*.hip) and the host context.
#pragma once
#include <hip/hip_runtime.h>

#include "fdh_types.h"

namespace fdh {

struct BinParams {
  const BinRec* binrec;  // per draw: clipped pixel bounds (ops that must reach every tile carry the frame), saturated core, entry flags
  const uint32_t* binbox; // per draw: x0 | y0 << 8 | (127 - x1) << 16 | (127 - y1) << 24 in (64 << binbox_shift)-px units, padded to 4 draws
  const uint32_t* chunkbox;  // per 256 draws: byte-wise min of their bin boxes = the union box, same format
  int n_draws, binbox_shift;
  uint2* lists;         // [phase][bin][stride] entries {draw index, 16-bit strip mask}
  uint32_t* counts;     // [phase][bin]
  const int* phase_first;  // [n_phases + 1]
  int n_phases, bins_x, bins_y, stride;
  const DrawRec* draws;    // the frame's records: the bin kernel pulls them into every XCD's L2 for the compositor (see k_bin_draws)
  const QuadExt* exts;     // the edge functions of rotated quads (BR_GENERAL draws: strips outside the quad are dropped from the entry)
  int refine;              // the frame holds BR_GENERAL / BR_CURVE draws: the build of k_bin_draws with their per-strip tests
  uint32_t* seq_out = nullptr;  // pinned host word that receives `seq` when the launch starts (or null): see k_bin_draws
  uint32_t seq = 0;
  // Bins per phase: phase p's workgroups are sub_first[p] .. sub_first[p + 1] - 1 and cover the sub_nx[p] bins wide block at
  // (sub_x0[p], sub_y0[p]) -- what that phase's compositor launch will read (a later phase touches the bins of ITS draws; a
  // workgroup per bin of the whole grid for every phase was two thirds of the bench frame's bin launch doing nothing).
  // sub_n = 0: more than kBinSubs phases, every phase gets the whole grid.
  static constexpr int kBinSubs = 8;
  int sub_n = 0;
  int sub_first[kBinSubs + 1] = {};
  int sub_x0[kBinSubs] = {}, sub_y0[kBinSubs] = {}, sub_nx[kBinSubs] = {};
};

struct CompositeParams {
  // (what a wave needs first comes first: with kernel-argument preloading the leading dwords arrive in SGPRs with the wave)
  const int* order;         // bins of a full-grid launch sorted by list length, longest first, or null (built by the
                            // previous frame's launch)
  int* order_next;          // if set, one extra wavefront of this launch sorts this frame's counts into it
  const uint32_t* counts;  // this phase's [bin]
  const uint2* lists;      // this phase's [bin][stride]
  int bin_x0, bin_y0, bin_nx, bin_ny;  // sub-grid of bins this launch covers
  int bins_x, stride;
  int row_lo, row_hi;       // stripe: only rows in [row_lo,row_hi) are produced
  int W, H, pitch;          // pitch in pixels
  int load_fb;              // 0: start from clear_rgba8
  uint32_t clear_rgba8;
  int n_wg;                 // total workgroups (for the XCD remap)
  uint32_t* fb;
  const uint32_t* backdrop;  // blurred snapshot sampled by mode 17
  int has_masks;            // the phase holds clip / rect-mask ops (disables per-strip occlusion culling)
  int has_slow;             // the phase holds draws that need the one-pixel-slot path (k_composite_tiles<3>)
  int has_slow_atlas = 0;   // ... atlas quads among them: k_composite_tiles<19>, the same build at three waves per SIMD (168 registers)
  int has_rot;              // ... or rotated SDF quads the 4-wide path takes (k_composite_tiles<8>, or <3> with the others)
  int has_atlas;            // ... or axis-aligned atlas quads at >= 1:1 (k_composite_tiles<2>)
  uint32_t* mask_spill;     // clip-stack levels beyond kMaskDepth: [level - kMaskDepth][strip][lane] (null when no frame nests that deep)
  size_t spill_stride;      // dwords per level = strips of the frame x 64
  AtlasView atlas;
  // quarter strips (round 6, k_composite_tiles): the first deep_k8 positions of `order` (a multiple of 8) are shaded by four waves per
  // strip; deep_min / deep_out: the sorting waves count the bins with at least deep_min draws per class into deep_out[0..7]
  // (pinned host memory, or null)
  // a direct launch (round 6): the phase has at most 64 draws and no list -- the compositor's waves make their bin's entries themselves
  // from the bin records of draws [direct_first, direct_first + direct_n)
  int direct = 0, direct_first = 0, direct_n = 0;
  const BinRec* binrec = nullptr;
  int deep_k8 = 0;
  int deep_strip_min = 0;   // ... those of their strips that have at least this many draws to SHADE (strip_shade_count); the others keep their one wave
  int deep_min = 0;
  uint32_t* deep_out = nullptr;
};

struct BlurParams {
  const uint32_t* src;
  uint32_t* dst;
  int W, H, pitch;
  int x0, y0, x1, y1;  // output region
  // V pass only: when fuse_draw >= 0 the mode-17 quad that consumes this blur is composited straight into
  // `dst` (the live surface) instead of writing the blurred snapshot out and reading it back
  int fuse_draw;
  // Toeplitz weight fragments of this pass for k_blur_mx (Context::submit builds them: [k-step][hi, lo][lane] x 8 halves),
  // or null: the packed-FMA passes then take the launch
  const uint4* mx_w;
  BlurTaps taps;
  // pixels of the NODE's whole footprint (0: of this launch's region): which passes take the launch -- matrix pipe or VALU -- goes by
  // it, so that a row stripe of a large node runs the kernels the whole frame runs (their sums differ in the last bits)
  long long node_pixels = 0;
};

// profile mode: the next launch_* call stamps these events with its kernel's own start / end (nullptr: plain launches)
void set_launch_events(hipEvent_t start, hipEvent_t stop);
bool launch_events_used();
void launch_bin(hipStream_t s, const BinParams& P);
void launch_composite(hipStream_t s, const DrawRec* draws, const QuadExt* exts, CompositeParams P);
void launch_blur_h(hipStream_t s, const BlurParams& P);
void launch_blur_v(hipStream_t s, const BlurParams& P, const DrawRec* draws, const QuadExt* exts);
// a small region (blur_one_kernel_ok): both passes in ONE kernel, P.src -> the blurred snapshot in P.dst (out of place: the backdrop
// surface), rows [P.y0, P.y1) of columns [P.x0, P.x1); the phase's compositor launch samples it for the mode-17 quad
bool blur_one_kernel_ok(int w, int h, int reach);
void launch_blur_small(hipStream_t s, const BlurParams& P);
// both passes of a full-frame node in one kernel, out of place (P.src -> P.dst, the fused mode-17 composite blended over P.src);
// P.mx_w = the horizontal pass's weight fragments, w_v = the vertical pass's.  false: no instantiation for this filter width
bool blur_fused_supported(int reach, int W, int pitch);
bool launch_blur_fused(hipStream_t s, const BlurParams& P, const uint4* w_v, const DrawRec* draws, const QuadExt* exts);
void launch_fill(hipStream_t s, uint32_t* p, uint32_t v, size_t n);
// glyph images: LCD filter (pixie_raster.nim:12-43), one minifyBy2 step, copy into an atlas level
// glyph outline (flattened to lines x0, y0, x1, y1) -> premultiplied white coverage; scratch: h x (w + 2) floats
void launch_rasterize_lines(hipStream_t s, const float4* lines, int n, int w, int h, float* scratch, uint32_t* out);
void launch_lcd_filter(hipStream_t s, const uint32_t* src, uint32_t* dst, int w, int h);
void launch_minify2(hipStream_t s, const uint32_t* src, uint32_t* dst, int sw, int sh);  // dst: ((sw + 1) / 2) x ((sh + 1) / 2)
void launch_atlas_blit(hipStream_t s, uint32_t* level, int LS, int x, int y, const uint32_t* src, int w, int h);
// The frame upload (k_upload_frame): a table of runs, each `bytes` of pinned host memory (its device view) going to byte offset
// dst_off of the frame block.  kind 0: 16-byte units; 1: BinRecs -- copied in 8-byte units, and the lane that carries a record's
// pixel bounds also writes the draw's 4-byte bin box; 2: DrawRecs -- 16-byte units, and the `ext` of every F_GENERAL record gets
// ext_add (extension indices are relative to the recording thread's array).  The table travels in the kernel arguments (4 KB at most).
constexpr int kMaxUploadRuns = 96;
struct UploadRun { const void* src; uint32_t dst_off, bytes, ext_add, kind; };
struct UploadTable {
  uint32_t n_runs, copy_units, n_draws, binbox_shift;
  uint32_t bins_off, box_off, _pad0, _pad1;     // byte offsets of the BinRec / bin box arrays in the block
  uint32_t unit_first[kMaxUploadRuns];           // the first 1-KB unit of run r (ascending)
  UploadRun run[kMaxUploadRuns];
};
static_assert(sizeof(UploadTable) <= 4096, "the table must fit the kernel-argument segment");
void launch_upload_frame(hipStream_t s, void* dst, const UploadTable& T);

}  // namespace fdh
