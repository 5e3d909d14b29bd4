// fdh_record.cpp -- the BackendContext state machine: draw calls -> draw records of one lane.
//
// What glcontext.nim does with ten vertex streams and a batch flush, this does with one 128-byte record
// per call.  The record carries exactly what the reference's vertex attributes carry (ceil-snapped quad,
// un-snapped half extents, packed radii, mode word, factors, colours) so the kernels can restate the
// fragment shaders.  Clip masks become push/pop records evaluated analytically per pixel, backdrop
// blurs split the list into phases (a blur is a global barrier in painter's order, glcontext.nim:1788-1841).
// Everything a record needs on the device -- its bin record, its share of the list-stride count, its phase's summary -- is
// produced here, when the record's bounds are final (commit_bins), not in a second pass at submit.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "fdh_context.h"
#include "fdh_host.h"

namespace fdh {

// The record just emitted (an upright atlas quad sampling level 0 of `e`) covers nothing outside the image of the ink box at
// level `level_t` (values <= level_t give coverage exactly 0 for this draw): its pixel bounds shrink to that image.  A bilinear
// sample at texel coordinate t reads texels floor(t) and floor(t) + 1, the sub-pixel shift moves t by less than one texel: the box
// is widened by three texels and the pixel range by one pixel on every side, far beyond any rounding of the linear map.
void Recorder::shrink_to_ink(const AtlasEntry& e, bool use_alpha, int level_t) {
  static const bool enabled = [] { const char* v = std::getenv("FDH_INK_BOUNDS"); return !v || std::atoi(v) != 0; }();
  if (!enabled || !e.has_ink || level_t < 0 || lane_->recs.empty()) return;
  DrawRec& r = lane_->recs.back();
  BBox& b = lane_->bins.back().box;
  if ((r.op_mode & F_GENERAL) || b.x1 <= b.x0 || b.y1 <= b.y0) return;
  const InkBox ib = (use_alpha ? e.ink_a : e.ink_rgb)[std::min(level_t / 16, kInkLevels - 1)];
  if (ib.x1 <= ib.x0 || ib.y1 <= ib.y0) { b = BBox{0, 0, 0, 0}; r.bx0 = r.by0 = r.bx1 = r.by1 = 0; return; }  // nothing in the image reaches the level
  const double S = (double)cx_->atlas_size_;
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
  // (writes the flag only where it was on -- i.e. on the calling thread while the call recorder runs, when nothing is handed to the walk
  // pool: written unconditionally, false over false, it was a data race between the pool's threads; ThreadSanitizer, profiles/r06_host_asan.txt)
  bool& on; const bool was;
  explicit RecPause(bool& o) : on(o), was(o) { if (was) on = false; }
  ~RecPause() { if (was) on = true; }
};
// (the recorder only ever runs on the calling thread's recorder: while it is on nothing is handed to the walk pool)
#define FDH_REC(name) Rec rec_scope_(cx_->rec_, cx_->rec_first_, cx_->rec_on_ && is_main_, name, &cx_->rec_mark_, &cx_->rec_mark_first_); rec_scope_
// cull mode 2 (culling while the recorder runs): the draw call just recorded left no record -- it leaves no entry either
#define FDH_CULLED() do { culled_draws_++; if (cx_->rec_on_ && is_main_) { cx_->rec_.resize(cx_->rec_mark_); cx_->rec_first_ = cx_->rec_mark_first_; } } while (0)
void Context::record_begin() { rec_on_ = true; rec_first_ = true; rec_ = "["; }
const char* Context::record_json() {
  if (!rec_on_) return "[]";
  rec_ += "\n]";
  rec_on_ = false;
  return rec_.c_str();
}
void Recorder::set_aa(float aa) { { FDH_REC("set_aa_factor").f(aa); } aa_ = aa; }
void Recorder::set_subpixel_shift(float s) { { FDH_REC("set_text_subpixel_shift").f(s); } subpixel_shift_ = s; }  // setTextSubpixelShift figbackend.nim:663-686
bool Recorder::subpixel_enabled() const { return cx_->subpixel_enabled_; }
bool Recorder::subpixel_variants() const { return cx_->subpixel_variants_; }
bool Recorder::culling() const { return cx_->cull_mode_ == 2 || (cx_->cull_mode_ == 1 && !cx_->rec_on_); }
bool Recorder::bbox_visible(const BBox& b) const { return b.x1 > b.x0 && std::min<int>(b.y1, cx_->cull_y1_) > std::max<int>(b.y0, cx_->cull_y0_); }

void Recorder::save_transform() { { FDH_REC("save_transform"); } mats_.push_back(mat_); }
void Recorder::restore_transform() {
  { FDH_REC("restore_transform"); }
  if (mats_.empty()) throw Error(FDH_ERR_INVALID, "restoreTransform: empty transform stack");
  mat_ = mats_.back();
  mats_.pop_back();
}
void Recorder::translate(float x, float y) { { FDH_REC("translate").f(x).f(y); } Aff t; t.tx = x; t.ty = y; mat_ = aff_mul(mat_, t); }
void Recorder::rotate(float a) {
  { FDH_REC("rotate").f(a); }
  Aff r;  // vmath rotateZ: column 0 = (cos, -sin), column 1 = (sin, cos); pinned by tests/expected/render_line_rect.png
  r.a = std::cos(a); r.b = -std::sin(a); r.c = -r.b; r.d = r.a;
  mat_ = aff_mul(mat_, r);
}
void Recorder::scale(float sx, float sy) { { FDH_REC("scale").f(sx).f(sy); } Aff s; s.a = sx; s.d = sy; mat_ = aff_mul(mat_, s); }
void Recorder::apply_transform(const float m[16]) {  // column-major Mat4; `mat * vec3(x, y, 0)` uses its 2D affine part
  { FDH_REC("apply_transform").fv(m, 16); }
  Aff n;
  n.a = m[0]; n.b = m[1]; n.c = m[4]; n.d = m[5]; n.tx = m[12]; n.ty = m[13];
  mat_ = aff_mul(mat_, n);
}
bool Recorder::transform_mirrors_y() const { return mat_.a * mat_.d - mat_.b * mat_.c < 0.0f; }

// ------------------------------------------------------------------ records
DrawRec& Recorder::next_rec() {
  DrawRec* r = lane_->recs.slot();
  std::memset(static_cast<void*>(r), 0, sizeof *r);
  return *r;
}
// counts the slot next_rec() handed out; its BinRec starts as bare bounds (flags follow when the bounds are final: commit_bins)
void Recorder::push_rec(BBox b) {  // (by value: callers pass bounds that live in the very array slot() may move)
  BinRec* br = lane_->bins.slot();
  *br = BinRec{b, 0, 0, 0, 0, 0u, 0u};
  *lane_->boxes.slot() = 0x7f7f7f7fu;
  lane_->recs.n++;
  lane_->bins.n++;
  lane_->boxes.n++;
}

// What k_bin_draws scans: 7-bit inclusive bounds in bin units, upper bounds complemented (x0 | y0 << 8 | (127 - x1) << 16 |
// (127 - y1) << 24; x0 = y0 = 127, x1 = y1 = 0 never hits).  The device derives the same word from the BinRec (k_upload_frame).
uint32_t bin_box_of(const BBox& b, int shift) {
  if (bbox_empty(b)) return 0x7f7f7f7fu;
  return (uint32_t)(b.x0 >> shift) | ((uint32_t)(b.y0 >> shift) << 8) | ((127u - (uint32_t)((b.x1 - 1) >> shift)) << 16) | ((127u - (uint32_t)((b.y1 - 1) >> shift)) << 24);
}

// The bin-independent part of a draw's list entries (BinRec::flags): which straight-line path its edge strips take, what its
// core strips are.  Decided here so that the compositor branches on the list entry (already in an SGPR) before it has fetched
// the record.  `r` still holds its four vertex colours.
static uint32_t binrec_flags(const DrawRec& r) {
  const uint32_t om = r.op_mode, op = (om >> 12) & 15u, mode = om & 255u, fill_mode = (om >> 9) & 7u;
  const bool atlas_mode = mode == 0u || (mode >= 13u && mode <= 16u);
  const bool sdf = !(om & F_GENERAL) && !atlas_mode && mode < 18u && (op == OP_DRAW || op == OP_MASK_PUSH);
  const uint32_t ell = (om & F_ELLIP) ? 4u : 0u;
  uint32_t flags = 0;
  const bool rot = (om & F_GENERAL) && (om & F_EDGE32) && !atlas_mode && mode < 18u && op == OP_DRAW;  // (its core: QuadExt::core)
  if (mode >= 18u && mode <= 20u && op == OP_DRAW && r.inv_w != 0.0f) return BR_CURVE;  // (an upright quad: ox .. inv_h are set, emit_corners)
  if (!sdf && !rot) return 0u;
  flags |= BR_HAS_CORE | (rot ? BR_GENERAL : 0u);
  if (op == OP_DRAW && (mode == 9u || mode == 11u || mode == 12u)) {
    flags |= BR_CORE_REMOVED;  // the stroke's interior, or so deep inside an inner shadow that no 8-bit channel moves
    if (!rot && (om & F_SOLID) && fill_mode == 0u && mode != 11u) flags |= ((mode == 9u ? 3u : 4u) + ell) << LE_PATH_SHIFT;
  } else {
    if (op == OP_DRAW && (om & F_SOLID) && fill_mode == 0u && mode != 17u) {
      flags |= LE_PLAIN;
      const uint32_t code = mode == 3u ? 1u : mode == 7u ? 2u : 0u;
      if (code && !rot) flags |= (code + ell) << LE_PATH_SHIFT;
    }
    if (op == OP_DRAW && mode == 3u) {
      uint32_t a = r.col[0] & r.col[1] & r.col[2] & r.col[3];
      if (fill_mode != 0u) a &= r.mid & r.stop;
      if ((a >> 24) == 255u) flags |= LE_OPAQUE;
    }
  }
  return flags;
}
// The device's copy of a one-colour upright SDF draw carries the colour once more as three floats, c / 255, in the slots of
// the three redundant vertex colours: the compositor's uniform-blend and packed edge paths (the only readers: LE_PLAIN and
// the path codes go to exactly these records) take them as they are instead of converting and scaling three
// bytes per strip.  The same IEEE product the kernels formed (one multiply by the float 1 / 255): bit-identical frames.
static inline bool colour_as_floats(uint32_t om) {
  const uint32_t mode = om & 255u;
  const bool atlas_mode = mode == 0u || (mode >= 13u && mode <= 16u);
  return !(((om & F_GENERAL) && !(om & F_EDGE32)) || atlas_mode || mode >= 18u || ((om >> 12) & 15u) != OP_DRAW || !(om & F_SOLID));  // (F_EDGE32: the 4-wide rotated path)
}
void record_host_form(DrawRec& r) {  // what the record held before commit_bins (fdh_debug_record_digest hashes that form)
  if (colour_as_floats(r.op_mode)) r.col[1] = r.col[2] = r.col[3] = r.col[0];
}

// LE_SHARE: record idx is drawn over the same quad with the same shape as record idx - 1 -- a node's fill, then its stroke, then
// its inner shadows (figrender.nim:806-873, 716-744): both one-colour fill / stroke / inner shadow (path codes 1, 3, 4 and their
// elliptical twins) over the same quad, radii and AA factor, the same shape half extents, in the same phase and piece.
void Recorder::link_share(uint32_t idx) {
  if ((int)idx <= phase_floor_) return;
  const DrawRec& b = lane_->recs[idx];
  BinRec& pa = lane_->bins[idx - 1];
  const uint32_t code = (pa.flags >> LE_PATH_SHIFT) & 15u;
  if (code == 0u) return;
  const DrawRec& r = lane_->recs[idx - 1];
  const uint32_t om = r.op_mode, mode = om & 255u;
  if (mode == 7u) return;
  const uint32_t omb = b.op_mode, modeb = omb & 255u;
  const bool simple_b = ((omb >> 12) & 15u) == OP_DRAW && !(omb & F_GENERAL) && (omb & F_SOLID) && ((omb >> 9) & 7u) == 0u &&
                        (modeb == 3u || modeb == 9u || modeb == 12u) && ((omb ^ om) & F_ELLIP) == 0u;
  if (simple_b && std::memcmp(&r.ox, &b.ox, 6 * sizeof(float)) == 0 && std::memcmp(r.r, b.r, sizeof r.r) == 0 &&
      std::memcmp(&r.bx0, &b.bx0, 4 * sizeof(int16_t)) == 0 && r.aa == b.aa) {
    const float sax = mode == 9u ? r.p0 : r.p2, say = mode == 9u ? r.p1 : r.p3, sbx = modeb == 9u ? b.p0 : b.p2, sby = modeb == 9u ? b.p1 : b.p3;
    if (sax == sbx && say == sby) pa.flags |= LE_SHARE;
  }
}

// The record's bounds are final (a draw: at the end of its call; a clip push: at its pop, when its content's union is known):
// its BinRec, its share of the list-stride count and of its phase's summary, and the device form of its colours.
void Recorder::commit_bins(uint32_t idx) {
  DrawRec& r = lane_->recs[idx];
  BinRec& br = lane_->bins[idx];
  const uint32_t om = r.op_mode, op = (om >> 12) & 15u, mode = om & 255u;
  br.ix0 = r.ix0; br.iy0 = r.iy0; br.ix1 = r.ix1; br.iy1 = r.iy1;
  br.flags = binrec_flags(r);
  const BBox b = br.box;
  if (op == OP_DRAW && (br.flags & BR_HAS_CORE) && !(br.flags & BR_GENERAL) && b.x0 == r.bx0 && b.y0 == r.by0 && b.x1 == r.bx1 && b.y1 == r.by1) br.flags |= BR_BOX_EXACT;
  lane_->boxes[idx] = bin_box_of(b, 6 + cx_->binbox_shift_);
  lane_->count_add(b);
  bbox_union(sum_.u, b);
  // which compositor build the phase needs (mirrors the path selection of k_composite_tiles)
  const bool atlas_mode = mode == 0u || (mode >= 13u && mode <= 16u);
  if (op != OP_DRAW) sum_.has_masks = true;
  // 4-wide atlas path for axis-aligned atlas quads sampled from level 0
  const bool atlas4 = atlas_mode && !(om & F_GENERAL) && op == OP_DRAW && !(mode == 0u && r.aux2 > 0.0f && cx_->n_levels_ >= 2);
  if (atlas4) sum_.has_atlas = true;
  // (a rect mask under a rotated transform -- matY.x != 0 -- is set up one pixel slot at a time; an upright one runs 4-wide)
  else if ((op == OP_DRAW || op == OP_MASK_PUSH) && (om & F_GENERAL) && (om & F_EDGE32) && !atlas_mode && mode < 18u) sum_.has_rot = true;
  else if ((op == OP_RMASK_BEGIN && r.inv_h != 0.0f) || ((op == OP_DRAW || op == OP_MASK_PUSH) && ((om & F_GENERAL) || atlas_mode || mode >= 18u))) {
    sum_.has_slow = true;
    // (a quad textured with one white texel -- drawRect / drawFilledQuad: uvAt == uvTo -- samples nothing worth registers)
    if (atlas_mode && !(mode == 0u && r.r[0] == r.r[2] && r.r[1] == r.r[3])) sum_.has_slow_atlas = true;
  }
  if (op == OP_DRAW && !bbox_empty(b)) {  // covered fragments by mode (counted for phase 0 by the context)
    const int64_t area = (int64_t)(b.x1 - b.x0) * (b.y1 - b.y0);
    if (mode == 3u) sum_.frag_mode[0] += area; else if (mode == 7u) sum_.frag_mode[1] += area; else if (mode == 9u) sum_.frag_mode[2] += area;
    else if (mode == 12u) sum_.frag_mode[3] += area; else sum_.frag_other += area;
    if (om & F_ELLIP) sum_.frag_ellip += area;
  }
  if (op == OP_DRAW) {
    link_share(idx);
    if (colour_as_floats(om)) {
      const uint32_t c = r.col[0];
      const float inv255 = 1.0f / 255.0f;
      const float u[3] = {(float)(c & 255u) * inv255, (float)((c >> 8) & 255u) * inv255, (float)((c >> 16) & 255u) * inv255};
      std::memcpy(&r.col[1], u, sizeof u);
    }
  }
}

// ------------------------------------------------------------------ publishing (Lane)
// The mirrors grow with the lane (never shrink, never hold more than the lane); a mirror that had to move is filled again from the
// lane up to what had been published -- from ordinary memory: pinned memory is not read.
static void mirror_fit(Lane& L) {
  auto fit = [](auto& up, const auto& src, size_t published, const uint8_t*& dev) {
    if (up.cap >= src.cap) return;
    up.n = 0;  // (nothing to carry over: reserve() would READ the old pinned block)
    up.reserve(src.cap);
    if (published) std::memcpy(static_cast<void*>(up.p), static_cast<const void*>(src.p), published * sizeof(*src.p));
    dev = up.device_view();
  };
  fit(L.up_recs, L.recs, L.pub_recs, L.d_recs);
  fit(L.up_bins, L.bins, L.pub_recs, L.d_bins);
  fit(L.up_exts, L.exts, L.pub_exts, L.d_exts);
}
void Lane::publish(uint32_t first, uint32_t n, uint32_t ext_first, uint32_t n_ext) {
  if (!device) return;
  mirror_fit(*this);
  if (n) {
    std::memcpy(static_cast<void*>(up_recs.p + first), static_cast<const void*>(recs.p + first), (size_t)n * sizeof(DrawRec));
    std::memcpy(static_cast<void*>(up_bins.p + first), static_cast<const void*>(bins.p + first), (size_t)n * sizeof(BinRec));
    pub_recs = std::max(pub_recs, (size_t)first + n);
  }
  if (n_ext) {
    std::memcpy(static_cast<void*>(up_exts.p + ext_first), static_cast<const void*>(exts.p + ext_first), (size_t)n_ext * sizeof(QuadExt));
    pub_exts = std::max(pub_exts, (size_t)ext_first + n_ext);
  }
  if (up_recs.vram) store_fence();  // (write-combined stores into device memory: drained before this thread says the piece is there)
}
void Lane::publish_bytes(int array, size_t at, size_t len) {
  if (!device || !len) return;
  mirror_fit(*this);
  uint8_t* dst = array == 0 ? reinterpret_cast<uint8_t*>(up_recs.p) : array == 1 ? reinterpret_cast<uint8_t*>(up_bins.p) : reinterpret_cast<uint8_t*>(up_exts.p);
  const uint8_t* src = array == 0 ? reinterpret_cast<const uint8_t*>(recs.p) : array == 1 ? reinterpret_cast<const uint8_t*>(bins.p) : reinterpret_cast<const uint8_t*>(exts.p);
  std::memcpy(dst + at, src + at, len);
  if (up_recs.vram) store_fence();
}

// ------------------------------------------------------------------ list stride (Lane)
void Lane::count_begin(int bins_x, int bins_y) {
  if (dw != bins_x + 1 || dh != bins_y + 1) {
    dw = bins_x + 1; dh = bins_y + 1;
    diff.assign((size_t)dw * dh, 0);
  } else if (touched) {
    for (int y = ty0; y <= ty1; y++) std::fill(diff.begin() + (size_t)y * dw + tx0, diff.begin() + (size_t)y * dw + tx1 + 1, 0);
  }
  touched = false;
}
void Lane::count_add(const BBox& b) {
  if (bbox_empty(b)) return;
  constexpr int kBinShift = 6;
  static_assert((1 << kBinShift) == kBin, "bins are 64 px");
  const int bx0 = b.x0 >> kBinShift, by0 = b.y0 >> kBinShift, bx1 = ((b.x1 - 1) >> kBinShift) + 1, by1 = ((b.y1 - 1) >> kBinShift) + 1;  // [bx0,bx1) x [by0,by1)
  if (bx1 >= dw || by1 >= dh) return;  // (bounds are clipped to the frame: cannot happen)
  int* d = diff.data();
  d[(size_t)by0 * dw + bx0]++; d[(size_t)by0 * dw + bx1]--; d[(size_t)by1 * dw + bx0]--; d[(size_t)by1 * dw + bx1]++;
  if (!touched) { tx0 = bx0; ty0 = by0; tx1 = bx1; ty1 = by1; touched = true; }
  else { tx0 = std::min(tx0, bx0); ty0 = std::min(ty0, by0); tx1 = std::max(tx1, bx1); ty1 = std::max(ty1, by1); }
}
// The largest number of records any bin received since the last close, counted exactly from the 2-D difference array
// (O(records + bins reached)): sizing the lists for "every draw of the phase in every bin" cost 163 MB for the 10 001-draw glyph
// frame; the exact bound is 2040 bins x a few dozen entries.
int Lane::count_close() {
  if (!touched) return 0;
  int mx = 0;
  for (int y = ty0; y < ty1; y++) {
    int run = 0;
    int* row = diff.data() + (size_t)y * dw;
    const int* above = y > ty0 ? row - dw : nullptr;
    for (int x = tx0; x < tx1; x++) {
      run += row[x];
      const int cell = run + (above ? above[x] : 0);  // column prefix over the row prefixes
      row[x] = cell;
      mx = std::max(mx, cell);
    }
  }
  for (int y = ty0; y <= ty1; y++) std::fill(diff.begin() + (size_t)y * dw + tx0, diff.begin() + (size_t)y * dw + tx1 + 1, 0);
  touched = false;
  return mx;
}

// The part of an axis-aligned SDF quad where the coverage term is saturated (DrawRec::ix0..iy1).  Works in the
// shader's local frame (atlas.frag:252-262: p = (uv - 0.5) * 2 * quadHalfExtents, y up) and maps back to pixels.
// {dist <= -e} of sdRoundedBox(b, r) is the rounded box (b - e, max(r - e, 0)); an axis-aligned rectangle whose
// corners are pulled in by (1 - 1/sqrt 2) r per corner lies inside it.  Elliptical corners use an approximate
// distance (atlas.frag:71-79), so there the core stays out of the corner cells, where the distance is the plain
// box distance max(|p| - b).  One pixel of slack on every side absorbs all float rounding.
// (first half: the rectangle {xl..xr} x {yb..yt} of the local frame, y up; false = no core)
static bool local_core(const DrawRec& r, double& xl, double& xr, double& yb, double& yt) {
  const uint32_t mode = r.op_mode & 255u, op = (r.op_mode >> 12) & 15u;
  const uint32_t fill_mode = (r.op_mode >> 9) & 7u;
  if (!(op == OP_DRAW || op == OP_MASK_PUSH) || !(r.aa > 0.0f)) return false;
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
  } else return false;
  const bool inset = mode == FDH_SDF_INSET_SHADOW;
  const double qhx = r.p0, qhy = r.p1, bx = inset ? qhx : (double)r.p2, by = inset ? qhy : (double)r.p3;
  if (!(qhx > 0.0 && qhy > 0.0 && bx > 0.0 && by > 0.0)) return false;
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
  return xr > xl && yt > yb;
}
static void set_saturated_core(DrawRec& r, float w_px, float h_px) {
  r.ix0 = r.iy0 = r.ix1 = r.iy1 = 0;
  double xl, xr, yb, yt;
  if (!local_core(r, xl, xr, yb, yt)) return;
  const double qhx = r.p0, qhy = r.p1;
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
bool Recorder::emit_quad(DrawRec& r, float x0, float y0, float x1, float y1, bool count_fragments) {
  const float vx[4] = {x0, x1, x1, x0}, vy[4] = {y1, y1, y0, y0};  // BL, BR, TR, TL
  return emit_quad_pts(r, vx, vy, count_fragments);
}

// The pixel bounds emit_quad_pts gives a quad over `rect` (same arithmetic), grown by `pad`: does it reach a row / column the
// frame will produce?  The scene front-end asks before it opens a clip: content under a mask that lies outside is invisible.
bool Recorder::rect_visible(const float rect[4], float pad) const {
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
  const float W = (float)cx_->W_, H = (float)cx_->H_;
  BBox b;
  b.x0 = (int16_t)clampf(minx - pad, 0.0f, W); b.x1 = (int16_t)clampf(maxx + pad, 0.0f, W);
  b.y0 = (int16_t)clampf(miny - pad, 0.0f, H); b.y1 = (int16_t)clampf(maxy + pad, 0.0f, H);
  return bbox_visible(b);
}

// Four pre-transform vertices in the reference's vertex order 0..3 (triangles (3,0,1) and (2,3,1), glcontext.nim:418-429).
// `r` is the lane's next record slot (next_rec): counted here unless the draw is culled.
// the quad's vertices on the pixel grid -- ceil(ctx.mat * corner) per vertex (glcontext.nim:1498-1509) -- and its clipped bounds
void Recorder::quad_corners(const float vx[4], const float vy[4], QuadPx& q) const {
  for (int i = 0; i < 4; i++) {
    q.px[i] = std::ceil(mat_.a * vx[i] + mat_.c * vy[i] + mat_.tx);
    q.py[i] = std::ceil(mat_.b * vx[i] + mat_.d * vy[i] + mat_.ty);
  }
  float minx = q.px[0], maxx = q.px[0], miny = q.py[0], maxy = q.py[0];
  for (int i = 1; i < 4; i++) {
    minx = std::min(minx, q.px[i]); maxx = std::max(maxx, q.px[i]);
    miny = std::min(miny, q.py[i]); maxy = std::max(maxy, q.py[i]);
  }
  const float lim = 1.0e6f;  // keep the integer edge functions far from overflow
  q.b = BBox{0, 0, 0, 0};
  q.finite = minx > -lim && maxx < lim && miny > -lim && maxy < lim;
  if (!q.finite) return;
  const float W = (float)cx_->W_, H = (float)cx_->H_;
  q.b.x0 = (int16_t)clampf(minx, 0.0f, W); q.b.x1 = (int16_t)clampf(maxx, 0.0f, W);
  q.b.y0 = (int16_t)clampf(miny, 0.0f, H); q.b.y1 = (int16_t)clampf(maxy, 0.0f, H);
}
bool Recorder::emit_quad_pts(DrawRec& r, const float vx[4], const float vy[4], bool count_fragments) {
  QuadPx q;
  quad_corners(vx, vy, q);
  return emit_corners(r, q, count_fragments);
}
bool Recorder::emit_corners(DrawRec& r, const QuadPx& q, bool count_fragments) {
  const float* px = q.px;
  const float* py = q.py;
  BBox b = q.b;
  // A draw that reaches no pixel the frame will produce leaves no trace (it would never be binned).  Clip pushes stay: their
  // bounds grow to their content's, and a push that is not there would let that content through.
  const bool cullable = ((r.op_mode >> 12) & 15u) == OP_DRAW && culling();
  // The same for a draw with a NaN or an infinity among its shader parameters (half extents, factors, radii / uv rect, AA factor,
  // sub-pixel shift, LOD): the kernels are compiled with -fno-honor-nans (csrc/Makefile) on the guarantee that no such value reaches
  // them, and this is where it is kept -- the reference's shader would produce NaN alphas there, i.e. no defined pixel either.
  const float fields[] = {r.p0, r.p1, r.p2, r.p3, r.f0, r.f1, r.r[0], r.r[1], r.r[2], r.r[3], r.aa, r.aux, r.aux2};
  bool params_finite = true;
  for (const float v : fields) params_finite = params_finite && std::isfinite(v);
  if (!q.finite || !params_finite) {
    if (cullable) { FDH_CULLED(); return false; }
    r.bx0 = r.by0 = r.bx1 = r.by1 = 0; push_rec(b); return true;
  }
  if (cullable && !bbox_visible(b)) { FDH_CULLED(); return false; }
  r.bx0 = b.x0; r.by0 = b.y0; r.bx1 = b.x1; r.by1 = b.y1;
  const bool aligned = px[3] == px[0] && px[2] == px[1] && py[3] == py[2] && py[0] == py[1] && px[1] > px[0] && py[0] > py[3];
  // A bezier stroke (modes 18 - 20) always takes the two-triangle form, upright or not: its distance is a closed-form cubic that turns a
  // last-bit difference of its input into pixels, so the kernels form its uv exactly as the reference's rasteriser does (barycentrics
  // of the quad's two triangles, in double precision: tri_bary(exact) in k_composite_tiles), not as (x - ox) / w.  An upright one
  // keeps ox .. inv_h too: k_bin_draws maps strips through them (BR_CURVE).
  const bool bezier = (r.op_mode & 255u) >= 18u && (r.op_mode & 255u) <= 20u;
  if (aligned) {
    r.ox = px[3];
    r.oy = py[3];
    r.inv_w = 1.0f / (px[1] - px[0]);
    r.inv_h = 1.0f / (py[0] - py[3]);
    r.kx = 2.0f * r.p0 * r.inv_w;  // (meaningful for SDF quads, where p0, p1 are the quad's half extents)
    r.ky = 2.0f * r.p1 * r.inv_h;
    if (!bezier) set_saturated_core(r, px[1] - px[0], py[0] - py[3]);
  }
  if (!aligned || bezier) {
    QuadExt& q = *lane_->exts.slot();
    std::memset(static_cast<void*>(&q), 0, sizeof q);
    static const int TRI[2][3] = {{3, 0, 1}, {2, 3, 1}};  // glcontext.nim:418-429
    const uint32_t mode = r.op_mode & 255u;
    const bool atlas_mode = mode == 0u || (mode >= 13u && mode <= 16u);
    const float uax = atlas_mode ? r.r[0] : 0.0f, uay = atlas_mode ? r.r[1] : 0.0f, utx = atlas_mode ? r.r[2] : 1.0f, uty = atlas_mode ? r.r[3] : 1.0f;
    const float vu[4] = {uax, utx, utx, uax}, vv[4] = {uty, uty, uay, uay};
    // F_EDGE32 (fdh_types.h): may the compositor evaluate the edge functions in 32-bit arithmetic at every pixel of the frame's strips?
    bool edge32 = true;
    const long long xm = 2LL * cx_->W_ + 129, ym = 2LL * cx_->H_ + 129;  // (every pixel of every 64 x 64 bin that overlaps the frame)
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
        {
          const long long aa = A < 0 ? -A : A, ab = B < 0 ? -B : B, ac = C < 0 ? -(C + 1) : C;
          edge32 = edge32 && aa < (1LL << 23) && ab < (1LL << 23) && ac < (1LL << 30) && aa * xm + ab * ym + ac + 2 < (1LL << 31);
        }
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
        const double S = (double)cx_->atlas_size_;
        const double rho = std::max(std::sqrt(dudx * dudx + dvdx * dvdx), std::sqrt(dudy * dudy + dvdy * dvdy)) * S;
        q.lod[t] = rho > 0.0 ? (float)std::log2(rho) : 0.0f;
      } else {
        q.fw_u[t] = q.fw_v[t] = 1.0f;
      }
    }
    r.op_mode |= F_GENERAL;
    if (edge32) r.op_mode |= F_EDGE32;
    // the saturated core of a rotated SDF draw, in the local frame, and the two triangles' maps into that frame (QuadExt::core, lm)
    q.core[0] = q.core[1] = q.core[2] = q.core[3] = 0.0f;
    double xl, xr, yb, yt;
    if (edge32 && !atlas_mode && mode < 18u && ((r.op_mode >> 12) & 15u) == OP_DRAW && q.inv_sum[0] != 0.0f && q.inv_sum[1] != 0.0f && local_core(r, xl, xr, yb, yt)) {
      // tri 0 = (TL, BL, BR): u = E2 is, v = (E1 + E2) is;  tri 1 = (TR, TL, BR): u = (E0 + E2) is, v = E2 is;  local x = (u - 0.5) 2 qhx,
      // local y (up) = -(v - 0.5) 2 qhy (atlas.frag:252-262)
      const double qhx = r.p0, qhy = r.p1;
      double sx = 0.0, sy = 0.0;
      for (int t = 0; t < 2; t++) {
        const QuadExt::Edge &E0 = q.e[t][0], &E1 = q.e[t][1], &E2 = q.e[t][2];
        const double is = (double)q.inv_sum[t];
        const double ua = t == 0 ? (double)E2.a : (double)E0.a + E2.a, ub = t == 0 ? (double)E2.b : (double)E0.b + E2.b, uc = t == 0 ? (double)E2.c : (double)E0.c + (double)E2.c;
        const double va = t == 0 ? (double)E1.a + E2.a : (double)E2.a, vb = t == 0 ? (double)E1.b + E2.b : (double)E2.b, vc = t == 0 ? (double)E1.c + (double)E2.c : (double)E2.c;
        const double kx = 2.0 * qhx * is, ky = -2.0 * qhy * is;
        const double m[6] = {kx * ua, kx * ub, kx * uc - qhx, ky * va, ky * vb, ky * vc + qhy};
        for (int k = 0; k < 6; k++) q.lm[t][k] = (float)m[k];
        sx = std::max(sx, 2.0 * (std::fabs(m[0]) + std::fabs(m[1])));  // local units per pixel step
        sy = std::max(sy, 2.0 * (std::fabs(m[3]) + std::fabs(m[4])));
      }
#ifndef FDH_CORE_SLACK
#define FDH_CORE_SLACK (1.0 / 16.0)
#endif
      xl += sx * FDH_CORE_SLACK; xr -= sx * FDH_CORE_SLACK; yb += sy * FDH_CORE_SLACK; yt -= sy * FDH_CORE_SLACK;
      if (xr > xl && yt > yb) { q.core[0] = (float)xl; q.core[1] = (float)xr; q.core[2] = (float)yb; q.core[3] = (float)yt; }
    }
    r.ext = (uint32_t)lane_->exts.n;  // lane-relative: the upload re-bases it (k_upload_frame)
    lane_->exts.n++;
  }
  if (count_fragments) fragments_ += (int64_t)(b.x1 - b.x0) * (b.y1 - b.y0);
  for (auto idx : open_ops_) bbox_union(lane_->bins[idx].box, b);  // clip pushes only need to reach tiles their content touches
  if (outer_open_) bbox_union(outer_union_, b);                      // ... and so do the ones open around a pool thread's sibling group
  push_rec(b);
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
                         float aa) {  // (`r` arrives zeroed)
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
  std::memset(static_cast<void*>(&r), 0, sizeof r);
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
void Recorder::draw_rounded_rect_sdf(const float rect[4], const FdhColor colors[4], const float rx[4], const float ry[4], int mode,
                                    float factor, float spread, const float shape[2], int fill_mode, FdhColor mid, FdhColor stop,
                                    float mid_pos) {
  { FDH_REC("draw_rounded_rect_sdf").fv(rect, 4).cols(colors).fv(rx, 4).fv(ry, 4).i(mode).f(factor).f(spread).fv(shape, 2).i(fill_mode).col(mid).col(stop).f(mid_pos); }
  if (!cx_->frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  if (!(rect[2] > 0.0f) || !(rect[3] > 0.0f)) return;  // (a NaN extent draws nothing)
  if (mode >= FDH_SDF_BEZIER_STROKE_AA) throw Error(FDH_ERR_INVALID, "bezier stroke modes go through drawQuadraticBezierSdf");
  QuadPx q;
  {
    const float x0 = rect[0], y0 = rect[1], x1 = rect[0] + rect[2], y1 = rect[1] + rect[3];
    const float vx[4] = {x0, x1, x1, x0}, vy[4] = {y1, y1, y0, y0};  // BL, BR, TR, TL
    quad_corners(vx, vy, q);
  }
  // (before the record is built: most of a long table is below the window)
  if (culling() && (!q.finite || !bbox_visible(q.b))) { FDH_CULLED(); return; }
  DrawRec& r = next_rec();
  fill_sdf_rec(r, rect, colors, rx, ry, mode, factor, spread, shape, fill_mode, mid, stop, mid_pos, aa_);
  if (mode == FDH_SDF_BACKDROP_BLUR) r.op_mode |= F_SELF_BACKDROP;  // a bare mode-17 call has no snapshot of its own
  if (emit_corners(r, q, true)) commit_bins((uint32_t)lane_->recs.n - 1);
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
void Recorder::draw_rounded_rect_fill(const float rect[4], const FdhFill& fill, const float rx[4], const float ry[4], int mode,
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
void Recorder::draw_image(int64_t key, const float pos[2], const FdhColor colors[4], const float size[2], bool flip_y) {
  { FDH_REC("draw_image").i(key).fv(pos, 2).cols(colors).fv(size, 2).i(flip_y ? 1 : 0); }
  if (!cx_->frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  auto it = cx_->entries_.find(key);
  if (it == cx_->entries_.end()) return;  // "missing image in context": warn + no-op (glcontext.nim:1310-1315)
  const AtlasEntry& e = it->second;
  const float S = (float)cx_->atlas_size_;
  const float ex = (float)e.x / S, ey = (float)e.y / S, ew = (float)e.w / S, eh = (float)e.h / S;  // entries = rect / atlasSize
  const bool sized = size[0] > 0.0f && size[1] > 0.0f;
  const float dw = sized ? size[0] : ew * S, dh = sized ? size[1] : eh * S;
  DrawRec& r = next_rec();
  r.op_mode = FDH_SDF_ATLAS;
  if (flip_y) { r.r[0] = ex; r.r[1] = ey + eh; r.r[2] = ex + ew; r.r[3] = ey; }
  else { r.r[0] = ex; r.r[1] = ey; r.r[2] = ex + ew; r.r[3] = ey + eh; }
  for (int i = 0; i < 4; i++) r.col[i] = pack_color(colors[i]);
  if (r.col[0] == r.col[1] && r.col[1] == r.col[2] && r.col[2] == r.col[3]) r.op_mode |= F_SOLID;
  r.aa = aa_;
  const bool subpixel_enabled_ = cx_->subpixel_enabled_;
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
  if (!emit_quad(r, x0, y0, x1, y1, true)) return;
  if (one_to_one && !(r.op_mode & F_GENERAL)) {
    DrawRec& rr = r;
    rr.op_mode |= F_TEXEL_1TO1;
    rr.ext = (uint32_t)(int32_t)(e.x - (int)qx0);   // texel x = pixel x + tdx
    rr._pad = (uint32_t)(int32_t)(e.y - (int)qy0);  // texel y = pixel y + tdy
  }
  // atlas.frag:284-295: the source alpha is texel alpha x vertex alpha -- 0 wherever all four taps have alpha 0.  (Level 0 only:
  // a minified image, aux2 > 0, takes its taps from coarser levels.)
  if (!(r.aux2 > 0.0f)) shrink_to_ink(e, true, 0);
  commit_bins((uint32_t)lane_->recs.n - 1);
}

// drawImageAdj: glcontext.nim:1369-1381 -- drawImage with the uv rect pulled in by two texels on every side (2 / atlasSize)
void Recorder::draw_image_adj(int64_t key, const float pos[2], FdhColor color, const float size[2]) {
  { FDH_REC("draw_image_adj").i(key).fv(pos, 2).col(color).fv(size, 2); }
  if (!cx_->frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  auto it = cx_->entries_.find(key);
  if (it == cx_->entries_.end()) return;
  const AtlasEntry& e = it->second;
  const float S = (float)cx_->atlas_size_;
  const float ex = (float)e.x / S, ey = (float)e.y / S, ew = (float)e.w / S, eh = (float)e.h / S, adj = 2.0f / S;
  DrawRec& r = next_rec();
  r.op_mode = FDH_SDF_ATLAS | F_SOLID;
  r.r[0] = ex + adj; r.r[1] = ey + adj; r.r[2] = ex + ew - adj; r.r[3] = ey + eh - adj;
  for (int i = 0; i < 4; i++) r.col[i] = pack_color(color);
  r.aa = aa_;
  if (cx_->subpixel_enabled_) {
    r.op_mode |= F_SUBPIXEL;
    r.aux = std::max(0.0f, std::min(subpixel_shift_, 0.999f));
  }
  const float x0 = pos[0], y0 = pos[1], x1 = pos[0] + size[0], y1 = pos[1] + size[1];
  {  // LOD of the axis-aligned form, as in draw_image
    const float qx0 = std::ceil(mat_.a * x0 + mat_.tx), qy0 = std::ceil(mat_.d * y0 + mat_.ty);
    const float qx1 = std::ceil(mat_.a * x1 + mat_.tx), qy1 = std::ceil(mat_.d * y1 + mat_.ty);
    const float rw = std::fabs(qx1 - qx0), rh = std::fabs(qy1 - qy0);
    if (rw > 0.0f && rh > 0.0f) {
      const float rho = std::max(std::fabs(r.r[2] - r.r[0]) * S / rw, std::fabs(r.r[3] - r.r[1]) * S / rh);
      r.aux2 = rho > 0.0f ? std::log2(rho) : 0.0f;
    }
  }
  if (emit_quad(r, x0, y0, x1, y1, true)) commit_bins((uint32_t)lane_->recs.n - 1);
}

// drawMsdfImage / drawMtsdfImage: glcontext.nim:1097-1155, drawUvRectAtlasSdf :1022-1093
void Recorder::draw_msdf(int64_t key, const float pos[2], FdhColor color, const float size[2], float px_range, float sd_threshold,
                        float stroke_weight, bool mtsdf, bool flip_y) {
  { FDH_REC("draw_msdf").i(key).fv(pos, 2).col(color).fv(size, 2).f(px_range).f(sd_threshold).f(stroke_weight).i(mtsdf ? 1 : 0).i(flip_y ? 1 : 0); }
  if (!cx_->frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  auto it = cx_->entries_.find(key);
  if (it == cx_->entries_.end()) return;
  const AtlasEntry& e = it->second;
  const float S = (float)cx_->atlas_size_;
  const float ex = (float)e.x / S, ey = (float)e.y / S, ew = (float)e.w / S, eh = (float)e.h / S;
  const float sw = std::max(0.0f, stroke_weight);
  DrawRec& r = next_rec();
  r.op_mode = (uint32_t)(mtsdf ? (sw > 0.0f ? FDH_SDF_MTSDF_ANNULAR : FDH_SDF_MTSDF) : (sw > 0.0f ? FDH_SDF_MSDF_ANNULAR : FDH_SDF_MSDF)) | F_SOLID;
  if (flip_y) { r.r[0] = ex; r.r[1] = ey + eh; r.r[2] = ex + ew; r.r[3] = ey; }
  else { r.r[0] = ex; r.r[1] = ey; r.r[2] = ex + ew; r.r[3] = ey + eh; }
  r.p0 = S; r.p1 = sw;
  r.f0 = px_range; r.f1 = sd_threshold;
  for (int i = 0; i < 4; i++) r.col[i] = pack_color(color);
  r.aa = aa_;
  if (!emit_quad(r, pos[0], pos[1], pos[0] + size[0], pos[1] + size[1], true)) return;
  // atlas.frag:296-318, the fill variants: alpha = clamp(spr (sd - threshold) + 0.5), sd the median of the filtered r, g, b
  // (MTSDF: the filtered alpha) -- exactly 0 wherever sd <= threshold - 0.5 / spr.  Where all four taps have every channel <= t
  // the filtered channels, hence their median, are <= t: the box of texels above a level safely below that bound is all the
  // draw can touch.  (Stroke variants cover a band around the outline whatever sd is beyond it: left alone.)
  if (!(sw > 0.0f) && !(r.op_mode & F_GENERAL)) {
    const DrawRec& rr = r;
    const double unit = (double)px_range / (double)S;
    const double fw_u = std::fabs((double)(rr.r[2] - rr.r[0]) * rr.inv_w), fw_v = std::fabs((double)(rr.r[3] - rr.r[1]) * rr.inv_h);
    if (fw_u > 0.0 && fw_v > 0.0) {
      const double spr = std::max(0.5 * (unit / fw_u + unit / fw_v), 1.0);
      const double cut = (double)sd_threshold - 0.5 / spr - 0.008;  // two 8-bit steps below the bound (the kernel's rcp is good to 1e-7)
      if (cut > 0.0 && cut <= 1.0) shrink_to_ink(e, mtsdf, (int)std::floor(cut * 255.0) - 1);  // (NaN parameters: no shrink)
    }
  }
  commit_bins((uint32_t)lane_->recs.n - 1);
}

// drawQuadraticBezierSdf: glcontext.nim:1619-1741
void Recorder::draw_quadratic_bezier_sdf(const float rect[4], const FdhFill& fill, const float p0[2], const float p1[2],
                                        const float p2[2], float stroke_weight, int cap) {
  { FDH_REC("draw_quadratic_bezier_sdf").fv(rect, 4).fill(fill).fv(p0, 2).fv(p1, 2).fv(p2, 2).f(stroke_weight).i(cap); }
  if (!cx_->frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  if (!(rect[2] > 0.0f) || !(rect[3] > 0.0f) || !(stroke_weight > 0.0f)) return;
  DrawRec& r = next_rec();
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
  if (emit_quad(r, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], true)) commit_bins((uint32_t)lane_->recs.n - 1);
}

// The 4x4 white "rect" atlas image drawRect / drawFilledQuad sample (glcontext.nim:966-970, 1411-1415); it takes
// atlas space on first use exactly like the reference's -- on the calling thread: a pool thread that finds it missing hands its
// sibling group back (SerialOnly).
const AtlasEntry& Recorder::rect_entry() {
  auto it = cx_->entries_.find(kRectImageKey);
  if (it == cx_->entries_.end()) {
    if (!is_main_) throw SerialOnly{};
    uint8_t white[4 * 4 * 4];
    std::memset(white, 255, sizeof white);
    cx_->put_image(kRectImageKey, 4, 4, white, nullptr);
    it = cx_->entries_.find(kRectImageKey);
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
void Recorder::draw_filled_quad(const float verts[8], const FdhColor colors[4]) {
  { FDH_REC("draw_filled_quad").fv(verts, 8).cols(colors); }
  if (!cx_->frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  const AtlasEntry& white = rect_entry();  // (may upload the image: before the record slot is taken)
  DrawRec& r = next_rec();
  r.op_mode = FDH_SDF_ATLAS;
  white_texel_uv(white, cx_->atlas_size_, r);
  for (int i = 0; i < 4; i++) r.col[i] = pack_color(colors[i]);
  if (r.col[0] == r.col[1] && r.col[1] == r.col[2] && r.col[2] == r.col[3]) r.op_mode |= F_SOLID;
  r.aa = aa_;
  const float vx[4] = {verts[0], verts[2], verts[4], verts[6]}, vy[4] = {verts[1], verts[3], verts[5], verts[7]};
  if (emit_quad_pts(r, vx, vy, true)) commit_bins((uint32_t)lane_->recs.n - 1);
}
// drawRect: glcontext.nim:1410-1426
void Recorder::draw_rect(const float rect[4], FdhColor color) {
  { FDH_REC("draw_rect").fv(rect, 4).col(color); }
  if (!cx_->frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  const AtlasEntry& white = rect_entry();
  DrawRec& r = next_rec();
  r.op_mode = FDH_SDF_ATLAS | F_SOLID;
  white_texel_uv(white, cx_->atlas_size_, r);
  for (int i = 0; i < 4; i++) r.col[i] = pack_color(color);
  r.aa = aa_;
  if (emit_quad(r, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], true)) commit_bins((uint32_t)lane_->recs.n - 1);
}

// ------------------------------------------------------------------ masks (glcontext.nim:1873-1949)
void Recorder::begin_mask(const float rect[4], const float rx[4], const float ry[4]) {
  { FDH_REC("begin_mask").fv(rect, 4).fv(rx, 4).fv(ry, 4); }
  if (!cx_->frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  if (mask_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginMask has already been called.");
  mask_begun_ = true;
  mask_depth_++;  // (beyond kMaskDepth levels the compositor's stack spills to a global plane: Context::prepare)
  const FdhColor red{255, 0, 0, 255}, zero{0, 0, 0, 0};
  const FdhColor cols[4] = {red, red, red, red};
  const float shape[2] = {0, 0};
  DrawRec& r = next_rec();
  fill_sdf_rec(r, rect, cols, rx, ry, FDH_SDF_CLIP_AA, 4.0f, 0.0f, shape, 0, zero, zero, 0.5f, aa_);
  r.op_mode |= OP_MASK_PUSH << 12;
  const uint32_t before = (uint32_t)lane_->recs.n;
  if (rect[2] > 0.0f && rect[3] > 0.0f) {
    emit_quad(r, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], false);
  } else {  // drawRoundedRectSdf returns early: the mask plane stays cleared to 0
    r.bx0 = r.by0 = r.bx1 = r.by1 = 0;
    push_rec(BBox{0, 0, 0, 0});
  }
  lane_->bins[before].box = BBox{0, 0, 0, 0};  // grows to the union of the content drawn under it; final (commit_bins) at the pop
  open_ops_.push_back(before);
  depth_now_++;
  sum_.deepest = std::max(sum_.deepest, depth_now_);
}
void Recorder::end_mask() {
  { FDH_REC("end_mask"); }
  if (!mask_begun_) throw Error(FDH_ERR_INVALID, "ctx.maskBegun has not been called.");
  mask_begun_ = false;
}
void Recorder::pop_mask() {
  { FDH_REC("pop_mask"); }
  if (mask_depth_ <= 0 || open_ops_.empty()) throw Error(FDH_ERR_INVALID, "popMask without beginMask");
  const uint32_t push_idx = open_ops_.back();
  open_ops_.pop_back();
  mask_depth_--;
  commit_bins(push_idx);
  DrawRec& r = next_rec();
  r.op_mode = OP_MASK_POP << 12;
  push_rec(lane_->bins[push_idx].box);
  commit_bins((uint32_t)lane_->recs.n - 1);
  depth_now_--;
}
// makeRectMask glcontext.nim:831-850; beginRectMask :1932-1943
void Recorder::begin_rect_mask(const float rect[4], const float rx[4], const float ry[4]) {
  { FDH_REC("begin_rect_mask").fv(rect, 4).fv(rx, 4).fv(ry, 4); }
  if (!cx_->frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  if (mask_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginRectMask cannot start inside a mask.");
  if (rect_masks_.empty() && outer_rect_masks_ == 0 && rect[2] > 0.0f && rect[3] > 0.0f) {
    const float hx = rect[2] * 0.5f, hy = rect[3] * 0.5f;
    DrawRec& r = next_rec();
    const bool ellip = rounded_radii_vec(rx, ry, hx, hy, r.r);
    const float det = mat_.a * mat_.d - mat_.b * mat_.c, id = 1.0f / det;
    const float ia = mat_.d * id, ib = -mat_.b * id, ic = -mat_.c * id, idd = mat_.a * id;
    const float itx = -(ia * mat_.tx + ic * mat_.ty), ity = -(ib * mat_.tx + idd * mat_.ty);
    r.ox = ia; r.oy = ic; r.inv_w = itx;   // matX
    r.inv_h = ib; r.f0 = idd; r.f1 = ity;  // matY
    r.p0 = rect[0] + hx; r.p1 = rect[1] + hy; r.p2 = hx; r.p3 = hy;
    r.aa = aa_;
    r.op_mode = (OP_RMASK_BEGIN << 12) | (ellip ? F_ELLIP : 0u);
    open_ops_.push_back((uint32_t)lane_->recs.n);
    push_rec(BBox{0, 0, 0, 0});
    rect_masks_.push_back(RectMaskEntry{1});
  } else {
    { const RecPause quiet(cx_->rec_on_);  // the fallback's own begin/end are this backend's business, not the caller's
      begin_mask(rect, rx, ry);
      end_mask(); }
    rect_masks_.push_back(RectMaskEntry{2});
  }
}
void Recorder::pop_rect_mask() {
  { FDH_REC("pop_rect_mask"); }
  if (rect_masks_.empty()) throw Error(FDH_ERR_INVALID, "No rect mask has been pushed.");
  const RectMaskEntry e = rect_masks_.back();
  rect_masks_.pop_back();
  if (e.kind == 2) { const RecPause quiet(cx_->rec_on_); pop_mask(); return; }
  const uint32_t begin_idx = open_ops_.back();
  open_ops_.pop_back();
  commit_bins(begin_idx);
  DrawRec& r = next_rec();
  r.op_mode = OP_RMASK_END << 12;
  push_rec(lane_->bins[begin_idx].box);
  commit_bins((uint32_t)lane_->recs.n - 1);
}

// ------------------------------------------------------------------ backdrop blur (glcontext.nim:1743-1841, blur.frag:11-32)
BlurTaps make_taps(float blur_radius) {
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

void Recorder::draw_backdrop_blur(const float rect[4], const float rx[4], const float ry[4], float blur_radius) {
  { FDH_REC("draw_backdrop_blur").fv(rect, 4).fv(rx, 4).fv(ry, 4).f(blur_radius); }
  if (!cx_->frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has not been called.");
  if (!(blur_radius > 0.0f) || !(rect[2] > 0.0f) || !(rect[3] > 0.0f)) return;  // (written so that a NaN draws nothing)
  const FdhColor white{255, 255, 255, 255}, zero{0, 0, 0, 0};
  const FdhColor cols[4] = {white, white, white, white};
  const float shape[2] = {0, 0};
  DrawRec quad;
  std::memset(static_cast<void*>(&quad), 0, sizeof quad);
  fill_sdf_rec(quad, rect, cols, rx, ry, FDH_SDF_BACKDROP_BLUR, blur_radius, 0.0f, shape, 0, zero, zero, 0.5f, aa_);
  if (blur_radius <= 0.5f) {  // runBackdropSeparableBlur returns early: the snapshot is the live frame
    quad.op_mode |= F_SELF_BACKDROP;
    DrawRec& r = next_rec();
    r = quad;
    if (emit_quad(r, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], true)) commit_bins((uint32_t)lane_->recs.n - 1);
    return;
  }
  if (culling() && !rect_visible(rect, 0.0f)) { FDH_CULLED(); return; }  // a blurred backdrop nobody sees: no snapshot, no phase
  if (!is_main_) throw SerialOnly{};  // phases are the calling thread's
  Context& C = *cx_;
  // A blurred snapshot is a barrier in painter's order: close the phase, blur, continue in a new phase.  The open clips end
  // with the phase (their bounds are final there) and are re-established, as new records, at the start of the next.
  std::vector<DrawRec> reopen;
  for (auto idx : open_ops_) { reopen.push_back(lane_->recs[idx]); commit_bins(idx); }
  C.split_phase((int)C.blurs_.size());
  depth_now_ = 0;
  for (size_t i = 0; i < reopen.size(); i++) {
    open_ops_[i] = (uint32_t)lane_->recs.n;
    DrawRec& nr = next_rec();
    nr = reopen[i];
    push_rec(BBox{0, 0, 0, 0});
    if (((nr.op_mode >> 12) & 15u) == OP_MASK_PUSH) { depth_now_++; sum_.deepest = std::max(sum_.deepest, depth_now_); }
  }
  // no clip state to carry: the V pass can composite the quad itself -- unless the region is small enough for the one-kernel
  // route, whose snapshot goes to the backdrop surface and is sampled by the phase's compositor launch (k_blur_small)
  DrawRec& r = next_rec();
  r = quad;
  if (!emit_quad(r, rect[0], rect[1], rect[0] + rect[2], rect[1] + rect[3], true)) throw Error(FDH_ERR_INVALID, "drawBackdropBlur: rect_visible and emit_quad disagree");
  const uint32_t idx = (uint32_t)lane_->recs.n - 1;
  const BBox fb = lane_->bins[idx].box;
  BlurJob job;
  job.taps = make_taps(blur_radius);
  const bool fuse = open_ops_.empty() && !(C.latency_routes_ && blur_one_kernel_ok(fb.x1 - fb.x0, fb.y1 - fb.y0, job.taps.reach));
  job.fuse_draw = -1;
  if (fuse) {
    job.fuse_draw = (int)C.global_index(idx);
    lane_->bins[idx].box = BBox{0, 0, 0, 0};  // never binned: k_blur_v blends it
  }
  commit_bins(idx);
  job.radius = blur_radius;
  job.x0 = fb.x0; job.y0 = fb.y0; job.x1 = fb.x1; job.y1 = fb.y1;
  C.blurs_.push_back(job);
}

// ------------------------------------------------------------------ the frame: pieces and phases (calling thread)
Lane& Context::ensure_lane(int i) {
  auto& v = lanes_[(size_t)staging_i_];
  while ((int)v.size() <= i) {
    v.emplace_back(new Lane());
    v.back()->set_pinned(!host_only_, device_);
  }
  return *v[(size_t)i];
}
uint32_t Context::global_count() const { return piece_open_ ? (uint32_t)lanes_[(size_t)staging_i_][0]->recs.n + g0_delta_ : n_total_; }
void Context::open_piece() {
  Lane& L = lane(0);
  Piece p;
  p.lane = 0; p.first = (uint32_t)L.recs.n; p.ext_first = (uint32_t)L.exts.n;
  pieces_.push_back(p);
  piece_open_ = true;
  g0_delta_ = n_total_ - p.first;
  phase_floor_ = (int)p.first;  // (LE_SHARE looks one record back in the lane: never across a piece boundary)
}
void Context::close_piece() {
  if (!piece_open_) return;
  Lane& L = lane(0);
  Piece& p = pieces_.back();
  p.n = (uint32_t)L.recs.n - p.first;
  p.n_ext = (uint32_t)L.exts.n - p.ext_first;
  n_total_ += p.n;
  n_ext_total_ += p.n_ext;
  piece_open_ = false;
  if (p.n == 0) pieces_.pop_back();
}
void Context::add_sum(const PhaseSum& s, int depth_base) {
  Phase& ph = phases_.back();
  ph.has_masks = ph.has_masks || s.has_masks;
  ph.has_atlas = ph.has_atlas || s.has_atlas;
  ph.has_slow = ph.has_slow || s.has_slow;
  ph.has_slow_atlas = ph.has_slow_atlas || s.has_slow_atlas;
  ph.has_rot = ph.has_rot || s.has_rot;
  bbox_union(phase_u_, s.u);
  deepest_clip_ = std::max(deepest_clip_, depth_base + s.deepest);
  if (phases_.size() == 1) {  // the phase-0 composite launch's work units
    for (int k = 0; k < 4; k++) frag_mode_[k] += s.frag_mode[k];
    frag_ellip_ += s.frag_ellip;
    frag_other_ += s.frag_other;
  }
}
// a pool thread's chunk of a sibling group takes its place in painter's order (the caller closed lane 0's piece before the group)
void Context::add_piece(const Piece& p, const PhaseSum& s, const BBox& outer_union, int64_t fragments, int64_t culled) {
  if (p.n) {
    pieces_.push_back(p);
    n_total_ += p.n;
    n_ext_total_ += p.n_ext;
  }
  Lane& L = lane(0);
  for (auto idx : open_ops_) bbox_union(L.bins[idx].box, outer_union);  // the clips open around the group reach what it drew
  add_sum(s, depth_now_);
  fragments_ += fragments;
  culled_draws_ += culled;
}
// the phase ends here (a blur node, or the frame's end): its summary and its share of the list stride
void Context::close_phase() {
  add_sum(sum_, 0);
  sum_ = PhaseSum{};
  Phase& ph = phases_.back();
  constexpr int kBinShift = 6;
  const BBox u = phase_u_;
  ph.bin_x0 = u.x0 >> kBinShift; ph.bin_y0 = u.y0 >> kBinShift;
  ph.bin_x1 = bbox_empty(u) ? ph.bin_x0 : (u.x1 + kBin - 1) >> kBinShift;
  ph.bin_y1 = bbox_empty(u) ? ph.bin_y0 : (u.y1 + kBin - 1) >> kBinShift;
  phase_u_ = BBox{0, 0, 0, 0};
  stride_max_ = std::max(stride_max_, lane(0).count_close() + phase_extra_);
  phase_extra_ = 0;
  ph.count = (int)(global_count() - (uint32_t)ph.first);
}
void Context::split_phase(int blur) {
  close_phase();
  Phase next;
  next.first = (int)global_count();
  next.blur = blur;
  phases_.push_back(next);
  phase_floor_ = (int)lane(0).recs.n;
}

void Context::begin_frame(int w, int h, bool clear, const float rgba[4]) {  // glcontext.nim:2080-2092, 1951-1980
  { FDH_REC("begin_frame").i(clear ? 1 : 0).fv(rgba, 4); }
  if (frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame has already been called.");
  if (w <= 0 || h <= 0 || w > 16384 || h > 16384) throw Error(FDH_ERR_INVALID, "beginFrame: frame size must be in 1..16384");
  t_begin_frame_ = std::chrono::steady_clock::now();
  for (auto& v : host_ns_) v = 0;
  if (!host_only_) FDH_HIP(hipSetDevice(device_));
  W_ = w;
  H_ = h;
  ensure_surfaces();
  clear_ = clear;
  if (clear) {
    auto q = [](float v) { return (uint32_t)std::floor(clampf(v, 0.0f, 1.0f) * 255.0f + 0.5f); };
    clear_rgba8_ = q(rgba[0]) | (q(rgba[1]) << 8) | (q(rgba[2]) << 16) | (q(rgba[3]) << 24);
  }
  // The records of this frame go into the next set of lanes: the set's last user was the frame kStaging frames ago, whose upload
  // kernel has run by now (that frame's issue was waited for by the end_frame after it: the event is recorded).
  staging_i_ = (staging_i_ + 1) % kStaging;
  frame_no_++;
  if (!host_only_ && staging_busy_[staging_i_]) { HostTimer t(host_ns_[1]); wait_staging(staging_i_); }
  Lane& L0 = ensure_lane(0);
  L0.clear();
  L0.count_begin((w + kBin - 1) / kBin, (h + kBin - 1) / kBin);
  binbox_shift_ = ((w + kBin - 1) / kBin > 128 || (h + kBin - 1) / kBin > 128) ? 1 : 0;
  lane_ = &L0;
  frame_begun_ = true;
  mask_begun_ = false;
  mask_depth_ = 0;
  rect_masks_.clear();
  open_ops_.clear();
  outer_rect_masks_ = 0;
  outer_open_ = false;
  outer_union_ = BBox{0, 0, 0, 0};
  depth_now_ = 0;
  sum_ = PhaseSum{};
  fragments_ = 0;
  culled_draws_ = 0;
  pieces_.clear();
  n_total_ = n_ext_total_ = 0;
  piece_open_ = false;
  phases_.clear();
  phases_.push_back(Phase{});
  blurs_.clear();
  phase_u_ = BBox{0, 0, 0, 0};
  stride_max_ = 1;
  phase_extra_ = 0;
  deepest_clip_ = 0;
  for (auto& f : frag_mode_) f = 0;
  frag_ellip_ = frag_other_ = 0;
  parallel_groups_ = 0;
  rec_diff_upload_ = false;
  open_piece();
  pick_routes();
  // rows a draw has to reach: the frame's, or -- under fdh_set_stripe, when the front-end has told how far the scene's blur nodes
  // reach (render_frame: the per-call path cannot know what is still to come) -- the stripe's, widened by that reach
  cull_y0_ = 0; cull_y1_ = H_;
  if (stripe_y1_ > stripe_y0_ && pending_reach_ >= 0) {
    cull_y0_ = std::max(0, std::min(H_, stripe_y0_) - pending_reach_);
    cull_y1_ = std::min(H_, std::max(0, stripe_y1_) + pending_reach_);
  }
  pending_reach_ = -1;
  t_walk_begin_ = std::chrono::steady_clock::now();
  host_ns_[0] = std::chrono::duration_cast<std::chrono::nanoseconds>(t_walk_begin_ - t_begin_frame_).count();
}

// end_frame = prepare (this thread) + issue (the context's submit thread).
//   prepare  lays the frame block out and lists the runs the upload kernel gathers; everything per record was produced while the
//            frame was recorded (commit_bins).  It runs on the CALLING thread.
//   issue    launches the upload kernel and the frame's kernels (~20 us of HIP runtime calls) from the submit thread, so the
//            caller is already walking the next frame's tree.  FDH_CREATE_SYNC_SUBMIT contexts run it inline.
void Context::end_frame() {  // glcontext.nim:1982-1989
  { FDH_REC("end_frame"); }
  if (!frame_begun_) throw Error(FDH_ERR_INVALID, "ctx.beginFrame was not called first.");
  if (mask_depth_ != 0) throw Error(FDH_ERR_INVALID, "Not all masks have been popped.");
  if (!rect_masks_.empty()) throw Error(FDH_ERR_INVALID, "Not all rect masks have been popped.");
  frame_begun_ = false;
  const auto t0 = std::chrono::steady_clock::now();
  host_ns_[2] = std::chrono::duration_cast<std::chrono::nanoseconds>(t0 - t_walk_begin_).count();
  close_phase();
  close_piece();
  culled_total_ = culled_draws_;
  const auto t1 = std::chrono::steady_clock::now();
  host_ns_[3] = std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
  // (without what begin_frame waited for the GPU -- the lane set's previous upload: back-pressure, not work)
  host_record_ms_ = std::chrono::duration<float, std::milli>(t1 - t_begin_frame_).count() - (float)host_ns_[1] * 1e-6f;
  // a list entry carries the draw index in 25 bits beside its path code and flags (k_bin_draws, LE_INDEX)
  if (n_total_ >= LE_INDEX) throw Error(FDH_ERR_INVALID, "more than 33 554 430 draw records in one frame");
  if (host_only_) return;
  // prepare() notes what the device block will hold once this frame's upload has run (blur tables, the retained path's shadow):
  // if the frame is dropped before it is handed over -- prepare or the wait for the previous frame's launches throws, or an
  // inline issue fails -- those notes are void (a later frame would skip uploads the device never received)
  try {
    { HostTimer t(host_ns_[4]); prepare(next_); }
    { HostTimer t(host_ns_[6]); drain(); }  // the previous frame's launches (normally long issued: they ran while this frame was being recorded)
    std::swap(job_, next_);
    have_frame_ = true;
    if (!worker_.joinable()) { issue(job_); return; }
  } catch (...) {
    tables_dev_ = nullptr; shadow_dev_ = nullptr; have_frame_ = false;
    throw;
  }
  {
    std::lock_guard<std::mutex> lk(mu_);
    pending_.store(true, std::memory_order_release);
  }
  cv_job_.notify_one();
}

// FNV-1a over the last frame's records in painter's order, in the form the calls produced them (four vertex colours, extension
// indices counted over the whole frame), their bounds, extensions and the phase table: two frames with equal digests hand the
// kernels identical input, however many threads recorded them.
uint64_t Context::record_digest() {
  drain();
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(p); for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
  const uint64_t n = n_total_;
  mix(&n, sizeof n);
  uint32_t ext_base = 0;
  for (const Piece& p : pieces_) {
    const Lane& L = lane(p.lane);
    for (uint32_t i = 0; i < p.n; i++) {
      DrawRec r = L.recs[p.first + i];
      record_host_form(r);
      if (r.op_mode & F_GENERAL) r.ext = r.ext - p.ext_first + ext_base;
      mix(&r, sizeof r);
    }
    ext_base += p.n_ext;
  }
  for (const Piece& p : pieces_) { const Lane& L = lane(p.lane); for (uint32_t i = 0; i < p.n; i++) mix(&L.bins[p.first + i].box, sizeof(BBox)); }
  for (const Piece& p : pieces_) { const Lane& L = lane(p.lane); for (uint32_t i = 0; i < p.n_ext; i++) mix(&L.exts[p.ext_first + i], sizeof(QuadExt)); }
  for (const Phase& ph : phases_) { mix(&ph.first, sizeof ph.first); mix(&ph.count, sizeof ph.count); mix(&ph.blur, sizeof ph.blur); }
  return h;
}

// Fault hunting (fdh_debug_verify_upload): the device's frame block -- what k_upload_frame gathered for the frame last submitted --
// read back and compared with the lanes the records were made in (ordinary host memory, untouched until the staging set comes
// round again).  out[0..2] = bytes that differ in records / bin records / extensions, out[3] = bytes compared; out[4..9] describe the
// first difference: array (0, 1, 2), byte offset in the device array, the piece's lane, device dword, host dword, dwords of the
// device run that are zero; out[10] = pieces, out[11] = records.
void Context::debug_verify_upload(uint32_t out[24]) {
  need_device("debug_verify_upload");
  drain();
  FDH_HIP(hipSetDevice(device_));
  FDH_HIP(hipStreamSynchronize(stream_));
  for (int i = 0; i < 24; i++) out[i] = 0;
  const LaunchJob& J = job_;
  out[10] = (uint32_t)pieces_.size(); out[11] = n_total_;
  if (!J.dv.recs || n_total_ == 0) return;
  std::vector<DrawRec> recs(n_total_);
  std::vector<BinRec> bins(n_total_);
  std::vector<QuadExt> exts(n_ext_total_);
  FDH_HIP(hipMemcpy(recs.data(), J.dv.recs, recs.size() * sizeof(DrawRec), hipMemcpyDeviceToHost));
  FDH_HIP(hipMemcpy(bins.data(), J.dv.binrecs, bins.size() * sizeof(BinRec), hipMemcpyDeviceToHost));
  if (!exts.empty()) FDH_HIP(hipMemcpy(exts.data(), J.dv.exts, exts.size() * sizeof(QuadExt), hipMemcpyDeviceToHost));
  bool first = true;
  auto cmp = [&](int array, int lane_no, const void* dev, const void* host, size_t bytes, size_t dev_off) {
    const uint32_t* d = static_cast<const uint32_t*>(dev);
    const uint32_t* h = static_cast<const uint32_t*>(host);
    out[3] += (uint32_t)bytes;
    for (size_t i = 0; i < bytes / 4; i++) {
      if (d[i] == h[i]) continue;
      out[array] += 4;
      if (first) {
        first = false;
        out[4] = (uint32_t)array; out[5] = (uint32_t)(dev_off + 4 * i); out[6] = (uint32_t)lane_no; out[7] = d[i]; out[8] = h[i];
        uint32_t z = 0;
        for (size_t k = 0; k < bytes / 4; k++) z += d[k] == 0u;
        out[9] = z;
      }
    }
  };
  uint32_t r0 = 0, e0 = 0;
  for (const Piece& p : pieces_) {
    const Lane& L = lane(p.lane);
    std::vector<DrawRec> want(L.recs.p + p.first, L.recs.p + p.first + p.n);
    for (DrawRec& r : want) if (r.op_mode & F_GENERAL) r.ext = r.ext - p.ext_first + e0;
    cmp(0, p.lane, recs.data() + r0, want.data(), (size_t)p.n * sizeof(DrawRec), (size_t)r0 * sizeof(DrawRec));
    cmp(1, p.lane, bins.data() + r0, L.bins.p + p.first, (size_t)p.n * sizeof(BinRec), (size_t)r0 * sizeof(BinRec));
    if (p.n_ext) cmp(2, p.lane, exts.data() + e0, L.exts.p + p.ext_first, (size_t)p.n_ext * sizeof(QuadExt), (size_t)e0 * sizeof(QuadExt));
    r0 += p.n; e0 += p.n_ext;
  }
  // the block from the chunk boxes on (chunk boxes, phase table, blur weight tables as far as this frame staged them): out[12] bytes
  // that differ, out[13] first offset (in that block), out[14] device dword, out[15] host dword, out[16] bytes compared
  {
    const size_t o_misc = (size_t)(reinterpret_cast<const uint8_t*>(J.dv.chunkbox) - d_frame_.ptr);
    std::vector<uint8_t> dev(misc_host_.size());
    if (!dev.empty()) FDH_HIP(hipMemcpy(dev.data(), d_frame_.ptr + o_misc, dev.size(), hipMemcpyDeviceToHost));
    out[16] = (uint32_t)dev.size();
    for (size_t i = 0; i + 4 <= dev.size(); i += 4) {
      uint32_t a, b;
      std::memcpy(&a, dev.data() + i, 4); std::memcpy(&b, misc_host_.data() + i, 4);
      if (a == b) continue;
      if (!out[12]) { out[13] = (uint32_t)i; out[14] = a; out[15] = b; }
      out[12] += 4;
    }
  }
  // the bin boxes the upload kernel derives from the bin records: out[17] boxes that differ from the host's, out[18] first index,
  // out[19] device value, out[20] host value
  {
    std::vector<uint32_t> box(n_total_);
    FDH_HIP(hipMemcpy(box.data(), J.dv.binbox, box.size() * 4, hipMemcpyDeviceToHost));
    uint32_t g0 = 0;
    for (const Piece& p : pieces_) {
      const Lane& L = lane(p.lane);
      for (uint32_t i = 0; i < p.n; i++, g0++) {
        if (g0 == 0 && out[1]) continue;  // (a folded clear emptied record 0's box on the device side)
        if (box[g0] == L.boxes.p[p.first + i]) continue;
        if (!out[17]) { out[18] = g0; out[19] = box[g0]; out[20] = L.boxes.p[p.first + i]; }
        out[17]++;
      }
    }
  }
}

// Fault hunting (fdh_debug_bin_digest): what the bin kernel left for the frame last submitted -- per phase and bin the count and the
// list entries it covers.  out[0] = FNV-1a over them, out[1] = sum of the counts, out[2] = bins with count 0, out[3] = list entries
// whose first word is 0, out[4] = bins whose count exceeds the list stride (garbage).
void Context::debug_bin_digest(uint64_t out[8]) {
  need_device("debug_bin_digest");
  drain();
  FDH_HIP(hipSetDevice(device_));
  FDH_HIP(hipStreamSynchronize(stream_));
  for (int i = 0; i < 8; i++) out[i] = 0;
  const LaunchJob& J = job_;
  const size_t nb = (size_t)J.bins_x * J.bins_y, np = J.phases.size(), stride = (size_t)J.list_stride;
  if (!nb || !np || !J.counts || !J.lists) return;
  std::vector<uint32_t> counts(np * nb);
  std::vector<uint2> lists(np * nb * stride);
  FDH_HIP(hipMemcpy(counts.data(), J.counts, counts.size() * 4, hipMemcpyDeviceToHost));
  FDH_HIP(hipMemcpy(lists.data(), J.lists, lists.size() * sizeof(uint2), hipMemcpyDeviceToHost));
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](uint32_t v) { for (int k = 0; k < 4; k++) { h ^= (v >> (8 * k)) & 255u; h *= 1099511628211ull; } };
  for (size_t p = 0; p < np; p++) {
    const Phase& ph = J.phases[p];
    const bool whole = p == 0;
    for (int by = 0; by < J.bins_y; by++)
      for (int bx = 0; bx < J.bins_x; bx++) {
        if (!whole && (bx < ph.bin_x0 || bx >= ph.bin_x1 || by < ph.bin_y0 || by >= ph.bin_y1)) continue;  // (bins the phase's launches never look at)
        const size_t b = p * nb + (size_t)by * J.bins_x + bx;
        const uint32_t c = counts[b];
        mix(c);
        out[1] += c;
        if (c == 0) out[2]++;
        if (c > stride) { out[4]++; continue; }
        for (uint32_t e = 0; e < c; e++) { const uint2 v = lists[b * stride + e]; mix(v.x); mix(v.y); if (v.x == 0) out[3]++; }
      }
  }
  out[0] = h;
}

// A retained root's cached records take their place in lane 0 (fdh_scene_render): a memcpy per array, the extension indices
// moved to where the extensions landed, the list-stride count and the phase summary brought up to date.
void Context::splice_cached(const RetainedRoot& C) {
  Lane& L = lane(0);
  const uint32_t r0 = (uint32_t)L.recs.n, e0 = (uint32_t)L.exts.n;
  L.recs.append(C.recs.data(), C.recs.size());
  L.bins.append(C.bins.data(), C.bins.size());
  L.exts.append(C.exts.data(), C.exts.size());
  if (!C.exts.empty())
    for (size_t i = r0; i < L.recs.n; i++) if (L.recs[i].op_mode & F_GENERAL) L.recs[i].ext += e0;
  L.boxes.reserve(L.bins.n);
  for (size_t i = r0; i < L.bins.n; i++) { L.count_add(L.bins[i].box); L.boxes[i] = bin_box_of(L.bins[i].box, 6 + binbox_shift_); }
  L.boxes.n = L.bins.n;
  bbox_union(sum_.u, C.sum.u);
  sum_.has_masks = sum_.has_masks || C.sum.has_masks;
  sum_.has_atlas = sum_.has_atlas || C.sum.has_atlas;
  sum_.has_slow = sum_.has_slow || C.sum.has_slow;
  sum_.has_slow_atlas = sum_.has_slow_atlas || C.sum.has_slow_atlas;
  sum_.has_rot = sum_.has_rot || C.sum.has_rot;
  sum_.deepest = std::max(sum_.deepest, depth_now_ + C.sum.deepest);
  for (int k = 0; k < 4; k++) sum_.frag_mode[k] += C.sum.frag_mode[k];
  sum_.frag_ellip += C.sum.frag_ellip;
  sum_.frag_other += C.sum.frag_other;
  fragments_ += C.fragments;
  if (!C.recs.empty()) link_share(r0);  // the record in front of the splice may share its distance field with the first one here
}

}  // namespace fdh
