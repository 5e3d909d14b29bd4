// fdh_host.h -- small host-side helpers shared by the library's .cpp files.
#pragma once
#include <cmath>

#include "fdh_context.h"

namespace fdh {

inline float nim_round(float x) { return x >= 0.0f ? std::floor(x + 0.5f) : -std::floor(-x + 0.5f); }  // Nim math.round
inline float clampf(float x, float lo, float hi) { return !(x >= lo) ? lo : (x > hi ? hi : x); }  // (a NaN comes out as lo)
inline uint32_t pack_color(FdhColor c) { return (uint32_t)c.r | ((uint32_t)c.g << 8) | ((uint32_t)c.b << 16) | ((uint32_t)c.a << 24); }

inline Aff aff_mul(const Aff& m, const Aff& n) {
  Aff r;
  r.a = m.a * n.a + m.c * n.b;
  r.b = m.b * n.a + m.d * n.b;
  r.c = m.a * n.c + m.c * n.d;
  r.d = m.b * n.c + m.d * n.d;
  r.tx = m.a * n.tx + m.c * n.ty + m.tx;
  r.ty = m.b * n.tx + m.d * n.ty + m.ty;
  return r;
}

inline bool bbox_empty(const BBox& b) { return b.x1 <= b.x0 || b.y1 <= b.y0; }
inline void bbox_union(BBox& a, const BBox& b) {
  if (bbox_empty(b)) return;
  if (bbox_empty(a)) { a = b; return; }
  a.x0 = std::min(a.x0, b.x0); a.y0 = std::min(a.y0, b.y0); a.x1 = std::max(a.x1, b.x1); a.y1 = std::max(a.y1, b.y1);
}
BlurTaps make_taps(float blur_radius);
uint32_t bin_box_of(const BBox& b, int shift);
FdhColor sample_fill(const FdhFill& f, float t);
void gradient_colors(const FdhFill& f, FdhColor out[4]);
constexpr int64_t kRectImageKey = 0x7265637452454354LL;  // the 4x4 white image drawRect / drawFilledQuad sample (glcontext.nim:966-970)

}  // namespace fdh
