// fdh_frontend.cpp -- the renderer front-end: node tree -> backend calls.
//
// Restates figrender.nim's `renderFrame` (:1960-1995), `renderRoot` (:1946-1955) and the recursive
// `render` (:1756-1839) whose stage order is fixed by the `renderStages` macro (:501-547): stages run
// top-down, their `finally` blocks run in reverse after the children.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "fdh_context.h"
#include "fdh_host.h"
#include "fdh_walkpool.h"

namespace fdh {

// Who is whose child, found once per layer: first_child[i] / next_sibling[i] are the nodes whose parent is i, in index order (a
// parent precedes its children: a node that names a later node, or none, is nobody's child -- exactly what the reference's forward
// scan finds, childIndex fignodes.nim:165-177); blur_below[i]: a backdrop-blur node sits in i's subtree (i included) -- such a
// subtree splits the frame into phases and is always walked by the calling thread.
struct LayerLinks {
  const FdhLayer* layer = nullptr;
  std::vector<int> first_child, next_sibling, last_child;
  std::vector<uint8_t> blur_below;
  void build(const FdhLayer& L) {
    if (layer == &L) return;
    layer = &L;
    const size_t n = (size_t)std::max(L.n_nodes, 0);
    first_child.assign(n, -1); next_sibling.assign(n, -1); last_child.assign(n, -1);
    blur_below.assign(n, 0);
    bool any_blur = false;
    for (size_t i = 0; i < n; i++) {
      if (L.nodes[i].kind == FDH_NK_BACKDROP_BLUR) { blur_below[i] = 1; any_blur = true; }
      const int p = L.nodes[i].parent;
      if (p < 0 || (size_t)p >= i) continue;
      if (last_child[(size_t)p] < 0) first_child[(size_t)p] = (int)i; else next_sibling[(size_t)last_child[(size_t)p]] = (int)i;
      last_child[(size_t)p] = (int)i;
    }
    for (size_t i = n; any_blur && i-- > 0;) {  // (a second walk over the node array only where a blur node has to mark its ancestors)
      const int p = L.nodes[i].parent;
      if (blur_below[i] && p >= 0 && (size_t)p < i) blur_below[(size_t)p] = 1;
    }
  }
};

struct Walker;
struct ParallelWalk {  // (a friend of Context: the pool threads' recorders and lanes are the context's)
  static bool group(Walker& main, const FdhLayer& L, const int* items, int n);
};

struct Walker {
  Recorder& ctx;
  const FdhScene& scene;
  float ui;
  LayerLinks* links;       // shared by the calling thread's walker and the pool threads' (built by the former, before a group starts)
  bool can_fork = false;   // the calling thread's walker outside any group: large sibling groups go to the pool

  float scaled(float v) const { return v * ui; }  // common/shared.nim:94-95

  static uint8_t fill_alpha_max(const FdhFill& f) {  // figrender.nim:587-594
    if (f.kind == FDH_FILL_COLOR) return f.start.a;
    if (f.kind == FDH_FILL_LINEAR2) return std::max(f.start.a, f.stop.a);
    return std::max(f.start.a, std::max(f.mid.a, f.stop.a));
  }

  void radii(const FdhFig& n, float rx[4], float ry[4]) const {  // resolvedCorners + scaledCorners :549-571
    for (int i = 0; i < 4; i++) {
      rx[i] = scaled((float)n.corners[i]);
      ry[i] = (n.flags & FDH_NF_ELLIPTICAL_CORNERS) ? scaled((float)n.corner_radii_y[i]) : rx[i];
    }
  }
  void box_of(const FdhFig& n, float b[4]) const { for (int i = 0; i < 4; i++) b[i] = n.box[i] * ui; }

  void drop_shadows(const FdhFig& n) {  // renderDropShadows :654-689
    for (const FdhShadow& sh : n.shadows) {
      if (sh.style != FDH_SHADOW_DROP) continue;
      if (sh.blur <= 0.0f && sh.spread <= 0.0f) continue;
      if (fill_alpha_max(sh.fill) == 0) continue;
      float box[4], rx[4], ry[4];
      box_of(n, box);
      radii(n, rx, ry);
      const float sx = scaled(sh.x), sy = scaled(sh.y), sb = scaled(sh.blur), ss = scaled(sh.spread);
      auto nround = [](float x) { return x >= 0.0f ? std::floor(x + 0.5f) : -std::floor(-x + 0.5f); };
      const float pad = std::max(nround(ss) + nround(1.5f * sb), 0.0f);
      const float quad[4] = {box[0] + sx - pad, box[1] + sy - pad, box[2] + 2.0f * pad, box[3] + 2.0f * pad};
      const float shape[2] = {box[2], box[3]};
      ctx.draw_rounded_rect_fill(quad, sh.fill, rx, ry, FDH_SDF_DROP_SHADOW, sb, ss, shape);
    }
  }
  void inner_shadows(const FdhFig& n) {  // renderInnerShadows :716-744 (hasActiveInnerShadow :778-789 = same skip rules)
    for (const FdhShadow& sh : n.shadows) {
      if (sh.style != FDH_SHADOW_INNER) continue;
      if (sh.blur <= 0.0f && sh.spread <= 0.0f) continue;
      if (fill_alpha_max(sh.fill) == 0) continue;
      float box[4], rx[4], ry[4];
      box_of(n, box);
      radii(n, rx, ry);
      const float off[2] = {scaled(sh.x), scaled(sh.y)};
      ctx.draw_rounded_rect_fill(box, sh.fill, rx, ry, FDH_SDF_INSET_SHADOW, scaled(sh.blur), scaled(sh.spread), off);
    }
  }
  void rounded_shape(const FdhFig& n, const FdhFill& fill, const FdhStroke* stroke) {  // renderRoundedShapeScaledCorners :806-873
    float box[4], rx[4], ry[4];
    box_of(n, box);
    radii(n, rx, ry);
    const float shape[2] = {0, 0};
    const bool gradient = (fill.kind == FDH_FILL_LINEAR2 || fill.kind == FDH_FILL_LINEAR3) && fill_alpha_max(fill) > 0;
    if (gradient) {
      ctx.draw_rounded_rect_fill(box, fill, rx, ry, FDH_SDF_CLIP_AA, 4.0f, 0.0f, shape);
    } else if (fill_alpha_max(fill) > 0) {
      FdhFill solid = fill;
      solid.kind = FDH_FILL_COLOR;
      solid.start = sample_fill(fill, 0.5f);  // fillCenterColor
      ctx.draw_rounded_rect_fill(box, solid, rx, ry, FDH_SDF_CLIP_AA, 4.0f, 0.0f, shape);
    }
    if (stroke && fill_alpha_max(stroke->fill) > 0 && stroke->weight > 0.0f)
      ctx.draw_rounded_rect_fill(box, stroke->fill, rx, ry, FDH_SDF_ANNULAR_AA, scaled(stroke->weight), 0.0f, shape);
  }


  // ------------------------------------------------------------------ nkDrawable (figrender.nim:910-1667)
  struct V2 { float x, y; };
  struct Span { V2 p0, p1, p2; };
  static V2 add(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
  static V2 sub(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
  static V2 mul(V2 a, float k) { return {a.x * k, a.y * k}; }
  static float len(V2 v) { return std::sqrt(v.x * v.x + v.y * v.y); }
  static V2 normalized_or(V2 v, V2 fb) { const float l = len(v); return l <= 0.000001f ? fb : V2{v.x / l, v.y / l}; }
  static V2 normal_left(V2 d) { return {-d.y, d.x}; }
  static float cross2(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
  static float nround(float x) { return x >= 0.0f ? std::floor(x + 0.5f) : -std::floor(-x + 0.5f); }
  static uint16_t radius_corner(float r) {  // :797-802
    if (r <= 0.0f) return 0;
    if (r >= 65535.0f) return 65535;
    return (uint16_t)nround(r);
  }
  static constexpr int kMaxAdaptiveSteps = 192;  // max(DefaultDrawableBezierSteps * 4, 64)
  static constexpr int kMaxCurveDepth = 8;

  // renderRoundedShape on an explicit (unscaled) box: fill + optional stroke
  void shape(const float box_unscaled[4], const FdhFill& fill, const FdhStroke* stroke, const float rx[4], const float ry[4]) {
    float box[4];
    for (int i = 0; i < 4; i++) box[i] = box_unscaled[i] * ui;
    const float shp[2] = {0, 0};
    const bool gradient = (fill.kind == FDH_FILL_LINEAR2 || fill.kind == FDH_FILL_LINEAR3) && fill_alpha_max(fill) > 0;
    if (gradient) {
      ctx.draw_rounded_rect_fill(box, fill, rx, ry, FDH_SDF_CLIP_AA, 4.0f, 0.0f, shp);
    } else if (fill_alpha_max(fill) > 0) {
      FdhFill solid = fill;
      solid.kind = FDH_FILL_COLOR;
      solid.start = sample_fill(fill, 0.5f);
      ctx.draw_rounded_rect_fill(box, solid, rx, ry, FDH_SDF_CLIP_AA, 4.0f, 0.0f, shp);
    }
    if (stroke && fill_alpha_max(stroke->fill) > 0 && stroke->weight > 0.0f)
      ctx.draw_rounded_rect_fill(box, stroke->fill, rx, ry, FDH_SDF_ANNULAR_AA, scaled(stroke->weight), 0.0f, shp);
  }
  void shape_u16(const float box[4], const FdhFill& fill, const FdhStroke* stroke, const uint16_t corners[4]) {
    float rx[4];
    for (int i = 0; i < 4; i++) rx[i] = scaled((float)corners[i]);
    shape(box, fill, stroke, rx, rx);
  }
  void stroke_cap(V2 center, float radius, const FdhFill& fill) {  // renderDrawableStrokeCap :993-1005
    if (radius <= 0.0f || fill_alpha_max(fill) == 0) return;
    const float box[4] = {center.x - radius, center.y - radius, radius * 2.0f, radius * 2.0f};
    const uint16_t rc = radius_corner(radius);
    const uint16_t corners[4] = {rc, rc, rc, rc};
    shape_u16(box, fill, nullptr, corners);
  }
  void line(V2 origin, V2 pa, V2 pb, const FdhStroke& stroke) {  // renderDrawableLine :943-991
    const float weight = std::max(0.0f, stroke.weight);
    if (weight <= 0.0f || fill_alpha_max(stroke.fill) == 0) return;
    const V2 a = add(origin, pa), b = add(origin, pb), delta = sub(b, a);
    const float length = len(delta);
    if (length <= 0.0f) return;
    const int cap = stroke.cap == FDH_CAP_AUTO ? FDH_CAP_BUTT : stroke.cap;
    const float cap_radius = weight * 0.5f;
    const V2 dir{delta.x / length, delta.y / length};
    V2 da = a, db = b;
    float dl = length;
    if (cap == FDH_CAP_SQUARE) { da = sub(a, mul(dir, cap_radius)); db = add(b, mul(dir, cap_radius)); dl = length + weight; }
    const V2 center{(da.x + db.x) / 2.0f, (da.y + db.y) / 2.0f};
    const float box[4] = {center.x - dl / 2.0f, center.y - weight / 2.0f, dl, weight};
    const float pvx = box[0] * ui + box[2] * ui / 2.0f, pvy = box[1] * ui + box[3] * ui / 2.0f;
    ctx.save_transform();
    ctx.translate(pvx, pvy);
    ctx.rotate(std::atan2(delta.y, delta.x));
    ctx.translate(-pvx, -pvy);
    const uint16_t zero[4] = {0, 0, 0, 0};
    shape_u16(box, stroke.fill, nullptr, zero);
    ctx.restore_transform();
    if (cap == FDH_CAP_ROUND) { stroke_cap(a, cap_radius, stroke.fill); stroke_cap(b, cap_radius, stroke.fill); }
  }
  void endpoint_cap(V2 origin, V2 point, V2 tangent, float radius, const FdhStroke& stroke, int cap, bool is_start) {  // :1007-1040
    if (radius <= 0.0f || fill_alpha_max(stroke.fill) == 0) return;
    if (cap == FDH_CAP_ROUND) { stroke_cap(add(origin, point), radius, stroke.fill); return; }
    if (cap == FDH_CAP_SQUARE) {
      const V2 dir = normalized_or(tangent, {1.0f, 0.0f});
      const V2 a = is_start ? sub(point, mul(dir, radius)) : point;
      const V2 b = is_start ? point : add(point, mul(dir, radius));
      FdhStroke s2 = stroke;
      s2.cap = FDH_CAP_BUTT;
      line(origin, a, b, s2);
    }
  }
  void filled_quad(const V2 v[4], const FdhFill& fill) {  // renderDrawableFilledQuad :1050-1058
    if (fill_alpha_max(fill) == 0) return;
    const FdhColor k = sample_fill(fill, 0.5f);
    const FdhColor cols[4] = {k, k, k, k};
    float vv[8];
    for (int i = 0; i < 4; i++) { vv[2 * i] = v[i].x * ui; vv[2 * i + 1] = v[i].y * ui; }
    ctx.draw_filled_quad(vv, cols);
  }
  void stroke_join(V2 origin, V2 point, V2 in_t, V2 out_t, float radius, const FdhFill& fill, int join) {  // :1060-1109
    if (radius <= 0.0f || fill_alpha_max(fill) == 0) return;
    if (join == FDH_JOIN_ROUND) { stroke_cap(add(origin, point), radius, fill); return; }
    if (join != FDH_JOIN_BEVEL && join != FDH_JOIN_MITER) return;
    const V2 incoming = normalized_or(in_t, {1.0f, 0.0f});
    const V2 outgoing = normalized_or(out_t, incoming);
    const float turn = cross2(incoming, outgoing);
    if (std::fabs(turn) <= 0.0001f) return;
    const float side = turn > 0.0f ? -1.0f : 1.0f;
    const V2 in_outer = add(point, mul(normal_left(incoming), radius * side));
    const V2 out_outer = add(point, mul(normal_left(outgoing), radius * side));
    if (join == FDH_JOIN_MITER) {
      const float denom = cross2(incoming, outgoing);  // lineIntersection :1042-1048
      if (std::fabs(denom) > 0.000001f) {
        const float t = cross2(sub(out_outer, in_outer), outgoing) / denom;
        const V2 miter = add(in_outer, mul(incoming, t));
        if (len(sub(miter, point)) <= radius * 4.0f) {
          const V2 q[4] = {add(origin, point), add(origin, in_outer), add(origin, miter), add(origin, out_outer)};
          filled_quad(q, fill);
          return;
        }
      }
    }
    const V2 q[4] = {add(origin, point), add(origin, in_outer), add(origin, out_outer), add(origin, out_outer)};
    filled_quad(q, fill);
  }
  static V2 bezier_point(const float* ctrl, int n, float t) {  // de Casteljau, bezierPoint :1139-1153
    V2 work[32];
    if (n <= 0) return {0, 0};
    n = std::min(n, 32);
    for (int i = 0; i < n; i++) work[i] = {ctrl[2 * i], ctrl[2 * i + 1]};
    for (int count = n; count > 1; count--)
      for (int i = 0; i < count - 1; i++) work[i] = add(mul(work[i], 1.0f - t), mul(work[i + 1], t));
    return work[0];
  }
  static V2 quadratic_point(V2 p0, V2 p1, V2 p2, float t) {
    const float it = 1.0f - t;
    return add(add(mul(p0, it * it), mul(p1, 2.0f * it * t)), mul(p2, t * t));
  }
  static void quadratic_bounds(V2 p0, V2 p1, V2 p2, float pad, float out[4]) {  // :1177-1202
    V2 mn{std::min(p0.x, p2.x), std::min(p0.y, p2.y)}, mx{std::max(p0.x, p2.x), std::max(p0.y, p2.y)};
    auto include = [&](V2 q) { mn.x = std::min(mn.x, q.x); mn.y = std::min(mn.y, q.y); mx.x = std::max(mx.x, q.x); mx.y = std::max(mx.y, q.y); };
    const float dx = p0.x - 2.0f * p1.x + p2.x;
    if (std::fabs(dx) > 0.000001f) { const float t = (p0.x - p1.x) / dx; if (t > 0.0f && t < 1.0f) include(quadratic_point(p0, p1, p2, t)); }
    const float dy = p0.y - 2.0f * p1.y + p2.y;
    if (std::fabs(dy) > 0.000001f) { const float t = (p0.y - p1.y) / dy; if (t > 0.0f && t < 1.0f) include(quadratic_point(p0, p1, p2, t)); }
    out[0] = mn.x - pad; out[1] = mn.y - pad; out[2] = mx.x - mn.x + pad * 2.0f; out[3] = mx.y - mn.y + pad * 2.0f;
  }
  static V2 start_tangent(const Span& s) { return normalized_or(sub(s.p1, s.p0), normalized_or(sub(s.p2, s.p0), {1.0f, 0.0f})); }
  static V2 end_tangent(const Span& s) { return normalized_or(sub(s.p2, s.p1), normalized_or(sub(s.p2, s.p0), {1.0f, 0.0f})); }
  static Span bezier_span(const float* ctrl, int n, float t0, float t2) {  // bezierQuadraticSpan :1238-1247
    const float tm = (t0 + t2) * 0.5f;
    Span s;
    s.p0 = bezier_point(ctrl, n, t0);
    const V2 pm = bezier_point(ctrl, n, tm);
    s.p2 = bezier_point(ctrl, n, t2);
    s.p1 = sub(mul(pm, 2.0f), mul(add(s.p0, s.p2), 0.5f));
    return s;
  }
  void adaptive_spans(const float* ctrl, int n, float t0, float t2, int depth, std::vector<Span>& spans) {  // :1267-1282
    const Span s = bezier_span(ctrl, n, t0, t2);
    float err = 0.0f;
    for (float lt : {0.25f, 0.75f}) {  // quadraticApproxErrorPx :1256-1265
      const float t = t0 + (t2 - t0) * lt;
      err = std::max(err, len(mul(sub(bezier_point(ctrl, n, t), quadratic_point(s.p0, s.p1, s.p2, lt)), ui)));
    }
    if (err <= 0.5f || depth >= kMaxCurveDepth || (int)spans.size() >= kMaxAdaptiveSteps - 1) { spans.push_back(s); return; }
    const float tm = (t0 + t2) * 0.5f;
    adaptive_spans(ctrl, n, t0, tm, depth + 1, spans);
    adaptive_spans(ctrl, n, tm, t2, depth + 1, spans);
  }
  void quadratic_sdf(V2 origin, V2 p0, V2 p1, V2 p2, const FdhStroke& stroke, int cap) {  // renderDrawableQuadraticBezierSdf :1335-1376
    const int rcap = cap == FDH_CAP_AUTO ? (stroke.cap == FDH_CAP_AUTO ? FDH_CAP_ROUND : stroke.cap) : cap;
    if (std::fabs(cross2(sub(p1, p0), sub(p2, p1))) <= 0.0001f) {  // isFlatQuadratic
      FdhStroke s2 = stroke;
      s2.cap = rcap;
      line(origin, p0, p2, s2);
      return;
    }
    const float sw = std::max(0.0f, stroke.weight);
    const float padding = sw * 0.5f + 2.0f / ui;  // DrawableSdfPaddingPx.descaled()
    const V2 a = add(origin, p0), b = add(origin, p1), c = add(origin, p2);
    float box[4];
    quadratic_bounds(a, b, c, padding, box);
    if (box[2] <= 0.0f || box[3] <= 0.0f) return;
    const V2 center{box[0] + box[2] * 0.5f, box[1] + box[3] * 0.5f};
    const float la[2] = {(a.x - center.x) * ui, (a.y - center.y) * ui}, lb[2] = {(b.x - center.x) * ui, (b.y - center.y) * ui};
    const float lc[2] = {(c.x - center.x) * ui, (c.y - center.y) * ui};
    const float sbox[4] = {box[0] * ui, box[1] * ui, box[2] * ui, box[3] * ui};
    ctx.draw_quadratic_bezier_sdf(sbox, stroke.fill, la, lb, lc, sw * ui, rcap);
  }
  void draw_spans(V2 origin, const std::vector<Span>& spans, const FdhStroke& stroke) {  // :1412-1455, :1548-1604
    const int cap = stroke.cap == FDH_CAP_AUTO ? FDH_CAP_ROUND : stroke.cap;
    const int join = stroke.join == FDH_JOIN_AUTO ? FDH_JOIN_ROUND : stroke.join;
    const bool simple = cap == FDH_CAP_ROUND && join == FDH_JOIN_ROUND;
    const int span_cap = simple ? FDH_CAP_ROUND : FDH_CAP_BUTT;
    const float cap_radius = std::max(0.0f, stroke.weight) / 2.0f;
    const int n = (int)spans.size();
    for (int i = 0; i < n; i++) {
      quadratic_sdf(origin, spans[i].p0, spans[i].p1, spans[i].p2, stroke, span_cap);
      if (simple) continue;
      if (i == 0) endpoint_cap(origin, spans[i].p0, start_tangent(spans[i]), cap_radius, stroke, cap, true);
      else stroke_join(origin, spans[i].p0, end_tangent(spans[i - 1]), start_tangent(spans[i]), cap_radius, stroke.fill, join);
      if (i == n - 1) endpoint_cap(origin, spans[i].p2, end_tangent(spans[i]), cap_radius, stroke, cap, false);
    }
  }
  static int explicit_steps(uint16_t steps, uint16_t node_steps) {  // :1204-1210
    if (steps != 0) return std::max<int>(1, steps);
    if (node_steps != 0) return std::max<int>(1, node_steps);
    return 0;
  }
  void segment_points(const float* ctrl, int n, float t0, float t2, int depth, std::vector<V2>& pts) {  // :1297-1313
    const V2 p0 = bezier_point(ctrl, n, t0), p2 = bezier_point(ctrl, n, t2);
    const float tm = (t0 + t2) * 0.5f;
    const V2 P = mul(bezier_point(ctrl, n, tm), ui), A = mul(p0, ui), B = mul(p2, ui), ab = sub(B, A);
    const float denom = ab.x * ab.x + ab.y * ab.y;
    float err;
    if (denom <= 0.000001f) err = len(sub(P, A));
    else {
      const float h = std::min(std::max(((P.x - A.x) * ab.x + (P.y - A.y) * ab.y) / denom, 0.0f), 1.0f);
      err = len(sub(P, add(A, mul(ab, h))));
    }
    if (err <= 0.5f || depth >= kMaxCurveDepth || (int)pts.size() >= kMaxAdaptiveSteps) { pts.push_back(p2); return; }
    segment_points(ctrl, n, t0, tm, depth + 1, pts);
    segment_points(ctrl, n, tm, t2, depth + 1, pts);
  }
  void bezier_segments(V2 origin, const float* ctrl, int n, uint16_t steps, const FdhStroke& stroke, uint16_t node_steps) {  // :1378-1410
    if (n < 2 || stroke.weight <= 0.0f || fill_alpha_max(stroke.fill) == 0) return;
    std::vector<V2> pts;
    const int fixed = explicit_steps(steps, node_steps);
    pts.push_back(bezier_point(ctrl, n, 0.0f));
    if (fixed > 0) for (int s = 1; s <= fixed; s++) pts.push_back(bezier_point(ctrl, n, (float)s / (float)fixed));
    else segment_points(ctrl, n, 0.0f, 1.0f, 0, pts);
    if (pts.size() < 2) return;
    const int cap = stroke.cap == FDH_CAP_AUTO ? FDH_CAP_ROUND : stroke.cap;
    const int join = stroke.join == FDH_JOIN_AUTO ? FDH_JOIN_ROUND : stroke.join;
    const float cap_radius = std::max(0.0f, stroke.weight) / 2.0f;
    FdhStroke seg = stroke;
    seg.cap = FDH_CAP_BUTT;
    V2 prev = pts[0], prev_t{1.0f, 0.0f};
    for (size_t s = 1; s < pts.size(); s++) {
      const V2 cur = pts[s], tan = sub(cur, prev);
      line(origin, prev, cur, seg);
      if (s == 1) endpoint_cap(origin, prev, tan, cap_radius, stroke, cap, true);
      else stroke_join(origin, prev, prev_t, tan, cap_radius, stroke.fill, join);
      if (s == pts.size() - 1) endpoint_cap(origin, cur, tan, cap_radius, stroke, cap, false);
      prev = cur;
      prev_t = tan;
    }
  }
  void drawable_ops(const FdhFig& n) {  // renderDrawableOps :1627-1645
    const V2 origin{n.box[0], n.box[1]};
    const FdhStroke& stroke = n.draw_stroke;
    // (ranges into the scene's side arrays are clipped to them: a negative start or a count past the end reads nothing it should not)
    for (long long oi = std::max(n.op_first, 0); scene.ops && oi < (long long)n.op_first + n.op_count && oi < scene.n_ops; oi++) {
      const FdhDrawOp& op = scene.ops[oi];
      switch (op.kind) {
        case FDH_DK_LINE: line(origin, {op.v[0], op.v[1]}, {op.v[2], op.v[3]}, stroke); break;
        case FDH_DK_CIRCLE: {  // :1111-1125
          const float r = std::max(0.0f, op.v[2]);
          if (r <= 0.0f) break;
          const float box[4] = {origin.x + op.v[0] - r, origin.y + op.v[1] - r, r * 2.0f, r * 2.0f};
          const uint16_t rc = radius_corner(r);
          const uint16_t corners[4] = {rc, rc, rc, rc};
          shape_u16(box, n.fill, &stroke, corners);
          break;
        }
        case FDH_DK_RECTANGLE: {  // :1127-1131
          const float box[4] = {origin.x + op.v[0], origin.y + op.v[1], op.v[2], op.v[3]};
          shape_u16(box, n.fill, &stroke, op.corners);
          break;
        }
        case FDH_DK_ELLIPSE: {  // :1606-1625
          const float rx0 = std::max(0.0f, op.v[2]), ry0 = std::max(0.0f, op.v[3]);
          if (rx0 <= 0.0f || ry0 <= 0.0f) break;
          const float box[4] = {origin.x + op.v[0] - rx0, origin.y + op.v[1] - ry0, rx0 * 2.0f, ry0 * 2.0f};
          float rx[4], ry[4];
          for (int i = 0; i < 4; i++) { rx[i] = scaled(rx0); ry[i] = scaled(ry0); }
          shape(box, n.fill, &stroke, rx, ry);
          break;
        }
        case FDH_DK_BEZIER: {  // renderDrawableBezier :1457-1486
          const int nc = op.ctrl_count;
          if (nc < 2 || !scene.controls || op.ctrl_first < 0 || (long long)op.ctrl_first + nc > scene.n_controls) break;
          const float* ctrl = scene.controls + 2 * op.ctrl_first;
          if (stroke.weight <= 0.0f || fill_alpha_max(stroke.fill) == 0) break;
          if (nc == 3) {
            quadratic_sdf(origin, {ctrl[0], ctrl[1]}, {ctrl[2], ctrl[3]}, {ctrl[4], ctrl[5]}, stroke,
                          stroke.cap == FDH_CAP_AUTO ? FDH_CAP_ROUND : stroke.cap);
          } else if (nc > 3) {
            std::vector<Span> spans;
            const int fixed = explicit_steps(op.steps, n.draw_steps);
            if (fixed > 0) for (int st = 0; st < fixed; st++) spans.push_back(bezier_span(ctrl, nc, (float)st / (float)fixed, (float)(st + 1) / (float)fixed));
            else adaptive_spans(ctrl, nc, 0.0f, 1.0f, 0, spans);
            draw_spans(origin, spans, stroke);
          } else {
            bezier_segments(origin, ctrl, nc, op.steps, stroke, n.draw_steps);
          }
          break;
        }
        case FDH_DK_ARC: {  // renderDrawableArc :1606-1625, arcQuadraticSpan :1531-1546, adaptiveArcStepCount :1315-1333
          const float radius = std::max(0.0f, op.v[2]), start = op.v[3], sweep = op.v[4];
          if (radius <= 0.0f || sweep == 0.0f || stroke.weight <= 0.0f || fill_alpha_max(stroke.fill) == 0) break;
          int steps = explicit_steps(op.steps, n.draw_steps);
          if (steps <= 0) {
            const float rpx = std::max(0.0f, scaled(radius)), asw = std::fabs(sweep);
            if (rpx <= 0.0f || asw <= 0.0f) steps = 1;
            else {
              const float cl = std::min(std::max(1.0f - 0.5f / rpx, -1.0f), 1.0f);
              const float max_angle = std::max(0.01f, 2.0f * std::acos(cl));
              steps = std::min(std::max((int)std::ceil(asw / max_angle), 1), kMaxAdaptiveSteps);
            }
          }
          std::vector<Span> spans;
          const V2 cen{op.v[0], op.v[1]};
          for (int st = 0; st < steps; st++) {
            const float t0 = (float)st / (float)steps, t2 = (float)(st + 1) / (float)steps, tm = (t0 + t2) * 0.5f;
            const float a0 = start + sweep * t0, a2 = start + sweep * t2, am = start + sweep * tm;
            Span sp;
            sp.p0 = add(cen, {std::cos(a0) * radius, std::sin(a0) * radius});
            const V2 pm = add(cen, {std::cos(am) * radius, std::sin(am) * radius});
            sp.p2 = add(cen, {std::cos(a2) * radius, std::sin(a2) * radius});
            sp.p1 = sub(mul(pm, 2.0f), mul(add(sp.p0, sp.p2), 0.5f));
            spans.push_back(sp);
          }
          draw_spans(origin, spans, stroke);
          break;
        }
        default: break;
      }
    }
  }
  void drawable(const FdhFig& n) {  // renderDrawable :1647-1667
    if (n.draw_aa <= 0.0f || ctx.aa() == n.draw_aa) { drawable_ops(n); return; }
    const float old = ctx.aa();
    ctx.set_aa(n.draw_aa);
    drawable_ops(n);
    ctx.set_aa(old);
  }

  void text_rect(float x, float y, float w, float h, const FdhFill& f) {  // figrender.nim:355-369, 444-452
    const float rect[4] = {scaled(x), scaled(y), scaled(w), scaled(h)};
    const float zero[4] = {0, 0, 0, 0}, shape[2] = {0, 0};
    ctx.draw_rounded_rect_fill(rect, f, zero, zero, FDH_SDF_CLIP_AA, 4.0f, 0.0f, shape);
  }
  void text(const FdhFig& n) {  // renderText :417-497 (typesetting happened on the caller's side)
    ctx.save_transform();
    ctx.translate(scaled(n.box[0]), scaled(n.box[1]));
    if (n.flags & FDH_NF_INVERT_Y) {
      ctx.translate(0.0f, scaled(n.box[3]));
      ctx.scale(1.0f, -1.0f);
    }
    // selection rectangles first (:435-452), then underline / strikethrough (:371-415), then the glyphs
    for (int pass = 0; pass < 2; pass++) {
      for (long long k = std::max(n.text_rect_first, 0); scene.text_rects && k < (long long)n.text_rect_first + n.text_rect_count && k < scene.n_text_rects; k++) {
        const FdhTextRect& tr = scene.text_rects[k];
        if (tr.kind != pass) continue;
        if (pass == 0) {
          if (!(n.flags & FDH_NF_SELECT_TEXT) || fill_alpha_max(n.fill) == 0 || !(tr.h > 0.0f)) continue;
          text_rect(tr.x, tr.y, std::max(tr.w, 1.0f), tr.h, n.fill);
        } else {
          if (tr.w <= 0.0f || tr.h <= 0.0f) continue;
          text_rect(tr.x, tr.y, tr.w, tr.h, tr.fill);
        }
      }
    }
    for (long long g = std::max(n.glyph_first, 0); scene.glyphs && g < (long long)n.glyph_first + n.glyph_count && g < scene.n_glyphs; g++) {
      const FdhGlyph& gl = scene.glyphs[g];
      float pos[2] = {gl.x, gl.y};
      const float size[2] = {0, 0};
      int64_t key = gl.image_id;
      float shift = gl.subpixel_shift;
      if (shift < 0.0f) {  // figrender.nim:464-471
        shift = 0.0f;
        if (ctx.subpixel_enabled()) {
          const float snapped = std::floor(pos[0]);
          const float frac = std::max(0.0f, std::min(pos[0] - snapped, 0.999f));
          pos[0] = snapped;
          if (ctx.subpixel_variants() && scene.glyph_variant_ids) {  // toGlyphVariantSubpixelStep common/fontglyphs.nim:50-52
            const int step = std::min((int)(frac * (float)FDH_GLYPH_VARIANT_STEPS), FDH_GLYPH_VARIANT_STEPS - 1);
            key = scene.glyph_variant_ids[(size_t)g * FDH_GLYPH_VARIANT_STEPS + step];
          } else {
            shift = frac;
          }
        }
      }
      ctx.set_subpixel_shift(shift);
      ctx.draw_image(key, pos, gl.colors, size, false);
    }
    ctx.set_subpixel_shift(0.0f);
    ctx.restore_transform();
  }

  int depth = 0;
  int64_t culled_subtrees = 0;
  bool has_blur(const FdhLayer& L, int idx) {
    const FdhFig& n = L.nodes[idx];
    if (n.kind == FDH_NK_BACKDROP_BLUR) return true;
    if (n.child_count <= 0) return false;
    links->build(L);
    return links->blur_below[(size_t)idx] != 0;
  }
  // A run of siblings in painter's order (the roots of a layer, the children of a node).  Runs of at least kForkMin nodes
  // without a blur node below them go to the walk pool when this is the calling thread's walker (ParallelWalk::group); everything
  // else is walked here, one node after the other, exactly as the reference does.
  static constexpr int kForkMin = 48;
  // the next sibling's node (488 bytes, eight cache lines) on its way while this one is decomposed: a tree the application has
  // just rebuilt sits in ITS core's cache, and a pool thread pays a cross-core miss per line it has not asked for in advance
  static void prefetch_node(const FdhLayer& L, int idx) {
    const char* p = reinterpret_cast<const char*>(&L.nodes[idx]);
    for (size_t o = 0; o < sizeof(FdhFig); o += 64) __builtin_prefetch(p + o, 0, 3);
  }
  void siblings(const FdhLayer& L, const int* items, int n) {
    if (!can_fork || n < kForkMin) { for (int k = 0; k < n; k++) { if (k + 1 < n) prefetch_node(L, items[k + 1]); node(L, items[k]); } return; }
    int i = 0;
    while (i < n) {
      if (has_blur(L, items[i])) { node(L, items[i]); i++; continue; }
      int j = i;
      while (j < n && !has_blur(L, items[j])) j++;
      if (j - i < kForkMin || !ParallelWalk::group(*this, L, items + i, j - i))
        for (int k = i; k < j; k++) node(L, items[k]);
      i = j;
    }
  }
  void node(const FdhLayer& L, int idx) {
    const FdhFig& n = L.nodes[idx];
    if (n.flags & FDH_NF_DISABLE_RENDER) return;
    // (the walk recurses like the reference's: a chain of thousands of only children would run the thread's stack out)
    if (depth >= 2048) throw Error(FDH_ERR_UNSUPPORTED, "scene: nodes nested deeper than 2048");
    struct Deeper { int& d; explicit Deeper(int& x) : d(x) { d++; } ~Deeper() { d--; } } deeper(depth);
    float box[4], rx[4], ry[4];
    box_of(n, box);
    radii(n, rx, ry);
    const bool rot = n.rotation != 0.0f, xf = n.kind == FDH_NK_TRANSFORM;
    const bool clip = (n.flags & FDH_NF_CLIP_CONTENT) != 0, rmask = (n.flags & FDH_NF_RECT_MASK_CONTENT) != 0;
    if (rot) {
      ctx.save_transform();
      const float cx = box[0] + box[2] / 2.0f, cy = box[1] + box[3] / 2.0f;
      ctx.translate(cx, cy);
      ctx.rotate(n.rotation / 180.0f * 3.14159265358979323846f);
      ctx.translate(-cx, -cy);
    }
    if (xf) {
      ctx.save_transform();
      if (n.translation[0] != 0.0f || n.translation[1] != 0.0f) ctx.translate(scaled(n.translation[0]), scaled(n.translation[1]));
      if (n.use_matrix) ctx.apply_transform(n.matrix);
    }
    if (n.kind == FDH_NK_RECTANGLE) drop_shadows(n);
    // Culling: a node that clips its content to a mask no produced pixel lies under draws nothing from here to its pop -- the
    // mask plane is 0 there (glcontext.nim:1886-1914: the mask quad never reaches those pixels; the analytic rect mask,
    // atlas_rect_mask.frag:222-237, is 0 beyond its AA fringe of 0.5 / aa px: hence the pad) -- so neither the node nor its
    // subtree is walked.  (The reference's own table benchmark scrolls 180 rows through a window that shows 30.)
    if ((clip || rmask) && ctx.culling()) {
      const bool dead = (clip && !ctx.rect_visible(box, 0.0f)) || (rmask && ctx.aa() >= 0.05f && !ctx.rect_visible(box, 1.0f + 0.5f / ctx.aa()));
      if (dead) {
        culled_subtrees++;
        if (xf) ctx.restore_transform();
        if (rot) ctx.restore_transform();
        return;
      }
    }
    if (clip) {
      ctx.begin_mask(box, rx, ry);
      ctx.end_mask();
    }
    if (rmask) ctx.begin_rect_mask(box, rx, ry);
    switch (n.kind) {
      case FDH_NK_TEXT: text(n); break;
      case FDH_NK_RECTANGLE: rounded_shape(n, n.fill, &n.stroke); break;  // renderBoxes :1669-1671
      case FDH_NK_IMAGE: {                                                 // renderImage :1673-1684
        if (n.image_id == 0) break;
        const FdhColor k = sample_fill(n.image_fill, 0.5f);
        const FdhColor cols[4] = {k, k, k, k};
        const float pos[2] = {box[0], box[1]}, size[2] = {box[2], box[3]};
        ctx.draw_image(n.image_id, pos, cols, size, (n.flags & FDH_NF_INVERT_Y) != 0);
        break;
      }
      case FDH_NK_MSDF_IMAGE:
      case FDH_NK_MTSDF_IMAGE: {  // renderMsdfImage / renderMtsdfImage :1686-1732
        if (n.image_id == 0) break;
        const float pr = n.px_range > 0.0f ? n.px_range : 4.0f;
        const float th = (n.sd_threshold > 0.0f && n.sd_threshold < 1.0f) ? n.sd_threshold : 0.5f;
        const float sw = scaled(std::max(0.0f, n.stroke_weight));
        const float pos[2] = {box[0], box[1]}, size[2] = {box[2], box[3]};
        ctx.draw_msdf(n.image_id, pos, sample_fill(n.image_fill, 0.5f), size, pr, th, sw, n.kind == FDH_NK_MTSDF_IMAGE,
                      (n.flags & FDH_NF_INVERT_Y) != 0);
        break;
      }
      case FDH_NK_BACKDROP_BLUR: {  // renderBackdropBlur :1734-1754
        if (n.blur > 0.0f) ctx.draw_backdrop_blur(box, rx, ry, scaled(n.blur));
        if (fill_alpha_max(n.fill) != 0) rounded_shape(n, n.fill, nullptr);
        break;
      }
      case FDH_NK_DRAWABLE: drawable(n); break;
      default: break;  // nkFrame / nkScrollBar / nkTransform draw nothing themselves
    }
    if (n.kind == FDH_NK_RECTANGLE) inner_shadows(n);
    // childIndex fignodes.nim:165-177 scans forward for the nodes whose parent is this one, the first childCount of them.  The
    // same nodes in the same order come out of sibling links built in one pass per layer (children_of): a parent of a thousand
    // children -- a table's viewport -- otherwise strides over all its grandchildren to find them, a few MB of node structs.
    if (n.child_count > 0) {
      links->build(L);
      if (can_fork && n.child_count >= kForkMin) {
        std::vector<int> kids;
        kids.reserve((size_t)n.child_count);
        for (int i = links->first_child[(size_t)idx]; i >= 0 && (int)kids.size() < n.child_count; i = links->next_sibling[(size_t)i]) kids.push_back(i);
        siblings(L, kids.data(), (int)kids.size());
      } else {
        int seen = 0;
        for (int i = links->first_child[(size_t)idx]; i >= 0 && seen < n.child_count; i = links->next_sibling[(size_t)i]) {
          seen++;
          const int nx = links->next_sibling[(size_t)i];
          if (nx >= 0) prefetch_node(L, nx);
          node(L, i);
        }
      }
    }
    if (rmask) ctx.pop_rect_mask();
    if (clip) ctx.pop_mask();
    if (xf) ctx.restore_transform();
    if (rot) ctx.restore_transform();
  }
};

// ------------------------------------------------------------------ a sibling group on the walk pool
// The reference walks the tree on one thread, every frame (figrender.nim:1756-1839, 1960-2002) -- and so did this library until
// round 4: 50 ns per draw record, which for the reference's own benchmark trees (thousands of nodes, examples/
// windy_non_clip_benchmark.nim:82-147) was ten times what the GPU needs for the frame.  A run of siblings is independent work
// once the state they inherit is fixed -- the transform, the AA factor, how many clips and rect masks are open around them -- so
// it is cut into chunks that pool threads (and this one) decompose into their own lanes of records; the chunks' pieces are
// then listed in painter's order.  What crosses chunk boundaries is put right by the calling thread: the bounds of the clips open
// around the group grow by each chunk's union, the phase takes each chunk's summary, the list stride each thread's maximum.
// Subtrees holding a backdrop-blur node never get here (Walker::siblings): a blur splits the frame into phases.
bool ParallelWalk::group(Walker& mw, const FdhLayer& L, const int* items, int n) {
  Context& C = *mw.ctx.cx_;
  const int helpers = C.walk_threads();
  if (helpers <= 0 || C.rec_on_ || !C.frame_begun_) return false;
  Context::HostTimer t_group(C.host_ns_[7]);
  const int slots = helpers + 1;
  // (every chunk is a piece of the frame -- three upload runs, a merge step --, and a frame's run table holds 31 pieces; three chunks per
  // slot: a viewport's visible rows lie in the first few chunks and the slots that got the culled ones must find something to steal)
  const int n_chunks = std::min(n / 12, slots * (slots > 4 ? 3 : 4));
  if (n_chunks < 2) return false;
  struct alignas(128) Out {
    Piece p;
    PhaseSum s;
    BBox outer{0, 0, 0, 0};
    int64_t frags = 0, culled = 0;
    bool serial_only = false;
    std::exception_ptr err;
    int slot = -1;
  };
  std::vector<Out> outs((size_t)n_chunks);
  // chunk c covers items [cut[c], cut[c + 1]): equal counts the first time a group is seen, then by what the chunks cost last time
  // (cost spread evenly over a chunk's items; the new cuts are the quantiles of that piecewise-constant density)
  Context::GroupCuts* gc = nullptr;
  for (auto& g : C.group_cuts_) if (g.first_item == items[0] && g.n == n && (int)g.cut.size() == n_chunks + 1) { gc = &g; break; }
  if (!gc) {
    if (C.group_cuts_.size() >= 16) {  // (forget the one that has not come for the longest)
      size_t old = 0;
      for (size_t i = 1; i < C.group_cuts_.size(); i++) if (C.group_cuts_[i].used < C.group_cuts_[old].used) old = i;
      C.group_cuts_.erase(C.group_cuts_.begin() + (long)old);
    }
    C.group_cuts_.emplace_back();
    gc = &C.group_cuts_.back();
    gc->first_item = items[0]; gc->n = n;
    gc->cut.resize((size_t)n_chunks + 1);
    for (int c = 0; c <= n_chunks; c++) gc->cut[(size_t)c] = (int)((int64_t)c * n / n_chunks);
  } else if ((int)gc->cost.size() == n_chunks) {
    double total = 0;
    for (float v : gc->cost) total += v;
    if (total > 0) {
      std::vector<int> cut((size_t)n_chunks + 1, 0);
      cut[(size_t)n_chunks] = n;
      int c_old = 0;
      double before = 0;  // cost of the old chunks before c_old
      for (int c = 1; c < n_chunks; c++) {
        const double want = total * c / n_chunks;
        while (c_old < n_chunks - 1 && before + gc->cost[(size_t)c_old] < want) { before += gc->cost[(size_t)c_old]; c_old++; }
        const int a = gc->cut[(size_t)c_old], b = gc->cut[(size_t)c_old + 1];
        const double f = gc->cost[(size_t)c_old] > 0 ? (want - before) / gc->cost[(size_t)c_old] : 0.0;
        int at = a + (int)std::lround(f * (b - a));
        at = std::max(at, cut[(size_t)c - 1] + 1);           // every chunk keeps at least one item ...
        at = std::min(at, n - (n_chunks - c));                // ... and leaves one for each chunk behind it
        cut[(size_t)c] = at;
      }
      gc->cut = cut;
    }
  }
  gc->used = C.frame_no_;
  const std::vector<int> cut = gc->cut;  // (a copy: the pool threads read it while nothing else may touch the cache)
  mw.links->build(L);  // (the pool threads read the links: built before they start)
  C.pool_slots(slots);
  struct Mark { size_t recs, exts; };
  std::vector<Mark> marks((size_t)slots);
  for (int s = 0; s < slots; s++) { Lane& Ln = C.lane(s + 1); marks[(size_t)s] = Mark{Ln.recs.n, Ln.exts.n}; }
  std::vector<int> slot_max((size_t)slots, 0);
  const Recorder& base = mw.ctx;
  const int outer_rmasks = (int)base.rect_masks_.size() + base.outer_rect_masks_;
  const bool outer_open = !base.open_ops_.empty() || base.outer_open_;
  C.close_piece();
  auto fn = [&](int slot, int c) {
    Recorder& R = *C.pool_recs_[(size_t)slot];
    Lane& Ln = C.lane(slot + 1);
    if (c < 0) { slot_max[(size_t)slot] = Ln.count_close(); return; }
    Out& o = outs[(size_t)c];
    thread_local int t_dev = -1;
    if (!C.host_only_ && t_dev != C.device_) { (void)hipSetDevice(C.device_); t_dev = C.device_; }  // (a lane's pinned mirror is allocated by the thread that publishes into it)
    R.lane_ = &Ln;
    R.mat_ = base.mat_; R.mats_.clear();
    R.aa_ = base.aa_; R.subpixel_shift_ = base.subpixel_shift_;
    R.mask_begun_ = false; R.mask_depth_ = 0;
    R.rect_masks_.clear(); R.outer_rect_masks_ = outer_rmasks;
    R.open_ops_.clear(); R.outer_open_ = outer_open; R.outer_union_ = BBox{0, 0, 0, 0};
    R.depth_now_ = 0;
    R.sum_ = PhaseSum{};
    R.fragments_ = 0; R.culled_draws_ = 0;
    R.phase_floor_ = (int)Ln.recs.n;
    o.p.lane = slot + 1; o.p.first = (uint32_t)Ln.recs.n; o.p.ext_first = (uint32_t)Ln.exts.n;
    try {
      Walker w{R, mw.scene, mw.ui, mw.links, false};
      w.depth = mw.depth;
      const int i0 = cut[(size_t)c], i1 = cut[(size_t)c + 1];
      for (int k = i0; k < i1; k++) { if (k + 1 < i1) Walker::prefetch_node(L, items[k + 1]); w.node(L, items[k]); }
    } catch (const SerialOnly&) {
      o.serial_only = true;
    } catch (...) {
      o.err = std::current_exception();
    }
    o.p.n = (uint32_t)Ln.recs.n - o.p.first; o.p.n_ext = (uint32_t)Ln.exts.n - o.p.ext_first;
    if (!o.serial_only && !o.err) {
      try { Ln.publish(o.p.first, o.p.n, o.p.ext_first, o.p.n_ext); }  // (final: every clip the chunk opened it closed)
      catch (...) { o.err = std::current_exception(); }
    }
    o.s = R.sum_; o.outer = R.outer_union_; o.frags = R.fragments_; o.culled = R.culled_draws_;
  };
  bool ran;
  { Context::HostTimer t(C.host_ns_[8]); ran = WalkPool::get().run(helpers, n_chunks, fn); }
  bool failed = !ran;
  std::exception_ptr err;
  for (const Out& o : outs) { if (o.serial_only || o.err) failed = true; if (o.err && !err) err = o.err; }
  if (failed) {  // nothing of the group stays: the calling thread walks it itself (or the frame ends with the error)
    for (int s = 0; s < slots; s++) { Lane& Ln = C.lane(s + 1); Ln.recs.n = Ln.bins.n = Ln.boxes.n = marks[(size_t)s].recs; Ln.exts.n = marks[(size_t)s].exts; }
    C.open_piece();
    if (err) std::rethrow_exception(err);
    return false;
  }
  Context::HostTimer t_merge(C.host_ns_[9]);
  gc->cost.resize((size_t)n_chunks);
  for (int c = 0; c < n_chunks; c++) gc->cost[(size_t)c] = (float)outs[(size_t)c].p.n + 0.25f * (float)(cut[(size_t)c + 1] - cut[(size_t)c]);
  Piece run{};
  PhaseSum none{};
  for (const Out& o : outs) {
    // (chunks one thread took back to back lie back to back in its lane: one piece)
    if (run.n && run.lane == o.p.lane && run.first + run.n == o.p.first && run.ext_first + run.n_ext == o.p.ext_first) { run.n += o.p.n; run.n_ext += o.p.n_ext; }
    else { if (run.n) C.add_piece(run, none, BBox{0, 0, 0, 0}, 0, 0); run = o.p; }
    Piece nothing{};
    C.add_piece(nothing, o.s, o.outer, o.frags, o.culled);
  }
  if (run.n) C.add_piece(run, none, BBox{0, 0, 0, 0}, 0, 0);
  for (int s = 0; s < slots; s++) C.phase_extra_ += slot_max[(size_t)s];
  C.open_piece();
  C.parallel_groups_++;
  return true;
}

// How far, in rows, the scene's backdrop blurs can carry a pixel's influence: the sum over its blur nodes of the tap reach of
// their filters (make_taps: at most max(radius, 8) + 1 rows, 66 for the clamped radius 64).  A row stripe is rendered with that
// much halo at most (launch_frame widens the stripe phase by phase), so draws beyond it can be culled.
static int scene_blur_reach(const FdhScene& scene, float ui) {
  long long reach = 0;
  for (int l = 0; l < scene.n_layers; l++) {
    const FdhLayer& L = scene.layers[l];
    for (int i = 0; i < L.n_nodes && L.nodes; i++) {
      const FdhFig& n = L.nodes[i];
      if (n.kind != FDH_NK_BACKDROP_BLUR || !(n.blur > 0.0f)) continue;
      const float r = std::min(std::max(n.blur * ui, 8.0f), 64.0f);
      reach += (long long)std::ceil(r) + 2;
    }
  }
  return (int)std::min<long long>(reach, 1 << 20);
}

void Context::render_frame(const FdhScene* scene, float fw, float fh, bool clear, const float rgba[4]) {
  if (!scene || (scene->n_layers > 0 && !scene->layers)) throw Error(FDH_ERR_INVALID, "render_frame: null scene");
  const float w = fw * ui_scale_, h = fh * ui_scale_;  // frameSize.scaled()
  if (w <= 0.0f || h <= 0.0f) return;
  if (stripe_y1_ > stripe_y0_ && culling()) pending_reach_ = scene_blur_reach(*scene, ui_scale_);
  begin_frame((int)w, (int)h, clear, rgba);
  try {
    save_transform();
    scale(pixel_scale_, pixel_scale_);
    static thread_local LayerLinks links;  // (the arrays keep their capacity from frame to frame)
    links.layer = nullptr;
    Walker wk{*this, *scene, ui_scale_, &links, true};
    for (int l = 0; l < scene->n_layers; l++) {
      const FdhLayer& L = scene->layers[l];
      if ((L.n_nodes > 0 && !L.nodes) || (L.n_roots > 0 && !L.root_ids)) throw Error(FDH_ERR_INVALID, "render_frame: a layer's node or root array is null");
      for (int r = 0; r < L.n_roots; r++)
        if (L.root_ids[r] < 0 || L.root_ids[r] >= L.n_nodes) throw Error(FDH_ERR_INVALID, "render_frame: root index out of range");
      wk.siblings(L, L.root_ids, L.n_roots);
    }
    restore_transform();
  } catch (...) {
    frame_begun_ = false;
    throw;
  }
  end_frame();
}

// ------------------------------------------------------------------ retained scenes (renderfragments.nim:426-544, common/transfer.nim)
// The reference keeps a base `Renders` and lets the application insert / append / replace fragments of it between frames
// (insertChildren, addChildren, insertRoot, updateFragment :523); its renderer still walks the whole tree every frame.  Here
// the tree lives in the context: fdh_scene_retain copies it, fdh_scene_update_nodes / fdh_scene_replace_root / fdh_scene_insert_root
// edit it, and fdh_scene_render re-decomposes only the roots an edit touched -- the draw records of every other root are
// spliced back from the per-root cache (a memcpy), so a frame after a small edit costs the launches, not a tree walk.
void Context::rebase_side(FdhFig* nodes, int n, const FdhScene* side) {
  // the new nodes index glyph / op / control / text-rect arrays of `side`: append what they use to the retained arrays
  RetainedScene& R = retained_;
  for (int i = 0; i < n; i++) {
    FdhFig& f = nodes[i];
    if (f.glyph_count > 0) {
      if (!side || !side->glyphs || f.glyph_first < 0 || f.glyph_first + f.glyph_count > side->n_glyphs) throw Error(FDH_ERR_INVALID, "scene update: glyph range outside the side arrays");
      const int base = (int)R.glyphs.size();
      R.glyphs.insert(R.glyphs.end(), side->glyphs + f.glyph_first, side->glyphs + f.glyph_first + f.glyph_count);
      if (!R.variant_ids.empty() || side->glyph_variant_ids) {
        // glyphs retained before any variant table existed fall back to their own image, as the new-glyph branch below does.
        // (The table's mere presence changes how renderText treats EVERY glyph under variant positioning -- key lookup with
        // shift 0 instead of the fractional shift -- so records cached without it are stale: table_epoch.)
        if (R.variant_ids.empty()) R.table_epoch++;
        for (size_t g0 = R.variant_ids.size() / FDH_GLYPH_VARIANT_STEPS; g0 < (size_t)base; g0++)
          for (int st = 0; st < FDH_GLYPH_VARIANT_STEPS; st++) R.variant_ids.push_back(R.glyphs[g0].image_id);
        for (int g = 0; g < f.glyph_count; g++)
          for (int st = 0; st < FDH_GLYPH_VARIANT_STEPS; st++)
            R.variant_ids.push_back(side->glyph_variant_ids ? side->glyph_variant_ids[(size_t)(f.glyph_first + g) * FDH_GLYPH_VARIANT_STEPS + st] : side->glyphs[f.glyph_first + g].image_id);
      }
      f.glyph_first = base;
    }
    if (f.text_rect_count > 0) {
      if (!side || !side->text_rects || f.text_rect_first < 0 || f.text_rect_first + f.text_rect_count > side->n_text_rects) throw Error(FDH_ERR_INVALID, "scene update: text-rect range outside the side arrays");
      const int base = (int)R.text_rects.size();
      R.text_rects.insert(R.text_rects.end(), side->text_rects + f.text_rect_first, side->text_rects + f.text_rect_first + f.text_rect_count);
      f.text_rect_first = base;
    }
    if (f.op_count > 0) {
      if (!side || !side->ops || f.op_first < 0 || f.op_first + f.op_count > side->n_ops) throw Error(FDH_ERR_INVALID, "scene update: drawable-op range outside the side arrays");
      const int base = (int)R.ops.size();
      for (int k = 0; k < f.op_count; k++) {
        FdhDrawOp op = side->ops[f.op_first + k];
        if (op.ctrl_count > 0) {
          if (!side->controls || op.ctrl_first < 0 || op.ctrl_first + op.ctrl_count > side->n_controls) throw Error(FDH_ERR_INVALID, "scene update: control-point range outside the side arrays");
          const int cb = (int)(R.controls.size() / 2);
          R.controls.insert(R.controls.end(), side->controls + 2 * op.ctrl_first, side->controls + 2 * (op.ctrl_first + op.ctrl_count));
          op.ctrl_first = cb;
        }
        R.ops.push_back(op);
      }
      f.op_first = base;
    }
  }
}

// An edit either lands whole or not at all: rebase_side appends to the retained side arrays before it has seen every node of
// the edit, so a bad range found late must take the appended entries back.
namespace {
struct SideMark {
  RetainedScene& R;
  const size_t g, v, o, c, t;
  bool keep = false;
  explicit SideMark(RetainedScene& r) : R(r), g(r.glyphs.size()), v(r.variant_ids.size()), o(r.ops.size()), c(r.controls.size()), t(r.text_rects.size()) {}
  ~SideMark() {
    if (keep) return;
    R.glyphs.resize(g); R.variant_ids.resize(v); R.ops.resize(o); R.controls.resize(c); R.text_rects.resize(t);
  }
};
}  // namespace

// Every update_nodes / replace_root appends the side entries of the new nodes and orphans those of the old ones; an animated
// text or drawable node would grow the arrays without bound (and n_glyphs past int32).  When the live entries are less than
// half of an array that has grown past a few thousand, the arrays are rebuilt from the nodes that still reference them.  Node
// ranges move, draw records do not depend on them: the per-root caches stay valid.
void Context::compact_side() {
  RetainedScene& R = retained_;
  size_t live_g = 0, live_o = 0, live_t = 0;
  for (const RetainedLayer& D : R.layers)
    for (const FdhFig& f : D.nodes) { live_g += (size_t)std::max(f.glyph_count, 0); live_o += (size_t)std::max(f.op_count, 0); live_t += (size_t)std::max(f.text_rect_count, 0); }
  auto bloated = [](size_t live, size_t have) { return have > 4096 && live * 2 < have; };
  if (!bloated(live_g, R.glyphs.size()) && !bloated(live_o, R.ops.size()) && !bloated(live_t, R.text_rects.size())) return;
  std::vector<FdhGlyph> glyphs;
  std::vector<int64_t> variants;
  std::vector<FdhDrawOp> ops;
  std::vector<float> controls;
  std::vector<FdhTextRect> rects;
  const bool has_var = !R.variant_ids.empty();
  for (RetainedLayer& D : R.layers)
    for (FdhFig& f : D.nodes) {
      if (f.glyph_count > 0 && f.glyph_first >= 0 && (size_t)f.glyph_first + (size_t)f.glyph_count <= R.glyphs.size()) {
        const int base = (int)glyphs.size();
        glyphs.insert(glyphs.end(), R.glyphs.begin() + f.glyph_first, R.glyphs.begin() + f.glyph_first + f.glyph_count);
        if (has_var)
          for (int g = f.glyph_first; g < f.glyph_first + f.glyph_count; g++)
            for (int st = 0; st < FDH_GLYPH_VARIANT_STEPS; st++) {
              const size_t at = (size_t)g * FDH_GLYPH_VARIANT_STEPS + st;
              variants.push_back(at < R.variant_ids.size() ? R.variant_ids[at] : R.glyphs[(size_t)g].image_id);
            }
        f.glyph_first = base;
      } else if (f.glyph_count > 0) f.glyph_count = 0;
      if (f.text_rect_count > 0 && f.text_rect_first >= 0 && (size_t)f.text_rect_first + (size_t)f.text_rect_count <= R.text_rects.size()) {
        const int base = (int)rects.size();
        rects.insert(rects.end(), R.text_rects.begin() + f.text_rect_first, R.text_rects.begin() + f.text_rect_first + f.text_rect_count);
        f.text_rect_first = base;
      } else if (f.text_rect_count > 0) f.text_rect_count = 0;
      if (f.op_count > 0 && f.op_first >= 0 && (size_t)f.op_first + (size_t)f.op_count <= R.ops.size()) {
        const int base = (int)ops.size();
        for (int k = 0; k < f.op_count; k++) {
          FdhDrawOp op = R.ops[(size_t)(f.op_first + k)];
          if (op.ctrl_count > 0 && op.ctrl_first >= 0 && 2 * ((size_t)op.ctrl_first + (size_t)op.ctrl_count) <= R.controls.size()) {
            const int cb = (int)(controls.size() / 2);
            controls.insert(controls.end(), R.controls.begin() + 2 * op.ctrl_first, R.controls.begin() + 2 * (op.ctrl_first + op.ctrl_count));
            op.ctrl_first = cb;
          } else op.ctrl_count = 0;
          ops.push_back(op);
        }
        f.op_first = base;
      } else if (f.op_count > 0) f.op_count = 0;
    }
  R.glyphs.swap(glyphs); R.variant_ids.swap(variants); R.ops.swap(ops); R.controls.swap(controls); R.text_rects.swap(rects);
}

// A retained layer is edited in place (subtrees compacted out, parents remapped): every parent index it holds must be -1 or
// that of an EARLIER node (parents precede their children in a RenderList, fignodes.nim:119-163).
static void check_parents(const FdhFig* nodes, int n, int first_index, const char* who) {
  for (int k = 0; k < n; k++)
    if (nodes[k].parent < -1 || nodes[k].parent >= first_index + k) throw Error(FDH_ERR_INVALID, std::string(who) + ": a node's parent must be -1 or an earlier node of the layer");
}

void Context::scene_retain(const FdhScene* scene, float fw, float fh, bool clear, const float rgba[4]) {
  if (!scene || (scene->n_layers > 0 && !scene->layers)) throw Error(FDH_ERR_INVALID, "scene_retain: null scene");
  for (int l = 0; l < scene->n_layers; l++) {
    const FdhLayer& L = scene->layers[l];
    if ((L.n_nodes > 0 && !L.nodes) || (L.n_roots > 0 && !L.root_ids)) throw Error(FDH_ERR_INVALID, "scene_retain: a layer's node or root array is null");
    if (L.n_nodes > 32767) throw Error(FDH_ERR_INVALID, "scene_retain: more than 32767 nodes in a layer (FigIdx is int16, fignodes.nim:119)");
    check_parents(L.nodes, L.n_nodes, 0, "scene_retain");
  }
  RetainedScene& R = retained_;
  R = RetainedScene{};
  R.fw = fw; R.fh = fh; R.clear = clear;
  for (int i = 0; i < 4; i++) R.rgba[i] = rgba[i];
  if (scene->glyphs && scene->n_glyphs > 0) R.glyphs.assign(scene->glyphs, scene->glyphs + scene->n_glyphs);
  if (scene->glyph_variant_ids && scene->n_glyphs > 0) R.variant_ids.assign(scene->glyph_variant_ids, scene->glyph_variant_ids + (size_t)scene->n_glyphs * FDH_GLYPH_VARIANT_STEPS);
  if (scene->ops && scene->n_ops > 0) R.ops.assign(scene->ops, scene->ops + scene->n_ops);
  if (scene->controls && scene->n_controls > 0) R.controls.assign(scene->controls, scene->controls + 2 * (size_t)scene->n_controls);
  if (scene->text_rects && scene->n_text_rects > 0) R.text_rects.assign(scene->text_rects, scene->text_rects + scene->n_text_rects);
  R.layers.resize((size_t)std::max(scene->n_layers, 0));
  for (int l = 0; l < scene->n_layers; l++) {
    const FdhLayer& L = scene->layers[l];
    RetainedLayer& D = R.layers[(size_t)l];
    D.zlevel = L.zlevel;
    if (L.n_nodes > 0) D.nodes.assign(L.nodes, L.nodes + L.n_nodes);
    if (L.n_roots > 0) D.roots.assign(L.root_ids, L.root_ids + L.n_roots);
    for (int r : D.roots) if (r < 0 || r >= L.n_nodes) throw Error(FDH_ERR_INVALID, "scene_retain: root index out of range");
    D.cache.assign(D.roots.size(), RetainedRoot{});
  }
  R.valid = true;
  scene_render();
}

// the root (index into the layer's nodes) each node hangs under; parents precede their children in a RenderList (fignodes.nim:119-163)
static std::vector<int> roots_of(const RetainedLayer& D) {
  std::vector<int> ro(D.nodes.size());
  for (size_t i = 0; i < D.nodes.size(); i++) {
    const int p = D.nodes[i].parent;
    ro[i] = (p < 0 || (size_t)p >= i) ? (int)i : ro[(size_t)p];
  }
  return ro;
}

void Context::scene_update_nodes(int layer, int first, int count, const FdhFig* nodes, const FdhScene* side) {
  RetainedScene& R = retained_;
  if (!R.valid) throw Error(FDH_ERR_INVALID, "scene_update_nodes: no retained scene (fdh_scene_retain first)");
  if (layer < 0 || (size_t)layer >= R.layers.size()) throw Error(FDH_ERR_INVALID, "scene_update_nodes: layer out of range");
  RetainedLayer& D = R.layers[(size_t)layer];
  if (count <= 0) return;
  if (!nodes || first < 0 || (size_t)first + (size_t)count > D.nodes.size()) throw Error(FDH_ERR_INVALID, "scene_update_nodes: node range out of bounds");
  check_parents(nodes, count, first, "scene_update_nodes");
  // the roots above the range before the edit (a node may change its parent) ...
  std::vector<int> before = roots_of(D);
  std::vector<FdhFig> fresh(nodes, nodes + count);
  {
    SideMark mark(R);  // a bad side range throws out of rebase_side: the entries it had appended go with it
    rebase_side(fresh.data(), count, side);
    mark.keep = true;
  }
  std::copy(fresh.begin(), fresh.end(), D.nodes.begin() + first);
  std::vector<int> after = roots_of(D);  // ... and after it
  for (size_t s = 0; s < D.roots.size(); s++)
    for (int i = first; i < first + count; i++)
      if (before[(size_t)i] == D.roots[s] || after[(size_t)i] == D.roots[s]) { D.cache[s].dirty = true; break; }
  compact_side();
}

void Context::scene_replace_root(int layer, int slot, const FdhFig* subtree, int n, const FdhScene* side, bool insert) {
  RetainedScene& R = retained_;
  if (!R.valid) throw Error(FDH_ERR_INVALID, "scene_replace_root: no retained scene (fdh_scene_retain first)");
  if (layer < 0 || (size_t)layer >= R.layers.size()) throw Error(FDH_ERR_INVALID, "scene_replace_root: layer out of range");
  RetainedLayer& D = R.layers[(size_t)layer];
  if (slot < 0 || (size_t)slot > D.roots.size() || (!insert && (size_t)slot == D.roots.size())) throw Error(FDH_ERR_INVALID, "scene_replace_root: root slot out of range");
  if (n < 0 || (n > 0 && !subtree)) throw Error(FDH_ERR_INVALID, "scene_replace_root: bad subtree");
  if (n > 0 && subtree[0].parent >= 0) throw Error(FDH_ERR_INVALID, "scene_replace_root: the subtree's first node must be its root (parent -1)");
  for (int i = 1; i < n; i++)
    if (subtree[i].parent < 0 || subtree[i].parent >= i) throw Error(FDH_ERR_INVALID, "scene_replace_root: subtree parents must precede their children");
  // Everything that can fail happens BEFORE the retained layer is touched: the node budget, and the re-basing of the new
  // nodes' side ranges (into a copy; what it appended to the side arrays is taken back if it throws).  A failed call leaves
  // the scene exactly as it was.
  std::vector<int> ro;
  size_t kept_count = D.nodes.size();
  int old_root = -1;
  if (!insert) {
    ro = roots_of(D);
    old_root = D.roots[(size_t)slot];
    kept_count = 0;
    for (size_t i = 0; i < D.nodes.size(); i++) if (ro[i] != old_root) kept_count++;
  }
  if (kept_count + (size_t)n > 32767u) throw Error(FDH_ERR_INVALID, "scene_replace_root: more than 32767 nodes in a layer (FigIdx is int16, fignodes.nim:119)");
  std::vector<FdhFig> fresh;
  if (n > 0) {
    fresh.assign(subtree, subtree + n);
    SideMark mark(R);
    rebase_side(fresh.data(), n, side);
    mark.keep = true;
  }
  // ---- commit (nothing below throws but std::bad_alloc)
  if (!insert) {  // drop the old subtree, compacting the node array
    std::vector<int> remap(D.nodes.size(), -1);
    std::vector<FdhFig> kept;
    kept.reserve(kept_count + (size_t)n);
    for (size_t i = 0; i < D.nodes.size(); i++)
      if (ro[i] != old_root) { remap[i] = (int)kept.size(); kept.push_back(D.nodes[i]); }
    for (FdhFig& f : kept) if (f.parent >= 0) f.parent = remap[(size_t)f.parent];
    // (a root slot whose node hangs inside the removed subtree -- fdh_scene_update_nodes may have given a listed root a parent --
    // goes with it: it would name a node that no longer exists)
    for (size_t s = D.roots.size(); s-- > 0;) {
      if ((int)s == slot) continue;
      const int to = remap[(size_t)D.roots[s]];
      if (to >= 0) { D.roots[s] = to; continue; }
      D.roots.erase(D.roots.begin() + (std::ptrdiff_t)s);
      D.cache.erase(D.cache.begin() + (std::ptrdiff_t)s);
      if ((int)s < slot) slot--;
    }
    D.nodes.swap(kept);
    if (n == 0) { D.roots.erase(D.roots.begin() + slot); D.cache.erase(D.cache.begin() + slot); compact_side(); return; }
  } else {
    if (n == 0) return;
    D.roots.insert(D.roots.begin() + slot, 0);
    D.cache.insert(D.cache.begin() + slot, RetainedRoot{});
  }
  const int base = (int)D.nodes.size();
  for (int i = 1; i < n; i++) fresh[(size_t)i].parent += base;
  D.nodes.insert(D.nodes.end(), fresh.begin(), fresh.end());
  D.roots[(size_t)slot] = base;
  D.cache[(size_t)slot] = RetainedRoot{};
  compact_side();
}

void Context::scene_render() {
  RetainedScene& R = retained_;
  if (!R.valid) throw Error(FDH_ERR_INVALID, "scene_render: no retained scene (fdh_scene_retain first)");
  const float w = R.fw * ui_scale_, h = R.fh * ui_scale_;
  if (w <= 0.0f || h <= 0.0f) return;
  // every cached record depends on these front-end settings (the text path snaps glyph positions and picks shifts / variant
  // images by the two sub-pixel switches, figrender.nim:464-476)
  const bool config_changed = R.ui_scale != ui_scale_ || R.aa != aa_ || R.subpixel != subpixel_enabled_ || R.variants != subpixel_variants_ ||
                              R.table_epoch != R.table_epoch_seen;
  R.ui_scale = ui_scale_; R.aa = aa_; R.subpixel = subpixel_enabled_; R.variants = subpixel_variants_; R.table_epoch_seen = R.table_epoch;
  for (const RetainedLayer& D : R.layers)
    for (int r : D.roots) if (r < 0 || (size_t)r >= D.nodes.size()) throw Error(FDH_ERR_INVALID, "scene_render: root index out of range");
  std::vector<FdhLayer> views(R.layers.size());
  for (size_t l = 0; l < R.layers.size(); l++) {
    RetainedLayer& D = R.layers[l];
    views[l] = FdhLayer{D.zlevel, (int32_t)D.nodes.size(), (int32_t)D.roots.size(), 0, D.nodes.data(), D.roots.data()};
  }
  FdhScene view{};
  view.layers = views.data(); view.n_layers = (int32_t)views.size();
  view.glyphs = R.glyphs.data(); view.n_glyphs = (int32_t)R.glyphs.size();
  view.glyph_variant_ids = R.variant_ids.empty() ? nullptr : R.variant_ids.data();
  view.ops = R.ops.data(); view.n_ops = (int32_t)R.ops.size();
  view.controls = R.controls.data(); view.n_controls = (int32_t)(R.controls.size() / 2);
  view.text_rects = R.text_rects.data(); view.n_text_rects = (int32_t)R.text_rects.size();
  R.roots_walked = R.roots_reused = 0;
  if (stripe_y1_ > stripe_y0_ && culling()) pending_reach_ = scene_blur_reach(view, ui_scale_);
  begin_frame((int)w, (int)h, R.clear, R.rgba);
  rec_diff_upload_ = true;  // consecutive frames of a retained scene differ in a few records: Context::prepare uploads the difference
  try {
    save_transform();
    scale(pixel_scale_, pixel_scale_);
    static thread_local LayerLinks links;  // (the arrays keep their capacity from frame to frame)
    links.layer = nullptr;
    Walker wk{*this, view, ui_scale_, &links, false};  // (roots are walked one by one: each has its cache entry)
    Lane& L0 = lane(0);
    for (size_t l = 0; l < R.layers.size(); l++) {
      RetainedLayer& D = R.layers[l];
      for (size_t s = 0; s < D.roots.size(); s++) {
        RetainedRoot& C = D.cache[s];
        // (cached records were culled to the rows in force when they were made: good for any frame that produces no row beyond them)
        if (!C.dirty && C.cacheable && !config_changed && C.cull_y0 <= cull_y0_ && C.cull_y1 >= cull_y1_ && C.atlas_epoch == atlas_epoch_) {
          splice_cached(C);
          R.roots_reused++;
          continue;
        }
        const size_t r0 = L0.recs.n, e0 = L0.exts.n, p0 = phases_.size(), b0 = blurs_.size();
        const int64_t f0 = fragments_;
        const int d0 = depth_now_;
        add_sum(sum_, 0);  // (what the phase has so far goes to the phase: the root's own summary starts from nothing)
        sum_ = PhaseSum{};
        wk.node(views[l], D.roots[s]);
        R.roots_walked++;
        C.dirty = false;
        C.atlas_epoch = atlas_epoch_;
        C.cull_y0 = cull_y0_; C.cull_y1 = cull_y1_;
        C.cacheable = phases_.size() == p0 && blurs_.size() == b0;
        C.recs.clear(); C.bins.clear(); C.exts.clear();
        if (C.cacheable) {
          C.recs.assign(L0.recs.p + r0, L0.recs.p + L0.recs.n);
          C.bins.assign(L0.bins.p + r0, L0.bins.p + L0.bins.n);
          C.exts.assign(L0.exts.p + e0, L0.exts.p + L0.exts.n);
          for (DrawRec& r : C.recs) if (r.op_mode & F_GENERAL) r.ext -= (uint32_t)e0;
          if (!C.bins.empty()) C.bins.back().flags &= ~LE_SHARE;  // (what follows the root is decided at every splice: Context::splice_cached)
          C.fragments = fragments_ - f0;
          C.sum = sum_;
          C.sum.deepest = std::max(0, sum_.deepest - d0);
        }
      }
    }
    restore_transform();
  } catch (...) {
    frame_begun_ = false;
    throw;
  }
  end_frame();
}

}  // namespace fdh
