// fdh_frontend.cpp -- the renderer front-end: node tree -> backend calls.
//
// Restates figrender.nim's `renderFrame` (:1960-1995), `renderRoot` (:1946-1955) and the recursive
// `render` (:1756-1839) whose stage order is fixed by the `renderStages` macro (:501-547): stages run
// top-down, their `finally` blocks run in reverse after the children.
#include <cmath>
#include <cstring>

#include "fdh_context.h"

namespace fdh {

FdhColor sample_fill(const FdhFill& f, float t);

namespace {

struct Walker {
  Context& ctx;
  const FdhScene& scene;
  float ui;

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

  void text(const FdhFig& n) {  // renderText :417-497, glyph loop (layout happened on the caller's side)
    ctx.save_transform();
    ctx.translate(scaled(n.box[0]), scaled(n.box[1]));
    if (n.flags & FDH_NF_INVERT_Y) {
      ctx.translate(0.0f, scaled(n.box[3]));
      ctx.scale(1.0f, -1.0f);
    }
    for (int g = n.glyph_first; g < n.glyph_first + n.glyph_count && g < scene.n_glyphs; g++) {
      const FdhGlyph& gl = scene.glyphs[g];
      const float pos[2] = {gl.x, gl.y}, size[2] = {0, 0};
      ctx.set_subpixel_shift(gl.subpixel_shift);
      ctx.draw_image(gl.image_id, pos, gl.colors, size, false);
    }
    ctx.set_subpixel_shift(0.0f);
    ctx.restore_transform();
  }

  void node(const FdhLayer& L, int idx) {
    const FdhFig& n = L.nodes[idx];
    if (n.flags & FDH_NF_DISABLE_RENDER) return;
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
      case FDH_NK_DRAWABLE:
        throw Error(FDH_ERR_UNSUPPORTED, "nkDrawable (lines / circles / beziers) is not on the implemented path yet");
      default: break;  // nkFrame / nkScrollBar / nkTransform draw nothing themselves
    }
    if (n.kind == FDH_NK_RECTANGLE) inner_shadows(n);
    int seen = 0;  // childIndex fignodes.nim:165-177: scan forward for nodes whose parent is this one
    for (int i = idx + 1; i < L.n_nodes && seen < n.child_count; i++)
      if (L.nodes[i].parent == idx) { seen++; node(L, i); }
    if (rmask) ctx.pop_rect_mask();
    if (clip) ctx.pop_mask();
    if (xf) ctx.restore_transform();
    if (rot) ctx.restore_transform();
  }
};

}  // namespace

void Context::render_frame(const FdhScene* scene, float fw, float fh, bool clear, const float rgba[4]) {
  if (!scene) throw Error(FDH_ERR_INVALID, "render_frame: null scene");
  const float w = fw * ui_scale_, h = fh * ui_scale_;  // frameSize.scaled()
  if (w <= 0.0f || h <= 0.0f) return;
  begin_frame((int)w, (int)h, clear, rgba);
  try {
    save_transform();
    scale(pixel_scale_, pixel_scale_);
    Walker wk{*this, *scene, ui_scale_};
    for (int l = 0; l < scene->n_layers; l++) {
      const FdhLayer& L = scene->layers[l];
      for (int r = 0; r < L.n_roots; r++) {
        const int idx = L.root_ids[r];
        if (idx < 0 || idx >= L.n_nodes) throw Error(FDH_ERR_INVALID, "render_frame: root index out of range");
        wk.node(L, idx);
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
