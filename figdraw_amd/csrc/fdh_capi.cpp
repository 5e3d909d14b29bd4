// fdh_capi.cpp -- extern "C" surface of libfigdraw_hip.so (include/figdraw_hip.h).  Every entry point
// converts C++ exceptions into a negative FdhStatus + a thread-local message: nothing throws across the ABI.
#include <string>

#include "fdh_context.h"

using fdh::Context;

namespace {
thread_local std::string g_last_error;

template <typename F>
int guard(F&& f) {
  try {
    f();
    return FDH_OK;
  } catch (const fdh::Error& e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return FDH_ERR_INVALID;
  } catch (...) {
    g_last_error = "unknown error";
    return FDH_ERR_INVALID;
  }
}
inline Context* C(FdhContext* c) {
  if (!c) throw fdh::Error(FDH_ERR_INVALID, "null FdhContext");
  return reinterpret_cast<Context*>(c);
}
}  // namespace

extern "C" {

const char* fdh_last_error(void) { return g_last_error.c_str(); }
#if FDH_STATS
extern "C++" { namespace fdh { void debug_counters(unsigned long long out[128], bool reset); void debug_wave_times(unsigned long long* out); } }
__attribute__((visibility("default"))) int fdh_debug_wave_times(unsigned long long* out) { fdh::debug_wave_times(out); return 0; }
__attribute__((visibility("default"))) int fdh_debug_counters(unsigned long long out[128], int reset) { fdh::debug_counters(out, reset != 0); return 0; }
#endif
int fdh_saturated_core(const float rect[4], const float rx[4], const float ry[4], int mode, float factor, float spread,
                       const float shape[2], float aa, int out_px[4]) {
  return guard([&] { fdh::saturated_core_of(rect, rx, ry, mode, factor, spread, shape, aa, out_px); });
}
int fdh_blur_weight_fragments(float blur_radius, int vertical, float* dense, uint16_t* frag_bits, int* reach, int* k_steps) {
  return guard([&] {
    if (!dense || !frag_bits || !reach || !k_steps) throw fdh::Error(FDH_ERR_INVALID, "fdh_blur_weight_fragments: null pointer");
    fdh::blur_weight_fragments(blur_radius, vertical != 0, dense, frag_bits, reach, k_steps);
  });
}
const char* fdh_version(void) { return "figdraw_hip 0.1.0 (gfx950)"; }
int fdh_sizeof_fig(void) { return (int)sizeof(FdhFig); }
int fdh_sizeof_glyph(void) { return (int)sizeof(FdhGlyph); }
int fdh_sizeof_draw_op(void) { return (int)sizeof(FdhDrawOp); }
int fdh_sizeof_text_rect(void) { return (int)sizeof(FdhTextRect); }

int fdh_create(FdhContext** out, int atlas_size, float pixel_scale, int device, uint32_t flags) {
  return guard([&] {
    if (!out) throw fdh::Error(FDH_ERR_INVALID, "fdh_create: null out pointer");
    *out = nullptr;
    *out = reinterpret_cast<FdhContext*>(new Context(atlas_size, pixel_scale, device, flags));
  });
}
int fdh_destroy(FdhContext* c) { return guard([&] { delete C(c); }); }
int fdh_set_stream(FdhContext* c, void* s) { return guard([&] { C(c)->set_stream(s); }); }

int fdh_begin_frame(FdhContext* c, int w, int h, int clear, const float rgba[4]) {
  return guard([&] {
    const float white[4] = {1, 1, 1, 1};
    C(c)->begin_frame(w, h, clear != 0, rgba ? rgba : white);
  });
}
int fdh_end_frame(FdhContext* c) { return guard([&] { C(c)->end_frame(); }); }
int fdh_save_transform(FdhContext* c) { return guard([&] { C(c)->save_transform(); }); }
int fdh_restore_transform(FdhContext* c) { return guard([&] { C(c)->restore_transform(); }); }
int fdh_translate(FdhContext* c, float x, float y) { return guard([&] { C(c)->translate(x, y); }); }
int fdh_rotate(FdhContext* c, float a) { return guard([&] { C(c)->rotate(a); }); }
int fdh_scale(FdhContext* c, float sx, float sy) { return guard([&] { C(c)->scale(sx, sy); }); }
int fdh_apply_transform(FdhContext* c, const float m[16]) { return guard([&] { C(c)->apply_transform(m); }); }
int fdh_transform_mirrors_y(FdhContext* c, int* out) { return guard([&] { *out = C(c)->transform_mirrors_y() ? 1 : 0; }); }
int fdh_set_aa_factor(FdhContext* c, float aa) { return guard([&] { C(c)->set_aa(aa); }); }
int fdh_get_aa_factor(FdhContext* c, float* out) { return guard([&] { *out = C(c)->aa(); }); }
int fdh_get_pixel_scale(FdhContext* c, float* out) { return guard([&] { *out = C(c)->pixel_scale(); }); }

int fdh_draw_rounded_rect_sdf(FdhContext* c, const float rect[4], const FdhColor colors[4], const float rx[4], const float ry[4],
                              int mode, float factor, float spread, const float shape[2], int fill_mode, FdhColor mid,
                              FdhColor stop, float mid_pos) {
  return guard([&] {
    const float zero2[2] = {0, 0};
    C(c)->draw_rounded_rect_sdf(rect, colors, rx, ry ? ry : rx, mode, factor, spread, shape ? shape : zero2, fill_mode, mid, stop, mid_pos);
  });
}
int fdh_draw_rounded_rect_fill(FdhContext* c, const float rect[4], const FdhFill* fill, const float rx[4], const float ry[4],
                               int mode, float factor, float spread, const float shape[2]) {
  return guard([&] {
    if (!fill) throw fdh::Error(FDH_ERR_INVALID, "null fill");
    const float zero2[2] = {0, 0};
    C(c)->draw_rounded_rect_fill(rect, *fill, rx, ry ? ry : rx, mode, factor, spread, shape ? shape : zero2);
  });
}
int fdh_draw_image(FdhContext* c, int64_t key, const float pos[2], const FdhColor colors[4], const float size[2], int flip_y) {
  return guard([&] {
    const float zero2[2] = {0, 0};
    C(c)->draw_image(key, pos, colors, size ? size : zero2, flip_y != 0);
  });
}
int fdh_draw_image_adj(FdhContext* c, int64_t key, const float pos[2], FdhColor color, const float size[2]) {
  return guard([&] {
    if (!pos || !size) throw fdh::Error(FDH_ERR_INVALID, "fdh_draw_image_adj: null pointer");
    C(c)->draw_image_adj(key, pos, color, size);
  });
}
int fdh_draw_msdf(FdhContext* c, int64_t key, const float pos[2], FdhColor color, const float size[2], float px_range,
                  float sd_threshold, float stroke_weight, int mtsdf, int flip_y) {
  return guard([&] { C(c)->draw_msdf(key, pos, color, size, px_range, sd_threshold, stroke_weight, mtsdf != 0, flip_y != 0); });
}
int fdh_draw_backdrop_blur(FdhContext* c, const float rect[4], const float rx[4], const float ry[4], float blur_radius) {
  return guard([&] { C(c)->draw_backdrop_blur(rect, rx, ry ? ry : rx, blur_radius); });
}
int fdh_begin_mask(FdhContext* c, const float rect[4], const float rx[4], const float ry[4]) {
  return guard([&] { C(c)->begin_mask(rect, rx, ry ? ry : rx); });
}
int fdh_end_mask(FdhContext* c) { return guard([&] { C(c)->end_mask(); }); }
int fdh_pop_mask(FdhContext* c) { return guard([&] { C(c)->pop_mask(); }); }
int fdh_begin_rect_mask(FdhContext* c, const float rect[4], const float rx[4], const float ry[4]) {
  return guard([&] { C(c)->begin_rect_mask(rect, rx, ry ? ry : rx); });
}
int fdh_pop_rect_mask(FdhContext* c) { return guard([&] { C(c)->pop_rect_mask(); }); }
int fdh_draw_quadratic_bezier_sdf(FdhContext* c, const float rect[4], const FdhFill* fill, const float p0[2], const float p1[2],
                                  const float p2[2], float stroke_weight, int cap) {
  return guard([&] {
    if (!fill) throw fdh::Error(FDH_ERR_INVALID, "null fill");
    C(c)->draw_quadratic_bezier_sdf(rect, *fill, p0, p1, p2, stroke_weight, cap);
  });
}
int fdh_draw_filled_quad(FdhContext* c, const float verts[8], const FdhColor colors[4]) {
  return guard([&] { C(c)->draw_filled_quad(verts, colors); });
}
int fdh_draw_rect(FdhContext* c, const float rect[4], FdhColor color) { return guard([&] { C(c)->draw_rect(rect, color); }); }
int fdh_set_text_subpixel_positioning(FdhContext* c, int e) { return guard([&] { C(c)->set_subpixel_enabled(e != 0); }); }
int fdh_set_text_subpixel_glyph_variants(FdhContext* c, int e) { return guard([&] { C(c)->set_subpixel_variants(e != 0); }); }
int fdh_set_text_subpixel_shift(FdhContext* c, float s) { return guard([&] { C(c)->set_subpixel_shift(s); }); }
int fdh_set_text_lcd_filtering(FdhContext* c, int e) { return guard([&] { C(c)->set_text_lcd_filtering(e != 0); }); }
int fdh_get_text_lcd_filtering(FdhContext* c, int* out) {
  return guard([&] {
    if (!out) throw fdh::Error(FDH_ERR_INVALID, "null output");
    *out = C(c)->text_lcd_filtering() ? 1 : 0;
  });
}

int fdh_put_image(FdhContext* c, int64_t key, int w, int h, const uint8_t* rgba, int out_rect[4]) {
  return guard([&] { C(c)->put_image(key, w, h, rgba, out_rect); });
}
int fdh_put_glyph_image(FdhContext* c, int64_t key, int w, int h, const uint8_t* rgba, uint32_t flags, int out_rect[4]) {
  return guard([&] { C(c)->put_glyph_image(key, w, h, rgba, flags, out_rect); });
}
int fdh_put_glyph_outline(FdhContext* c, int64_t key, int w, int h, const float* segs, int n, uint32_t flags, int out_rect[4]) {
  return guard([&] { C(c)->put_glyph_outline(key, w, h, segs, n, flags, out_rect); });
}
int fdh_put_image_mips(FdhContext* c, int64_t key, int n_levels, const int* widths, const int* heights,
                       const uint8_t* const* premul_rgba8, int out_rect[4]) {
  return guard([&] { C(c)->put_mips(key, n_levels, widths, heights, premul_rgba8, out_rect); });
}
int fdh_put_flippy(FdhContext* c, int64_t key, const uint8_t* bytes, size_t n, int out_rect[4]) {
  return guard([&] { C(c)->put_flippy(key, bytes, n, out_rect); });
}
int fdh_update_image(FdhContext* c, int64_t key, int w, int h, const uint8_t* rgba) {
  return guard([&] { C(c)->update_image(key, w, h, rgba); });
}
int fdh_remove_image(FdhContext* c, int64_t key) { return guard([&] { C(c)->remove_image(key); }); }
int fdh_has_image(FdhContext* c, int64_t key, int* out) { return guard([&] { *out = C(c)->has_image(key) ? 1 : 0; }); }
int fdh_reset_atlas(FdhContext* c, int minimum_size) { return guard([&] { C(c)->reset_atlas(minimum_size); }); }
int fdh_atlas_size(FdhContext* c, int* out) { return guard([&] { *out = C(c)->atlas_size(); }); }
int fdh_atlas_packed_area(FdhContext* c, int64_t* out) { return guard([&] { *out = C(c)->atlas_packed_area(); }); }

int fdh_read_pixels(FdhContext* c, int x, int y, int w, int h, uint8_t* out) {
  return guard([&] {
    if (!out) throw fdh::Error(FDH_ERR_INVALID, "null output buffer");
    C(c)->read_pixels(x, y, w, h, out);
  });
}
int fdh_frame_device_ptr(FdhContext* c, void** p, int* w, int* h, int64_t* pitch) {
  return guard([&] { C(c)->frame_device_ptr(p, w, h, pitch); });
}
int fdh_record_begin(FdhContext* c) { return guard([&] { C(c)->record_begin(); }); }
const char* fdh_record_json(FdhContext* c) {
  const char* out = "[]";
  guard([&] { out = C(c)->record_json(); });
  return out;
}
int fdh_debug_read_surface(FdhContext* c, int which, uint8_t* out) {
  return guard([&] {
    if (!out) throw fdh::Error(FDH_ERR_INVALID, "null output buffer");
    C(c)->debug_read_surface(which, out);
  });
}
int fdh_sync(FdhContext* c) { return guard([&] { C(c)->sync(); }); }
int fdh_flush(FdhContext* c) { return guard([&] { C(c)->flush(); }); }
int fdh_set_ui_scale(FdhContext* c, float s) { return guard([&] { C(c)->set_ui_scale(s); }); }
int fdh_render_frame(FdhContext* c, const FdhScene* scene, float fw, float fh, int clear, const float rgba[4]) {
  return guard([&] {
    const float white[4] = {1, 1, 1, 1};
    C(c)->render_frame(scene, fw, fh, clear != 0, rgba ? rgba : white);
  });
}
int fdh_scene_retain(FdhContext* c, const FdhScene* scene, float fw, float fh, int clear, const float rgba[4]) {
  return guard([&] {
    const float white[4] = {1, 1, 1, 1};
    C(c)->scene_retain(scene, fw, fh, clear != 0, rgba ? rgba : white);
  });
}
int fdh_scene_update_nodes(FdhContext* c, int layer, int first, int count, const FdhFig* nodes, const FdhScene* side) {
  return guard([&] { C(c)->scene_update_nodes(layer, first, count, nodes, side); });
}
int fdh_scene_replace_root(FdhContext* c, int layer, int slot, const FdhFig* subtree, int n, const FdhScene* side) {
  return guard([&] { C(c)->scene_replace_root(layer, slot, subtree, n, side, false); });
}
int fdh_scene_insert_root(FdhContext* c, int layer, int slot, const FdhFig* subtree, int n, const FdhScene* side) {
  return guard([&] { C(c)->scene_replace_root(layer, slot, subtree, n, side, true); });
}
int fdh_scene_render(FdhContext* c) { return guard([&] { C(c)->scene_render(); }); }
int fdh_last_upload_bytes(FdhContext* c, int64_t* out) {
  return guard([&] {
    if (!out) throw fdh::Error(FDH_ERR_INVALID, "null output");
    *out = C(c)->uploaded_bytes();
  });
}
int fdh_scene_stats(FdhContext* c, int64_t* walked, int64_t* reused) {
  return guard([&] {
    if (!walked || !reused) throw fdh::Error(FDH_ERR_INVALID, "null output");
    C(c)->scene_stats(walked, reused);
  });
}
int fdh_debug_record_digest(FdhContext* c, uint64_t* out) {
  return guard([&] {
    if (!out) throw fdh::Error(FDH_ERR_INVALID, "null output");
    *out = C(c)->record_digest();
  });
}
int fdh_debug_verify_upload(FdhContext* c, uint32_t out[24]) {
  return guard([&] {
    if (!out) throw fdh::Error(FDH_ERR_INVALID, "null output");
    C(c)->debug_verify_upload(out);
  });
}
int fdh_debug_staging_store_bytes(int device, int64_t* out) {
  return guard([&] {
    if (!out) throw fdh::Error(FDH_ERR_INVALID, "null output");
    *out = (int64_t)fdh::vram_store_bytes(device);
  });
}
int fdh_debug_bin_digest(FdhContext* c, uint64_t out[8]) {
  return guard([&] {
    if (!out) throw fdh::Error(FDH_ERR_INVALID, "null output");
    C(c)->debug_bin_digest(out);
  });
}
int fdh_stripe_rows(int height, int world, int rank, int* y0, int* y1) {
  return guard([&] {
    if (!y0 || !y1) throw fdh::Error(FDH_ERR_INVALID, "null output");
    fdh::stripe_rows(height, world, rank, y0, y1);
  });
}
int fdh_comm_unique_id(uint8_t out[FDH_COMM_ID_BYTES]) {
  return guard([&] {
    if (!out) throw fdh::Error(FDH_ERR_INVALID, "null output");
    fdh::comm_unique_id(out);
  });
}
int fdh_comm_init(FdhContext* c, const uint8_t id[FDH_COMM_ID_BYTES], int rank, int world) {
  return guard([&] {
    if (!id) throw fdh::Error(FDH_ERR_INVALID, "null communicator id");
    C(c)->comm_init(id, rank, world);
  });
}
int fdh_comm_info(FdhContext* c, int* rank, int* world) {
  return guard([&] {
    if (!rank || !world) throw fdh::Error(FDH_ERR_INVALID, "null output");
    C(c)->comm_info(rank, world);
  });
}
int fdh_comm_share(FdhContext* c, FdhContext* owner) { return guard([&] { C(c)->comm_share(C(owner)); }); }
int fdh_comm_destroy(FdhContext* c) { return guard([&] { C(c)->comm_destroy(); }); }
int fdh_gather_stripes(FdhContext* c, int dst_rank, void* dst_image) { return guard([&] { C(c)->gather_stripes(dst_rank, dst_image); }); }
int fdh_gather_frames(FdhContext* c, int dst_rank, void* const* dst_images) { return guard([&] { C(c)->gather_frames(dst_rank, dst_images); }); }
int fdh_set_blur_route(FdhContext* c, int route) { return guard([&] { C(c)->set_blur_route(route); }); }
int fdh_set_stripe(FdhContext* c, int y0, int y1) { return guard([&] { C(c)->set_stripe(y0, y1); }); }
int fdh_set_cull(FdhContext* c, int mode) { return guard([&] { C(c)->set_cull(mode); }); }
int fdh_debug_host_times(FdhContext* c, int64_t out_ns[12]) {
  return guard([&] {
    if (!out_ns) throw fdh::Error(FDH_ERR_INVALID, "null output");
    for (int i = 0; i < 12; i++) out_ns[i] = C(c)->host_ns_[i];
  });
}
int fdh_set_walk_threads(FdhContext* c, int n) { return guard([&] { C(c)->set_walk_threads(n); }); }
int fdh_walk_stats(FdhContext* c, int* threads, int64_t* parallel_groups) {
  return guard([&] {
    if (threads) *threads = C(c)->walk_threads();
    if (parallel_groups) *parallel_groups = C(c)->parallel_groups();
  });
}
int fdh_culled_draws(FdhContext* c, int64_t* out) {
  return guard([&] {
    if (!out) throw fdh::Error(FDH_ERR_INVALID, "null output");
    *out = C(c)->culled_draws();
  });
}
int fdh_replay(FdhContext* c, int times) { return guard([&] { C(c)->replay(times); }); }
int fdh_replay_async(FdhContext* c, int times) { return guard([&] { C(c)->replay_async(times); }); }
int fdh_replay_timed(FdhContext* c, int times, float* ms_out) { return guard([&] { C(c)->replay_timed(times, ms_out); }); }
int fdh_profile(FdhContext* c, int times) { return guard([&] { C(c)->profile(times); }); }
int fdh_get_frame_stats(FdhContext* c, FdhFrameStats* out) { return guard([&] { C(c)->frame_stats(out); }); }

}  // extern "C"
