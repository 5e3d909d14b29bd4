/* abi_smoke.c -- every entry point include/figdraw_hip.h declares, called from C99.
 *
 * Test infrastructure (tests/test_abi_and_sharding.py compiles it with gcc -std=c99 -Wall -Wextra -Werror -pedantic and runs it
 * in the CPU suite).  It drives a FDH_CREATE_RECORD_ONLY context -- the call recorder: front-end, record building and atlas
 * packer run, no device is touched -- so header / library drift (a changed parameter, a struct that grew) fails here without
 * a GPU: the program either does not compile, or a known answer below is off.  Entry points that need pixels must fail with
 * FDH_ERR_NO_DEVICE on such a context; that is checked too.
 * usage: abi_smoke <path to a .flippy file> */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "figdraw_hip.h"

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { printf("abi_smoke: FAILED %s:%d: %s   (last error: %s)\n", __FILE__, __LINE__, #cond, fdh_last_error()); failures++; } } while (0)
#define OK(call) CHECK((call) == FDH_OK)

static FdhFill solid(uint8_t r, uint8_t g, uint8_t b, uint8_t a) {
  FdhFill f;
  memset(&f, 0, sizeof f);
  f.kind = FDH_FILL_COLOR;
  f.start.r = r; f.start.g = g; f.start.b = b; f.start.a = a;
  return f;
}
static FdhFig rect_node(float x, float y, float w, float h, FdhFill fill) {
  FdhFig n;
  memset(&n, 0, sizeof n);
  n.kind = FDH_NK_RECTANGLE;
  n.parent = -1;
  n.box[0] = x; n.box[1] = y; n.box[2] = w; n.box[3] = h;
  n.fill = fill;
  n.corners[0] = n.corners[1] = n.corners[2] = n.corners[3] = 6;
  return n;
}

int main(int argc, char** argv) {
  FdhContext* c = NULL;
  const float white[4] = {1.0f, 1.0f, 1.0f, 1.0f};
  const float rect[4] = {10.0f, 20.0f, 100.0f, 60.0f}, radii[4] = {8.0f, 8.0f, 8.0f, 8.0f}, radii_y[4] = {4.0f, 8.0f, 12.0f, 16.0f}, shape0[2] = {0.0f, 0.0f};
  const FdhColor red = {255, 0, 0, 255}, none = {0, 0, 0, 0};
  FdhColor cols[4];
  FdhFill grad;
  uint8_t img[16 * 12 * 4];
  int out_rect[4], i, flag = -1, atlas = 0;
  int64_t area = 0, walked = -1, reused = -1, bytes = -1;
  uint64_t d_frame = 0, d_retained = 0, d_edit = 0;
  float f = 0.0f;
  const char* json;

  /* sizes the other bindings mirror */
  CHECK(fdh_sizeof_fig() == (int)sizeof(FdhFig));
  CHECK(fdh_sizeof_glyph() == (int)sizeof(FdhGlyph));
  CHECK(fdh_sizeof_draw_op() == (int)sizeof(FdhDrawOp));
  CHECK(fdh_sizeof_text_rect() == (int)sizeof(FdhTextRect));
  CHECK(strstr(fdh_version(), "gfx950") != NULL);

  CHECK(fdh_end_frame(NULL) == FDH_ERR_INVALID);
  CHECK(strstr(fdh_last_error(), "null") != NULL);
  OK(fdh_create(&c, 256, 1.0f, 0, FDH_CREATE_RECORD_ONLY | FDH_CREATE_SYNC_SUBMIT));
  CHECK(c != NULL);
  if (!c) return 1;

  OK(fdh_get_pixel_scale(c, &f)); CHECK(f == 1.0f);
  OK(fdh_set_aa_factor(c, 1.5f)); OK(fdh_get_aa_factor(c, &f)); CHECK(f == 1.5f);
  OK(fdh_set_aa_factor(c, 1.2f));
  OK(fdh_set_ui_scale(c, 1.0f));
  OK(fdh_set_text_subpixel_positioning(c, 0));
  OK(fdh_set_text_subpixel_glyph_variants(c, 0));

  /* atlas: skyline packer with margin 4 (glcontext.nim:541-586) */
  memset(img, 200, sizeof img);
  OK(fdh_put_image(c, 42, 16, 12, img, out_rect));
  CHECK(out_rect[0] == 4 && out_rect[1] == 4 && out_rect[2] == 16 && out_rect[3] == 12);
  OK(fdh_has_image(c, 42, &flag)); CHECK(flag == 1);
  OK(fdh_has_image(c, 43, &flag)); CHECK(flag == 0);
  OK(fdh_update_image(c, 42, 16, 12, img));
  CHECK(fdh_update_image(c, 42, 8, 8, img) == FDH_ERR_INVALID);
  OK(fdh_put_glyph_image(c, 44, 16, 12, img, FDH_GLYPH_LCD_FILTER, out_rect)); CHECK(out_rect[2] == 16);
  {
    const float segs[4 * 6] = {2.0f, 2.0f, NAN, NAN, 10.0f, 2.0f, 10.0f, 2.0f, NAN, NAN, 10.0f, 9.0f, 10.0f, 9.0f, NAN, NAN, 2.0f, 9.0f, 2.0f, 9.0f, NAN, NAN, 2.0f, 2.0f};
    OK(fdh_put_glyph_outline(c, 45, 12, 12, segs, 4, 0, out_rect)); CHECK(out_rect[2] == 12 && out_rect[3] == 12);
  }
  {
    const int ws[2] = {8, 4}, hs[2] = {8, 4};
    const uint8_t* levels[2];
    levels[0] = img; levels[1] = img;
    OK(fdh_put_image_mips(c, 46, 2, ws, hs, levels, out_rect)); CHECK(out_rect[2] == 8);
  }
  if (argc > 1) {
    FILE* fp = fopen(argv[1], "rb");
    CHECK(fp != NULL);
    if (fp) {
      static uint8_t buf[1 << 20];
      const size_t n = fread(buf, 1, sizeof buf, fp);
      fclose(fp);
      OK(fdh_put_flippy(c, 47, buf, n, out_rect)); CHECK(out_rect[2] == 100 && out_rect[3] == 100); /* data/img1.flippy: 100 x 100 */
      CHECK(fdh_put_flippy(c, 48, buf, 6, out_rect) == FDH_ERR_INVALID);
    }
  }
  OK(fdh_atlas_size(c, &atlas)); CHECK(atlas == 256);
  OK(fdh_atlas_packed_area(c, &area)); CHECK(area > 0);
  OK(fdh_remove_image(c, 46));
  OK(fdh_has_image(c, 46, &flag)); CHECK(flag == 0);

  /* one frame through the per-call entry points, recorded */
  OK(fdh_record_begin(c));
  CHECK(fdh_draw_rect(c, rect, red) == FDH_ERR_INVALID); /* "ctx.beginFrame has not been called." */
  OK(fdh_begin_frame(c, 320, 240, 1, white));
  CHECK(fdh_begin_frame(c, 320, 240, 1, white) == FDH_ERR_INVALID);
  OK(fdh_save_transform(c));
  OK(fdh_translate(c, 5.0f, -4.0f));
  OK(fdh_scale(c, 1.0f, 1.0f));
  OK(fdh_rotate(c, 0.0f));
  {
    const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    OK(fdh_apply_transform(c, ident));
  }
  OK(fdh_transform_mirrors_y(c, &flag)); CHECK(flag == 0);
  for (i = 0; i < 4; i++) cols[i] = red;
  OK(fdh_draw_rounded_rect_sdf(c, rect, cols, radii, radii_y, FDH_SDF_CLIP_AA, 4.0f, 0.0f, shape0, 0, none, none, 0.5f));
  grad = solid(10, 20, 30, 255);
  grad.kind = FDH_FILL_LINEAR3; grad.axis = FDH_AXIS_Y; grad.mid = red; grad.stop = none; grad.mid_pos = 128;
  OK(fdh_draw_rounded_rect_fill(c, rect, &grad, radii, NULL, FDH_SDF_ANNULAR_AA, 3.0f, 0.0f, shape0));
  {
    const float pos[2] = {30.0f, 40.0f}, size[2] = {32.0f, 24.0f}, p0[2] = {-20.0f, 0.0f}, p1[2] = {0.0f, 15.0f}, p2[2] = {20.0f, 0.0f};
    const float quad[8] = {10, 10, 60, 14, 55, 50, 12, 44};
    OK(fdh_set_text_subpixel_shift(c, 0.25f));
    OK(fdh_draw_image(c, 42, pos, cols, size, 0));
    OK(fdh_set_text_subpixel_shift(c, 0.0f));
    OK(fdh_draw_image(c, 999, pos, cols, size, 0)); /* unknown key: warn-and-skip (glcontext.nim:1310-1315) */
    OK(fdh_draw_image_adj(c, 42, pos, red, size));
    OK(fdh_draw_msdf(c, 42, pos, red, size, 4.0f, 0.5f, 0.0f, 0, 0));
    OK(fdh_draw_quadratic_bezier_sdf(c, rect, &grad, p0, p1, p2, 3.0f, FDH_CAP_ROUND));
    OK(fdh_draw_filled_quad(c, quad, cols));
    OK(fdh_draw_rect(c, rect, red));
  }
  OK(fdh_begin_mask(c, rect, radii, radii));
  CHECK(fdh_begin_mask(c, rect, radii, radii) == FDH_ERR_INVALID);
  OK(fdh_end_mask(c));
  OK(fdh_begin_rect_mask(c, rect, radii, radii));
  OK(fdh_draw_backdrop_blur(c, rect, radii, radii, 12.0f));
  OK(fdh_pop_rect_mask(c));
  CHECK(fdh_end_frame(c) == FDH_ERR_INVALID); /* "Not all masks have been popped." */
  OK(fdh_pop_mask(c));
  OK(fdh_restore_transform(c));
  CHECK(fdh_restore_transform(c) == FDH_ERR_INVALID);
  OK(fdh_end_frame(c));
  json = fdh_record_json(c);
  CHECK(json != NULL && strstr(json, "\"draw_backdrop_blur\"") != NULL && strstr(json, "\"draw_quadratic_bezier_sdf\"") != NULL);
  CHECK(json != NULL && strstr(json, "[\"translate\",5,-4]") != NULL);
  OK(fdh_debug_record_digest(c, &d_frame)); CHECK(d_frame != 0);
  { uint32_t vu[24]; CHECK(fdh_debug_verify_upload(c, vu) == FDH_ERR_NO_DEVICE); } /* (a recorder holds no device block) */
  { uint64_t bd[8]; CHECK(fdh_debug_bin_digest(c, bd) == FDH_ERR_NO_DEVICE); }
  { int64_t held = -1; OK(fdh_debug_staging_store_bytes(7, &held)); CHECK(held == 0); } /* (an ordinal nothing has used) */

  /* whole scenes and retained scenes */
  {
    FdhFig nodes[3], moved;
    int32_t roots[3] = {0, 1, 2};
    FdhLayer layer;
    FdhScene scene;
    nodes[0] = rect_node(0, 0, 320, 240, solid(240, 240, 240, 255));
    nodes[1] = rect_node(40, 30, 120, 80, solid(200, 40, 40, 255));
    nodes[2] = rect_node(90, 70, 150, 100, solid(40, 60, 200, 180));
    nodes[2].stroke.weight = 3.0f; nodes[2].stroke.fill = solid(0, 0, 0, 255);
    memset(&layer, 0, sizeof layer); memset(&scene, 0, sizeof scene);
    layer.n_nodes = 3; layer.n_roots = 3; layer.nodes = nodes; layer.root_ids = roots;
    scene.layers = &layer; scene.n_layers = 1;
    OK(fdh_render_frame(c, &scene, 320.0f, 240.0f, 1, white));
    OK(fdh_debug_record_digest(c, &d_frame));
    OK(fdh_scene_retain(c, &scene, 320.0f, 240.0f, 1, white));
    OK(fdh_debug_record_digest(c, &d_retained)); CHECK(d_retained == d_frame);
    OK(fdh_scene_stats(c, &walked, &reused)); CHECK(walked == 3 && reused == 0);
    moved = nodes[1]; moved.box[0] += 7.0f;
    OK(fdh_scene_update_nodes(c, 0, 1, 1, &moved, NULL));
    OK(fdh_scene_render(c));
    OK(fdh_scene_stats(c, &walked, &reused)); CHECK(walked == 1 && reused == 2);
    OK(fdh_debug_record_digest(c, &d_edit)); CHECK(d_edit != d_frame);
    OK(fdh_scene_replace_root(c, 0, 1, &nodes[1], 1, NULL)); /* back to the original */
    OK(fdh_scene_render(c));
    OK(fdh_debug_record_digest(c, &d_edit)); CHECK(d_edit == d_frame);
    OK(fdh_scene_insert_root(c, 0, 3, &moved, 1, NULL));
    OK(fdh_scene_render(c));
    OK(fdh_scene_stats(c, &walked, &reused)); CHECK(walked + reused == 4);
    CHECK(fdh_scene_replace_root(c, 0, 9, &moved, 1, NULL) == FDH_ERR_INVALID);
    OK(fdh_last_upload_bytes(c, &bytes)); CHECK(bytes == 0); /* nothing is ever uploaded by a recorder */
  }

  /* host-only diagnostics */
  {
    int core[4] = {0, 0, 0, 0}, reach = 0, k_steps = 0;
    static float dense[160];
    static uint16_t frags[11 * 2 * 64 * 8];
    OK(fdh_saturated_core(rect, radii, radii, FDH_SDF_CLIP_AA, 4.0f, 0.0f, shape0, 1.2f, core));
    CHECK(core[0] > 10 && core[2] < 110 && core[1] > 20 && core[3] < 80 && core[2] > core[0] && core[3] > core[1]);
    OK(fdh_blur_weight_fragments(18.0f, 0, dense, frags, &reach, &k_steps));
    CHECK(reach == 18 && k_steps == 5); /* radius 18: 17 taps 2.25 px apart, +-8 x 2.25 = +-18 (blur.frag:11-32) */
    {
      double sum = 0.0;
      for (i = 0; i <= 2 * reach; i++) sum += dense[i];
      CHECK(fabs(sum - 1.0) < 1e-5);
    }
  }

  /* what needs pixels refuses loudly on a recorder: there is no CPU fallback behind this ABI */
  {
    void* p = NULL;
    int w = 0, h = 0;
    int64_t pitch = 0;
    float ms[2];
    FdhFrameStats st;
    static uint8_t px[320 * 240 * 4];
    CHECK(fdh_read_pixels(c, 0, 0, 0, 0, px) == FDH_ERR_NO_DEVICE);
    CHECK(fdh_frame_device_ptr(c, &p, &w, &h, &pitch) == FDH_ERR_NO_DEVICE);
    CHECK(fdh_debug_read_surface(c, 0, px) == FDH_ERR_NO_DEVICE);
    CHECK(fdh_replay(c, 1) == FDH_ERR_NO_DEVICE);
    CHECK(fdh_replay_async(c, 1) == FDH_ERR_NO_DEVICE);
    CHECK(fdh_replay_timed(c, 2, ms) == FDH_ERR_NO_DEVICE);
    CHECK(fdh_profile(c, 1) == FDH_ERR_NO_DEVICE);
    CHECK(fdh_set_stream(c, NULL) == FDH_ERR_NO_DEVICE);
    {
      int y0 = -1, y1 = -1, r, covered = 0;
      uint8_t id[FDH_COMM_ID_BYTES];
      for (r = 0; r < 3; r++) { OK(fdh_stripe_rows(2160, 3, r, &y0, &y1)); CHECK(y0 == covered && y0 % 8 == 0 && y1 > y0); covered = y1; }
      CHECK(covered == 2160);
      CHECK(fdh_stripe_rows(2160, 3, 3, &y0, &y1) == FDH_ERR_INVALID);
      memset(id, 0, sizeof id);
      CHECK(fdh_comm_init(c, id, 0, 1) == FDH_ERR_NO_DEVICE);     /* a communicator lives on a device */
      CHECK(fdh_gather_stripes(c, 0, NULL) == FDH_ERR_NO_DEVICE);
      CHECK(fdh_gather_frames(c, 0, NULL) == FDH_ERR_NO_DEVICE);
      CHECK(fdh_comm_share(c, c) == FDH_ERR_NO_DEVICE);
      OK(fdh_comm_destroy(c));                                    /* nothing to destroy: fine */
      r = fdh_comm_unique_id(id);                                 /* needs librccl and a device: either answer is an answer, no crash */
      CHECK(r == FDH_OK || r == FDH_ERR_UNSUPPORTED || r == FDH_ERR_HIP);
    }
    OK(fdh_set_stripe(c, 0, 0));
    { int64_t ns[12]; OK(fdh_debug_host_times(c, ns)); CHECK(ns[0] >= 0); }
    { int lcd = -1, rank = -1, world = -1; OK(fdh_set_text_lcd_filtering(c, 1)); OK(fdh_get_text_lcd_filtering(c, &lcd)); CHECK(lcd == 1); OK(fdh_set_text_lcd_filtering(c, 0));
      OK(fdh_comm_info(c, &rank, &world)); CHECK(rank == 0 && world == 1); }
    { int th = -1; int64_t groups = -1; OK(fdh_set_walk_threads(c, 2)); OK(fdh_walk_stats(c, &th, &groups)); CHECK(th == 2 && groups >= 0); OK(fdh_set_walk_threads(c, -1)); }
    { int64_t culled = -1; OK(fdh_set_cull(c, 0)); OK(fdh_set_cull(c, 1)); OK(fdh_culled_draws(c, &culled)); CHECK(culled >= 0); }
    OK(fdh_set_blur_route(c, 1)); OK(fdh_set_blur_route(c, -1));
    OK(fdh_sync(c));
    OK(fdh_flush(c));
    OK(fdh_get_frame_stats(c, &st));
    OK(fdh_reset_atlas(c, 0));
    OK(fdh_has_image(c, 42, &flag)); CHECK(flag == 0);
  }
  OK(fdh_destroy(c));
  {
    FdhContext* gpu = NULL; /* a real context needs a gfx950 device; without one the call must say so, not fall back */
    const int rc = fdh_create(&gpu, 256, 1.0f, 0, 0);
    CHECK(rc == FDH_OK || rc == FDH_ERR_NO_DEVICE);
    if (rc == FDH_OK) OK(fdh_destroy(gpu));
  }
  if (failures) { printf("abi_smoke: %d check(s) failed\n", failures); return 1; }
  printf("abi_smoke: OK\n");
  return 0;
}
