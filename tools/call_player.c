/* call_player.c -- a C99 driver of libfigdraw_hip.so's per-call BackendContext path.
 *
 * The reference's renderer talks to its backend one method call at a time (figbackend.nim:468-634; ~710 calls for the bench
 * frame) and its benchmark loop times exactly that, `renderFrame(renders, frameSize)` per frame
 * (examples/windy_non_clip_benchmark.nim:113-147).  This file plays recorded call streams -- what fdh_record_json returns,
 * packed to binary by figdraw_amd/call_stream.py -- through the same C entry points a Nim shim would bind, from plain C with
 * no Python between the calls, so that the per-call path can be timed (bench.py `per_call_path`) and tested (tests/).
 * Built with `gcc -std=c99 -Wall -Werror -Iinclude`: it is also the proof that include/figdraw_hip.h is a C header.
 *
 * Stream format: little-endian 32-bit words.  Each call = opcode word + its arguments (floats as IEEE-754 bit patterns,
 * colours as one RGBA8 word, image keys as two words lo, hi).  Opcodes below; the stream of one frame starts with
 * OP_BEGIN_FRAME and ends with OP_END_FRAME.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

#include "figdraw_hip.h"

enum {
  OP_BEGIN_FRAME = 1, OP_END_FRAME, OP_SAVE_TRANSFORM, OP_RESTORE_TRANSFORM, OP_TRANSLATE, OP_ROTATE, OP_SCALE, OP_APPLY_TRANSFORM,
  OP_SET_AA, OP_DRAW_ROUNDED_RECT_SDF, OP_DRAW_IMAGE, OP_DRAW_MSDF, OP_DRAW_BACKDROP_BLUR, OP_BEGIN_MASK, OP_END_MASK, OP_POP_MASK,
  OP_BEGIN_RECT_MASK, OP_POP_RECT_MASK, OP_DRAW_QUADRATIC_BEZIER_SDF, OP_DRAW_FILLED_QUAD, OP_DRAW_RECT, OP_SET_SUBPIXEL_SHIFT,
  OP_DRAW_IMAGE_ADJ
};

static float f32(uint32_t w) { float f; memcpy(&f, &w, 4); return f; }
static FdhColor rgba8(uint32_t w) { FdhColor c; c.r = (uint8_t)w; c.g = (uint8_t)(w >> 8); c.b = (uint8_t)(w >> 16); c.a = (uint8_t)(w >> 24); return c; }
static void floats(const uint32_t* w, int n, float* out) { int i; for (i = 0; i < n; i++) out[i] = f32(w[i]); }
static FdhFill fill_of(const uint32_t* w) { /* kind, axis, start, mid, stop, mid_pos */
  FdhFill f;
  memset(&f, 0, sizeof f);
  f.kind = (int32_t)w[0]; f.axis = (int32_t)w[1]; f.start = rgba8(w[2]); f.mid = rgba8(w[3]); f.stop = rgba8(w[4]); f.mid_pos = (uint8_t)w[5];
  return f;
}

/* Play one frame's stream on one context.  Returns 0, the failing call's FdhStatus, or -100 for a malformed stream. */
FDH_API int fdh_play_calls(FdhContext* ctx, const uint32_t* w, size_t n_words, int width, int height) {
  size_t i = 0;
  while (i < n_words) {
    const uint32_t op = w[i++];
    const uint32_t* a = w + i;
    int rc = 0, used = 0;
    float r4[4], x4[4], y4[4], p2[2], q2[2], s2[2], m16[16], v8[8];
    FdhColor cols[4];
    int k;
    switch (op) {
      case OP_BEGIN_FRAME: used = 5; if (i + used > n_words) return -100; floats(a + 1, 4, r4); rc = fdh_begin_frame(ctx, width, height, (int)a[0], r4); break;
      case OP_END_FRAME: rc = fdh_end_frame(ctx); break;
      case OP_SAVE_TRANSFORM: rc = fdh_save_transform(ctx); break;
      case OP_RESTORE_TRANSFORM: rc = fdh_restore_transform(ctx); break;
      case OP_TRANSLATE: used = 2; if (i + used > n_words) return -100; rc = fdh_translate(ctx, f32(a[0]), f32(a[1])); break;
      case OP_ROTATE: used = 1; if (i + used > n_words) return -100; rc = fdh_rotate(ctx, f32(a[0])); break;
      case OP_SCALE: used = 2; if (i + used > n_words) return -100; rc = fdh_scale(ctx, f32(a[0]), f32(a[1])); break;
      case OP_APPLY_TRANSFORM: used = 16; if (i + used > n_words) return -100; floats(a, 16, m16); rc = fdh_apply_transform(ctx, m16); break;
      case OP_SET_AA: used = 1; if (i + used > n_words) return -100; rc = fdh_set_aa_factor(ctx, f32(a[0])); break;
      case OP_DRAW_ROUNDED_RECT_SDF: /* rect4 cols4 rx4 ry4 mode factor spread shape2 fill_mode mid stop mid_pos */
        used = 25; if (i + used > n_words) return -100;
        floats(a, 4, r4); for (k = 0; k < 4; k++) cols[k] = rgba8(a[4 + k]);
        floats(a + 8, 4, x4); floats(a + 12, 4, y4); floats(a + 19, 2, s2);
        rc = fdh_draw_rounded_rect_sdf(ctx, r4, cols, x4, y4, (int)a[16], f32(a[17]), f32(a[18]), s2, (int)a[21], rgba8(a[22]), rgba8(a[23]), f32(a[24]));
        break;
      case OP_DRAW_IMAGE: /* key lo hi, pos2, cols4, size2, flip */
        used = 11; if (i + used > n_words) return -100;
        floats(a + 2, 2, p2); for (k = 0; k < 4; k++) cols[k] = rgba8(a[4 + k]); floats(a + 8, 2, s2);
        rc = fdh_draw_image(ctx, (int64_t)((uint64_t)a[0] | ((uint64_t)a[1] << 32)), p2, cols, s2, (int)a[10]);
        break;
      case OP_DRAW_MSDF: /* key lo hi, pos2, color, size2, px_range, threshold, stroke, mtsdf, flip */
        used = 12; if (i + used > n_words) return -100;
        floats(a + 2, 2, p2); floats(a + 5, 2, s2);
        rc = fdh_draw_msdf(ctx, (int64_t)((uint64_t)a[0] | ((uint64_t)a[1] << 32)), p2, rgba8(a[4]), s2, f32(a[7]), f32(a[8]), f32(a[9]), (int)a[10], (int)a[11]);
        break;
      case OP_DRAW_BACKDROP_BLUR: used = 13; if (i + used > n_words) return -100; floats(a, 4, r4); floats(a + 4, 4, x4); floats(a + 8, 4, y4);
        rc = fdh_draw_backdrop_blur(ctx, r4, x4, y4, f32(a[12])); break;
      case OP_BEGIN_MASK: used = 12; if (i + used > n_words) return -100; floats(a, 4, r4); floats(a + 4, 4, x4); floats(a + 8, 4, y4); rc = fdh_begin_mask(ctx, r4, x4, y4); break;
      case OP_END_MASK: rc = fdh_end_mask(ctx); break;
      case OP_POP_MASK: rc = fdh_pop_mask(ctx); break;
      case OP_BEGIN_RECT_MASK: used = 12; if (i + used > n_words) return -100; floats(a, 4, r4); floats(a + 4, 4, x4); floats(a + 8, 4, y4); rc = fdh_begin_rect_mask(ctx, r4, x4, y4); break;
      case OP_POP_RECT_MASK: rc = fdh_pop_rect_mask(ctx); break;
      case OP_DRAW_QUADRATIC_BEZIER_SDF: { /* rect4 fill6 p0 p1 p2 weight cap */
        FdhFill f;
        used = 18; if (i + used > n_words) return -100;
        floats(a, 4, r4); f = fill_of(a + 4); floats(a + 10, 2, p2); floats(a + 12, 2, q2); floats(a + 14, 2, s2);
        rc = fdh_draw_quadratic_bezier_sdf(ctx, r4, &f, p2, q2, s2, f32(a[16]), (int)a[17]);
        break;
      }
      case OP_DRAW_FILLED_QUAD: used = 12; if (i + used > n_words) return -100; floats(a, 8, v8); for (k = 0; k < 4; k++) cols[k] = rgba8(a[8 + k]); rc = fdh_draw_filled_quad(ctx, v8, cols); break;
      case OP_DRAW_RECT: used = 5; if (i + used > n_words) return -100; floats(a, 4, r4); rc = fdh_draw_rect(ctx, r4, rgba8(a[4])); break;
      case OP_DRAW_IMAGE_ADJ: /* key lo hi, pos2, color, size2 */
        used = 7; if (i + used > n_words) return -100;
        floats(a + 2, 2, p2); floats(a + 5, 2, s2);
        rc = fdh_draw_image_adj(ctx, (int64_t)((uint64_t)a[0] | ((uint64_t)a[1] << 32)), p2, rgba8(a[4]), s2);
        break;
      case OP_SET_SUBPIXEL_SHIFT: used = 1; if (i + used > n_words) return -100; rc = fdh_set_text_subpixel_shift(ctx, f32(a[0])); break;
      default: return -100;
    }
    if (rc != 0) return rc;
    i += (size_t)used;
  }
  return 0;
}

/* `frames` frames, frame k = stream k % n_streams on context k % n_ctx; then every context is waited for.  The wall time of the
 * whole loop INCLUDING those waits goes to *seconds.  What the reference's benchmark loop times per frame. */
FDH_API int fdh_play_frames(FdhContext* const* ctxs, int n_ctx, const uint32_t* const* streams, const size_t* n_words, int n_streams,
                            int frames, int width, int height, double* seconds) {
  struct timespec t0, t1;
  int k, rc = 0;
  if (n_ctx <= 0 || n_streams <= 0) return -100;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (k = 0; k < frames && rc == 0; k++) rc = fdh_play_calls(ctxs[k % n_ctx], streams[k % n_streams], n_words[k % n_streams], width, height);
  for (k = 0; k < n_ctx; k++) { const int r2 = fdh_sync(ctxs[k]); if (rc == 0) rc = r2; }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return rc;
}

/* The whole-scene entry timed the same way: frame k = fdh_render_frame(scene k % n_scenes) on context k % n_ctx. */
FDH_API int fdh_play_scenes(FdhContext* const* ctxs, int n_ctx, const FdhScene* const* scenes, int n_scenes, int frames, float frame_w,
                            float frame_h, double* seconds) {
  struct timespec t0, t1;
  const float white[4] = {1.0f, 1.0f, 1.0f, 1.0f};
  int k, rc = 0;
  if (n_ctx <= 0 || n_scenes <= 0) return -100;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (k = 0; k < frames && rc == 0; k++) rc = fdh_render_frame(ctxs[k % n_ctx], scenes[k % n_scenes], frame_w, frame_h, 1, white);
  for (k = 0; k < n_ctx; k++) { const int r2 = fdh_sync(ctxs[k]); if (rc == 0) rc = r2; }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return rc;
}

/* The same frames with one host thread per group of contexts: thread t drives the contexts c with c % n_threads == t (a context is
 * only ever driven by its one thread; frame k still goes to context k % n_ctx with scene k % n_scenes).  What an application with
 * several windows does -- one render thread per window -- and what takes the tree walk of frame k + 1 off the path of frame k's
 * neighbours: with one calling thread the walks of all contexts' frames queue up behind each other. */
struct fdh_play_gate { pthread_mutex_t mu; pthread_cond_t cv; int go; };
struct fdh_play_thread {
  FdhContext* const* ctxs;
  const FdhScene* const* scenes;
  int n_ctx, n_scenes, frames, n_threads, t, rc;
  float frame_w, frame_h;
  struct fdh_play_gate* gate;
};
static void* fdh_play_thread_main(void* arg) {
  struct fdh_play_thread* a = (struct fdh_play_thread*)arg;
  const float white[4] = {1.0f, 1.0f, 1.0f, 1.0f};
  int k, rc = 0, go;
  pthread_mutex_lock(&a->gate->mu);
  while ((go = a->gate->go) == 0) pthread_cond_wait(&a->gate->cv, &a->gate->mu);
  pthread_mutex_unlock(&a->gate->mu);
  if (go < 0) return NULL;  /* a sibling could not be started: nothing is drawn */
  for (k = 0; k < a->frames && rc == 0; k++)
    if ((k % a->n_ctx) % a->n_threads == a->t) rc = fdh_render_frame(a->ctxs[k % a->n_ctx], a->scenes[k % a->n_scenes], a->frame_w, a->frame_h, 1, white);
  for (k = a->t; k < a->n_ctx; k += a->n_threads) { const int r2 = fdh_sync(a->ctxs[k]); if (rc == 0) rc = r2; }
  a->rc = rc;
  return NULL;
}
FDH_API int fdh_play_scenes_threads(FdhContext* const* ctxs, int n_ctx, const FdhScene* const* scenes, int n_scenes, int frames, float frame_w,
                                    float frame_h, int n_threads, double* seconds) {
  enum { MAX_THREADS = 64 };
  struct fdh_play_thread args[MAX_THREADS];
  pthread_t th[MAX_THREADS];
  struct fdh_play_gate gate;
  struct timespec t0, t1;
  int t, made = 0, rc = 0;
  if (n_ctx <= 0 || n_scenes <= 0 || n_threads <= 0) return -100;
  if (n_threads > n_ctx) n_threads = n_ctx;
  if (n_threads > MAX_THREADS) n_threads = MAX_THREADS;
  if (pthread_mutex_init(&gate.mu, NULL) != 0) return -101;
  if (pthread_cond_init(&gate.cv, NULL) != 0) { pthread_mutex_destroy(&gate.mu); return -101; }
  gate.go = 0;
  for (t = 0; t < n_threads; t++) {
    args[t].ctxs = ctxs; args[t].scenes = scenes; args[t].n_ctx = n_ctx; args[t].n_scenes = n_scenes; args[t].frames = frames;
    args[t].n_threads = n_threads; args[t].t = t; args[t].rc = 0; args[t].frame_w = frame_w; args[t].frame_h = frame_h; args[t].gate = &gate;
    if (pthread_create(&th[t], NULL, fdh_play_thread_main, &args[t]) != 0) break;
    made++;
  }
  if (made != n_threads) rc = -102;
  clock_gettime(CLOCK_MONOTONIC, &t0);  /* the threads exist and wait at the gate: the timed region is frames + syncs + joins */
  pthread_mutex_lock(&gate.mu);
  gate.go = made == n_threads ? 1 : -1;
  pthread_cond_broadcast(&gate.cv);
  pthread_mutex_unlock(&gate.mu);
  for (t = 0; t < made; t++) { pthread_join(th[t], NULL); if (rc == 0) rc = args[t].rc; }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  pthread_cond_destroy(&gate.cv);
  pthread_mutex_destroy(&gate.mu);
  if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return rc;
}
