// Standalone probe (no figdraw): does a VALU kernel's packed-f32 arithmetic stay exact while ANOTHER kernel keeps the matrix
// pipe of the same SIMDs busy?  Background: DESIGN.md section 4, "cross-context anomaly".  With several figdraw contexts in
// flight, pixels of rows 6 and 7 of a compositor strip (lanes 48-63 of the wave) came out with their red and blue channels
// (the .x halves of the packed colour pairs) short of a blend term -- but only while another context's matrix-pipe blur
// passes were running.  This program reproduces the two ingredients in isolation:
//   victim    one-wave workgroups, each lane runs v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (inline assembly, so the
//             compiler can neither fold nor re-pack them) next to the same arithmetic done with scalar v_fma_f32 /
//             v_mul_f32 / v_add_f32, and counts bitwise mismatches per (component, lane);
//   aggressor one-wave workgroups issuing v_mfma_f32_32x32x16_f16 back to back (or, as a control, v_fma_f32 only).
// Both run on their own streams at the same time; the victim alone and the victim beside the VALU aggressor are controls.
//   hipcc --offload-arch=gfx950 -O3 pk_vs_mfma.hip -o pk_vs_mfma && ./pk_vs_mfma [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-result"

typedef float f2 __attribute__((ext_vector_type(2)));
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// hist[op][component][lane], sample[op][8]: {block, iteration, lane, comp, got, want, ...}
struct Report { unsigned hist[16][2][64]; unsigned samples[16][8]; unsigned n[16]; unsigned when[16][4]; };  // when: iteration 0, 1, 2..15, later

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f2 pk_fma_s(f2 a, f2 b_uniform, f2 c) { f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c)); return r; }
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) { f2 r; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) { f2 r; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float s_mul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_add(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ __forceinline__ void note(Report* rep, int op, int comp, int it, float got, float want) {
  atomicAdd(&rep->hist[op][comp][threadIdx.x], 1u);
  atomicAdd(&rep->when[op][it == 0 ? 0 : it == 1 ? 1 : it < 16 ? 2 : 3], 1u);
  if (atomicAdd(&rep->n[op], 1u) == 0u) {
    unsigned* s = rep->samples[op];
    s[0] = blockIdx.x; s[1] = (unsigned)it; s[2] = threadIdx.x; s[3] = (unsigned)comp; s[4] = __float_as_uint(got); s[5] = __float_as_uint(want);
  }
}

// the compositor's blend has this shape: F = rint(fma(F, 1 - sa, src * 255 * sa)) on (r, g) and (b, a) pairs
__global__ __launch_bounds__(64) void k_victim(Report* rep, int iters, float ua, float ub) {
  const int lane = threadIdx.x;
  f2 x = {1.0f + 0.001f * lane, 2.0f + 0.003f * lane + 0.0001f * (blockIdx.x & 255)};
  f2 y = {0.75f + 0.0005f * lane, 0.5f + 0.0007f * lane};
  f2 z = {3.0f + lane, 100.0f - lane};
  const f2 u = {ua, ub};  // wave-uniform (SGPR pair)
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    // 0: pk_fma vgpr
    { const f2 r = pk_fma(x, y, z); const float wx = s_fma(x.x, y.x, z.x), wy = s_fma(x.y, y.y, z.y);
      if (__float_as_uint(r.x) != __float_as_uint(wx)) note(rep, 0, 0, it, r.x, wx);
      if (__float_as_uint(r.y) != __float_as_uint(wy)) note(rep, 0, 1, it, r.y, wy); }
    // 1: pk_fma with an SGPR pair as multiplier
    { const f2 r = pk_fma_s(x, u, z); const float wx = s_fma(x.x, u.x, z.x), wy = s_fma(x.y, u.y, z.y);
      if (__float_as_uint(r.x) != __float_as_uint(wx)) note(rep, 1, 0, it, r.x, wx);
      if (__float_as_uint(r.y) != __float_as_uint(wy)) note(rep, 1, 1, it, r.y, wy); }
    // 2: pk_mul
    { const f2 r = pk_mul(x, y); const float wx = s_mul(x.x, y.x), wy = s_mul(x.y, y.y);
      if (__float_as_uint(r.x) != __float_as_uint(wx)) note(rep, 2, 0, it, r.x, wx);
      if (__float_as_uint(r.y) != __float_as_uint(wy)) note(rep, 2, 1, it, r.y, wy); }
    // 3: pk_add
    { const f2 r = pk_add(x, z); const float wx = s_add(x.x, z.x), wy = s_add(x.y, z.y);
      if (__float_as_uint(r.x) != __float_as_uint(wx)) note(rep, 3, 0, it, r.x, wx);
      if (__float_as_uint(r.y) != __float_as_uint(wy)) note(rep, 3, 1, it, r.y, wy); }
    // 4: pk_fma, low half reads the HIGH register of the multiplier pair (op_sel:[0,1,0]) -- the form the compositor's
    //    packed edge path uses for pixels 1 and 3 of a lane
    { f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
      const float wx = s_fma(x.x, y.y, z.x), wy = s_fma(x.y, y.y, z.y);
      if (__float_as_uint(r.x) != __float_as_uint(wx)) note(rep, 4, 0, it, r.x, wx);
      if (__float_as_uint(r.y) != __float_as_uint(wy)) note(rep, 4, 1, it, r.y, wy); }
    // 8: the same, result written over the addend pair (dst = src2)
    { f2 r = z; asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(r) : "v"(x), "v"(y));
      const float wx = s_fma(x.x, y.y, z.x), wy = s_fma(x.y, y.y, z.y);
      if (__float_as_uint(r.x) != __float_as_uint(wx)) note(rep, 8, 0, it, r.x, wx);
      if (__float_as_uint(r.y) != __float_as_uint(wy)) note(rep, 8, 1, it, r.y, wy); }
    // 9: the same, result written over the multiplier pair whose high register both halves read (dst = src1)
    { f2 r = y; asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0]" : "+v"(r) : "v"(x), "v"(z));
      const float wx = s_fma(x.x, y.y, z.x), wy = s_fma(x.y, y.y, z.y);
      if (__float_as_uint(r.x) != __float_as_uint(wx)) note(rep, 9, 0, it, r.x, wx);
      if (__float_as_uint(r.y) != __float_as_uint(wy)) note(rep, 9, 1, it, r.y, wy); }
    // 10: a chain as in the edge path: multiply by the high half, then two in-place fmas back to back
    { f2 t, u2 = y; asm volatile("v_pk_mul_f32 %0, %2, %3 op_sel:[0,1]\n\tv_pk_fma_f32 %0, %4, %1, %0 op_sel:[0,1,0]\n\tv_pk_fma_f32 %1, %5, %1, %3 op_sel:[0,1,0]"
                                 : "=&v"(t), "+v"(u2) : "v"(x), "v"(z), "v"(y), "v"(x));
      const float m0 = s_mul(x.x, z.y), m1 = s_mul(x.y, z.y);
      const float wx = s_fma(y.x, y.y, m0), wy = s_fma(y.y, y.y, m1);
      const float vx = s_fma(x.x, y.y, z.x), vy = s_fma(x.y, y.y, z.y);
      if (__float_as_uint(t.x) != __float_as_uint(wx)) note(rep, 10, 0, it, t.x, wx);
      if (__float_as_uint(t.y) != __float_as_uint(wy)) note(rep, 10, 1, it, t.y, wy);
      if (__float_as_uint(u2.x) != __float_as_uint(vx)) note(rep, 11, 0, it, u2.x, vx);
      if (__float_as_uint(u2.y) != __float_as_uint(vy)) note(rep, 11, 1, it, u2.y, vy); }
    // 12: low half reads the high register of the FIRST operand (op_sel:[1,0,0])
    { f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
      const float wx = s_fma(x.y, y.x, z.x), wy = s_fma(x.y, y.y, z.y);
      if (__float_as_uint(r.x) != __float_as_uint(wx)) note(rep, 12, 0, it, r.x, wx);
      if (__float_as_uint(r.y) != __float_as_uint(wy)) note(rep, 12, 1, it, r.y, wy); }
    // 13: low half reads the high register of the ADDEND (op_sel:[0,0,1])
    { f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
      const float wx = s_fma(x.x, y.x, z.y), wy = s_fma(x.y, y.y, z.y);
      if (__float_as_uint(r.x) != __float_as_uint(wx)) note(rep, 13, 0, it, r.x, wx);
      if (__float_as_uint(r.y) != __float_as_uint(wy)) note(rep, 13, 1, it, r.y, wy); }
    // 14: v_pk_mul_f32, low half reads the high register of the second operand (op_sel:[0,1])
    { f2 r; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(y));
      const float wx = s_mul(x.x, y.y), wy = s_mul(x.y, y.y);
      if (__float_as_uint(r.x) != __float_as_uint(wx)) note(rep, 14, 0, it, r.x, wx);
      if (__float_as_uint(r.y) != __float_as_uint(wy)) note(rep, 14, 1, it, r.y, wy); }
    // 15: high half reads the LOW register of the second operand (op_sel_hi:[1,0,1]) -- the broadcast used for pixels 0 and 2
    { f2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(y), "v"(z));
      const float wx = s_fma(x.x, y.x, z.x), wy = s_fma(x.y, y.x, z.y);
      if (__float_as_uint(r.x) != __float_as_uint(wx)) note(rep, 15, 0, it, r.x, wx);
      if (__float_as_uint(r.y) != __float_as_uint(wy)) note(rep, 15, 1, it, r.y, wy); }
    // 5: scalar control: v_fma_f32 twice must agree with itself
    { const float a = s_fma(x.x, y.x, z.x), b = s_fma(x.x, y.x, z.x);
      if (__float_as_uint(a) != __float_as_uint(b)) note(rep, 5, 0, it, a, b); }
    x = {x.x * 1.0000001f + 0.000001f, x.y * 0.9999999f + 0.000002f};
    z = {z.x + 0.5f, z.y - 0.25f};
  }
  if (x.x == 12345.678f) rep->samples[7][7] = 1;  // keep x live
}


// Victim 2: the compositor's core-strip blend as hipcc compiles it (k_composite.hip, k_composite_tiles, LE_PLAIN entries): one
// colour per step arrives through a scalar load, is unpacked with v_cvt_f32_ubyte{0..3} from the SGPR, scaled with packed
// multiplies (op_sel / op_sel_hi broadcasts) and blended into four pixels per lane with v_pk_fma_f32 + v_rndne_f32.  Every
// lane computes the SAME values (the inputs are wave-uniform), so a lane whose result differs from lane 0's is wrong:
// hist[6 + (pixel slot > 1)][component & 1] counts them per lane, samples[6] keeps the first.
struct F4 { float x, y, z, w; };
__device__ __forceinline__ F4 unpack255(unsigned c) { F4 r; r.x = (float)(c & 255u); r.y = (float)((c >> 8) & 255u); r.z = (float)((c >> 16) & 255u); r.w = (float)(c >> 24); return r; }
__device__ __forceinline__ void blend_pre(F4& F, f2 c_rg, f2 c_ba, float ia) {
  f2 xy = {F.x, F.y}, zw = {F.z, F.w};
  const f2 ia2 = {ia, ia};
  xy = __builtin_elementwise_fma(xy, ia2, c_rg);
  zw = __builtin_elementwise_fma(zw, ia2, c_ba);
  F.x = __builtin_rintf(xy.x); F.y = __builtin_rintf(xy.y); F.z = __builtin_rintf(zw.x); F.w = __builtin_rintf(zw.y);
}
__global__ __launch_bounds__(64, 5) void k_victim2(Report* rep, const unsigned* __restrict__ colours, int n_colours, int iters) {
  const float inv255 = 1.0f / 255.0f;
  F4 F[4];
  for (int k = 0; k < 4; k++) F[k] = unpack255(0xffffffffu);
  unsigned bad_any = 0;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    const int ci = __builtin_amdgcn_readfirstlane((it * 7 + (int)blockIdx.x) % n_colours);
    const F4 c0 = unpack255(colours[ci]);  // s_load_dword + v_cvt_f32_ubyte* from the SGPR
    const float sa = c0.w * inv255, A = 255.0f * sa, ia = 1.0f - sa;
    const f2 c_rg = {c0.x * inv255 * A, c0.y * inv255 * A}, c_ba = {c0.z * inv255 * A, A};
    blend_pre(F[0], c_rg, c_ba, ia); blend_pre(F[1], c_rg, c_ba, ia); blend_pre(F[2], c_rg, c_ba, ia); blend_pre(F[3], c_rg, c_ba, ia);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float v[4] = {F[k].x, F[k].y, F[k].z, F[k].w};
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const unsigned mine = __float_as_uint(v[c]), first = __builtin_amdgcn_readfirstlane(mine);
        if (mine != first) {
          atomicAdd(&rep->hist[6 + (k >> 1)][c & 1][threadIdx.x], 1u);
          if (atomicAdd(&rep->n[6], 1u) == 0u) { unsigned* s = rep->samples[6]; s[0] = blockIdx.x; s[1] = (unsigned)it; s[2] = threadIdx.x; s[3] = (unsigned)(4 * k + c); s[4] = mine; s[5] = first; }
          bad_any = 1;
        }
      }
    }
    if (bad_any) { for (int k = 0; k < 4; k++) F[k] = unpack255(0xffffffffu); bad_any = 0; }  // resynchronise the lanes
  }
  if (F[0].x == 12345.678f) rep->samples[7][7] = 1;
}

__global__ __launch_bounds__(64) void k_mfma(float* out, int iters) {
  h8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
  f32x16 acc[4];
  for (int c = 0; c < 4; c++) for (int e = 0; e < 16; e++) acc[c][e] = 0.0f;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.0f;
  for (int c = 0; c < 4; c++) for (int e = 0; e < 16; e++) s += acc[c][e];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

__global__ __launch_bounds__(64) void k_valu(float* out, int iters) {  // control aggressor: same shape, no matrix pipe
  float acc[8];
  for (int i = 0; i < 8; i++) acc[i] = threadIdx.x + i;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = s_fma(acc[i], 1.0000001f, 0.5f);
  }
  float s = 0.0f;
  for (int i = 0; i < 8; i++) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

static void print_report2(const Report& r) {
  for (int half = 0; half < 2; half++) {
    unsigned long long tx = 0, ty = 0;
    for (int l = 0; l < 64; l++) { tx += r.hist[6 + half][0][l]; ty += r.hist[6 + half][1][l]; }
    printf("  compositor blend, pixel slots %d-%d    lanes differing from lane 0: red/blue (.x of the pairs) %llu  green/alpha (.y) %llu\n", 2 * half, 2 * half + 1, tx, ty);
    if (tx + ty == 0) continue;
    for (int comp = 0; comp < 2; comp++) {
      printf("    .%c by 16-lane quarter:", comp ? 'y' : 'x');
      for (int q = 0; q < 4; q++) { unsigned long long t = 0; for (int l = 16 * q; l < 16 * q + 16; l++) t += r.hist[6 + half][comp][l]; printf(" %llu", t); }
      printf("\n");
    }
  }
  if (r.n[6]) {
    const unsigned* s = r.samples[6];
    float got, want; memcpy(&got, &s[4], 4); memcpy(&want, &s[5], 4);
    printf("    first: block %u step %u lane %u pixel slot %u channel %u got %.9g want (lane 0) %.9g\n", s[0], s[1], s[2], s[3] >> 2, s[3] & 3, got, want);
  }
}
static void print_report(const char* title, const Report& r) {
  static const char* names[16] = {"v_pk_fma_f32 vgpr", "v_pk_fma_f32 sgpr multiplier", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32 op_sel:[0,1,0]", "v_fma_f32 (control)", "", "",
                                  "... dst = addend pair", "... dst = multiplier pair", "chain: first fma", "chain: second fma",
                                  "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_fma_f32 op_sel_hi:[1,0,1]"};
  printf("== %s\n", title);
  for (int op = 0; op < 16; op++) {
    if (op == 6 || op == 7) continue;
    unsigned long long tx = 0, ty = 0;
    for (int l = 0; l < 64; l++) { tx += r.hist[op][0][l]; ty += r.hist[op][1][l]; }
    printf("  %-32s mismatches .x %llu  .y %llu\n", names[op], tx, ty);
    if (tx + ty == 0) continue;
    for (int comp = 0; comp < 2; comp++) {
      printf("    .%c by 16-lane quarter:", comp ? 'y' : 'x');
      for (int q = 0; q < 4; q++) { unsigned long long t = 0; for (int l = 16 * q; l < 16 * q + 16; l++) t += r.hist[op][comp][l]; printf(" %llu", t); }
      printf("\n");
    }
    printf("    by loop iteration: first %u, second %u, 3rd-16th %u, later %u\n", r.when[op][0], r.when[op][1], r.when[op][2], r.when[op][3]);
    const unsigned* s = r.samples[op];
    float got, want; memcpy(&got, &s[4], 4); memcpy(&want, &s[5], 4);
    printf("    first: block %u iteration %u lane %u .%c got %.9g (%08x) want %.9g (%08x)\n", s[0], s[1], s[2], s[3] ? 'y' : 'x', got, s[4], want, s[5]);
  }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20;
  const int victim_grid = 65536, victim_iters = 64, agg_grid = 4096, agg_iters = 100000, victims_per_round = 40;  // short victims: the misreads happen in a wave's first pass through the code
  Report* d_rep; float* d_out;
  CK(hipMalloc(&d_rep, sizeof(Report)));
  CK(hipMalloc(&d_out, (size_t)agg_grid * 64 * 4));
  unsigned* d_col; CK(hipMalloc(&d_col, 256 * 4));
  { unsigned hc[256]; unsigned x = 12345u; for (int i = 0; i < 256; i++) { x = x * 1664525u + 1013904223u; hc[i] = x | ((i & 3) == 0 ? 0xff000000u : 0u); } CK(hipMemcpy(d_col, hc, sizeof hc, hipMemcpyHostToDevice)); }
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Report h;
  for (int mode = 0; mode < 3; mode++) {  // 0: victim alone, 1: beside the VALU aggressor, 2: beside the matrix-pipe aggressor
    CK(hipMemset(d_rep, 0, sizeof(Report)));
    float ms_total = 0.0f, agg_ms = 0.0f;
    for (int r = 0; r < rounds; r++) {
      hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
      CK(hipEventRecord(a0, s2));
      if (mode == 1) hipLaunchKernelGGL(k_valu, dim3(agg_grid), dim3(64), 0, s2, d_out, agg_iters / 4);
      if (mode == 2) hipLaunchKernelGGL(k_mfma, dim3(agg_grid), dim3(64), 0, s2, d_out, agg_iters);
      CK(hipEventRecord(a1, s2));
      CK(hipEventRecord(e0, s1));
      for (int v = 0; v < victims_per_round; v++) {
        hipLaunchKernelGGL(k_victim, dim3(victim_grid), dim3(64), 0, s1, d_rep, victim_iters / 8, 0.625f, 1.375f);
        hipLaunchKernelGGL(k_victim2, dim3(victim_grid), dim3(64), 0, s1, d_rep, d_col, 256, victim_iters);
      }
      CK(hipEventRecord(e1, s1));
      CK(hipStreamSynchronize(s1));
      CK(hipStreamSynchronize(s2));
      float ms, ams; CK(hipEventElapsedTime(&ms, e0, e1)); ms_total += ms; CK(hipEventElapsedTime(&ams, a0, a1)); agg_ms += ams;
      CK(hipEventDestroy(a0)); CK(hipEventDestroy(a1));
    }
    CK(hipMemcpy(&h, d_rep, sizeof(Report), hipMemcpyDeviceToHost));
    char title[160];
    snprintf(title, sizeof title, "%s (%d rounds, victims %.2f ms per round, aggressor %.2f ms)", mode == 0 ? "victim alone" : mode == 1 ? "victim beside a VALU-only kernel" : "victim beside a v_mfma_f32_32x32x16_f16 kernel", rounds, ms_total / rounds, agg_ms / rounds);
    print_report(title, h);
    print_report2(h);
  }
  return 0;
}
