// VALU issue-rate microbenchmark for gfx950: v_fma_f32 vs v_pk_fma_f32 (VGPR and SGPR multiplier), v_cvt_f32_ubyte, v_rndne.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIter = 4096;
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  f2 acc[8];
  for (int i = 0; i < 8; i++) acc[i] = {(float)threadIdx.x, (float)i};
  f2 m = {a + threadIdx.x * 1e-9f, b};
  f2 ms = {a, b};  // uniform
  for (int it = 0; it < kIter; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) { acc[i].x = __builtin_fmaf(acc[i].x, m.x, m.y); acc[i].y = __builtin_fmaf(acc[i].y, m.y, m.x); }   // 2 scalar fma
      if (MODE == 1) acc[i] = __builtin_elementwise_fma(acc[i], m, m);                                                      // 1 pk_fma, VGPR operands
      if (MODE == 2) acc[i] = __builtin_elementwise_fma(acc[i], ms, acc[i]);                                                // pk_fma with SGPR operand
      if (MODE == 3) { acc[i].x = __builtin_fmaf(acc[i].x, m.x, m.y); }                                                     // 1 scalar fma
      if (MODE == 6) { acc[i].x = __builtin_amdgcn_exp2f(acc[i].x); }                                                       // v_exp_f32
      if (MODE == 7) { acc[i].x = __builtin_amdgcn_rcpf(acc[i].x); }                                                        // v_rcp_f32
      if (MODE == 8) { acc[i].x = __builtin_amdgcn_sqrtf(acc[i].x); }                                                       // v_sqrt_f32
      if (MODE == 9) { acc[i].x = __builtin_fminf(acc[i].x, m.x); acc[i].y = acc[i].y > m.y ? acc[i].x : acc[i].y; }         // v_min_f32 + v_cmp + v_cndmask
      if (MODE == 10) { acc[i].x = __builtin_rintf(acc[i].x + m.x); }                                                       // v_add_f32 + v_rndne_f32
      if (MODE == 11) { acc[0].x = __builtin_fmaf(acc[0].x, m.x, m.y); }                                                     // ONE dependent chain
      if (MODE == 12) { acc[i & 1].x = __builtin_fmaf(acc[i & 1].x, m.x, m.y); }                                             // two chains
    }
    if (MODE == 4) {  // the blur inner loop mix: one texel = 4 byte->float converts + 16 packed FMAs with uniform coefficients
      const unsigned t = __float_as_uint(acc[0].x) + it;
      f2 trg = {(float)(t & 255u), (float)((t >> 8) & 255u)}, tba = {(float)((t >> 16) & 255u), (float)(t >> 24)};
#pragma unroll
      for (int i = 0; i < 8; i++) { acc[i] += trg * (a + i); }
      f2 acc2[8];
#pragma unroll
      for (int i = 0; i < 8; i++) { acc2[i] = acc[i]; acc2[i] += tba * (b + i); acc[i] = acc2[i]; }
    }
    if (MODE == 5) {  // converts only
      const unsigned t = __float_as_uint(acc[0].x) + it;
#pragma unroll
      for (int i = 0; i < 8; i++) { acc[i].x += (float)((t >> (8 * (i & 3))) & 255u); }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> double run(const char* name, double ops_per_iter_per_lane) {
  float* d; hipMalloc(&d, 256 * 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 8 * 4;  // 8 waves/SIMD x 4 rounds
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1.0000001f, 1e-9f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1.0000001f, 1e-9f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)grid * 4 /*waves*/ * kIter * 8 * ops_per_iter_per_lane;  // wave-instructions
  const double per_simd = insts / 1024.0;
  printf("%-28s %8.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  hipFree(d);
  return ms;
}
// occupancy sweep: W waves per SIMD (256 x W workgroups of 4 waves), how fast does ONE SIMD issue?
template <int MODE> void sweep(const char* name) {
  float* d; hipMalloc(&d, 256 * 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-34s", name);
  for (int W : {1, 2, 3, 4, 5, 6, 8}) {
    const int grid = 256 * W;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1.0000001f, 1e-9f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1.0000001f, 1e-9f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)grid * 4 * kIter * 8 / 1024.0;
    printf("  W=%d %.2f", W, ms * 1e6 / per_simd);
  }
  printf("   (ns per wave-instr per SIMD)\n");
  hipFree(d);
}
int main() {
  printf("# VALU issue rates on this GPU: ns (and cycles at 2.4 GHz) one wave64 instruction occupies a SIMD, 8 waves per SIMD, 8 independent chains per wave\n");
  run<0>("2x v_fma_f32", 2);
  run<1>("v_pk_fma_f32 (vgpr)", 1);
  run<2>("v_pk_fma_f32 (sgpr mult)", 1);
  run<3>("1x v_fma_f32", 1);
  run<4>("blur mix (4 cvt + 16 pk_fma)/8", 20.0 / 8);
  run<5>("cvt_ubyte + add", 2);
  run<6>("v_exp_f32", 1);
  run<7>("v_rcp_f32", 1);
  run<8>("v_sqrt_f32", 1);
  run<9>("v_min + v_cmp + v_cndmask", 3);
  run<10>("v_add_f32 + v_rndne_f32", 2);
  printf("# the same against occupancy\n");
  sweep<3>("v_fma_f32, 8 independent chains");
  sweep<12>("v_fma_f32, 2 chains");
  sweep<11>("v_fma_f32, 1 dependent chain");
  sweep<9>("min+cmp+cndmask (x3 instr)");
  sweep<6>("v_exp_f32, 8 chains");
  return 0;
}
