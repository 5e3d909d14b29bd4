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
  const unsigned long long mask = __ballot(threadIdx.x & 1);
  float m2[8];
  for (int i = 0; i < 8; i++) m2[i] = a * i + threadIdx.x;
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
      if (MODE == 13) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].x) : "v"(m.x), "v"(m.y));       // VOP3, three VGPR sources
      if (MODE == 14) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].x) : "v"(m.x), "v"(m.y));           // VOP2, D += S0 * S1
      if (MODE == 15) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].x) : "s"(a), "v"(m.y));          // VOP3, one SGPR source
      if (MODE == 16) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(m.x));                      // VOP2
      if (MODE == 17) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(acc[i].x) : "v"(m.x));                  // VOP3, a source used twice
      if (MODE == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i].x) : "v"(m.x));             // VOP2 select on VCC
      if (MODE == 19) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(acc[i].x) : "v"(m.x), "s"(mask));   // VOP3 select on an SGPR pair
      if (MODE == 20) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(acc[i].x), "v"(m.x) : "vcc");           // compare into VCC
      if (MODE == 21) asm volatile("v_max_f32 %0, %0, %1" : "+v"(acc[i].x) : "v"(m.x));                      // VOP2
      if (MODE == 22) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i].x) : "s"(a));                        // VOP2, SGPR src0
      if (MODE == 23) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i].x) : "s"(a));                        // VOP2, SGPR src0
      if (MODE == 24) asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(m.x));                      // VOP2
      if (MODE == 25) asm volatile("v_min_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(m.x));                      // VOP2
      if (MODE == 26) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(acc[i].x) : "v"(m.x));                 // VOP3, inline constant
      if (MODE == 27) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].x) : "s"(a), "v"(m.y));             // VOP2, SGPR src0
      if (MODE == 28) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(acc[i].x) : "v"(m.x), "v"(m.y));       // VOP3
      if (MODE == 29) asm volatile("v_rndne_f32 %0, %0" : "+v"(acc[i].x));                                   // VOP1
      if (MODE == 30) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(acc[i].x));                              // VOP1
      if (MODE == 31) asm volatile("v_mul_f32 %0, 0x3b808081, %0" : "+v"(acc[i].x));                         // VOP2, 32-bit literal
      if (MODE == 32) asm volatile("v_max_f32 %0, 0, %0" : "+v"(acc[i].x));                                  // VOP2, inline constant
      if (MODE == 33) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(acc[i].x) : "v"(m.x));                      // VOP2
      if (MODE == 34) asm volatile("v_and_b32 %0, %1, %0" : "+v"(acc[i].x) : "v"(m.x));                      // VOP2 integer
      if (MODE == 35) asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc[i].x) : "v"(m.x));                      // VOP2 integer
      if (MODE == 36) asm volatile("v_mov_b32 %0, %1" : "=v"(acc[i].x) : "v"(acc[(i + 1) & 7].y));           // VOP1
      if (MODE == 37) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(acc[i].x) : "v"(m.x), "v"(m.y));  // VOP3 + clamp
      // two kinds of VALU work side by side: M = v_rndne (4-cycle class), F = v_fma (2-cycle class)
      if (MODE == 40 || MODE == 41 || MODE == 42) asm volatile("v_rndne_f32 %0, %0" : "+v"(acc[i].x));
      if (MODE == 41 || MODE == 42 || MODE == 43) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].y) : "v"(m.x), "v"(m.y));
      if (MODE == 42 || MODE == 43) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(m2[i]) : "v"(m.x), "v"(m.y));
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
  for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y + m2[i];
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
  printf("# encodings (inline assembly, 8 independent chains)\n");
  run<13>("v_fma_f32 d, d, v, v  (VOP3)", 1);
  run<17>("v_fma_f32 d, d, v, v (same v)", 1);
  run<15>("v_fma_f32 d, d, s, v  (VOP3)", 1);
  run<14>("v_fmac_f32 d, v, v   (VOP2)", 1);
  run<16>("v_mul_f32 d, v, d    (VOP2)", 1);
  run<21>("v_max_f32 d, d, v    (VOP2)", 1);
  run<18>("v_cndmask_b32 .. vcc (VOP2)", 1);
  run<19>("v_cndmask_b32 .. s[] (VOP3)", 1);
  run<20>("v_cmp_gt_f32 vcc     (VOPC)", 1);
  run<24>("v_add_f32 d, v, d    (VOP2)", 1);
  run<33>("v_sub_f32 d, d, v    (VOP2)", 1);
  run<25>("v_min_f32 d, v, d    (VOP2)", 1);
  run<32>("v_max_f32 d, 0, d    (VOP2)", 1);
  run<28>("v_med3_f32           (VOP3)", 1);
  run<37>("v_fma_f32 .. clamp   (VOP3)", 1);
  run<26>("v_fma_f32 d, d, v, 0.5", 1);
  run<22>("v_mul_f32 d, s, d    (VOP2)", 1);
  run<23>("v_add_f32 d, s, d    (VOP2)", 1);
  run<27>("v_fmac_f32 d, s, v   (VOP2)", 1);
  run<31>("v_mul_f32 d, literal, d", 1);
  run<29>("v_rndne_f32          (VOP1)", 1);
  run<30>("v_cvt_f32_ubyte1     (VOP1)", 1);
  run<36>("v_mov_b32            (VOP1)", 1);
  run<34>("v_and_b32            (VOP2)", 1);
  run<35>("v_add_u32            (VOP2)", 1);
  printf("# do the two classes overlap?  time per loop step (8 x the instructions named), ns per SIMD\n");
  run<40>("8 rndne", 1);
  run<41>("8 rndne + 8 fma   (per pair)", 1);
  run<42>("8 rndne + 16 fma  (per triple)", 1);
  run<43>("16 fma            (per pair)", 1);
  printf("# the same against occupancy\n");
  sweep<3>("v_fma_f32, 8 independent chains");
  sweep<12>("v_fma_f32, 2 chains");
  sweep<11>("v_fma_f32, 1 dependent chain");
  sweep<9>("min+cmp+cndmask (x3 instr)");
  sweep<6>("v_exp_f32, 8 chains");
  return 0;
}
