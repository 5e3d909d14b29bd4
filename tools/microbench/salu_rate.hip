// Scalar-unit issue rate on gfx950, alone and mixed with VALU work, against occupancy.
// hipcc --offload-arch=gfx950 -O3 salu_rate.hip -o salu_rate && ./salu_rate
// MODE 0: 64 s_add_u32 per iteration (8 independent chains)      MODE 1: 64 v_fma_f32 (8 chains)
// MODE 2: 64 x (v_fma_f32, s_add_u32) interleaved in ONE wave     MODE 3: 64 s_and_b64 on 8 chains (64-bit SALU)
// MODE 4: even waves run MODE 0's body, odd waves MODE 1's (the two kinds of work come from DIFFERENT waves of a SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kIter = 4096;
#define S8(op) op " %0, %0, 1\n" op " %1, %1, 1\n" op " %2, %2, 1\n" op " %3, %3, 1\n" op " %4, %4, 1\n" op " %5, %5, 1\n" op " %6, %6, 1\n" op " %7, %7, 1\n"
#define V8 "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
#define VS(i, j) "v_fma_f32 %" #i ", %" #i ", %16, %16\n s_add_u32 %" #j ", %" #j ", 1\n"
#define VS8 VS(0, 8) VS(1, 9) VS(2, 10) VS(3, 11) VS(4, 12) VS(5, 13) VS(6, 14) VS(7, 15)
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, float a) {
  float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
  unsigned long long m0 = blockIdx.x, m1 = 1, m2 = 2, m3 = 3, m4 = 4, m5 = 5, m6 = 6, m7 = 7;
  const bool odd = (threadIdx.x >> 6) & 1;  // MODE 4: wave parity inside the workgroup... one wave per SIMD, so use the block's parity
  const bool second = (blockIdx.x & 1) != 0;
  (void)odd;
  for (int it = 0; it < kIter; it++) {
    if (MODE == 0 || (MODE == 4 && !second)) {
#pragma unroll
      for (int r = 0; r < 8; r++) asm volatile(S8("s_add_u32") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
    }
    if (MODE == 1 || (MODE == 4 && second)) {
#pragma unroll
      for (int r = 0; r < 8; r++) asm volatile(V8 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a));
    }
    if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 8; r++)
        asm volatile(VS8 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4),
                     "+s"(s5), "+s"(s6), "+s"(s7) : "v"(a) : "scc");
    }
    if (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 8; r++)
        asm volatile("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %2\n s_and_b64 %2, %2, %3\n s_or_b64 %3, %3, %4\n s_and_b64 %4, %4, %5\n s_or_b64 %5, %5, %6\n s_and_b64 %6, %6, %7\n s_or_b64 %7, %7, %0\n"
                     : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3), "+s"(m4), "+s"(m5), "+s"(m6), "+s"(m7) : : "scc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7) + (float)(m0 ^ m1 ^ m2 ^ m3 ^ m4 ^ m5 ^ m6 ^ m7);
}
template <int MODE> void sweep(const char* name, double inst_per_iter) {
  float* d; hipMalloc(&d, 256 * 64 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-46s", name);
  for (int W : {1, 2, 4, 5, 8}) {
    const int grid = 256 * W;  // workgroups of 4 waves = one wave per SIMD each
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1.0000001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1.0000001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)grid * 4 * kIter * inst_per_iter / 1024.0;
    printf("  W=%d %.2f", W, ms * 1e6 / per_simd);
  }
  printf("   ns per instruction per SIMD\n");
  hipFree(d);
}
int main() {
  sweep<0>("64 s_add_u32 (8 chains)", 64);
  sweep<3>("64 s_and/or_b64 (chained)", 64);
  sweep<1>("64 v_fma_f32 (8 chains)", 64);
  sweep<2>("64 x (v_fma, s_add) in one wave, per PAIR", 64);
  sweep<4>("half the waves SALU, half VALU, per instr", 64);
  return 0;
}
