// Probe (GPU box): issue rate of v_mfma_f32_32x32x16_f16 with normal vs SUBNORMAL f16 operands (one wave, 4 independent accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using f16v = __attribute__((ext_vector_type(16))) float;
union HB { h8 v; uint16_t u[8]; };
__global__ void rate(uint16_t bits_a, uint16_t bits_b, int iters, long long* cycles, float* sink) {
  HB a, b;
  for (int t = 0; t < 8; t++) { a.u[t] = bits_a + (threadIdx.x & 3); b.u[t] = bits_b + ((threadIdx.x + t) & 7); }
  f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, c3, 0, 0, 0);
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  long long* d; float* s; hipMalloc(&d, 8 * 1024); hipMalloc(&s, 4 * 64 * 1024);
  const struct { const char* name; uint16_t a, b; } cases[] = {
    {"A normal (0x3c00..), B normal (0x4000..)", 0x3c00, 0x4000},
    {"A normal, B subnormal (bytes 0x00c8..)", 0x3c00, 0x00c8},
    {"A subnormal, B subnormal", 0x0010, 0x00c8},
    {"A normal, B zero..7", 0x3c00, 0x0000},
  };
  for (auto& c : cases) {
    rate<<<1, 64>>>(c.a, c.b, 2000, d, s);
    long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-45s %.1f cycles per MFMA (s_memtime ticks / 8000)\n", c.name, (double)h / 8000.0);
  }
  return 0;
}
