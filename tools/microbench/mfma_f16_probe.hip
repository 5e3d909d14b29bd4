// Probe (GPU box): v_mfma_f32_32x32x16_f16 operand layout, and whether f16 SUBNORMAL inputs are honoured (a byte b placed
// in the low bits of a half is b * 2^-24).  hipcc --offload-arch=gfx950 -O2 mfma_f16_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using f16v = __attribute__((ext_vector_type(16))) float;
__global__ void probe(const uint16_t* A /*[32][16] bits*/, const uint16_t* B /*[16][32] bits*/, float* D /*[32][32]*/) {
  const int l = threadIdx.x, g = l >> 5, r = l & 31;
  union { h8 v; uint16_t u[8]; } a, b;
  for (int t = 0; t < 8; t++) { a.u[t] = A[r * 16 + 8 * g + t]; b.u[t] = B[(8 * g + t) * 32 + r]; }
  f16v c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, c, 0, 0, 0);
  for (int reg = 0; reg < 16; reg++) D[((reg & 3) + 8 * (reg >> 2) + 4 * g) * 32 + r] = c[reg];
}
static float h2f(uint16_t h) {
  const int e = (h >> 10) & 31, m = h & 1023;
  float v = e == 0 ? std::ldexp((float)m, -24) : std::ldexp((float)(m + 1024), e - 25);
  return (h & 0x8000) ? -v : v;
}
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; __builtin_memcpy(&u, &h, 2); return u; }
int main() {
  uint16_t hA[32 * 16], hB[16 * 32];
  for (int mode = 0; mode < 2; mode++) {
    for (int i = 0; i < 32; i++) for (int k = 0; k < 16; k++) {
      const int byte = (i * 7 + k * 13 + 5) & 255;
      hA[i * 16 + k] = mode == 0 ? f2h((float)byte) : (uint16_t)byte;  // mode 1: subnormal bits
    }
    for (int k = 0; k < 16; k++) for (int j = 0; j < 32; j++) hB[k * 32 + j] = f2h((float)((k * 3 + j * 5) % 17) * 0.125f + (k == j % 16 ? 1.0f : 0.0f));
    uint16_t *dA, *dB; float* dD; float hD[1024];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
      double ref = 0;
      for (int k = 0; k < 16; k++) ref += (double)h2f(hA[i * 16 + k]) * (double)h2f(hB[k * 32 + j]);
      maxerr = std::fmax(maxerr, std::fabs(ref - hD[i * 32 + j])); maxref = std::fmax(maxref, std::fabs(ref));
    }
    printf("mode %d (%s): max |D - ref| = %.3g, max |ref| = %.3g\n", mode, mode ? "A = subnormal halves" : "A = normal halves", maxerr, maxref);
  }
  return 0;
}
