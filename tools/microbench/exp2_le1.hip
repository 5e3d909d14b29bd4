// Is v_exp_f32 of a non-positive input ever above 1?  The shadow falloff (atlas.frag:330-343: exp(-z^2 / 2) as exp2(-0.7213 z^2)) is
// written min(exp2(e), 1) in the kernels; e is never positive, so the min is dead weight IF the hardware's approximation keeps
// exp2(e) <= 1 for every e <= 0 -- checked here over every float bit pattern with the sign bit set, plus +0.
// hipcc --offload-arch=gfx950 -O2 tools/microbench/exp2_le1.hip -o build/exp2_le1
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k(unsigned long long* out) {
  const uint64_t n = (uint64_t)gridDim.x * blockDim.x, i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long above = 0, one = 0, nan = 0;
  for (uint64_t b = 0x80000000ull + i0; b <= 0xff800000ull; b += n) {  // -0 .. -inf
    const float y = __builtin_amdgcn_exp2f(__uint_as_float((uint32_t)b));
    above += y > 1.0f; one += y == 1.0f; nan += y != y;
  }
  if (above) atomicAdd(out + 0, above);
  if (one) atomicAdd(out + 1, one);
  if (nan) atomicAdd(out + 2, nan);
  if (i0 == 0) { out[3] = __float_as_uint(__builtin_amdgcn_exp2f(-0.0f)); out[4] = __float_as_uint(__builtin_amdgcn_exp2f(0.0f)); }
}
int main() {
  unsigned long long* d; unsigned long long h[5] = {};
  hipMalloc((void**)&d, sizeof h); hipMemset(d, 0, sizeof h);
  hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  std::printf("v_exp_f32 over every x in [-inf, -0]: above 1: %llu, equal to 1: %llu, NaN: %llu; exp2(-0) = %08llx, exp2(+0) = %08llx\n", h[0], h[1], h[2], h[3], h[4]);
  return h[0] || h[2] || h[3] != 0x3f800000ull ? 1 : 0;
}
