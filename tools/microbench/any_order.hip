// Does hipExtAnyOrderLaunch clear the barrier bit between two kernels of ONE stream on this part?  Two one-workgroup kernels that
// each spin for ~200 us: back to back they take ~400 us, side by side ~200.  (hip_ext.h says the flag is "not supported on AMD GFX9xx
// boards" for hipExtModuleLaunchKernel; this asks the hardware.)   hipcc --offload-arch=gfx950 -O2 any_order.hip -o any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (out) *out = 1;
}
int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const long long cyc = 20000;  // wall_clock64 ticks at 100 MHz: 200 us
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(a, s);
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, cyc, (int*)nullptr);
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, cyc, (int*)nullptr);
      hipEventRecord(b, s);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      printf("second launch %s: %.1f us for two 200-us kernels\n", mode ? "hipExtAnyOrderLaunch" : "in order", ms * 1000.0f);
    }
  }
  return 0;
}
