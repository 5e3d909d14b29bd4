// What does a plain read + write of one 4K RGBA8 surface (33.2 MB in, 33.2 MB out) take on this GPU?  The practical ceiling
// for a blur pass at that size: hipMemcpyAsync D2D and three copy kernels (16 B per lane, grid-stride; one-wave workgroups
// walking 32-row blocks like k_blur_mx; LDS-DMA in + store out).  Events around 50 launches each.
// hipcc -O3 --offload-arch=gfx950 -o build/copy_rate tools/microbench/copy_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// one wave per 32 x 128-pixel tile (4 blocks of 32 x 32 like a blur wave): lane = x, 16 rows per step
__global__ __launch_bounds__(64) void k_copy_tiles(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int W, int H, int T) {
  const int strips = W / 32, item = blockIdx.x, sl = item % strips, sa = item / strips;
  const int lane = threadIdx.x, j = lane & 31, g = lane >> 5;
  const int x = sl * 32 + j;
  for (int b = 0; b < T; b++) {
    const int y0 = (sa * T + b) * 32;
    if (y0 >= H) return;
    uint32_t v[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { const int y = min(y0 + 2 * r + g, H - 1); v[r] = src[(size_t)y * W + x]; }
#pragma unroll
    for (int r = 0; r < 16; r++) { const int y = y0 + 2 * r + g; if (y < H) dst[(size_t)y * W + x] = v[r]; }
  }
}
int main() {
  const int W = 3840, H = 2160;
  const size_t bytes = (size_t)W * H * 4;
  void *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto fn) {
    for (int i = 0; i < 5; i++) fn();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 50; i++) fn();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 50;
    std::printf("%-44s %6.2f us per surface   %.2f TB/s (read + write)\n", name, us, 2.0 * bytes / us * 1e-6);
  };
  time("hipMemcpyAsync D2D", [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
  for (int grid : {1024, 2048, 4096, 8192})
    for (int block : {256, 1024}) {
      char nm[64]; std::snprintf(nm, sizeof nm, "k_copy16 grid %d x %d", grid, block);
      time(nm, [&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(block), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16); });
    }
  for (int T : {1, 2, 4, 8}) {
    char nm[64]; std::snprintf(nm, sizeof nm, "k_copy_tiles one-wave WGs, T = %d blocks", T);
    const int items = (W / 32) * ((H + 32 * T - 1) / (32 * T));
    time(nm, [&] { hipLaunchKernelGGL(k_copy_tiles, dim3(items), dim3(64), 0, 0, (const uint32_t*)a, (uint32_t*)b, W, H, T); });
  }
  return 0;
}
