// Do 2040 one-wave workgroups with 14 / 18 KB of LDS each (a k_blur_mx pass at 4K) really run all at once on 256 CUs?
// Every wave notes where it ran (XCC, SE, CU from HW_ID / XCC_ID) and when (s_memtime, comparable inside one XCD), then
// spins ~10 us.  Host: waves per CU, and how many waves started late (after the first wave of their XCD had already finished).
// hipcc -O3 --offload-arch=gfx950 -o build/occupancy_probe tools/microbench/occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(64, 2) void k_probe(unsigned long long* out, int spin) {
  extern __shared__ uint32_t lds[];
  const unsigned long long t0 = __builtin_readcyclecounter();
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  lds[threadIdx.x] = hw;
  unsigned long long t1;
  do { __builtin_amdgcn_s_sleep(16); t1 = __builtin_readcyclecounter(); } while ((long long)(t1 - t0) < spin);
  if (threadIdx.x == 0) { out[4 * blockIdx.x + 0] = t0; out[4 * blockIdx.x + 1] = t1; out[4 * blockIdx.x + 2] = hw; out[4 * blockIdx.x + 3] = xcc + lds[0] * 0; }
}
int main() {
  const int n = 2040;
  unsigned long long* d; (void)hipMalloc(&d, n * 32);
  std::vector<unsigned long long> h(4 * n);
  for (int kb : {2, 14, 18, 20, 22}) {
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_probe, dim3(n), dim3(64), kb * 1024, 0, d, 20000);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, n * 32, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu;
    std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> per_xcc;
    for (int i = 0; i < n; i++) {
      const unsigned hw = (unsigned)h[4 * i + 2], xcc = (unsigned)h[4 * i + 3] & 15u;
      const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
      per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
      per_xcc[xcc].push_back({h[4 * i], h[4 * i + 1]});
    }
    int mx = 0, mn = 1 << 30;
    for (auto& kv : per_cu) { mx = std::max(mx, kv.second); mn = std::min(mn, kv.second); }
    int late = 0;
    double spread = 0;
    for (auto& kv : per_xcc) {
      unsigned long long first_end = ~0ull, first_start = ~0ull, last_start = 0;
      for (auto& w : kv.second) { first_end = std::min(first_end, w.second); first_start = std::min(first_start, w.first); last_start = std::max(last_start, w.first); }
      for (auto& w : kv.second) if (w.first >= first_end) late++;
      spread = std::max(spread, (double)(last_start - first_start));
    }
    std::printf("LDS %2d KB per wave: %zu CUs used, waves per CU min %d max %d; waves that started after another had finished (second round): %d of %d; widest start spread in an XCD: %.0f cycles (100 MHz counter: x10 ns)\n",
                kb, per_cu.size(), mn, mx, late, n, spread);
  }
  return 0;
}
