// Can the host write records straight into device memory (no upload kernel)?  Allocates device memory the host may address
// (hipExtMallocWithFlags fine-grained / hipMallocManaged variants), has the CPU store a changing pattern into it, launches a kernel
// that checks the pattern, many times; reports host write bandwidth, launch-to-verified latency and stale reads.
// hipcc --offload-arch=gfx950 -O2 tools/microbench/bar_write.hip -o build/bar_write
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <csignal>
#include <csetjmp>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_check(const uint32_t* p, size_t n, uint32_t want, uint32_t* bad) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) if (p[i] != want + (uint32_t)i) atomicAdd(bad, 1u);
}
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
int main() {
  const size_t bytes = 128 << 10, n = bytes / 4;
  uint32_t* bad; CK(hipHostMalloc((void**)&bad, 4, 0)); *bad = 0;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  struct { const char* name; unsigned flags; int kind; } modes[] = {{"hipExtMallocWithFlags(Finegrained)", hipDeviceMallocFinegrained, 0}, {"hipExtMallocWithFlags(Uncached)", hipDeviceMallocUncached, 0}, {"hipMalloc", 0, 1}};
  for (auto& m : modes) {
    uint32_t* d = nullptr;
    hipError_t e = m.kind == 1 ? hipMalloc((void**)&d, bytes) : hipExtMallocWithFlags((void**)&d, bytes, m.flags);
    if (e != hipSuccess) { std::printf("%s: allocation failed: %s\n", m.name, hipGetErrorString(e)); continue; }
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    if (sigsetjmp(jb, 1)) { std::printf("%s: the host cannot address it (fault)\n", m.name); continue; }
    volatile uint32_t probe = 0; d[0] = 1; probe = d[0]; (void)probe;  // faults here if not mapped
    uint32_t stale = 0; double wr_us = 0, tot_us = 0; const int rounds = 2000;
    for (int r = 1; r <= rounds; r++) {
      const auto t0 = std::chrono::steady_clock::now();
      for (size_t i = 0; i < n; i++) d[i] = (uint32_t)r * 2654435761u + (uint32_t)i;
      _mm_sfence();
      const auto t1 = std::chrono::steady_clock::now();
      *bad = 0;
      hipLaunchKernelGGL(k_check, dim3(64), dim3(256), 0, s, d, n, (uint32_t)r * 2654435761u, bad);
      CK(hipStreamSynchronize(s));
      const auto t2 = std::chrono::steady_clock::now();
      stale += *bad != 0;
      wr_us += std::chrono::duration<double, std::micro>(t1 - t0).count(); tot_us += std::chrono::duration<double, std::micro>(t2 - t1).count();
    }
    std::printf("%s: host-addressable; %zu KB written in %.1f us (%.1f GB/s), launch + check %.1f us, rounds with stale data %u of %d\n", m.name, bytes >> 10, wr_us / rounds,
                bytes / (wr_us / rounds) / 1e3, tot_us / rounds, stale, rounds);
    (void)hipFree(d);
  }
  return 0;
}
