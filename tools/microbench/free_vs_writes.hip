// Does giving device memory back to the driver disturb OTHER streams' kernels?  (Round 5's fault hunt: with staging blocks --
// hipExtMallocWithFlags(hipDeviceMallocUncached) -- going back to the driver when a context closes, ~7 % of tools/thread_churn.py runs
// render a frame wrong although the device's records AND its bin lists are, read back afterwards, exactly right, and the same GPU
// work replayed from them gives the right pixels: something the kernels WROTE or READ in flight went missing.)  No library here:
//   W writer threads loop, each on its own stream:  k_fill(X, v) -> k_copy(X -> Y) -> k_check(Y, v)   (three dependent launches;
//                                                   X, Y ordinary hipMalloc memory, 4 MB each)
//   one churn thread loops:                         mode 1: 16 x hipExtMallocWithFlags(Uncached, 4 KB .. 256 KB), CPU stores, hipFree
//                                                   mode 2: the same with hipMalloc (ROCr's pool serves these)
//                                                   mode 3: hipMalloc / hipFree of 8 MB blocks
//                                                   mode 4: hipExtMallocWithFlags(Uncached) once, kept (the process-wide store's way)
//                                                   mode 0: nothing (control)
// Every k_check reports words of Y that are not v: lost or stale data of the launches before it on its own stream.
// hipcc --offload-arch=gfx950 -O2 -pthread tools/microbench/free_vs_writes.hip -o build/free_vs_writes ; usage: free_vs_writes <mode> [seconds] [writers] [fresh]
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

__global__ void k_fill(uint32_t* p, size_t n, uint32_t v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (uint32_t)i;
}
__global__ void k_copy(const uint32_t* __restrict__ a, uint32_t* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[n - 1 - i] ^ 0x5a5a5a5au;
}
__global__ void k_check(const uint32_t* p, size_t n, uint32_t v, uint32_t* bad, uint32_t* first) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t want = (v + (uint32_t)(n - 1 - i)) ^ 0x5a5a5a5au;
    if (p[i] != want && atomicAdd(bad, 1u) == 0) { first[0] = (uint32_t)i; first[1] = p[i]; first[2] = want; }
  }
}

static std::atomic<bool> g_stop{false};
static std::atomic<uint64_t> g_rounds{0}, g_bad_rounds{0}, g_bad_words{0}, g_churn{0};

// "fresh" writers (usage: 5th argument 1): what a fresh context does -- every round allocates its buffers anew, clears them with
// hipMemsetAsync (a marker byte, as the library's FDH_POISON build does), renders into them and gives them back; a wrong word that
// holds the marker is the CLEAR landing after (or a stale line of it surviving) the kernel's stores
static int g_fresh = 0;
static std::atomic<uint64_t> g_marker_words{0};
__global__ void k_check_marker(const uint32_t* p, size_t n, uint32_t v, uint32_t marker, uint32_t* bad, uint32_t* first, uint32_t* n_marker) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t want = (v + (uint32_t)(n - 1 - i)) ^ 0x5a5a5a5au;
    if (p[i] != want) {
      if (atomicAdd(bad, 1u) == 0) { first[0] = (uint32_t)i; first[1] = p[i]; first[2] = want; }
      if (p[i] == marker) atomicAdd(n_marker, 1u);
    }
  }
}
static void writer_fresh(int t) {
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t n = (size_t)1 << 20;
  uint32_t* bad;
  CK(hipHostMalloc((void**)&bad, 64, 0));
  for (uint32_t it = 1; !g_stop.load(std::memory_order_relaxed); it++) {
    const uint32_t v = it * 2654435761u + (uint32_t)t;
    uint32_t *x, *y;
    CK(hipMalloc((void**)&x, n * 4));
    CK(hipMalloc((void**)&y, n * 4));
    CK(hipMemsetAsync(x, 0x11, n * 4, s));
    CK(hipMemsetAsync(y, 0xA5, n * 4, s));
    bad[0] = 0; bad[4] = 0;
    hipLaunchKernelGGL(k_fill, dim3(512), dim3(256), 0, s, x, n, v);
    hipLaunchKernelGGL(k_copy, dim3(512), dim3(256), 0, s, x, y, n);
    hipLaunchKernelGGL(k_check_marker, dim3(512), dim3(256), 0, s, y, n, v, 0xA5A5A5A5u, bad, bad + 1, bad + 4);
    CK(hipStreamSynchronize(s));
    g_rounds++;
    if (bad[0]) {
      g_bad_words += bad[0]; g_marker_words += bad[4];
      if (g_bad_rounds++ < 8) std::printf("  fresh writer %d round %u: %u of %zu words wrong (%u hold the clear's marker); first at %u: got %08x want %08x\n", t, it, bad[0], n, bad[4], bad[1], bad[2], bad[3]);
    }
    CK(hipFree(x)); CK(hipFree(y));
  }
  (void)hipHostFree(bad); (void)hipStreamDestroy(s);
}

static void writer(int t) {
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t n = (size_t)1 << 20;
  uint32_t *x, *y, *bad;
  CK(hipMalloc((void**)&x, n * 4));
  CK(hipMalloc((void**)&y, n * 4));
  CK(hipHostMalloc((void**)&bad, 64, 0));
  for (uint32_t it = 1; !g_stop.load(std::memory_order_relaxed); it++) {
    const uint32_t v = it * 2654435761u + (uint32_t)t;
    bad[0] = 0;
    hipLaunchKernelGGL(k_fill, dim3(512), dim3(256), 0, s, x, n, v);
    hipLaunchKernelGGL(k_copy, dim3(512), dim3(256), 0, s, x, y, n);
    hipLaunchKernelGGL(k_check, dim3(512), dim3(256), 0, s, y, n, v, bad, bad + 1);
    CK(hipStreamSynchronize(s));
    g_rounds++;
    if (bad[0]) {
      g_bad_words += bad[0];
      if (g_bad_rounds++ < 8) std::printf("  writer %d round %u: %u of %zu words wrong; first at %u: got %08x want %08x\n", t, it, bad[0], n, bad[1], bad[2], bad[3]);
    }
  }
  (void)hipFree(x); (void)hipFree(y); (void)hipHostFree(bad); (void)hipStreamDestroy(s);
}

static void churn(int mode) {
  CK(hipSetDevice(0));
  uint32_t rng = 12345u;
  std::vector<void*> kept;
  while (!g_stop.load(std::memory_order_relaxed)) {
    if (mode == 0) { std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; }
    void* p[16];
    size_t bytes[16];
    const int k = mode == 3 ? 2 : 16;
    if (mode == 4 && !kept.empty()) {  // blocks stay with the process: only the CPU stores repeat
      for (void* q : kept) { volatile uint32_t* d = (volatile uint32_t*)q; for (int i = 0; i < 1024; i++) d[i] = rng + i; }
      _mm_sfence();
      g_churn++;
      continue;
    }
    for (int i = 0; i < k; i++) {
      rng = rng * 1664525u + 1013904223u;
      bytes[i] = mode == 3 ? ((size_t)8 << 20) : ((size_t)4096 << ((rng >> 24) % 7));
      if (mode == 1 || mode == 4) CK(hipExtMallocWithFlags(&p[i], bytes[i], hipDeviceMallocUncached));
      else CK(hipMalloc(&p[i], bytes[i]));
      if (mode == 1 || mode == 4) { volatile uint32_t* d = (volatile uint32_t*)p[i]; for (size_t w = 0; w < bytes[i] / 4; w += 16) d[w] = rng; _mm_sfence(); }
    }
    if (mode == 4) { kept.assign(p, p + k); continue; }
    for (int i = 0; i < k; i++) CK(hipFree(p[i]));
    g_churn++;
  }
  for (void* q : kept) (void)hipFree(q);
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? std::atoi(argv[1]) : 1, seconds = argc > 2 ? std::atoi(argv[2]) : 10, writers = argc > 3 ? std::atoi(argv[3]) : 3;
  g_fresh = argc > 4 ? std::atoi(argv[4]) : 0;
  CK(hipSetDevice(0));
  std::vector<std::thread> th;
  for (int t = 0; t < writers; t++) th.emplace_back(g_fresh ? writer_fresh : writer, t);
  std::thread c(churn, mode);
  std::this_thread::sleep_for(std::chrono::seconds(seconds));
  g_stop = true;
  for (auto& x : th) x.join();
  c.join();
  static const char* names[] = {"no churn", "hipExtMallocWithFlags(Uncached) + hipFree", "hipMalloc + hipFree, 4 - 256 KB", "hipMalloc + hipFree, 8 MB", "Uncached blocks kept, CPU stores only"};
  std::printf("free_vs_writes: mode %d (%s), %d %swriters, %d s: %llu rounds, %llu churn batches; rounds with wrong words: %llu (words %llu, holding the clear's marker %llu)\n", mode, names[mode], writers,
              g_fresh ? "FRESH (malloc + memset + free per round) " : "", seconds,
              (unsigned long long)g_rounds.load(), (unsigned long long)g_churn.load(), (unsigned long long)g_bad_rounds.load(), (unsigned long long)g_bad_words.load(), (unsigned long long)g_marker_words.load());
  return g_bad_rounds.load() ? 1 : 0;
}
