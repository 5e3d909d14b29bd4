// How fast does the CPU write / read host memory the GPU can see, by how it was obtained?  (Round 4: draw records are produced
// by CPU threads and fetched by an upload kernel over PCIe.)  Per kind: memcpy of 128 KB INTO the buffer from ordinary memory,
// a read of every 64-byte line of it back (sum), and a read-modify-write pass -- each right after a GPU kernel has read the
// whole buffer (what a frame's upload does), best of 50.
// hipcc -O2 --offload-arch=gfx950 -o build/pinned_host_speed tools/microbench/pinned_host_speed.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_read(const uint4* p, size_t n16, uint4* out) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if (acc.x == 0x12345678u) out[0] = acc;
}
static void run(const char* name, void* host, size_t bytes) {
  void* dev = nullptr;
  if (hipHostGetDevicePointer(&dev, host, 0) != hipSuccess) { std::printf("%-32s no device pointer\n", name); (void)hipGetLastError(); return; }
  uint4* out = nullptr;
  (void)hipMalloc((void**)&out, 64);
  unsigned char* src = (unsigned char*)std::aligned_alloc(64, bytes);
  double w = 1e9, r = 1e9, rmw = 1e9, g = 1e9;
  volatile unsigned long long sink = 0;
  for (int rep = 0; rep < 50; rep++) {
    std::memset(src, rep + 1, bytes);
    double t0 = now();
    std::memcpy(host, src, bytes);
    double t1 = now();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_read, dim3(128), dim3(64), 0, 0, (const uint4*)dev, bytes / 16, out);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    double t2 = now();
    unsigned long long s = 0;
    for (size_t i = 0; i < bytes; i += 64) s += *(volatile unsigned long long*)((unsigned char*)host + i);
    double t3 = now();
    for (size_t i = 0; i < bytes; i += 64) { volatile unsigned int* q = (volatile unsigned int*)((unsigned char*)host + i); q[1] = q[0] + 1; }
    double t4 = now();
    sink = s;
    w = std::min(w, t1 - t0); r = std::min(r, t3 - t2); rmw = std::min(rmw, t4 - t3); g = std::min(g, (double)ms * 1e-3);
  }
  std::printf("%-32s CPU memcpy in %6.1f us (%5.1f GB/s)   CPU read %6.1f ns/line   CPU rmw %6.1f ns/line   GPU read %5.1f us\n", name, w * 1e6, bytes / w * 1e-9, r / (bytes / 64) * 1e9,
              rmw / (bytes / 64) * 1e9, g * 1e6);
  (void)hipFree(out);
  std::free(src);
}
int main() {
  const size_t bytes = 128 << 10;
  struct { const char* name; unsigned flags; } kinds[] = {{"hipHostMallocDefault", hipHostMallocDefault}, {"hipHostMallocNonCoherent", hipHostMallocNonCoherent},
                                                        {"hipHostMallocCoherent", hipHostMallocCoherent}, {"hipHostMallocNumaUser", hipHostMallocNumaUser},
                                                        {"hipHostMallocWriteCombined", hipHostMallocWriteCombined}, {"hipHostMallocMapped|Portable", hipHostMallocMapped | hipHostMallocPortable}};
  for (auto& k : kinds) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, k.flags) != hipSuccess) { std::printf("%s: failed\n", k.name); (void)hipGetLastError(); continue; }
    run(k.name, p, bytes);
    (void)hipHostFree(p);
  }
  void* r = std::aligned_alloc(4096, bytes);
  if (hipHostRegister(r, bytes, hipHostRegisterDefault) == hipSuccess) { run("malloc + hipHostRegister", r, bytes); (void)hipHostUnregister(r); }
  else { std::printf("hipHostRegister failed\n"); (void)hipGetLastError(); }
  void* r2 = std::aligned_alloc(4096, bytes);
  if (hipHostRegister(r2, bytes, hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess) { run("malloc + Register(Mapped)", r2, bytes); (void)hipHostUnregister(r2); }
  else { std::printf("hipHostRegister(mapped) failed\n"); (void)hipGetLastError(); }
  void* m = nullptr;
  if (hipMallocManaged(&m, bytes, hipMemAttachGlobal) == hipSuccess) { run("hipMallocManaged", m, bytes); (void)hipFree(m); } else (void)hipGetLastError();
  return 0;
}
