// How fast does the CPU read / write hipHostMalloc memory compared with malloc memory?  (Round 4: draw records are written
// straight into pinned lanes; a record is read back a few times while it is finished.)
// hipcc -O2 -o /tmp/pinned_host_speed tools/microbench/pinned_host_speed.cpp && /tmp/pinned_host_speed
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Rec { unsigned v[32]; };
static void run(const char* name, Rec* p, size_t n) {
  // (a) write records sequentially, (b) read-modify-write each record right after writing it, (c) read all back
  double best_w = 1e9, best_rmw = 1e9, best_r = 1e9;
  volatile unsigned sink = 0;
  for (int rep = 0; rep < 20; rep++) {
    double t0 = now();
    for (size_t i = 0; i < n; i++) { std::memset(&p[i], 0, sizeof(Rec)); p[i].v[0] = (unsigned)i; p[i].v[31] = (unsigned)i; }
    double t1 = now();
    for (size_t i = 0; i < n; i++) { std::memset(&p[i], 0, sizeof(Rec)); p[i].v[0] = (unsigned)i; unsigned a = p[i].v[0] + (i ? p[i - 1].v[5] : 0); p[i].v[5] = a; p[i].v[17] |= a; }
    double t2 = now();
    unsigned s = 0;
    for (size_t i = 0; i < n; i++) s += p[i].v[3] + p[i].v[20];
    double t3 = now();
    sink = s;
    best_w = std::min(best_w, t1 - t0); best_rmw = std::min(best_rmw, t2 - t1); best_r = std::min(best_r, t3 - t2);
  }
  std::printf("%-28s write %6.1f ns/rec   write+readback %6.1f ns/rec   read %6.1f ns/rec\n", name, best_w / n * 1e9, best_rmw / n * 1e9, best_r / n * 1e9);
}
int main() {
  const size_t n = 800;  // one bench frame's records
  Rec* a = (Rec*)std::aligned_alloc(64, n * sizeof(Rec));
  run("malloc", a, n);
  struct { const char* name; unsigned flags; } kinds[] = {{"hipHostMallocDefault", hipHostMallocDefault}, {"hipHostMallocNonCoherent", hipHostMallocNonCoherent},
                                                        {"hipHostMallocCoherent", hipHostMallocCoherent}, {"hipHostMallocNumaUser", hipHostMallocNumaUser},
                                                        {"hipHostMallocMapped", hipHostMallocMapped}, {"hipHostMallocWriteCombined", hipHostMallocWriteCombined}};
  for (auto& k : kinds) {
    Rec* p = nullptr;
    if (hipHostMalloc((void**)&p, n * sizeof(Rec), k.flags) != hipSuccess) { std::printf("%s: failed\n", k.name); (void)hipGetLastError(); continue; }
    run(k.name, p, n);
    (void)hipHostFree(p);
  }
  Rec* r = (Rec*)std::aligned_alloc(4096, (n * sizeof(Rec) + 4095) & ~(size_t)4095);
  if (hipHostRegister(r, (n * sizeof(Rec) + 4095) & ~(size_t)4095, hipHostRegisterDefault) == hipSuccess) { run("malloc + hipHostRegister", r, n); (void)hipHostUnregister(r); }
  else std::printf("hipHostRegister failed\n");
  return 0;
}
