// How long a spinning thread takes to see a flag another thread sets (the walk pool's fork latency), on this host.
// g++ -O2 -std=c++17 -pthread tools/microbench/pool_wake.cpp -o build/pool_wake && ./build/pool_wake [threads]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <algorithm>
#include <sched.h>
#include <unistd.h>
static inline void relax() { __builtin_ia32_pause(); }
int main(int argc, char** argv) {
  const int T = argc > 1 ? std::atoi(argv[1]) : 7, rounds = 2000;
  std::atomic<uint64_t> epoch{0};
  struct alignas(128) Slot { std::atomic<int64_t> seen_ns{0}; std::atomic<uint64_t> ack{0}; };
  std::vector<Slot> slots(T);
  std::atomic<bool> quit{false};
  auto now = [] { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++)
    th.emplace_back([&, t] {
      uint64_t seen = 0;
      while (!quit.load(std::memory_order_relaxed)) {
        uint64_t e;
        while ((e = epoch.load(std::memory_order_acquire)) == seen && !quit.load(std::memory_order_relaxed)) relax();
        seen = e;
        slots[t].seen_ns.store(now(), std::memory_order_relaxed);
        slots[t].ack.store(e, std::memory_order_release);
      }
    });
  std::vector<double> worst, mean;
  for (int r = 1; r <= rounds; r++) {
    const int64_t t0 = now();
    while (now() - t0 < 40000) relax();  // the calling thread's serial work between groups: 40 us
    const int64_t tb = now();
    epoch.store((uint64_t)r, std::memory_order_release);
    double w = 0, m = 0;
    for (int t = 0; t < T; t++) {
      while (slots[t].ack.load(std::memory_order_acquire) != (uint64_t)r) relax();
      const double d = (slots[t].seen_ns.load() - tb) * 1e-3;
      w = std::max(w, d); m += d / T;
    }
    worst.push_back(w); mean.push_back(m);
  }
  quit = true; epoch.store(1u << 30);
  for (auto& t : th) t.join();
  std::sort(worst.begin(), worst.end()); std::sort(mean.begin(), mean.end());
  std::printf("%d spinning threads, %d rounds: mean latency p50 %.2f us p90 %.2f; slowest thread p50 %.2f us p90 %.2f p99 %.2f; cpus allowed %d of %ld\n", T, rounds,
              mean[rounds / 2], mean[rounds * 9 / 10], worst[rounds / 2], worst[rounds * 9 / 10], worst[rounds * 99 / 100], CPU_COUNT(([]{ static cpu_set_t s; sched_getaffinity(0, sizeof s, &s); return &s; })()), sysconf(_SC_NPROCESSORS_ONLN));
  return 0;
}
