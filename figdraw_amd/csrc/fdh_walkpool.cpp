// fdh_walkpool.cpp -- a small fork-join pool for the scene front-end (fdh_frontend.cpp: ParallelWalk).
//
// The reference's renderer walks its node tree on one thread, every frame (figrender.nim:1960-2002), and its own benchmarks time
// that walk.  Here the walk of a large sibling group -- the roots of a layer, the cells of a table's viewport -- is split into
// chunks that pool threads decompose side by side, each into its own lane of draw records (fdh_context.h); the calling thread
// takes chunks too and then puts the pieces in painter's order.  Frames come tens of microseconds apart, so a pool thread that
// has just worked spins for a while before it sleeps: waking a sleeping thread costs more than a group's walk.
#include "fdh_walkpool.h"

#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace fdh {

namespace {
inline void relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}
struct alignas(128) Shared {
  // One word per group: (group number << 8) | helpers of that group.  A helper decides whether it takes part from the value it
  // LOADED, never from a field it reads afterwards: a helper that is not part of group N and is preempted after seeing N must not
  // mistake itself for part of group N + 1 (whose fields may be half written) and pass through that group twice -- run() would
  // return while another helper still works on the caller's stack.
  std::atomic<uint64_t> word{0};
  // A chunk is claimed by whoever sets its flag first.  Slot s starts with its HOME chunks s, s + slots, s + 2 slots, ...: the
  // same part of the same tree goes to the same thread frame after frame, whose core still holds those nodes (488 bytes each) and
  // its recorder's state -- handed out first come first served, every chunk's nodes came from another core's cache or from memory,
  // which on a host of many core complexes cost more than decomposing them.  A slot that has done its own then takes what others
  // have not started (a viewport's visible rows all lie in the first chunks).
  static constexpr int kMaxChunks = 512;
  std::atomic<uint8_t> claimed[kMaxChunks];
  std::atomic<int> through{0};        // helper slots that have finished the current group
  std::atomic<int> asleep{0};         // helpers blocked on the condition variable
  int slots = 1, n_chunks = 0;        // of the current group (written before the word is published; read only by its participants)
  const std::function<void(int, int)>* fn = nullptr;
  std::mutex mu;
  std::condition_variable cv;
  bool quit = false;
};
}  // namespace

// Where the helpers run.  A group's chunks read the same node array and write lanes the calling thread reads back microseconds
// later: on a host of many core complexes (the pool's boxes: 256 hardware threads, 16-thread complexes) a helper the scheduler
// put on another complex -- or the other socket -- took 2 - 5x as long over the same 44 records as the calling thread
// (a per-chunk trace, round 4: 3.1 us against 4 - 18), and a group ends with its slowest chunk.  Helpers are therefore confined to the hardware
// threads that share a last-level cache with the thread that first used the pool (Linux: cpuN/cache/index3/shared_cpu_list),
// that thread's own core left out.  An affinity MASK, not a pin: the scheduler still places them.  FDH_WALK_AFFINITY=0, a list
// that cannot be read, or one too short for the helpers: no mask.
#if defined(__linux__)
static bool parse_cpu_list(const char* path, cpu_set_t& out) {
  CPU_ZERO(&out);
  FILE* f = std::fopen(path, "r");
  if (!f) return false;
  char buf[1024];
  const bool ok = std::fgets(buf, sizeof buf, f) != nullptr;
  std::fclose(f);
  if (!ok) return false;
  for (const char* p = buf; *p && *p != '\n';) {
    char* e;
    const long a = std::strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    p = e;
    if (*p == '-') { b = std::strtol(p + 1, &e, 10); p = e; }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET((int)c, &out);
    if (*p == ',') p++;
  }
  return CPU_COUNT(&out) > 0;
}
static bool helper_mask(int helpers, cpu_set_t& mask) {
  if (const char* e = std::getenv("FDH_WALK_AFFINITY")) if (std::atoi(e) == 0) return false;
  const int cpu = sched_getcpu();
  if (cpu < 0) return false;
  cpu_set_t allowed, sibs;
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
  const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string(cpu);
  if (!parse_cpu_list((base + "/cache/index3/shared_cpu_list").c_str(), mask)) return false;
  CPU_AND(&mask, &mask, &allowed);
  if (parse_cpu_list((base + "/topology/thread_siblings_list").c_str(), sibs)) {  // the caller keeps its core to itself
    for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &sibs)) CPU_CLR(c, &mask);
  } else CPU_CLR(cpu, &mask);
  return CPU_COUNT(&mask) >= helpers;
}
#endif

static void take_chunks(Shared& sh, int slot, const std::function<void(int, int)>& fn) {
  const int n = sh.n_chunks, slots = sh.slots;
  for (int c = slot; c < n; c += slots)  // home chunks
    if (sh.claimed[c].exchange(1, std::memory_order_acq_rel) == 0) fn(slot, c);
  for (int c = 0; c < n; c++)            // then whatever nobody has started
    if (sh.claimed[c].load(std::memory_order_relaxed) == 0 && sh.claimed[c].exchange(1, std::memory_order_acq_rel) == 0) fn(slot, c);
}

struct WalkPoolImpl {
  Shared sh;
  std::mutex owner;  // one group at a time
  std::vector<std::thread> threads;
  int spin_us = 200;
  void helper_main(int slot) {
    uint64_t seen = 0;  // the group number this helper last looked at
    for (;;) {
      // wait for the next group: spin first (the next frame is tens of microseconds away), then sleep
      const auto t0 = std::chrono::steady_clock::now();
      uint64_t wd;
      int spins = 0;
      while (((wd = sh.word.load(std::memory_order_seq_cst)) >> 8) == seen) {
        relax();
        if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) {
          std::unique_lock<std::mutex> lk(sh.mu);
          sh.asleep.fetch_add(1, std::memory_order_seq_cst);
          // (seq_cst: run() publishes the word, THEN reads `asleep`; this side bumps `asleep`, THEN reads the word -- one of the two sees the other)
          sh.cv.wait(lk, [&] { return sh.quit || (sh.word.load(std::memory_order_seq_cst) >> 8) != seen; });
          sh.asleep.fetch_sub(1, std::memory_order_seq_cst);
          if (sh.quit) return;
        }
      }
      seen = wd >> 8;
      if (slot > (int)(wd & 0xff)) continue;  // this group uses fewer helpers (decided on the loaded word alone)
      // a participant: run() does not return, and no later group is published, before this helper has bumped `through` -- the
      // group's fields cannot change under it
      const std::function<void(int, int)>& fn = *sh.fn;
      take_chunks(sh, slot, fn);
      fn(slot, -1);
      sh.through.fetch_add(1, std::memory_order_acq_rel);
    }
  }
  ~WalkPoolImpl() {
    { std::lock_guard<std::mutex> lk(sh.mu); sh.quit = true; }
    sh.word.fetch_add(1u << 8, std::memory_order_seq_cst);
    sh.cv.notify_all();
    for (auto& t : threads) if (t.joinable()) t.join();
  }
};

static WalkPoolImpl& impl() {
  // (never destroyed: a pool thread may be mid-spin when the process exits; the OS takes the threads with it)
  static WalkPoolImpl* p = [] {
    auto* q = new WalkPoolImpl();
    return q;
  }();
  return *p;
}

WalkPool& WalkPool::get() {
  static WalkPool w;
  return w;
}

int WalkPool::default_helpers() {
  static const int v = [] {
    if (const char* e = std::getenv("FDH_WALK_THREADS")) return std::min(std::max(std::atoi(e), 0), 64);
    // the hardware threads this process may run on (a pinned rank, a container's cpuset), not the machine's: helpers that cannot be
    // scheduled beside the caller make every frame slower than the serial walk
    unsigned hw = std::thread::hardware_concurrency();
#if defined(__linux__)
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) hw = std::min<unsigned>(hw ? hw : 1u << 20, (unsigned)CPU_COUNT(&allowed));
#endif
    if (hw < 4) return 0;
    return (int)std::min(7u, hw / 8 + 1);  // a few: the walk of a frame is tens of microseconds, not a reason to take the machine
  }();
  return v;
}

bool WalkPool::run(int helpers, int n_chunks, const std::function<void(int, int)>& fn) {
  WalkPoolImpl& P = impl();
  if (helpers <= 0 || n_chunks <= 0 || n_chunks > Shared::kMaxChunks) return false;
  std::unique_lock<std::mutex> own(P.owner, std::try_to_lock);
  if (!own.owns_lock()) return false;
  helpers = std::min(helpers, 64);  // (fits the word's low byte)
  if ((int)P.threads.size() < helpers) {
#if defined(__linux__)
    cpu_set_t mask;
    const bool masked = helper_mask(helpers, mask);
#endif
    while ((int)P.threads.size() < helpers) {
      const int slot = (int)P.threads.size() + 1;
      P.threads.emplace_back([&P, slot] { P.helper_main(slot); });
#if defined(__linux__)
      if (masked) (void)pthread_setaffinity_np(P.threads.back().native_handle(), sizeof mask, &mask);
#endif
    }
  }
  Shared& sh = P.sh;
  sh.slots = helpers + 1;
  sh.n_chunks = n_chunks;
  sh.fn = &fn;
  for (int c = 0; c < n_chunks; c++) sh.claimed[c].store(0, std::memory_order_relaxed);
  sh.through.store(0, std::memory_order_relaxed);
  sh.word.store((((sh.word.load(std::memory_order_relaxed) >> 8) + 1) << 8) | (uint64_t)helpers, std::memory_order_seq_cst);  // (one writer: `owner` is held)
  if (sh.asleep.load(std::memory_order_seq_cst) > 0) {
    std::lock_guard<std::mutex> lk(sh.mu);
    sh.cv.notify_all();
  }
  take_chunks(sh, 0, fn);  // the calling thread is slot 0
  fn(0, -1);
  // every helper of this group passes through (one that slept through the work still bumps the counter when it wakes)
  // ... spinning briefly (they finish within microseconds of the caller), then yielding: a helper the scheduler has not placed -- a
  // CPU-limited container, a rank pinned to few cores -- needs this thread's core to run at all
  for (int spins = 0; sh.through.load(std::memory_order_acquire) < helpers; ) {
    if (++spins < 4096) relax(); else sched_yield();
  }
  return true;
}

}  // namespace fdh
