// fdh_walkpool.h -- the process-wide pool of threads the scene front-end decomposes large sibling groups on.
#pragma once
#include <functional>

namespace fdh {

// One group at a time: run() hands chunk numbers 0 .. n_chunks - 1 to `fn` on up to `helpers` pool threads (slots 1 .. helpers)
// and on the calling thread (slot 0) -- slot s its home chunks s, s + slots, ... first (the same nodes on the same core every frame),
// then any chunk nobody has started; when a slot finds no chunk left it is called once more with chunk = -1 (its chance to close what it kept per slot).  Returns when every slot is through;
// false, having done nothing, when another thread is using the pool (the caller then does the work itself).  `fn` must not throw.
class WalkPool {
 public:
  static WalkPool& get();
  bool run(int helpers, int n_chunks, const std::function<void(int slot, int chunk)>& fn);
  static int default_helpers();  // FDH_WALK_THREADS, or a few by the host's core count
};

}  // namespace fdh
