// fdh_comm.cpp -- the one exchange step of the multi-GPU path: the gather of finished rows / frames to one rank over RCCL.
//
// frames are independent and row stripes re-render their blur halo themselves (fdh_set_stripe), so nothing is
// exchanged while a frame is rendered; the only collective is the gather of the RGBA8 result -- grouped ncclSend / ncclRecv over
// the point-to-point xGMI links, one contiguous run of rows per rank (a stripe of a row-major surface is contiguous).  One
// process per GPU: each process creates its context on its device and joins the communicator with fdh_comm_init.
// librccl is loaded at the first fdh_comm_* call (dlopen), not linked: a single-GPU host needs no RCCL to use the library.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "fdh_context.h"

namespace fdh {

// The handful of RCCL declarations the gather needs, spelled out here (nccl.h: stable ABI since NCCL 2): the library builds on a
// host without the RCCL headers, and runs without RCCL until a fdh_comm_* entry point is called.
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
}

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // (optional: what the communicator itself reports, fdh_comm_info)
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};
Rccl& rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = std::getenv("FDH_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (env && *env) R.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);  // a library named explicitly wins (RTLD_LOCAL: its symbols stay its own)
    for (const char* n : names) {  // otherwise a copy the process already holds (a PyTorch host brings its own) is taken first
      if (n && !R.lib) R.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    }
    for (const char* n : names) {
      if (n && !R.lib) R.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!R.lib) return;
    auto sym = [&](const char* s) { return dlsym(R.lib, s); };
    R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
    R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
    R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
    R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
    R.Send = reinterpret_cast<decltype(R.Send)>(sym("ncclSend"));
    R.Recv = reinterpret_cast<decltype(R.Recv)>(sym("ncclRecv"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
    R.CommCount = reinterpret_cast<decltype(R.CommCount)>(sym("ncclCommCount"));
    R.CommUserRank = reinterpret_cast<decltype(R.CommUserRank)>(sym("ncclCommUserRank"));
  });
  if (!R.lib || !R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.GroupStart || !R.GroupEnd || !R.Send || !R.Recv)
    throw Error(FDH_ERR_UNSUPPORTED, "librccl could not be loaded (set FDH_RCCL_LIB to its path): the multi-GPU gather needs RCCL");
  return R;
}
void nccl_check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return;
  Rccl& R = rccl();
  throw Error(FDH_ERR_HIP, std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(r) : "RCCL error"));
}
#define FDH_NCCL(x) nccl_check((x), #x)
// ncclGroupStart .. ncclGroupEnd around the sends / receives of one gather; a call that fails in between still closes the group
struct Group {
  Rccl& R;
  bool open = true;
  explicit Group(Rccl& r) : R(r) { FDH_NCCL(R.GroupStart()); }
  void end() { open = false; FDH_NCCL(R.GroupEnd()); }
  ~Group() { if (open) (void)R.GroupEnd(); }
};
}  // namespace

void stripe_rows(int height, int world, int rank, int* y0, int* y1) {
  if (height < 0 || world <= 0 || rank < 0 || rank >= world) throw Error(FDH_ERR_INVALID, "stripe_rows: bad partition");
  // contiguous, 8-row aligned (a strip of the compositor is 8 rows), covering [0, height) exactly once
  const int align = 8, tiles = (height + align - 1) / align, base = tiles / world, extra = tiles % world;
  const int t0 = rank * base + (rank < extra ? rank : extra), t1 = t0 + base + (rank < extra ? 1 : 0);
  *y0 = t0 * align < height ? t0 * align : height;
  *y1 = t1 * align < height ? t1 * align : height;
}

void comm_unique_id(uint8_t out[FDH_COMM_ID_BYTES]) {
  static_assert(FDH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as 128 opaque bytes");
  ncclUniqueId id;
  FDH_NCCL(rccl().GetUniqueId(&id));
  std::memcpy(out, id.internal, FDH_COMM_ID_BYTES);
}

// What the contexts of a process share: the communicator and the lock that keeps one gather's group (ncclGroupStart .. End) from
// interleaving with another context's on another host thread.
struct SharedComm {
  ncclComm_t comm = nullptr;
  std::mutex mu;
  ~SharedComm() { if (comm) { try { (void)rccl().CommDestroy(comm); } catch (...) {} } }
};

void Context::comm_init(const uint8_t id_bytes[FDH_COMM_ID_BYTES], int rank, int world) {
  need_device("comm_init");
  if (world <= 0 || rank < 0 || rank >= world) throw Error(FDH_ERR_INVALID, "comm_init: bad rank / world size");
  comm_destroy();
  FDH_HIP(hipSetDevice(device_));
  ncclUniqueId id;
  std::memcpy(id.internal, id_bytes, FDH_COMM_ID_BYTES);
  ncclComm_t c = nullptr;
  FDH_NCCL(rccl().CommInitRank(&c, world, id, rank));
  auto sc = std::make_shared<SharedComm>();
  sc->comm = c;
  // fdh_comm_info reports what the COMMUNICATOR says about itself, not what the caller asked for: a bench line that prints N ranks
  // then proves RCCL saw N (a communicator that disagrees with the request is refused)
  int seen_world = world, seen_rank = rank;
  if (rccl().CommCount) FDH_NCCL(rccl().CommCount(c, &seen_world));
  if (rccl().CommUserRank) FDH_NCCL(rccl().CommUserRank(c, &seen_rank));
  if (seen_world != world || seen_rank != rank) throw Error(FDH_ERR_HIP, "comm_init: the communicator reports another size / rank than it was created with");
  comm_ = std::static_pointer_cast<void>(sc);
  comm_rank_ = seen_rank;
  comm_world_ = seen_world;
}
// Several contexts of one process (frames in flight) use ONE communicator: the others borrow the owner's (a shared reference:
// the communicator is destroyed when its last holder lets go, in whichever order the contexts are destroyed).  RCCL runs a
// communicator's operations in the order they were issued, whichever stream each was issued on: a lock inside the shared object
// keeps two contexts' gathers (possibly on two host threads) from interleaving their groups, and EVERY RANK MUST ISSUE THE GATHERS
// OF A SHARED COMMUNICATOR IN THE SAME ORDER (context k's gather of frame n on all ranks, then context k + 1's, ...): a rank that
// issues them in another order pairs its sends with the wrong receives and the collective hangs.
void Context::comm_share(Context* owner) {
  need_device("comm_share");
  if (!owner || owner == this) throw Error(FDH_ERR_INVALID, "comm_share: needs another context that owns a communicator");
  if (owner->device_ != device_) throw Error(FDH_ERR_INVALID, "comm_share: both contexts must live on the same device");
  if (!owner->comm_) throw Error(FDH_ERR_INVALID, "comm_share: the other context holds no communicator");
  comm_destroy();
  comm_ = owner->comm_;
  comm_rank_ = owner->comm_rank_;
  comm_world_ = owner->comm_world_;
}
void Context::comm_destroy() {
  if (!comm_) return;
  drain();
  (void)hipStreamSynchronize(stream_);  // this context's transfers are done; the last holder's reset destroys the communicator
  comm_.reset();
  comm_rank_ = 0;
  comm_world_ = 1;
}

// Row-stripe mode: rank r holds rows stripe_rows(H, world, r) of the frame (fdh_set_stripe with those rows before rendering).
// One grouped send / recv on the context's stream, behind the frame's kernels: no host synchronisation.
void Context::gather_stripes(int dst_rank, void* dst_image) {
  need_device("gather_stripes");
  if (!fb_) throw Error(FDH_ERR_INVALID, "gather_stripes: no frame surface yet");
  const int world = comm_ ? comm_world_ : 1, rank = comm_ ? comm_rank_ : 0;
  if (dst_rank < 0 || dst_rank >= world) throw Error(FDH_ERR_INVALID, "gather_stripes: destination rank out of range");
  drain();  // the frame's launches are on the stream: the transfers queue behind them
  FDH_HIP(hipSetDevice(device_));
  const size_t row_bytes = (size_t)W_ * 4;
  int my0, my1;
  stripe_rows(H_, world, rank, &my0, &my1);
  // what travels is the rule's stripe: a context that rendered another one (fdh_set_stripe) would send rows it never produced
  if (stripe_y1_ > stripe_y0_ && (std::max(0, stripe_y0_) != my0 || std::min(H_, stripe_y1_) != my1))
    throw Error(FDH_ERR_INVALID, "gather_stripes: the stripe set with fdh_set_stripe is not this rank's fdh_stripe_rows(height, world, rank)");
  uint8_t* own = reinterpret_cast<uint8_t*>(fb_);
  uint8_t* dst = dst_image ? static_cast<uint8_t*>(dst_image) : own;
  if (rank == dst_rank && dst != own && my1 > my0)  // the destination's own rows: a copy on the device
    FDH_HIP(hipMemcpyAsync(dst + (size_t)my0 * row_bytes, own + (size_t)my0 * row_bytes, (size_t)(my1 - my0) * row_bytes, hipMemcpyDeviceToDevice, stream_));
  if (world == 1) return;
  Rccl& R = rccl();
  SharedComm& sc = *static_cast<SharedComm*>(comm_.get());
  std::lock_guard<std::mutex> lock(sc.mu);  // one group at a time on a shared communicator
  ncclComm_t comm = sc.comm;
  Group g(R);
  if (rank == dst_rank) {
    for (int r = 0; r < world; r++) {
      int y0, y1;
      stripe_rows(H_, world, r, &y0, &y1);
      if (r != rank && y1 > y0) FDH_NCCL(R.Recv(dst + (size_t)y0 * row_bytes, (size_t)(y1 - y0) * row_bytes, ncclUint8, r, comm, stream_));
    }
  } else if (my1 > my0) {
    FDH_NCCL(R.Send(own + (size_t)my0 * row_bytes, (size_t)(my1 - my0) * row_bytes, ncclUint8, dst_rank, comm, stream_));
  }
  g.end();
}

// Frame-parallel mode: every rank holds a whole frame; dst_rank receives rank r's into dst_images[r] (its own by a device copy).
void Context::gather_frames(int dst_rank, void* const* dst_images) {
  need_device("gather_frames");
  if (!fb_) throw Error(FDH_ERR_INVALID, "gather_frames: no frame surface yet");
  const int world = comm_ ? comm_world_ : 1, rank = comm_ ? comm_rank_ : 0;
  if (dst_rank < 0 || dst_rank >= world) throw Error(FDH_ERR_INVALID, "gather_frames: destination rank out of range");
  if (rank == dst_rank && !dst_images) throw Error(FDH_ERR_INVALID, "gather_frames: the destination rank needs one image per rank");
  if (rank == dst_rank)
    for (int r = 0; r < world; r++)
      if (r != rank && !dst_images[r]) throw Error(FDH_ERR_INVALID, "gather_frames: null destination image");
  drain();
  FDH_HIP(hipSetDevice(device_));
  const size_t bytes = (size_t)W_ * H_ * 4;
  if (rank == dst_rank && dst_images[rank] && dst_images[rank] != fb_)
    FDH_HIP(hipMemcpyAsync(dst_images[rank], fb_, bytes, hipMemcpyDeviceToDevice, stream_));
  if (world == 1) return;
  Rccl& R = rccl();
  SharedComm& sc = *static_cast<SharedComm*>(comm_.get());
  std::lock_guard<std::mutex> lock(sc.mu);
  ncclComm_t comm = sc.comm;
  Group g(R);
  if (rank == dst_rank) {
    for (int r = 0; r < world; r++)
      if (r != rank) FDH_NCCL(R.Recv(dst_images[r], bytes, ncclUint8, r, comm, stream_));
  } else {
    FDH_NCCL(R.Send(fb_, bytes, ncclUint8, dst_rank, comm, stream_));
  }
  g.end();
}

}  // namespace fdh
