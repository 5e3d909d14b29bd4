/* TEST INFRASTRUCTURE -- a stand-in transport for the handful of RCCL entry points fdh_comm.cpp binds (FDH_RCCL_LIB points the
 * library at it).  RCCL refuses two ranks on one device ("Duplicate GPU detected") and the pool's boxes have one GPU, so the
 * N > 1 branch of fdh_gather_stripes / fdh_gather_frames -- who sends which rows where, the in-place receives, contexts sharing one
 * communicator from several host threads -- could never execute.  With this transport it does, between two PROCESSES on one GPU:
 * a send is a device-to-host copy into a mailbox in POSIX shared memory, a receive the host-to-device copy out of it, both
 * completed inside ncclGroupEnd (sends first).  It proves the library's call pattern, not RCCL or xGMI.
 * Build: hipcc -shared -fPIC tests/mock/mock_rccl.c -o build/libmock_rccl.so  (tests/test_abi_and_sharding.py does it). */
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define MAX_WORLD 8
#define BOX_BYTES ((size_t)40 << 20) /* one message: a 1080p frame or half an 8K stripe set does not occur in the tests */
typedef struct { _Atomic uint64_t written, taken; uint64_t bytes; uint8_t pad[104]; uint8_t data[BOX_BYTES]; } Box;
typedef struct { _Atomic int joined; int world; uint8_t pad[120]; Box box[MAX_WORLD][MAX_WORLD]; /* [src][dst] */ } Shm;
typedef struct { int rank, world; Shm* shm; char name[64]; } Comm;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { int send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; } Op;
static __thread Op ops[64];
static __thread int n_ops, depth;

static size_t shm_bytes(int world) { (void)world; return sizeof(Shm); }
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof *id);
  struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof id->internal, "/fdh_mock_rccl_%d_%ld", (int)getpid(), (long)ts.tv_nsec);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(Comm** out, int world, ncclUniqueId id, int rank) {
  if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return ncclInvalidArgument;
  Comm* c = calloc(1, sizeof *c);
  c->rank = rank; c->world = world;
  snprintf(c->name, sizeof c->name, "%s", id.internal);
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)shm_bytes(world)) != 0) return ncclSystemError;
  } else {
    for (int tries = 0; tries < 20000 && fd < 0; tries++) {  /* until rank 0 has created it at full size */
      fd = shm_open(c->name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < shm_bytes(world))) { close(fd); fd = -1; }
      if (fd < 0) usleep(1000);
    }
    if (fd < 0) return ncclSystemError;
  }
  c->shm = mmap(NULL, shm_bytes(world), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->shm == MAP_FAILED) return ncclSystemError;
  if (rank == 0) c->shm->world = world;
  atomic_fetch_add(&c->shm->joined, 1);
  for (int tries = 0; atomic_load(&c->shm->joined) < world; tries++) {  /* ncclCommInitRank is collective */
    if (tries > 60000) return ncclSystemError;
    usleep(1000);
  }
  *out = c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(Comm* c) {
  if (!c) return ncclSuccess;
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->shm, shm_bytes(c->world));
  free(c);
  return ncclSuccess;
}
ncclResult_t ncclCommCount(const Comm* c, int* n) { *n = c->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const Comm* c, int* r) { *r = c->rank; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "mock transport error"; }
ncclResult_t ncclGroupStart(void) { if (depth++ == 0) n_ops = 0; return ncclSuccess; }
static ncclResult_t queue(int send, void* buf, size_t count, int peer, Comm* c, hipStream_t s) {
  if (depth == 0 || n_ops >= 64 || !c || peer < 0 || peer >= c->world || peer == c->rank || count > BOX_BYTES) return ncclInvalidArgument;
  ops[n_ops++] = (Op){send, buf, count, peer, c, s};
  return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, int type, int peer, Comm* c, hipStream_t s) { (void)type; return queue(1, (void*)buf, count, peer, c, s); }
ncclResult_t ncclRecv(void* buf, size_t count, int type, int peer, Comm* c, hipStream_t s) { (void)type; return queue(0, buf, count, peer, c, s); }
static int wait_until(_Atomic uint64_t* a, uint64_t at_least) {
  for (long spins = 0; atomic_load(a) < at_least; spins++) {
    if (spins > 200000000L) return -1;
    if ((spins & 1023) == 1023) usleep(50);
  }
  return 0;
}
ncclResult_t ncclGroupEnd(void) {
  if (--depth > 0) return ncclSuccess;
  for (int pass = 1; pass >= 0; pass--)  /* sends, then receives */
    for (int i = 0; i < n_ops; i++) {
      Op* o = &ops[i];
      if (o->send != pass) continue;
      Box* b = o->send ? &o->comm->shm->box[o->comm->rank][o->peer] : &o->comm->shm->box[o->peer][o->comm->rank];
      if (o->send) {
        const uint64_t seq = atomic_load(&b->written);
        if (wait_until(&b->taken, seq) != 0) return ncclSystemError;  /* the last message of this pair has been taken */
        if (hipMemcpyAsync(b->data, o->buf, o->bytes, hipMemcpyDeviceToHost, o->stream) != hipSuccess || hipStreamSynchronize(o->stream) != hipSuccess) return ncclSystemError;
        b->bytes = o->bytes;
        atomic_store(&b->written, seq + 1);
      } else {
        const uint64_t seq = atomic_load(&b->taken);
        if (wait_until(&b->written, seq + 1) != 0) return ncclSystemError;
        if (b->bytes != o->bytes) return ncclInvalidArgument;  /* a send paired with the wrong receive */
        if (hipMemcpyAsync(o->buf, b->data, o->bytes, hipMemcpyHostToDevice, o->stream) != hipSuccess || hipStreamSynchronize(o->stream) != hipSuccess) return ncclSystemError;
        atomic_store(&b->taken, seq + 1);
      }
    }
  n_ops = 0;
  return ncclSuccess;
}
