// TEST INFRASTRUCTURE: a stand-in for librccl.so that lets SEPARATE PROCESSES on ONE GPU run the multi-rank branch of
// datatable_amd/csrc/comm.hip (one process per rank, `dthip_comm_init`, ncclAllGather + grouped ncclSend / ncclRecv).
// RCCL itself refuses two ranks on one device, and the pool has single-GPU boxes; this library implements the nine entry
// points comm.hip resolves with dlsym on top of a POSIX shared-memory file: every collective stages device data through
// the file (hipMemcpy D2H / H2D) between process-shared barriers.  Loaded instead of librccl when DTHIP_RCCL_LIB names it
// (tests/test_gpu_rccl_2proc.py).  Semantics kept: all-gather in rank order; inside one group the k-th send of rank s to
// rank d matches the k-th receive of d from s, sizes must agree; calls are collective and block until all ranks arrive.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

// per rank; FAKE_RCCL_OUTBOX_MB widens it (every rank must see the same value)
static const size_t OUTBOX_BYTES = [] { const char* e = getenv("FAKE_RCCL_OUTBOX_MB"); size_t mb = e ? strtoull(e, nullptr, 10) : 0; return (mb ? mb : 96) << 20; }();
constexpr int MAX_RANKS = 16, MAX_MSGS = 4096;

struct Msg { int dst; size_t off, bytes; };
struct Shared {
  pthread_barrier_t bar;
  int ready;
  int nmsg[MAX_RANKS];
  Msg msg[MAX_RANKS][MAX_MSGS];
};

struct Comm {
  int rank = 0, world = 1;
  Shared* sh = nullptr;
  unsigned char* box = nullptr;       // world outboxes of OUTBOX_BYTES
  size_t map_bytes = 0;
};

struct Pending { bool send; const void* sbuf; void* rbuf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int g_group = 0;
thread_local std::vector<Pending> g_pending;
Comm* g_comm = nullptr;                 // the process's communicator (the tests create one per process): a rank with nothing to
                                        // send or receive in a group still has to meet the others at the barriers

const char* shm_dir() { const char* d = getenv("FAKE_RCCL_DIR"); return d ? d : "/tmp"; }

ncclResult_t run_group() {
  Comm* c = g_pending.empty() ? g_comm : g_pending[0].comm;
  if (!c) return ncclSuccess;
  for (auto& p : g_pending) if (hipStreamSynchronize(p.stream) != hipSuccess) return ncclUnhandledCudaError;
  // my sends -> my outbox
  size_t off = 0; int n = 0;
  unsigned char* mine = c->box + (size_t)c->rank * OUTBOX_BYTES;
  for (auto& p : g_pending) {
    if (!p.send) continue;
    if (off + p.bytes > OUTBOX_BYTES || n >= MAX_MSGS) { fprintf(stderr, "fake_rccl: outbox too small\n"); return ncclInternalError; }
    if (p.bytes && hipMemcpy(mine + off, p.sbuf, p.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    c->sh->msg[c->rank][n++] = Msg{p.peer, off, p.bytes};
    off += (p.bytes + 63) & ~(size_t)63;
  }
  c->sh->nmsg[c->rank] = n;
  pthread_barrier_wait(&c->sh->bar);
  // my receives: the k-th receive from s is the k-th message s addressed to me
  std::vector<int> next(c->world, 0);
  ncclResult_t rc = ncclSuccess;
  for (auto& p : g_pending) {
    if (p.send) continue;
    const int s = p.peer;
    int k = next[s];
    while (k < c->sh->nmsg[s] && c->sh->msg[s][k].dst != c->rank) k++;
    if (k >= c->sh->nmsg[s] || c->sh->msg[s][k].bytes != p.bytes) {
      fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from %d, found %s\n", c->rank, p.bytes, s, k >= c->sh->nmsg[s] ? "nothing" : "another size");
      rc = ncclInvalidUsage;
    } else if (p.bytes && hipMemcpy(p.rbuf, c->box + (size_t)s * OUTBOX_BYTES + c->sh->msg[s][k].off, p.bytes, hipMemcpyHostToDevice) != hipSuccess) {
      rc = ncclUnhandledCudaError;
    }
    next[s] = k + 1;
  }
  pthread_barrier_wait(&c->sh->bar);
  g_pending.clear();
  return rc;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "fakerccl-%d-%ld", (int)getpid(), (long)random());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  char path[512];
  snprintf(path, sizeof(path), "%s/%s", shm_dir(), id.internal);
  Comm* c = new Comm();
  c->rank = rank; c->world = nranks;
  c->map_bytes = sizeof(Shared) + (size_t)nranks * OUTBOX_BYTES;
  int fd = -1;
  if (rank == 0) {
    fd = open(path, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return ncclSystemError; }
  } else {
    for (int tries = 0; tries < 20000 && fd < 0; tries++) {          // rank 0 creates the file: wait for it (<= 20 s)
      fd = open(path, O_RDWR);
      struct stat st;
      if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < c->map_bytes)) { close(fd); fd = -1; }
      if (fd < 0) usleep(1000);
    }
    if (fd < 0) { delete c; return ncclSystemError; }
  }
  void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { delete c; return ncclSystemError; }
  c->sh = static_cast<Shared*>(m);
  c->box = static_cast<unsigned char*>(m) + sizeof(Shared);
  if (rank == 0) {
    pthread_barrierattr_t a;
    pthread_barrierattr_init(&a);
    pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&c->sh->bar, &a, (unsigned)nranks);
    __atomic_store_n(&c->sh->ready, 1, __ATOMIC_RELEASE);
  } else {
    for (int tries = 0; tries < 20000 && !__atomic_load_n(&c->sh->ready, __ATOMIC_ACQUIRE); tries++) usleep(1000);
    if (!c->sh->ready) { delete c; return ncclSystemError; }
  }
  pthread_barrier_wait(&c->sh->bar);
  if (rank == 0) unlink(path);
  g_comm = c;
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (c) { if (g_comm == c) g_comm = nullptr; munmap(c->sh, c->map_bytes); delete c; }
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error"; }

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (count > OUTBOX_BYTES) return ncclInternalError;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (count && hipMemcpy(c->box + (size_t)c->rank * OUTBOX_BYTES, send, count, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  pthread_barrier_wait(&c->sh->bar);
  ncclResult_t rc = ncclSuccess;
  for (int r = 0; r < c->world && count; r++)
    if (hipMemcpy(static_cast<unsigned char*>(recv) + (size_t)r * count, c->box + (size_t)r * OUTBOX_BYTES, count, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
  pthread_barrier_wait(&c->sh->bar);
  return rc;
}

ncclResult_t ncclGroupStart() { g_group++; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
  if (--g_group > 0) return ncclSuccess;
  return run_group();
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t stream) {
  g_pending.push_back(Pending{true, buf, nullptr, count, peer, reinterpret_cast<Comm*>(comm), stream});
  return g_group > 0 ? ncclSuccess : run_group();
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t comm, hipStream_t stream) {
  g_pending.push_back(Pending{false, nullptr, buf, count, peer, reinterpret_cast<Comm*>(comm), stream});
  return g_group > 0 ? ncclSuccess : run_group();
}

}  // extern "C"
