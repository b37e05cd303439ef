// comm.hip -- multi-GPU DT[:, aggs, by(keys)] and DT[:, cols, by(keys)] INSIDE libdthip.so: one context per GPU,
// RCCL called directly (ncclAllGather for the few control words, ncclSend/ncclRecv inside one group for the
// all-to-all-v of the data), no PyTorch anywhere on the path.
//
// The reference is single-process; this exchange step is new.  Design = SURVEY 8(e):
//   * rows are sharded by row block (rank r owns a contiguous block of rows of the frame);
//   * aggregates: every rank first runs the fused local groupby-aggregate (the COMBINER: at most one partial per
//     local group crosses the fabric), partials are RANGE-partitioned on the first key and exchanged with one
//     all-to-all-v, the owner merges the <= world partials of each group with the same HIP kernels
//     (sum of sums, min of mins, ...; mean = sum(mean_i * n_i) / sum(n_i));
//   * row-returning queries: every row gets its destination from the same range partition, a stable local
//     partition by destination builds per-destination slabs in sender row order, one all-to-all-v moves them;
//     slabs arrive in source-rank order = global row order, so ONE stable local dthip_groupby_rows reproduces the
//     reference's permutation exactly (global row ids travel as a payload column);
//   * key-range (not hash) partitioning keeps the global group order = concatenation of the ranks' results in rank
//     order -- what the reference returns (groups ascending, NA group first / last);
//   * the splitters balance the destinations: every rank contributes quantiles of the order-preserving 64-bit image of
//     its first key -- exact ones on the aggregate path (its partial groups ascend), a stratified random sample's on the
//     rows path -- and the weighted union is cut at the world-quantiles (a plain even split of [min, max] sends skewed
//     keys to one rank).
// Collectives per call.  Aggregates: the partial groups of a rank are ASCENDING in the first key, so 1024 of them at
// evenly spaced positions are exact local quantiles: one all-gather of those samples gives every rank the same
// splitters (no key-image pass, no histogram over the partials), one all-gather carries the send counts, one 16-byte
// all-gather agrees on the status before the data moves.  Rows (round 5): 4096 sampled key images (32 KB), then send
// counts + status; a status-only round only when a share exceeds the receive bound.  Then one grouped all-to-all-v over all columns: xGMI is point-to-point, the grouped send/recv keeps
// all 7 links of a GPU busy at once.
// Failure handling: every all-gathered blob starts with {status, query signature, rows}; a rank whose local work failed
// keeps taking part in the all-gathers (with empty data), and after each of them ALL ranks see the failure and return
// together -- nobody is left waiting in the next collective for a peer that bailed out (no counterpart in the
// single-process reference; its errors are C++ exceptions on the one calling thread, sort.cc:672-673).
//
// A LOCAL communicator (dthip_comm_init_local) binds `world` contexts of one process -- on any devices, also all on
// the same one -- and runs exactly the same phases with device-to-device copies as the exchange: it is how the
// algorithm is tested on a single GPU (G logical shards) and has no RCCL dependency.
//
// librccl is opened with dlopen at dthip_comm_init: single-GPU users of libdthip.so never load it.
#include <dlfcn.h>
#include <algorithm>
#include <rccl/rccl.h>
#include <chrono>
#include "common.hpp"
#include "split_plan.hpp"

struct dthip_comm {
  int kind = 0;                       // 0: RCCL, 1: local (one process holds every rank)
  int world = 1;
  ncclComm_t nccl = nullptr;
  std::vector<dthip_ctx*> ranks;      // local: context of every rank
  int refs = 0;
  // RCCL: device staging of the small all-gathers, allocated ONCE at dthip_comm_init -- a rank must never fail between
  // deciding to all-gather and entering the collective (its peers would wait in it forever)
  unsigned char* ag_send = nullptr;
  unsigned char* ag_recv = nullptr;
  size_t ag_cap = 0;                  // bytes per rank
  // what the last sharded call moved (dthip_comm_last_stats): bytes / rows of the all-to-all-v that left this rank for
  // OTHER ranks, that stayed on it, rows received, all-gather rounds
  int64_t st_bytes_peer = 0, st_bytes_self = 0, st_rows_sent = 0, st_rows_recv = 0, st_allgathers = 0;
};

namespace dthip {

typedef unsigned long long u64;

// ---- RCCL through dlopen ----------------------------------------------------------------------
struct NcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static NcclApi g_nccl;

static int nccl_load() {
  if (g_nccl.handle) return DTHIP_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  // DTHIP_RCCL_LIB: another library with RCCL's entry points (tests/cpp/fake_rccl.cpp lets several PROCESSES share one
  // GPU, which RCCL refuses, so that the multi-rank branch below can run on a single-GPU box)
  if (const char* alt = getenv("DTHIP_RCCL_LIB")) {
    h = dlopen(alt, RTLD_NOW | RTLD_LOCAL);
    if (!h) { set_error("DTHIP_RCCL_LIB=%s could not be loaded: %s", alt, dlerror()); return DTHIP_EDEVICE; }
  }
  for (const char* nm : names) { if (h) break; h = dlopen(nm, RTLD_NOW | RTLD_LOCAL); }
  if (!h) { set_error("librccl.so could not be loaded: %s", dlerror()); return DTHIP_EDEVICE; }
#define NCCL_SYM(field, sym)                                                                  \
  g_nccl.field = reinterpret_cast<decltype(g_nccl.field)>(dlsym(h, #sym));                    \
  if (!g_nccl.field) { set_error("librccl: symbol %s missing", #sym); dlclose(h); return DTHIP_EDEVICE; }
  NCCL_SYM(GetUniqueId, ncclGetUniqueId)
  NCCL_SYM(CommInitRank, ncclCommInitRank)
  NCCL_SYM(CommDestroy, ncclCommDestroy)
  NCCL_SYM(AllGather, ncclAllGather)
  NCCL_SYM(Send, ncclSend)
  NCCL_SYM(Recv, ncclRecv)
  NCCL_SYM(GroupStart, ncclGroupStart)
  NCCL_SYM(GroupEnd, ncclGroupEnd)
  NCCL_SYM(GetErrorString, ncclGetErrorString)
#undef NCCL_SYM
  g_nccl.handle = h;
  return DTHIP_OK;
}

#define DTHIP_CHECK_NCCL(expr)                                                                           \
  do {                                                                                                   \
    ncclResult_t _r = (expr);                                                                            \
    if (_r != ncclSuccess) {                                                                             \
      set_error("%s failed: %s (%s:%d)", #expr, g_nccl.GetErrorString(_r), __FILE__, __LINE__);          \
      return DTHIP_EDEVICE;                                                                              \
    }                                                                                                    \
  } while (0)

// ---- kernels -------------------------------------------------------------------------------------
// Order-preserving 64-bit image of a key column's values: ascending integers / floats keep their order, NA maps to
// `na_img` (0 when the NA group comes first, ~0 when last).  A valid image is never 0 (that would be INT64_MIN, the
// NA sentinel itself); it IS ~0 for the int64 key INT64_MAX, which therefore counts as "not valid" in the range and
// the histogram when NAs sort last -- harmless, because the destination of a row depends on its image alone and both
// images ~0 (INT64_MAX and NA) belong to the last rank, where the local grouping tells them apart again
// (tests/test_gpu_sharded.py::test_int64_max_next_to_na_last).
__device__ __forceinline__ u64 key_image(const void* data, int stype, uint32_t i, u64 na_img) {
  switch (stype) {
    case DTHIP_BOOL: case DTHIP_INT8: { const int8_t v = static_cast<const int8_t*>(data)[i]; return v == INT8_MIN ? na_img : ((u64)(long long)v ^ 0x8000000000000000ULL); }
    case DTHIP_INT16: { const int16_t v = static_cast<const int16_t*>(data)[i]; return v == INT16_MIN ? na_img : ((u64)(long long)v ^ 0x8000000000000000ULL); }
    case DTHIP_INT32: { const int32_t v = static_cast<const int32_t*>(data)[i]; return v == INT32_MIN ? na_img : ((u64)(long long)v ^ 0x8000000000000000ULL); }
    case DTHIP_INT64: { const long long v = static_cast<const long long*>(data)[i]; return v == INT64_MIN ? na_img : ((u64)v ^ 0x8000000000000000ULL); }
    case DTHIP_FLOAT32: {
      const float f = static_cast<const float*>(data)[i];
      if (f != f) return na_img;
      const u64 t = (u64)__double_as_longlong((double)f);
      return t ^ (0x8000000000000000ULL | (0ULL - (t >> 63)));
    }
    default: {
      const double d = static_cast<const double*>(data)[i];
      if (d != d) return na_img;
      const u64 t = (u64)__double_as_longlong(d);
      return t ^ (0x8000000000000000ULL | (0ULL - (t >> 63)));
    }
  }
}


// grid-stride; one set of global atomics per workgroup (every wave hitting the same three addresses cost 5 ms per 1e7 keys)
__global__ void __launch_bounds__(256) key_image_kernel(const void* data, int stype, uint32_t n, u64 na_img, u64* img, RangeAcc* acc) {
  __shared__ u64 s_lo[4], s_hi[4];
  __shared__ uint32_t s_n[4];
  u64 lo = ~0ULL, hi = 0ULL; uint32_t ok = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const u64 v = key_image(data, stype, i, na_img);
    img[i] = v;
    if (v != na_img) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; ok++; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u64 l2 = ((u64)(uint32_t)__shfl_xor((int)(uint32_t)(lo >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)lo, o, 64);
    const u64 h2 = ((u64)(uint32_t)__shfl_xor((int)(uint32_t)(hi >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)hi, o, 64);
    lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    ok += (uint32_t)__shfl_xor((int)ok, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_lo[w] = lo; s_hi[w] = hi; s_n[w] = ok; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t cnt = 0;
    for (int k = 0; k < 4; k++) { lo = s_lo[k] < lo ? s_lo[k] : lo; hi = s_hi[k] > hi ? s_hi[k] : hi; cnt += s_n[k]; }
    if (cnt) { atomicMin(&acc->lo, lo); atomicMax(&acc->hi, hi); atomicAdd(&acc->nvalid, (u64)cnt); }
  }
}


// cuts[j] = first position of the key column (ASCENDING images) whose image is >= bounds[j]
__global__ void lower_bound_kernel(const void* data, int stype, uint32_t n, u64 na_img, const u64* bounds, int nb, uint32_t* cuts) {
  const int j = threadIdx.x;
  if (j >= nb) return;
  const u64 b = bounds[j];
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (key_image(data, stype, mid, na_img) < b) lo = mid + 1; else hi = mid; }
  cuts[j] = lo;
}

// out[i] = image of the key at position floor(i * n / q): the local q-quantiles of an ascending key column
__global__ void __launch_bounds__(256) sample_image_kernel(const void* data, int stype, uint32_t n, uint32_t q, u64 na_img, u64* out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < q) out[i] = key_image(data, stype, (uint32_t)(((u64)i * n) / q), na_img);
}

// out[i] = img[pos[i]]: the rows path's stratified sample of key images (positions from split_plan.hpp::row_sample_pos)
__global__ void __launch_bounds__(256) row_sample_kernel(const u64* __restrict__ img, const uint32_t* __restrict__ pos, uint32_t q, u64* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < q) out[i] = img[pos[i]];
}

// destination rank of every row: number of boundaries <= image (int8: world <= 127)
__global__ void __launch_bounds__(256) image_dest_kernel(const u64* img, uint32_t n, const u64* bounds, int nb, int8_t* dest) {
  __shared__ u64 sb[128];
  if ((int)threadIdx.x < nb) sb[threadIdx.x] = bounds[threadIdx.x];
  __syncthreads();
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 v = img[i];
  int lo = 0, hi = nb;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (sb[mid] <= v) lo = mid + 1; else hi = mid; }
  dest[i] = (int8_t)lo;
}

// weighted partial of a mean: mean_i * n_i (0 for an all-NA partial), float64
template <typename MT>
__global__ void __launch_bounds__(256) mean_weight_kernel(const MT* mean, const long long* cnt, uint32_t n, double* out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = cnt[i] > 0 ? (double)mean[i] * (double)cnt[i] : 0.0;
}

__global__ void __launch_bounds__(256) dest_count_kernel(const int8_t* dest, uint32_t n, u64* cnt) {
  __shared__ uint32_t h[128];
  if (threadIdx.x < 128) h[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) atomicAdd(&h[(uint32_t)dest[i] & 127u], 1u);
  __syncthreads();
  if (threadIdx.x < 128 && h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (u64)h[threadIdx.x]);
}

__global__ void __launch_bounds__(256) iota64_kernel(long long* out, uint32_t n, long long first) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = first + (long long)i;
}

// ---- one rank's state ---------------------------------------------------------------------------
struct XCol { const void* send = nullptr; void* recv = nullptr; int elem = 0; int stype = 0; };

struct Job {
  dthip_ctx* ctx = nullptr;
  int rank = 0;
  // small host blobs of the all-gathers
  std::vector<unsigned char> xin, xout;
  // all-to-all-v layout, in rows
  std::vector<int64_t> send_cnt, send_off, recv_cnt, recv_off;
  int64_t nsend = 0, nrecv = 0;
  std::vector<XCol> cols;
  // first-key images of the rows / partial groups that will be sent
  u64* img = nullptr; int64_t nimg = 0;
  u64 na_img = 0;
  RangeAcc range{~0ULL, 0ULL, 0ULL};
  std::vector<u64> bounds;
  Scratch* sc = nullptr;
  dthip_result* local = nullptr;       // the local partial result (agg) / the per-destination slabs (rows)
  dthip_result* out = nullptr;
  int rc = DTHIP_OK;                   // status of this rank's local work so far (travels in every blob)
  uint32_t sig = 0;                    // signature of the query this rank was called with
};

// Wall-clock phases of a sharded call, recorded next to the per-kernel times when profiling is on (dthip_profile_enable):
// "phase_local" (this rank's own work before the first exchange), "phase_allgather" (all small host-synchronous rounds),
// "phase_plan" (splitters, cuts, receive buffers), "phase_alltoallv", "phase_merge".  A lap synchronises the streams, so
// it is taken only while profiling -- the timed steps of bench.py run without it.
struct PhaseClock {
  std::vector<Job>* jobs; bool on; std::chrono::steady_clock::time_point t;
  explicit PhaseClock(std::vector<Job>& j) : jobs(&j), on(j[0].ctx->prof), t(std::chrono::steady_clock::now()) {}
  void lap(const char* name) {
    if (!on) return;
    for (auto& j : *jobs) { (void)hipSetDevice(j.ctx->device); (void)hipStreamSynchronize(j.ctx->stream); }
    const auto now = std::chrono::steady_clock::now();
    ProfAcc& a = (*jobs)[0].ctx->acc[name];
    a.ms += std::chrono::duration<double, std::milli>(now - t).count();
    a.n++;
    t = now;
  }
};

// ---- exchange primitives -------------------------------------------------------------------------
// every job's xin (same size on all ranks) -> every job's xout = concatenation over ranks
static int exchange_allgather(dthip_comm* comm, std::vector<Job>& jobs) {
  const size_t bytes = jobs[0].xin.size();
  comm->st_allgathers++;
  if (comm->kind == 1) {
    for (auto& j : jobs) {
      j.xout.resize(bytes * comm->world);
      for (auto& s : jobs) memcpy(j.xout.data() + (size_t)s.rank * bytes, s.xin.data(), bytes);
    }
    return DTHIP_OK;
  }
  Job& j = jobs[0];
  dthip_ctx* ctx = j.ctx;
  // the blob size is the same on every rank (a property of the protocol stage), so this refusal is taken by all ranks alike
  if (bytes > comm->ag_cap) { set_error("all-gather blob of %zu bytes exceeds the staging buffer (%zu)", bytes, comm->ag_cap); return DTHIP_EINVAL; }
  // from here on the collective is entered whatever happens locally: a failed copy is reported AFTER it
  int rc = DTHIP_OK;
  hipError_t he = hipMemcpyAsync(comm->ag_send, j.xin.data(), bytes, hipMemcpyHostToDevice, ctx->stream);
  if (he != hipSuccess) { set_error("all-gather staging copy failed: %s", hipGetErrorString(he)); rc = DTHIP_EDEVICE; }
  DTHIP_CHECK_NCCL(g_nccl.AllGather(comm->ag_send, comm->ag_recv, bytes, ncclInt8, comm->nccl, ctx->stream));
  j.xout.resize(bytes * comm->world);
  DTHIP_CHECK_HIP(hipMemcpyAsync(j.xout.data(), comm->ag_recv, bytes * comm->world, hipMemcpyDeviceToHost, ctx->stream));
  DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return rc;
}

// all-to-all-v of every column of every job (send_off / send_cnt / recv_off / recv_cnt in rows)
static int exchange_alltoallv(dthip_comm* comm, std::vector<Job>& jobs) {
  {
    const Job& j = jobs[0];
    int64_t rowb = 0;
    for (const auto& c : j.cols) rowb += c.elem;
    for (int p = 0; p < comm->world; p++) {
      if (p == j.rank) comm->st_bytes_self += j.send_cnt[p] * rowb; else comm->st_bytes_peer += j.send_cnt[p] * rowb;
      comm->st_rows_sent += j.send_cnt[p];
    }
    comm->st_rows_recv += j.nrecv;
  }
  if (comm->kind == 1) {
    for (auto& s : jobs) { DTHIP_CHECK_HIP(hipSetDevice(s.ctx->device)); DTHIP_CHECK_HIP(hipStreamSynchronize(s.ctx->stream)); }
    for (auto& d : jobs) {
      DTHIP_CHECK_HIP(hipSetDevice(d.ctx->device));
      for (auto& s : jobs) {
        const int64_t cnt = s.send_cnt[d.rank];
        if (cnt == 0) continue;
        for (size_t c = 0; c < d.cols.size(); c++) {
          const int e = d.cols[c].elem;
          const unsigned char* src = static_cast<const unsigned char*>(s.cols[c].send) + (size_t)s.send_off[d.rank] * e;
          unsigned char* dst = static_cast<unsigned char*>(d.cols[c].recv) + (size_t)d.recv_off[s.rank] * e;
          if (s.ctx->device == d.ctx->device)
            DTHIP_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)cnt * e, hipMemcpyDeviceToDevice, d.ctx->stream));
          else
            DTHIP_CHECK_HIP(hipMemcpyPeerAsync(dst, d.ctx->device, src, s.ctx->device, (size_t)cnt * e, d.ctx->stream));
        }
      }
    }
    for (auto& d : jobs) { DTHIP_CHECK_HIP(hipSetDevice(d.ctx->device)); DTHIP_CHECK_HIP(hipStreamSynchronize(d.ctx->stream)); }
    return DTHIP_OK;
  }
  Job& j = jobs[0];
  DTHIP_CHECK_NCCL(g_nccl.GroupStart());
  // an error inside the group must not leave it open: remember the first one, always reach GroupEnd
  ncclResult_t first = ncclSuccess;
  for (size_t c = 0; c < j.cols.size() && first == ncclSuccess; c++) {
    const int e = j.cols[c].elem;
    for (int p = 0; p < comm->world && first == ncclSuccess; p++) {
      if (j.send_cnt[p]) {
        const unsigned char* src = static_cast<const unsigned char*>(j.cols[c].send) + (size_t)j.send_off[p] * e;
        first = g_nccl.Send(src, (size_t)j.send_cnt[p] * e, ncclInt8, p, comm->nccl, j.ctx->stream);
      }
      if (j.recv_cnt[p] && first == ncclSuccess) {
        unsigned char* dst = static_cast<unsigned char*>(j.cols[c].recv) + (size_t)j.recv_off[p] * e;
        first = g_nccl.Recv(dst, (size_t)j.recv_cnt[p] * e, ncclInt8, p, comm->nccl, j.ctx->stream);
      }
    }
  }
  const ncclResult_t ge = g_nccl.GroupEnd();
  if (first != ncclSuccess) { set_error("ncclSend / ncclRecv failed inside the all-to-all-v group: %s", g_nccl.GetErrorString(first)); return DTHIP_EDEVICE; }
  if (ge != ncclSuccess) { set_error("ncclGroupEnd failed: %s", g_nccl.GetErrorString(ge)); return DTHIP_EDEVICE; }
  return DTHIP_OK;
}

// ---- blobs of the all-gathers: {status, signature, n} + payload ---------------------------------------------------
static void blob_set(Job& j, long long n, const void* payload, size_t bytes) {
  j.xin.assign(sizeof(ShardHdr) + bytes, 0);
  const ShardHdr h{j.rc, j.sig, n};
  memcpy(j.xin.data(), &h, sizeof(h));
  if (payload && bytes) memcpy(j.xin.data() + sizeof(h), payload, bytes);
}
static const unsigned char* blob_of(const Job& j, int r) { return j.xout.data() + (size_t)r * j.xin.size() + sizeof(ShardHdr); }
static ShardHdr hdr_of(const Job& j, int r) { ShardHdr h; memcpy(&h, j.xout.data() + (size_t)r * j.xin.size(), sizeof(h)); return h; }

// after an all-gather: did any rank fail so far, do all ranks run the same query?  Every rank sees the same blobs and
// takes the same decision, so either all go on or all return (the failing rank keeps its own error message).
static int agree(dthip_comm* comm, std::vector<Job>& jobs, const char* stage) {
  const Job& j0 = jobs[0];
  int rank = -1; bool sig_ok = true;
  const int rc = first_failure(j0.xout.data(), j0.xin.size(), comm->world, &rank, &sig_ok);
  if (!sig_ok) {
    set_error("sharded groupby: the ranks were called with different queries (key / value stypes, reducers or na_pos differ)");
    return DTHIP_EINVAL;
  }
  if (rc == DTHIP_OK) return DTHIP_OK;
  bool mine = false;
  for (const auto& j : jobs) if (j.rank == rank) mine = true;
  if (!mine) set_error("sharded groupby: rank %d failed during %s (code %d); every rank returns", rank, stage, rc);
  return rc;
}

// one 16-byte all-gather that only agrees on the status (before the data moves)
static int agree_round(dthip_comm* comm, std::vector<Job>& jobs, const char* stage) {
  for (auto& j : jobs) blob_set(j, 0, nullptr, 0);
  DTHIP_TRY(exchange_allgather(comm, jobs));
  return agree(comm, jobs, stage);
}

// ---- shared phases (rows) -----------------------------------------------------------------------------
// images of the first key of the rows + local range; the blob of the first all-gather
static int phase_images(Job& j, const void* key0, int stype, int64_t n, int na_pos, std::vector<u64>* samples = nullptr) {
  dthip_ctx* ctx = j.ctx;
  j.na_img = (na_pos == DTHIP_NA_LAST) ? ~0ULL : 0ULL;
  j.nimg = n;
  j.range = RangeAcc{~0ULL, 0ULL, 0ULL};
  if (n > 0) {
    DTHIP_TRY(j.sc->get<u64>((size_t)n, &j.img));
    RangeAcc* d_acc = nullptr;
    DTHIP_TRY(j.sc->get<RangeAcc>(1, &d_acc));
    DTHIP_CHECK_HIP(hipMemcpyAsync(d_acc, &j.range, sizeof(RangeAcc), hipMemcpyHostToDevice, ctx->stream));
    DTHIP_LAUNCH(ctx, "key_image_kernel", key_image_kernel, (unsigned)std::min<int64_t>((n + 255) / 256, 4096), 256, 0, key0, stype, (uint32_t)n, j.na_img, j.img, d_acc);
    if (samples) {
      // round 5: the rank's stratified sample of ROW_SAMPLES images, sorted = its approximate quantiles (the key range and
      // the histogram over it -- two all-gathers -- are not needed any more)
      std::vector<uint32_t> pos(ROW_SAMPLES);
      for (int i = 0; i < ROW_SAMPLES; i++) pos[i] = (uint32_t)row_sample_pos((unsigned)i, (unsigned long long)n);
      uint32_t* d_pos = nullptr; u64* d_s = nullptr;
      DTHIP_TRY(j.sc->get<uint32_t>(ROW_SAMPLES, &d_pos));
      DTHIP_TRY(j.sc->get<u64>(ROW_SAMPLES, &d_s));
      DTHIP_CHECK_HIP(hipMemcpyAsync(d_pos, pos.data(), sizeof(uint32_t) * ROW_SAMPLES, hipMemcpyHostToDevice, ctx->stream));
      DTHIP_LAUNCH(ctx, "row_sample_kernel", row_sample_kernel, ROW_SAMPLES / 256, 256, 0, j.img, d_pos, (uint32_t)ROW_SAMPLES, d_s);
      samples->resize(ROW_SAMPLES);
      DTHIP_TRY(read_back(ctx, samples->data(), d_s, sizeof(u64) * ROW_SAMPLES));
      std::sort(samples->begin(), samples->end());
    } else {
      DTHIP_TRY(read_back(ctx, &j.range, d_acc, sizeof(RangeAcc)));
    }
  }
  return DTHIP_OK;
}

static void layout_from_counts(Job& j, int world) {
  // blobs = world x world matrix of send counts (row = sender); recv counts = column `rank`
  j.recv_cnt.assign(world, 0); j.recv_off.assign(world, 0);
  int64_t off = 0;
  for (int s = 0; s < world; s++) {
    int64_t c = 0;
    memcpy(&c, blob_of(j, s) + sizeof(int64_t) * (size_t)j.rank, sizeof(c));
    j.recv_cnt[s] = c; j.recv_off[s] = off; off += c;
  }
  j.nrecv = off;
}

static void counts_blob(Job& j, int world) {
  if (j.rc != DTHIP_OK) { j.send_cnt.assign(world, 0); j.send_off.assign(world, 0); }
  blob_set(j, j.nsend, j.send_cnt.data(), sizeof(int64_t) * (size_t)world);
}

static int stage_dev(dthip_ctx* ctx, Scratch& sc, const dthip_col& c, int64_t n, int mem, dthip_col* out) {
  *out = c;
  if (mem == DTHIP_DEVICE || n == 0) return DTHIP_OK;
  unsigned char* d = nullptr;
  const size_t bytes = (size_t)n * stype_size(c.stype);
  DTHIP_TRY(sc.get<unsigned char>(bytes, &d));
  DTHIP_CHECK_HIP(hipMemcpyAsync(d, c.data, bytes, hipMemcpyHostToDevice, ctx->stream));
  out->data = d;
  return DTHIP_OK;
}

// ---- aggregates -------------------------------------------------------------------------------------
struct AggPlan {
  std::vector<dthip_agg> partial;                 // local partial reducers (deduplicated)
  struct Rec { int kind; int a, b; };             // kind: 0 copy partial a, 1 mean from (wsum a, count b)
  std::vector<Rec> recipe;
};

static int plan_partials(const dthip_agg* aggs, int naggs, AggPlan* p) {
  auto need = [&](int op, int col) {
    for (size_t i = 0; i < p->partial.size(); i++) if (p->partial[i].op == op && p->partial[i].col == col) return (int)i;
    p->partial.push_back(dthip_agg{op, col});
    return (int)p->partial.size() - 1;
  };
  for (int a = 0; a < naggs; a++) {
    const int op = aggs[a].op, col = aggs[a].col;
    switch (op) {
      case DTHIP_SUM: case DTHIP_MIN: case DTHIP_MAX: case DTHIP_COUNT: p->recipe.push_back({0, need(op, col), -1}); break;
      case DTHIP_COUNT0: p->recipe.push_back({0, need(DTHIP_COUNT0, -1), -1}); break;
      case DTHIP_MEAN: { const int m = need(DTHIP_MEAN, col); const int c = need(DTHIP_COUNT, col); p->recipe.push_back({1, m, c}); break; }
      default: set_error("sharded groupby: reducer op %d needs the row order of a whole group (first/last) and is not distributed", op); return DTHIP_ENOTIMPL;
    }
  }
  return DTHIP_OK;
}

struct AggArgs {
  const dthip_col* keys; int nkeys; const dthip_col* values; int nvalues; const dthip_agg* aggs; int naggs;
  int64_t nrows; int na_pos; int mem;
};

// Phases (every `local` lambda may fail on one rank; the failure travels in the next blob and all ranks return together):
//   1  local fused groupby-aggregate (the combiner) on device columns, float32 value columns widened to float64 so
//      that partial sums cross the fabric unrounded (the single-GPU path accumulates float32 sums in float64 and
//      rounds once; rounding every partial to float32 first would differ by more than a re-association);
//      1024 local quantiles of the first key's image                          -> all-gather A
//   2  splitters (same on every rank), cuts of the ascending partials, receive buffers of the splitters' guaranteed
//      bound                                                                   -> all-gather B (send counts + status)
//   3  exact layout; only if a share exceeds the bound (every rank sees that): exact buffers -> all-gather C (status)
//   4  all-to-all-v of keys + partial columns, merge on the owner
static int run_sharded_agg(dthip_comm* comm, std::vector<Job>& jobs, const std::vector<AggArgs>& args, const AggPlan& plan) {
  const int world = comm->world;
  const int nkeys = args[0].nkeys, nvalues = args[0].nvalues;
  const int np = (int)plan.partial.size();
  std::vector<std::vector<double*>> wsum(jobs.size());
  std::vector<std::vector<dthip_col>> kd(jobs.size()), vd(jobs.size());
  PhaseClock clock(jobs);
  // ---- 1: local combiner + quantile samples of the partial groups' first key
  for (size_t q = 0; q < jobs.size(); q++) {
    Job& j = jobs[q]; const AggArgs& a = args[q]; dthip_ctx* ctx = j.ctx;
    std::vector<u64> samples(SPLIT_SAMPLES, 0);
    j.na_img = (a.na_pos == DTHIP_NA_LAST) ? ~0ULL : 0ULL;
    j.nimg = 0;
    auto local = [&]() -> int {
      DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
      kd[q].resize(nkeys); vd[q].resize(nvalues);
      if (ctx->f32_sum_ref) {
        // option "f32_sum" = 1 promises the reference's float32 accumulation in grouped ROW order; partial sums of row
        // shards merged across ranks cannot reproduce that order, so the sharded call says so instead of ignoring it
        for (int i = 0; i < a.naggs; i++)
          if (a.aggs[i].op == DTHIP_SUM && a.aggs[i].col >= 0 && a.aggs[i].col < nvalues && a.values[a.aggs[i].col].stype == DTHIP_FLOAT32) {
            set_error("sharded groupby: option f32_sum = 1 (float32 sums accumulated row by row) applies to single-GPU calls only");
            return DTHIP_ENOTIMPL;
          }
      }
      for (int k = 0; k < nkeys; k++) DTHIP_TRY(stage_dev(ctx, *j.sc, a.keys[k], a.nrows, a.mem, &kd[q][k]));
      for (int c = 0; c < nvalues; c++) {
        DTHIP_TRY(stage_dev(ctx, *j.sc, a.values[c], a.nrows, a.mem, &vd[q][c]));
        if (vd[q][c].stype == DTHIP_FLOAT32 && a.nrows > 0) {
          double* wide = nullptr;
          DTHIP_TRY(j.sc->get<double>((size_t)a.nrows, &wide));
          DTHIP_TRY(launch_gather_f64(ctx, vd[q][c].data, DTHIP_FLOAT32, nullptr, a.nrows, wide));
          vd[q][c].data = wide;
        }
        if (vd[q][c].stype == DTHIP_FLOAT32) vd[q][c].stype = DTHIP_FLOAT64;
      }
      const int saved = ctx->agg_offsets;
      ctx->agg_offsets = 0;
      const int rc = dthip_groupby_agg(ctx, kd[q].data(), nkeys, vd[q].data(), nvalues, plan.partial.data(), np, a.nrows, a.na_pos, DTHIP_DEVICE, &j.local);
      ctx->agg_offsets = saved;
      DTHIP_TRY(rc);
      const int64_t ng = j.local->ngroups;
      // mean partials travel as weighted sums
      wsum[q].assign(np, nullptr);
      for (int i = 0; i < np; i++) {
        if (plan.partial[i].op != DTHIP_MEAN || ng == 0) continue;
        int ci = -1;
        for (int t = 0; t < np; t++) if (plan.partial[t].op == DTHIP_COUNT && plan.partial[t].col == plan.partial[i].col) ci = t;
        DTHIP_TRY(j.sc->get<double>((size_t)ng, &wsum[q][i]));
        const long long* cnt = static_cast<const long long*>(j.local->agg[ci]);
        if (j.local->agg_stype[i] == DTHIP_FLOAT32)
          DTHIP_LAUNCH(ctx, "mean_weight_kernel", mean_weight_kernel<float>, (unsigned)((ng + 255) / 256), 256, 0, static_cast<const float*>(j.local->agg[i]), cnt, (uint32_t)ng, wsum[q][i]);
        else
          DTHIP_LAUNCH(ctx, "mean_weight_kernel", mean_weight_kernel<double>, (unsigned)((ng + 255) / 256), 256, 0, static_cast<const double*>(j.local->agg[i]), cnt, (uint32_t)ng, wsum[q][i]);
      }
      if (ng > 0) {
        u64* d_s = nullptr;
        DTHIP_TRY(j.sc->get<u64>(SPLIT_SAMPLES, &d_s));
        DTHIP_LAUNCH(ctx, "sample_image_kernel", sample_image_kernel, SPLIT_SAMPLES / 256, 256, 0, j.local->key[0], a.keys[0].stype, (uint32_t)ng, (uint32_t)SPLIT_SAMPLES, j.na_img, d_s);
        DTHIP_TRY(read_back(ctx, samples.data(), d_s, sizeof(u64) * SPLIT_SAMPLES));
      }
      j.nimg = ng;
      return DTHIP_OK;
    };
    if (j.rc == DTHIP_OK) j.rc = local();
    blob_set(j, j.rc == DTHIP_OK ? j.nimg : 0, samples.data(), sizeof(u64) * SPLIT_SAMPLES);
  }
  clock.lap("phase_local");
  DTHIP_TRY(exchange_allgather(comm, jobs));
  DTHIP_TRY(agree(comm, jobs, "the local aggregation"));
  clock.lap("phase_allgather");
  // ---- 2: splitters -> contiguous slabs of the (ascending) partial groups
  int64_t recv_bound = 0;
  {
    long long total = 0;
    for (int r = 0; r < world; r++) total += hdr_of(jobs[0], r).n;
    recv_bound = total / world + total / SPLIT_SAMPLES + 2 * (int64_t)world + 16;
  }
  auto alloc_recv = [&](size_t q, int64_t rows) -> int {
    Job& j = jobs[q]; const AggArgs& a = args[q];
    for (auto& c : j.cols) if (c.recv) j.sc->release(c.recv);        // (a share above the bound: the bound-sized buffers go first)
    j.cols.clear();
    for (int k = 0; k < nkeys; k++) {
      XCol c; c.stype = a.keys[k].stype; c.elem = stype_size(c.stype); c.send = j.local->key[k];
      unsigned char* r = nullptr; DTHIP_TRY(j.sc->get<unsigned char>((size_t)rows * c.elem + 16, &r)); c.recv = r;
      j.cols.push_back(c);
    }
    for (int i = 0; i < np; i++) {
      XCol c;
      if (plan.partial[i].op == DTHIP_MEAN) { c.stype = DTHIP_FLOAT64; c.send = wsum[q][i]; }
      else { c.stype = j.local->agg_stype[i]; c.send = j.local->agg[i]; }
      c.elem = stype_size(c.stype);
      unsigned char* r = nullptr; DTHIP_TRY(j.sc->get<unsigned char>((size_t)rows * c.elem + 16, &r)); c.recv = r;
      j.cols.push_back(c);
    }
    return DTHIP_OK;
  };
  for (size_t q = 0; q < jobs.size(); q++) {
    Job& j = jobs[q]; const AggArgs& a = args[q]; dthip_ctx* ctx = j.ctx;
    std::vector<u64> smp((size_t)world * SPLIT_SAMPLES);
    std::vector<long long> cnt(world);
    for (int r = 0; r < world; r++) { memcpy(&smp[(size_t)r * SPLIT_SAMPLES], blob_of(j, r), sizeof(u64) * SPLIT_SAMPLES); cnt[r] = hdr_of(j, r).n; }
    sample_bounds(smp.data(), cnt.data(), world, &j.bounds);
    j.send_cnt.assign(world, 0); j.send_off.assign(world, 0);
    auto local = [&]() -> int {
      DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
      std::vector<uint32_t> cuts(world + 1, 0);
      cuts[world] = (uint32_t)j.nimg;
      if (world > 1 && j.nimg > 0) {
        u64* d_b = nullptr; uint32_t* d_c = nullptr;
        DTHIP_TRY(j.sc->get<u64>(world, &d_b));
        DTHIP_TRY(j.sc->get<uint32_t>(world, &d_c));
        DTHIP_CHECK_HIP(hipMemcpyAsync(d_b, j.bounds.data(), sizeof(u64) * (world - 1), hipMemcpyHostToDevice, ctx->stream));
        DTHIP_LAUNCH(ctx, "lower_bound_kernel", lower_bound_kernel, 1, 128, 0, j.local->key[0], a.keys[0].stype, (uint32_t)j.nimg, j.na_img, d_b, world - 1, d_c);
        DTHIP_TRY(read_back(ctx, cuts.data() + 1, d_c, sizeof(uint32_t) * (world - 1)));
      }
      for (int k = 0; k < world; k++) { j.send_off[k] = cuts[k]; j.send_cnt[k] = (int64_t)cuts[k + 1] - (int64_t)cuts[k]; }
      // receive buffers BEFORE the send counts are known (round 4: saves the status-only all-gather that used to follow
      // their allocation): quantile splitters give a rank at most its fair share + total / SPLIT_SAMPLES partial groups
      // + the <= world partials that share a boundary key (split_plan.hpp::sample_bounds) -- the same number on every
      // rank, checked against the exact counts after the next all-gather
      return alloc_recv(q, recv_bound);
    };
    j.rc = local();
    j.nsend = j.nimg;
    counts_blob(j, world);
  }
  clock.lap("phase_plan");
  DTHIP_TRY(exchange_allgather(comm, jobs));
  DTHIP_TRY(agree(comm, jobs, "the partition of the partial groups / the allocation of the receive buffers"));
  clock.lap("phase_allgather");
  // ---- 3: exact receive layout; every rank sees the whole count matrix, so all ranks agree WITHOUT another round on
  // whether somebody's share exceeds the bound (then: exact buffers and the status round of rounds 2-3)
  bool exceeded = false;
  for (auto& j : jobs) layout_from_counts(j, world);
  for (int r = 0; r < world; r++) {
    int64_t recv_r = 0;
    for (int s_ = 0; s_ < world; s_++) { int64_t c = 0; memcpy(&c, blob_of(jobs[0], s_) + sizeof(int64_t) * (size_t)r, sizeof(c)); recv_r += c; }
    if (recv_r > recv_bound) exceeded = true;
  }
  if (exceeded) {
    for (size_t q = 0; q < jobs.size(); q++) {
      Job& j = jobs[q];
      auto local = [&]() -> int { DTHIP_CHECK_HIP(hipSetDevice(j.ctx->device)); return alloc_recv(q, j.nrecv); };
      j.rc = local();
    }
    clock.lap("phase_plan");
    DTHIP_TRY(agree_round(comm, jobs, "the allocation of the receive buffers"));
    clock.lap("phase_allgather");
  }
  // ---- 4: all-to-all-v of keys + partial columns
  DTHIP_TRY(exchange_alltoallv(comm, jobs));
  clock.lap("phase_alltoallv");
  // ---- merge on the owner
  for (size_t q = 0; q < jobs.size(); q++) {
    Job& j = jobs[q]; const AggArgs& a = args[q]; dthip_ctx* ctx = j.ctx;
    DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
    std::vector<dthip_col> mk(nkeys), mv(np);
    std::vector<dthip_agg> ma(np);
    for (int k = 0; k < nkeys; k++) mk[k] = dthip_col{j.cols[k].recv, a.keys[k].stype, a.keys[k].flags};
    for (int i = 0; i < np; i++) {
      const int op = plan.partial[i].op;
      const bool summed = op == DTHIP_SUM || op == DTHIP_MEAN || op == DTHIP_COUNT || op == DTHIP_COUNT0;
      // partial sums are VALUES (a NaN / INT64_MIN partial must not be skipped as NA): DTHIP_FLAG_NONA
      mv[i] = dthip_col{j.cols[nkeys + i].recv, j.cols[nkeys + i].stype, summed ? DTHIP_FLAG_NONA : 0};
      ma[i] = dthip_agg{summed ? DTHIP_SUM : op, i};
    }
    dthip_result* m = nullptr;
    const int saved = ctx->agg_offsets; const bool saved_merge = ctx->in_merge;
    ctx->agg_offsets = 0; ctx->in_merge = true;
    const int rc = dthip_groupby_agg(ctx, mk.data(), nkeys, mv.data(), np, ma.data(), np, j.nrecv, a.na_pos, DTHIP_DEVICE, &m);
    ctx->agg_offsets = saved; ctx->in_merge = saved_merge;
    DTHIP_TRY(rc);
    // the merged result becomes the output: requested aggregates point at (or are computed from) its columns
    const int64_t ng = m->ngroups;
    std::vector<void*> pa = m->agg;
    std::vector<int> ps = m->agg_stype;
    m->naggs = a.naggs;
    m->agg.assign(a.naggs, nullptr); m->agg_stype.assign(a.naggs, 0);
    m->offsets = nullptr;
    j.out = m;
    for (int t = 0; t < a.naggs; t++) {
      const AggPlan::Rec& r = plan.recipe[t];
      const int ost = dthip_reduce_out_stype(a.aggs[t].op, a.aggs[t].op == DTHIP_COUNT0 ? DTHIP_INT64 : a.values[a.aggs[t].col].stype);
      m->agg_stype[t] = ost;
      if (r.kind == 0 && ps[r.a] == ost) { m->agg[t] = pa[r.a]; continue; }
      void* o = nullptr;
      DTHIP_TRY(result_alloc(ctx, m, (size_t)ng * stype_size(ost) + 16, &o));
      m->agg[t] = o;
      if (!ng) continue;
      if (r.kind == 0) {
        // sum / min / max of a float32 column: merged in float64, rounded to the column's stype once
        if (ps[r.a] != DTHIP_FLOAT64 || ost != DTHIP_FLOAT32) { set_error("sharded groupby: partial stype %d for output stype %d", ps[r.a], ost); return DTHIP_EDEVICE; }
        DTHIP_TRY(launch_cast_f64_f32(ctx, static_cast<const double*>(pa[r.a]), ng, static_cast<float*>(o)));
      } else {
        DTHIP_TRY(launch_mean_div(ctx, static_cast<const double*>(pa[r.a]), static_cast<const long long*>(pa[r.b]), ng, o, ost == DTHIP_FLOAT32));
      }
    }
    m->nrows = a.nrows;
  }
  clock.lap("phase_merge");
  return DTHIP_OK;
}

// ---- rows in grouped order ---------------------------------------------------------------------------
struct RowsArgs {
  const dthip_col* keys; int nkeys; const dthip_col* cols; int ncols;
  int64_t nrows; int64_t row_offset; int na_pos; int mem;
};

// Phases (round 5: two all-gathers instead of four): 1 key images + a stratified sample of them -> all-gather A (the samples);
// 2 splitters from the weighted union of the samples, destination of every row, stable partition by destination, receive
// buffers of the common bound -> all-gather B (send counts + status);  [a share above the bound: exact buffers + a status
// round];  3 all-to-all-v, one stable local grouping.
static int run_sharded_rows(dthip_comm* comm, std::vector<Job>& jobs, const std::vector<RowsArgs>& args) {
  const int world = comm->world;
  const int nkeys = args[0].nkeys, ncols = args[0].ncols;
  std::vector<std::vector<dthip_col>> kd(jobs.size()), cd(jobs.size());
  std::vector<long long*> rowid(jobs.size(), nullptr);
  PhaseClock clock(jobs);
  // ---- 1: key images + the rank's stratified sample of them (its approximate quantiles)
  for (size_t q = 0; q < jobs.size(); q++) {
    Job& j = jobs[q]; const RowsArgs& a = args[q]; dthip_ctx* ctx = j.ctx;
    std::vector<u64> smp(ROW_SAMPLES, 0);
    auto local = [&]() -> int {
      DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
      kd[q].resize(nkeys); cd[q].resize(ncols);
      for (int k = 0; k < nkeys; k++) DTHIP_TRY(stage_dev(ctx, *j.sc, a.keys[k], a.nrows, a.mem, &kd[q][k]));
      for (int c = 0; c < ncols; c++) DTHIP_TRY(stage_dev(ctx, *j.sc, a.cols[c], a.nrows, a.mem, &cd[q][c]));
      if (a.nrows) {
        DTHIP_TRY(j.sc->get<long long>((size_t)a.nrows, &rowid[q]));
        DTHIP_LAUNCH(ctx, "iota64_kernel", iota64_kernel, (unsigned)((a.nrows + 255) / 256), 256, 0, rowid[q], (uint32_t)a.nrows, (long long)a.row_offset);
      }
      return phase_images(j, kd[q][0].data, kd[q][0].stype, a.nrows, a.na_pos, &smp);
    };
    if (j.rc == DTHIP_OK) j.rc = local();
    if (j.rc != DTHIP_OK) j.nimg = 0;
    smp.resize(ROW_SAMPLES, 0);
    blob_set(j, j.nimg, smp.data(), sizeof(u64) * ROW_SAMPLES);
  }
  clock.lap("phase_local");
  DTHIP_TRY(exchange_allgather(comm, jobs));
  DTHIP_TRY(agree(comm, jobs, "the key images"));
  clock.lap("phase_allgather");
  // ---- 2: splitters from the weighted union of the samples, destination of every row, slabs in sender row order;
  // receive buffers of the bound every rank computes alike (fair share + 1/8 of all rows) BEFORE the counts travel
  const int npay = nkeys + ncols + 1;                 // keys, columns, global row id
  int64_t recv_bound = 0;
  {
    long long total = 0;
    for (int r = 0; r < world; r++) total += hdr_of(jobs[0], r).n;
    recv_bound = rows_recv_bound(total, world);
  }
  auto alloc_recv = [&](Job& j, int64_t rows) -> int {
    for (auto& c : j.cols) if (c.recv) { j.sc->release(c.recv); c.recv = nullptr; }     // (the bound-sized buffers, when a share exceeds them)
    for (auto& c : j.cols) { unsigned char* r = nullptr; DTHIP_TRY(j.sc->get<unsigned char>((size_t)rows * c.elem + 16, &r)); c.recv = r; }
    return DTHIP_OK;
  };
  for (size_t q = 0; q < jobs.size(); q++) {
    Job& j = jobs[q]; const RowsArgs& a = args[q]; dthip_ctx* ctx = j.ctx;
    {
      std::vector<u64> all((size_t)world * ROW_SAMPLES);
      std::vector<long long> cnt(world);
      for (int r = 0; r < world; r++) { memcpy(&all[(size_t)r * ROW_SAMPLES], blob_of(j, r), sizeof(u64) * ROW_SAMPLES); cnt[r] = hdr_of(j, r).n; }
      sample_bounds_q(all.data(), cnt.data(), world, ROW_SAMPLES, &j.bounds);
    }
    j.send_cnt.assign(world, 0); j.send_off.assign(world, 0);
    auto local = [&]() -> int {
      DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
      j.cols.assign(npay, XCol());
      for (int k = 0; k < nkeys; k++) { j.cols[k].stype = kd[q][k].stype; j.cols[k].send = kd[q][k].data; }
      for (int c = 0; c < ncols; c++) { j.cols[nkeys + c].stype = cd[q][c].stype; j.cols[nkeys + c].send = cd[q][c].data; }
      j.cols[npay - 1].stype = DTHIP_INT64; j.cols[npay - 1].send = rowid[q];
      for (auto& c : j.cols) c.elem = stype_size(c.stype);
      if (world > 1 && a.nrows > 0) {
        int8_t* dest = nullptr; u64* d_b = nullptr;
        DTHIP_TRY(j.sc->get<int8_t>((size_t)a.nrows, &dest));
        DTHIP_TRY(j.sc->get<u64>(world, &d_b));
        DTHIP_CHECK_HIP(hipMemcpyAsync(d_b, j.bounds.data(), sizeof(u64) * (world - 1), hipMemcpyHostToDevice, ctx->stream));
        DTHIP_LAUNCH(ctx, "image_dest_kernel", image_dest_kernel, (unsigned)((a.nrows + 255) / 256), 256, 0, j.img, (uint32_t)a.nrows, d_b, world - 1, dest);
        // stable partition by destination: the library's own rows-in-grouped-order on the int8 destination
        dthip_col dk{dest, DTHIP_INT8, 0};
        std::vector<dthip_col> pay(npay);
        for (int c = 0; c < npay; c++) pay[c] = dthip_col{j.cols[c].send, j.cols[c].stype, 0};
        DTHIP_TRY(dthip_groupby_rows(ctx, &dk, 1, pay.data(), npay, a.nrows, DTHIP_NA_FIRST, DTHIP_DEVICE, 0, &j.local));
        // slabs are contiguous and in ascending destination order: their sizes are the destination counts
        u64* d_cnt = nullptr;
        DTHIP_TRY(j.sc->get<u64>(128, &d_cnt));
        DTHIP_CHECK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(u64) * 128, ctx->stream));
        DTHIP_LAUNCH(ctx, "dest_count_kernel", dest_count_kernel, (unsigned)std::min<int64_t>((a.nrows + 255) / 256, 2048), 256, 0, dest, (uint32_t)a.nrows, d_cnt);
        u64 cnt[128];
        DTHIP_TRY(read_back(ctx, cnt, d_cnt, sizeof(cnt)));
        int64_t off = 0;
        for (int d = 0; d < world; d++) { j.send_off[d] = off; j.send_cnt[d] = (int64_t)cnt[d]; off += (int64_t)cnt[d]; }
        for (int c = 0; c < npay; c++) j.cols[c].send = j.local->col[c];
      } else if (world == 1) {
        j.send_cnt[0] = a.nrows;
      }
      return alloc_recv(j, recv_bound);
    };
    j.rc = local();
    j.nsend = a.nrows;
    counts_blob(j, world);
  }
  clock.lap("phase_local");
  DTHIP_TRY(exchange_allgather(comm, jobs));
  DTHIP_TRY(agree(comm, jobs, "the partition of the rows / the allocation of the receive buffers"));
  clock.lap("phase_allgather");
  // ---- 3: exact receive layout; every rank sees the whole count matrix, so all ranks agree WITHOUT another round on whether
  // somebody's share exceeds the bound (one key holding most of the rows: then exact buffers and a status round)
  bool exceeded = false;
  for (auto& j : jobs) layout_from_counts(j, world);
  for (int r = 0; r < world; r++) {
    int64_t recv_r = 0;
    for (int s_ = 0; s_ < world; s_++) { int64_t c = 0; memcpy(&c, blob_of(jobs[0], s_) + sizeof(int64_t) * (size_t)r, sizeof(c)); recv_r += c; }
    if (recv_r > recv_bound) exceeded = true;
  }
  if (exceeded) {
    for (auto& j : jobs) {
      auto local = [&]() -> int { DTHIP_CHECK_HIP(hipSetDevice(j.ctx->device)); return alloc_recv(j, j.nrecv); };
      j.rc = local();
    }
    clock.lap("phase_plan");
    DTHIP_TRY(agree_round(comm, jobs, "the allocation of the receive buffers"));
    clock.lap("phase_allgather");
  }
  // ---- 5
  DTHIP_TRY(exchange_alltoallv(comm, jobs));
  clock.lap("phase_alltoallv");
  // one stable local grouping of what arrived (source-rank order = global row order)
  for (size_t q = 0; q < jobs.size(); q++) {
    Job& j = jobs[q]; const RowsArgs& a = args[q]; dthip_ctx* ctx = j.ctx;
    DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
    std::vector<dthip_col> mk(nkeys), mc(ncols + 1);
    for (int k = 0; k < nkeys; k++) mk[k] = dthip_col{j.cols[k].recv, a.keys[k].stype, a.keys[k].flags};
    for (int c = 0; c <= ncols; c++) mc[c] = dthip_col{j.cols[nkeys + c].recv, j.cols[nkeys + c].stype, 0};
    DTHIP_TRY(dthip_groupby_rows(ctx, mk.data(), nkeys, mc.data(), ncols + 1, j.nrecv, a.na_pos, DTHIP_DEVICE, 0, &j.out));
  }
  clock.lap("phase_merge");
  return DTHIP_OK;
}

static int check_jobs(dthip_comm* comm, dthip_ctx* const* ctxs, int n) {
  if (!comm) { set_error("context is not part of a communicator: call dthip_comm_init / dthip_comm_init_local first"); return DTHIP_EINVAL; }
  if (comm->kind == 1 && n != comm->world) { set_error("local communicator of %d ranks called with %d contexts", comm->world, n); return DTHIP_EINVAL; }
  if (comm->kind == 0 && n != 1) { set_error("an RCCL communicator is driven one rank per call"); return DTHIP_EINVAL; }
  for (int i = 0; i < n; i++) if (!ctxs[i] || ctxs[i]->comm != comm) { set_error("contexts belong to different communicators"); return DTHIP_EINVAL; }
  return DTHIP_OK;
}

}  // namespace dthip

using namespace dthip;

extern "C" {

int dthip_comm_unique_id(void* id_out) {
  if (!id_out) { set_error("null id buffer"); return DTHIP_EINVAL; }
  DTHIP_TRY(nccl_load());
  static_assert(sizeof(ncclUniqueId) == DTHIP_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  DTHIP_CHECK_NCCL(g_nccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return DTHIP_OK;
}

int dthip_comm_init(dthip_ctx* ctx, int rank, int world, const void* id) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) { set_error("bad communicator arguments"); return DTHIP_EINVAL; }
  if (world > 127) { set_error("at most 127 ranks (destinations are int8)"); return DTHIP_EINVAL; }
  if (ctx->comm) { set_error("context already belongs to a communicator"); return DTHIP_EINVAL; }
  DTHIP_TRY(nccl_load());
  DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  dthip_comm* c = new dthip_comm();
  c->kind = 0; c->world = world; c->refs = 1;
  ncclResult_t r = g_nccl.CommInitRank(&c->nccl, world, uid, rank);
  if (r != ncclSuccess) { set_error("ncclCommInitRank failed: %s", g_nccl.GetErrorString(r)); delete c; return DTHIP_EDEVICE; }
  // staging of the all-gathers (largest blob: 4096 histogram bins of 8 bytes + header; 64 KB per rank leaves room)
  c->ag_cap = 64 * 1024;
  if (hipMalloc(reinterpret_cast<void**>(&c->ag_send), c->ag_cap) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->ag_recv), c->ag_cap * (size_t)world) != hipSuccess) {
    set_error("dthip_comm_init: out of device memory for the all-gather staging buffers");
    if (c->ag_send) (void)hipFree(c->ag_send);
    (void)g_nccl.CommDestroy(c->nccl);
    delete c;
    return DTHIP_ENOMEM;
  }
  ctx->comm = c; ctx->comm_rank = rank;
  return DTHIP_OK;
}

int dthip_comm_init_local(dthip_ctx* const* ctxs, int world) {
  if (!ctxs || world < 1 || world > 127) { set_error("bad communicator arguments"); return DTHIP_EINVAL; }
  for (int i = 0; i < world; i++) {
    if (!ctxs[i] || ctxs[i]->comm) { set_error("context %d is null or already belongs to a communicator", i); return DTHIP_EINVAL; }
    for (int k = 0; k < i; k++) if (ctxs[k] == ctxs[i]) { set_error("every rank needs its own context"); return DTHIP_EINVAL; }
  }
  dthip_comm* c = new dthip_comm();
  c->kind = 1; c->world = world; c->refs = world;
  c->ranks.assign(ctxs, ctxs + world);
  for (int i = 0; i < world; i++) { ctxs[i]->comm = c; ctxs[i]->comm_rank = i; }
  return DTHIP_OK;
}

int dthip_comm_destroy(dthip_ctx* ctx) {
  if (!ctx || !ctx->comm) return DTHIP_OK;
  dthip_comm* c = ctx->comm;
  ctx->comm = nullptr; ctx->comm_rank = 0;
  if (--c->refs > 0) return DTHIP_OK;
  if (c->kind == 0 && c->nccl) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)g_nccl.CommDestroy(c->nccl);
    if (c->ag_send) (void)hipFree(c->ag_send);
    if (c->ag_recv) (void)hipFree(c->ag_recv);
  }
  delete c;
  return DTHIP_OK;
}

int dthip_comm_last_stats(const dthip_ctx* ctx, int64_t* out, int n) {
  if (!ctx || !ctx->comm || !out || n < 1) { set_error("dthip_comm_last_stats: no communicator / bad arguments"); return DTHIP_EINVAL; }
  const dthip_comm* c = ctx->comm;
  const int64_t v[5] = {c->st_bytes_peer, c->st_bytes_self, c->st_rows_sent, c->st_rows_recv, c->st_allgathers};
  for (int i = 0; i < n; i++) out[i] = i < 5 ? v[i] : 0;
  return DTHIP_OK;
}

int dthip_comm_rank(const dthip_ctx* ctx) { return (ctx && ctx->comm) ? ctx->comm_rank : -1; }
int dthip_comm_world(const dthip_ctx* ctx) { return (ctx && ctx->comm) ? ctx->comm->world : 0; }

// query signature: ranks of an RCCL communicator are separate processes, nothing else tells them they were handed
// the same query
static uint32_t query_sig(const dthip_col* keys, int nkeys, const dthip_col* vals, int nvals, const dthip_agg* aggs, int naggs, int na_pos, int kind) {
  uint32_t h = 2166136261u;
  const int head[5] = {kind, nkeys, nvals, naggs, na_pos};
  h = fnv1a(h, head, sizeof(head));
  for (int k = 0; keys && k < nkeys && k < MAX_KEYCOLS; k++) { const int v[2] = {keys[k].stype, keys[k].flags}; h = fnv1a(h, v, sizeof(v)); }
  for (int c = 0; vals && c < nvals && c < 64; c++) { const int v[2] = {vals[c].stype, vals[c].flags}; h = fnv1a(h, v, sizeof(v)); }
  for (int a = 0; aggs && a < naggs && a < 256; a++) { const int v[2] = {aggs[a].op, aggs[a].col}; h = fnv1a(h, v, sizeof(v)); }
  return h;
}

static int sharded_agg_impl(dthip_ctx* const* ctxs, int n, const std::vector<AggArgs>& args, dthip_result** outs) {
  if (!ctxs || !outs || n < 1) { set_error("null argument"); return DTHIP_EINVAL; }
  dthip_comm* comm = ctxs[0] ? ctxs[0]->comm : nullptr;
  DTHIP_TRY(check_jobs(comm, ctxs, n));
  comm->st_bytes_peer = comm->st_bytes_self = comm->st_rows_sent = comm->st_rows_recv = comm->st_allgathers = 0;
  for (int i = 0; i < n; i++) outs[i] = nullptr;
  AggPlan plan;
  // argument errors of ONE rank of an RCCL communicator must not make it return alone (its peers would wait for it
  // in the first all-gather): they become the rank's status and every rank returns after that all-gather
  auto validate = [&](const AggArgs& a) -> int {
    if (!a.keys || a.nkeys < 1 || a.nkeys > MAX_KEYCOLS || (a.naggs > 0 && !a.aggs) || (a.nvalues > 0 && !a.values) || a.nvalues < 0 ||
        a.nrows < 0 || a.nrows > (int64_t)INT32_MAX) { set_error("bad argument"); return DTHIP_EINVAL; }
    if (a.na_pos != DTHIP_NA_FIRST && a.na_pos != DTHIP_NA_LAST) { set_error("na_pos %d not implemented", a.na_pos); return DTHIP_ENOTIMPL; }
    for (int t = 0; t < a.naggs; t++)
      if (a.aggs[t].op != DTHIP_COUNT0 && (a.aggs[t].col < 0 || a.aggs[t].col >= a.nvalues)) { set_error("agg %d refers to value column %d of %d", t, a.aggs[t].col, a.nvalues); return DTHIP_EINVAL; }
    if (a.keys[0].flags & DTHIP_FLAG_DESCENDING) { set_error("sharded groupby: the first key must be ascending (range partition)"); return DTHIP_ENOTIMPL; }
    if (a.nkeys != args[0].nkeys || a.naggs != args[0].naggs || a.nvalues != args[0].nvalues) { set_error("ranks disagree on the query"); return DTHIP_EINVAL; }
    return DTHIP_OK;
  };
  std::vector<int> pre(n, DTHIP_OK);
  for (int i = 0; i < n; i++) {
    pre[i] = validate(args[i]);
    if (pre[i] != DTHIP_OK && comm->kind == 1) return pre[i];          // one process holds every rank: nobody waits
  }
  int prc = pre[0] == DTHIP_OK ? plan_partials(args[0].aggs, args[0].naggs, &plan) : DTHIP_OK;
  if (prc != DTHIP_OK) { if (comm->kind == 1) return prc; pre[0] = prc; plan = AggPlan(); }
  std::vector<Job> jobs(n);
  std::vector<Scratch*> scs;
  for (int i = 0; i < n; i++) {
    jobs[i].ctx = ctxs[i]; jobs[i].rank = ctxs[i]->comm_rank; scs.push_back(new Scratch(ctxs[i])); jobs[i].sc = scs.back();
    jobs[i].rc = pre[i];
    jobs[i].sig = query_sig(args[i].keys, args[i].nkeys, args[i].values, args[i].nvalues, args[i].aggs, args[i].naggs, args[i].na_pos, 1);
  }
  std::sort(jobs.begin(), jobs.end(), [](const Job& a, const Job& b) { return a.rank < b.rank; });
  std::vector<AggArgs> sorted_args(n);
  for (int i = 0; i < n; i++) for (int q = 0; q < n; q++) if (jobs[q].ctx == ctxs[i]) sorted_args[q] = args[i];
  const int rc = run_sharded_agg(comm, jobs, sorted_args, plan);
  for (auto& j : jobs) {
    (void)hipSetDevice(j.ctx->device);
    if (j.local) dthip_result_free(j.ctx, j.local);
    if (rc != DTHIP_OK && j.out) { dthip_result_free(j.ctx, j.out); j.out = nullptr; }
  }
  for (int i = 0; i < n; i++) for (int q = 0; q < n; q++) if (jobs[q].ctx == ctxs[i]) outs[i] = jobs[q].out;
  for (auto* s : scs) { (void)hipSetDevice(s->ctx->device); delete s; }
  return rc;
}

int dthip_sharded_groupby_agg(dthip_ctx* ctx, const dthip_col* keys, int nkeys, const dthip_col* values, int nvalues,
                              const dthip_agg* aggs, int naggs, int64_t nrows_local, int na_pos, int mem, dthip_result** out) {
  std::vector<AggArgs> args(1);
  args[0] = AggArgs{keys, nkeys, values, nvalues, aggs, naggs, nrows_local, na_pos, mem};
  return sharded_agg_impl(&ctx, 1, args, out);
}

int dthip_sharded_groupby_agg_local(dthip_ctx* const* ctxs, int world, const dthip_col* const* keys, int nkeys,
                                    const dthip_col* const* values, int nvalues, const dthip_agg* aggs, int naggs,
                                    const int64_t* nrows, int na_pos, int mem, dthip_result** outs) {
  if (!keys || !nrows || (nvalues > 0 && !values)) { set_error("null argument"); return DTHIP_EINVAL; }
  std::vector<AggArgs> args(world);
  for (int r = 0; r < world; r++) args[r] = AggArgs{keys[r], nkeys, nvalues > 0 ? values[r] : nullptr, nvalues, aggs, naggs, nrows[r], na_pos, mem};
  return sharded_agg_impl(ctxs, world, args, outs);
}

static int sharded_rows_impl(dthip_ctx* const* ctxs, int n, const std::vector<RowsArgs>& args, dthip_result** outs) {
  if (!ctxs || !outs || n < 1) { set_error("null argument"); return DTHIP_EINVAL; }
  dthip_comm* comm = ctxs[0] ? ctxs[0]->comm : nullptr;
  DTHIP_TRY(check_jobs(comm, ctxs, n));
  comm->st_bytes_peer = comm->st_bytes_self = comm->st_rows_sent = comm->st_rows_recv = comm->st_allgathers = 0;
  for (int i = 0; i < n; i++) outs[i] = nullptr;
  auto validate = [&](const RowsArgs& a) -> int {
    if (!a.keys || a.nkeys < 1 || a.nkeys > MAX_KEYCOLS || a.ncols < 0 || (a.ncols > 0 && !a.cols) || a.nrows < 0 || a.nrows > (int64_t)INT32_MAX) {
      set_error("bad argument"); return DTHIP_EINVAL;
    }
    if (a.na_pos != DTHIP_NA_FIRST && a.na_pos != DTHIP_NA_LAST) { set_error("na_pos %d not implemented", a.na_pos); return DTHIP_ENOTIMPL; }
    if (a.keys[0].flags & DTHIP_FLAG_DESCENDING) { set_error("sharded groupby: the first key must be ascending (range partition)"); return DTHIP_ENOTIMPL; }
    if (a.nkeys != args[0].nkeys || a.ncols != args[0].ncols) { set_error("ranks disagree on the query"); return DTHIP_EINVAL; }
    return DTHIP_OK;
  };
  std::vector<int> pre(n, DTHIP_OK);
  for (int i = 0; i < n; i++) {
    pre[i] = validate(args[i]);
    if (pre[i] != DTHIP_OK && comm->kind == 1) return pre[i];
  }
  std::vector<Job> jobs(n);
  std::vector<Scratch*> scs;
  for (int i = 0; i < n; i++) {
    jobs[i].ctx = ctxs[i]; jobs[i].rank = ctxs[i]->comm_rank; scs.push_back(new Scratch(ctxs[i])); jobs[i].sc = scs.back();
    jobs[i].rc = pre[i];
    jobs[i].sig = query_sig(args[i].keys, args[i].nkeys, args[i].cols, args[i].ncols, nullptr, 0, args[i].na_pos, 2);
  }
  std::sort(jobs.begin(), jobs.end(), [](const Job& a, const Job& b) { return a.rank < b.rank; });
  std::vector<RowsArgs> sorted_args(n);
  for (int i = 0; i < n; i++) for (int q = 0; q < n; q++) if (jobs[q].ctx == ctxs[i]) sorted_args[q] = args[i];
  const int rc = run_sharded_rows(comm, jobs, sorted_args);
  for (auto& j : jobs) {
    (void)hipSetDevice(j.ctx->device);
    if (j.local) dthip_result_free(j.ctx, j.local);
    if (rc != DTHIP_OK && j.out) { dthip_result_free(j.ctx, j.out); j.out = nullptr; }
  }
  for (int i = 0; i < n; i++) for (int q = 0; q < n; q++) if (jobs[q].ctx == ctxs[i]) outs[i] = jobs[q].out;
  for (auto* s : scs) { (void)hipSetDevice(s->ctx->device); delete s; }
  return rc;
}

int dthip_sharded_groupby_rows(dthip_ctx* ctx, const dthip_col* keys, int nkeys, const dthip_col* cols, int ncols,
                               int64_t nrows_local, int64_t row_offset, int na_pos, int mem, dthip_result** out) {
  std::vector<RowsArgs> args(1);
  args[0] = RowsArgs{keys, nkeys, cols, ncols, nrows_local, row_offset, na_pos, mem};
  return sharded_rows_impl(&ctx, 1, args, out);
}

int dthip_sharded_groupby_rows_local(dthip_ctx* const* ctxs, int world, const dthip_col* const* keys, int nkeys,
                                     const dthip_col* const* cols, int ncols, const int64_t* nrows, const int64_t* row_offsets,
                                     int na_pos, int mem, dthip_result** outs) {
  if (!keys || !nrows || !row_offsets || (ncols > 0 && !cols)) { set_error("null argument"); return DTHIP_EINVAL; }
  std::vector<RowsArgs> args(world);
  for (int r = 0; r < world; r++) args[r] = RowsArgs{keys[r], nkeys, ncols > 0 ? cols[r] : nullptr, ncols, nrows[r], row_offsets[r], na_pos, mem};
  return sharded_rows_impl(ctxs, world, args, outs);
}

}  // extern "C"
