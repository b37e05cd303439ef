// api.hip -- the C ABI of libdthip.so (include/dthip.h): context management and
// the host-side orchestration of the groupby / reduce / RowIndex kernels.
//
// Orchestration mirrors what the reference does around its kernels:
//   dthip_groupby      ~ group()                    src/core/sort.cc:1411-1495
//   dthip_groupby_agg  ~ EvalContext::evaluate()    src/core/expr/eval_context.cc:144-172,
//                        compute_groupby_and_sort() :249-288, evaluate_select() :497-508
//   dthip_reduce       ~ FExpr_ReduceUnary::evaluate_n  src/core/expr/fexpr_reduce_unary.cc:32-69
#include <cstdarg>
#include <csignal>
#include <unistd.h>
#include <algorithm>
#include "host.hpp"

namespace dthip {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- guard-page mode ----------------------------------------------------------
// The name of the kernel in flight, for the SIGABRT handler: the ROCm runtime reports a GPU memory access fault
// ("Memory access fault by GPU node-N ...") and aborts; with one synchronised launch at a time the kernel named
// here is the one that touched the unmapped page.
static char g_guard_kernel[128] = "(no kernel in flight)";
// the most recent guarded buffers {first mapped byte, mapped bytes, user pointer, requested bytes, 1 = released}: printed by
// the handler so that the address the runtime reports can be placed next to (before / after / inside) a buffer
struct GuardLog { unsigned long long map, map_bytes, ptr, bytes, freed; };
static GuardLog g_guard_log[96];
static unsigned g_guard_logn = 0;
static void guard_sigabrt(int) {
  static const char pre[] = "\n[dthip guard] abort while kernel '";
  static const char post[] = "' was in flight (a GPU memory access fault above means it touched memory outside its buffers)\n";
  (void)!write(2, pre, sizeof(pre) - 1);
  (void)!write(2, g_guard_kernel, strnlen(g_guard_kernel, sizeof(g_guard_kernel)));
  (void)!write(2, post, sizeof(post) - 1);
  char line[160];
  const unsigned nlog = g_guard_logn < 96 ? g_guard_logn : 96;
  for (unsigned i = 0; i < nlog; i++) {
    const GuardLog& g = g_guard_log[(g_guard_logn - 1 - i) % 96];
    const int m = snprintf(line, sizeof(line), "[dthip guard]   buffer %u back: mapped [0x%llx, 0x%llx) user 0x%llx + %llu bytes%s\n", i,
                           g.map, g.map + g.map_bytes, g.ptr, g.bytes, g.freed ? " (released)" : "");
    if (m > 0) (void)!write(2, line, (size_t)m);
  }
  signal(SIGABRT, SIG_DFL);
  raise(SIGABRT);
}
static void guard_install_handler() {
  static bool done = false;
  if (!done) { signal(SIGABRT, guard_sigabrt); done = true; }
}
void guard_before_launch(dthip_ctx* ctx, const char* kname) {
  snprintf(g_guard_kernel, sizeof(g_guard_kernel), "%s", kname);
  ctx->guard_launches++;
}
int guard_after_launch(dthip_ctx* ctx, const char* kname) {
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { set_error("[guard] kernel %s failed: %s", kname, hipGetErrorString(e)); return DTHIP_EDEVICE; }
  snprintf(g_guard_kernel, sizeof(g_guard_kernel), "(none; last completed: %.90s)", kname);
  return DTHIP_OK;
}

static int guard_alloc(dthip_ctx* ctx, size_t bytes, void** out) {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = ctx->device;
  size_t gran = 0;
  DTHIP_CHECK_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  if (gran == 0) gran = 1 << 21;
  const size_t want = (std::max<size_t>(bytes, 1) + 15) & ~(size_t)15;      // 16-byte loads are the widest access
  dthip_ctx::GuardBlock b{};
  b.map_bytes = (want + gran - 1) / gran * gran;
  b.va_bytes = b.map_bytes + 2 * gran;                                       // one unmapped granule on either side
  DTHIP_CHECK_HIP(hipMemAddressReserve(&b.va, b.va_bytes, gran, nullptr, 0));
  hipError_t e = hipMemCreate(&b.h, b.map_bytes, &prop, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError(); (void)hipMemAddressFree(b.va, b.va_bytes);
    set_error("[guard] hipMemCreate(%zu bytes) failed: %s", b.map_bytes, hipGetErrorString(e));
    return DTHIP_ENOMEM;
  }
  b.map = static_cast<char*>(b.va) + gran;
  DTHIP_CHECK_HIP(hipMemMap(b.map, b.map_bytes, 0, b.h, 0));
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  DTHIP_CHECK_HIP(hipMemSetAccess(b.map, b.map_bytes, &acc, 1));
  // poison: fresh memory is not zero (nor is hipMalloc's), and slack inside the mapping is recognisable
  static const int poison = getenv("DTHIP_GUARD_POISON") ? atoi(getenv("DTHIP_GUARD_POISON")) : 0xA5;
  if (poison >= 0) {
    // synchronous: the next guarded allocation edits the page tables (hipMemMap / hipMemSetAccess), and doing that while
    // the fill of a multi-GB mapping was still running was seen to fault INSIDE the freshly mapped range
    DTHIP_CHECK_HIP(hipMemsetAsync(b.map, poison, b.map_bytes, ctx->stream));
    DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  void* p = ctx->guard == 2 ? b.map : static_cast<char*>(b.map) + (b.map_bytes - want);
  ctx->guarded[p] = b;
  ctx->guard_allocs++;
  g_guard_log[g_guard_logn % 96] = GuardLog{(unsigned long long)(uintptr_t)b.map, (unsigned long long)b.map_bytes,
                                            (unsigned long long)(uintptr_t)p, (unsigned long long)bytes, 0ULL};
  g_guard_logn++;
  *out = p;
  return DTHIP_OK;
}

// DTHIP_GUARD_FREE: 1 (default) a released buffer is unmapped at once (use after release faults too) but its address
// range stays reserved, so no later buffer ever appears at an address a stale pointer or a stale TLB entry still
// knows; 2 = the range is given back as well; 0 = released buffers stay mapped until dthip_trim / dthip_destroy
static int guard_free_mode() {
  static const int m = getenv("DTHIP_GUARD_FREE") ? atoi(getenv("DTHIP_GUARD_FREE")) : 1;
  return m;
}
static void guard_unmap(const dthip_ctx::GuardBlock& b, bool free_va) {
  (void)hipMemUnmap(b.map, b.map_bytes);
  (void)hipMemRelease(b.h);
  if (free_va) (void)hipMemAddressFree(b.va, b.va_bytes);
}
static void guard_free(dthip_ctx* ctx, void* p, bool final = false) {
  auto it = ctx->guarded.find(p);
  if (it == ctx->guarded.end()) return;
  const dthip_ctx::GuardBlock b = it->second;
  ctx->guarded.erase(it);
  for (unsigned i = 0; i < 96 && i < g_guard_logn; i++) if (g_guard_log[i].ptr == (unsigned long long)(uintptr_t)p) g_guard_log[i].freed = 1;
  (void)hipStreamSynchronize(ctx->stream);          // kernels queued on the block must be done before it disappears
  if (!final && guard_free_mode() == 0) { ctx->guard_limbo.push_back(b); return; }
  guard_unmap(b, final || guard_free_mode() == 2);
  if (!(final || guard_free_mode() == 2)) ctx->guard_vas.push_back(b);
}
static void guard_trim(dthip_ctx* ctx, bool final) {
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& b : ctx->guard_limbo) { guard_unmap(b, final); if (!final) ctx->guard_vas.push_back(b); }
  ctx->guard_limbo.clear();
  if (final) { for (auto& b : ctx->guard_vas) (void)hipMemAddressFree(b.va, b.va_bytes); ctx->guard_vas.clear(); }
}

// ---- caching device allocator ------------------------------------------------
int dev_alloc(dthip_ctx* ctx, size_t bytes, void** out) {
  if (ctx->guard == 1 || ctx->guard == 2) return guard_alloc(ctx, bytes, out);
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes == 0) bytes = 256;
  auto it = ctx->cache.lower_bound(bytes);
  if (it != ctx->cache.end() && it->first <= bytes + bytes / 4 + 4096) {
    *out = it->second;
    ctx->live[it->second] = it->first;
    ctx->cached_bytes -= it->first;
    ctx->cache.erase(it);
    return DTHIP_OK;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    dev_trim(ctx);
    e = hipMalloc(&p, bytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return DTHIP_ENOMEM;
  }
  ctx->live[p] = bytes;
  *out = p;
  return DTHIP_OK;
}

void dev_release(dthip_ctx* ctx, void* p) {
  if (!p) return;
  if (!ctx->guarded.empty() && ctx->guarded.count(p)) { guard_free(ctx, p); return; }
  auto it = ctx->live.find(p);
  if (it == ctx->live.end()) return;
  ctx->cache.emplace(it->second, p);
  ctx->cached_bytes += it->second;
  ctx->live.erase(it);
}

int dev_trim(dthip_ctx* ctx) {
  if (!ctx->guard_limbo.empty()) guard_trim(ctx, false);
  if (ctx->cache.empty()) return DTHIP_OK;
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->cache) (void)hipFree(kv.second);
  ctx->cache.clear();
  ctx->cached_bytes = 0;
  return DTHIP_OK;
}

int read_back(dthip_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
  if (bytes > ctx->pinned_bytes) {
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    size_t nb = std::max<size_t>(bytes, 1 << 16);
    DTHIP_CHECK_HIP(hipHostMalloc(&ctx->pinned, nb, hipHostMallocDefault));
    ctx->pinned_bytes = nb;
  }
  DTHIP_CHECK_HIP(hipMemcpyAsync(ctx->pinned, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(host_dst, ctx->pinned, bytes);
  return DTHIP_OK;
}

// 16 words of pinned host memory mapped into the device's address space (lazily; null when the runtime refuses)
bool host_words(dthip_ctx* ctx) {
  if (ctx->host_words) return true;
  void* h = nullptr; void* d = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return false; }
  ctx->host_words = static_cast<uint32_t*>(h);
  ctx->host_words_dev = static_cast<uint32_t*>(d);
  return true;
}

hipEvent_t prof_event(dthip_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

int prof_flush(dthip_ctx* ctx) {
  if (ctx->pending.empty()) return DTHIP_OK;
  DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  for (auto& r : ctx->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      auto& a = ctx->acc[r.name];
      a.ms += ms;
      a.n += 1;
    }
    ctx->event_pool.push_back(r.a);
    ctx->event_pool.push_back(r.b);
  }
  ctx->pending.clear();
  return DTHIP_OK;
}

// ---- staging of host columns ---------------------------------------------------
int stage_in(dthip_ctx* ctx, Scratch& sc, const void* src, size_t bytes, int mem, const void** dev) {
  if (mem == DTHIP_DEVICE || src == nullptr) { *dev = src; return DTHIP_OK; }
  unsigned char* d = nullptr;
  DTHIP_TRY(sc.get<unsigned char>(bytes, &d));
  if (bytes) DTHIP_CHECK_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  *dev = d;
  return DTHIP_OK;
}

int copy_out(dthip_ctx* ctx, void* dst, const void* dev_src, size_t bytes, int mem) {
  if (bytes == 0) return DTHIP_OK;
  if (!dst || !dev_src) { set_error("copy_out: null pointer"); return DTHIP_EINVAL; }
  if (mem == DTHIP_DEVICE) {
    DTHIP_CHECK_HIP(hipMemcpyAsync(dst, dev_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    DTHIP_CHECK_HIP(hipMemcpyAsync(dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  return DTHIP_OK;
}

int result_alloc(dthip_ctx* ctx, dthip_result* r, size_t bytes, void** out) {
  DTHIP_TRY(dev_alloc(ctx, bytes, out));
  r->owned.push_back(*out);
  return DTHIP_OK;
}

void result_adopt(Scratch& sc, dthip_result* r, void* p) {
  sc.disown(p);
  r->owned.push_back(p);
}

void result_destroy(dthip_ctx* ctx, dthip_result* r) {
  for (void* p : r->owned) dev_release(ctx, p);
  delete r;
}

int check_common(dthip_ctx* ctx, int64_t nrows, int mem) {
  if (!ctx) { set_error("null context"); return DTHIP_EINVAL; }
  if (nrows < 0 || nrows > (int64_t)INT32_MAX) {
    set_error("nrows=%lld is outside [0, 2^31-1]: RowIndex and group offsets are int32", (long long)nrows);
    return DTHIP_EINVAL;
  }
  if (mem != DTHIP_HOST && mem != DTHIP_DEVICE) { set_error("bad mem space %d", mem); return DTHIP_EINVAL; }
  DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
  return DTHIP_OK;
}

int stage_cols(dthip_ctx* ctx, Scratch& sc, const dthip_col* cols, int ncols, int64_t nrows, int mem,
                      std::vector<dthip_col>* out) {
  out->resize(ncols);
  for (int i = 0; i < ncols; i++) {
    const int sz = stype_size(cols[i].stype);
    if (sz == 0) { set_error("unsupported stype %d", cols[i].stype); return DTHIP_ENOTIMPL; }
    if (nrows > 0 && cols[i].data == nullptr) { set_error("null column data"); return DTHIP_EINVAL; }
    (*out)[i] = cols[i];
    DTHIP_TRY(stage_in(ctx, sc, cols[i].data, (size_t)nrows * sz, mem, &(*out)[i].data));
  }
  return DTHIP_OK;
}

int empty_result(dthip_ctx* ctx, dthip_result* res) {
  void* off = nullptr;
  DTHIP_TRY(result_alloc(ctx, res, sizeof(int32_t), &off));
  DTHIP_CHECK_HIP(hipMemsetAsync(off, 0, sizeof(int32_t), ctx->stream));
  res->offsets = static_cast<int32_t*>(off);
  res->nrows = 0; res->ngroups = 0;
  return DTHIP_OK;
}

int reduce_outs_for(int op, void* dst, ReduceOuts* o) {
  switch (op) {
    case DTHIP_SUM: o->sum = dst; break;
    case DTHIP_MEAN: o->mean = dst; break;
    case DTHIP_MIN: o->mn = dst; break;
    case DTHIP_MAX: o->mx = dst; break;
    case DTHIP_COUNT: o->count = static_cast<int64_t*>(dst); break;
    default: set_error("bad reducer op %d", op); return DTHIP_EINVAL;
  }
  return DTHIP_OK;
}

}  // namespace dthip

using namespace dthip;

static_assert(MAX_KEYCOLS == 8, "dthip_result::key is sized for MAX_KEYCOLS");

extern "C" {

int dthip_abi_version(void) { return DTHIP_ABI_VERSION; }
const char* dthip_build_id(void) {
  static const char id[] =
#if __has_include("build_id.inc")        // written by csrc/Makefile; a compile outside it (scripts/kernel_resources.sh) has none
#include "build_id.inc"
#else
      "unknown00000"
#endif
      ;
  return id;
}
const char* dthip_last_error(void) { return g_err; }

int dthip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int dthip_init(int device, void* stream, dthip_ctx** out) {
  if (!out) { set_error("null out"); return DTHIP_EINVAL; }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    (void)hipGetLastError();
    set_error("no HIP device available (%s): libdthip has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    return DTHIP_EDEVICE;
  }
  if (device < 0 || device >= n) { set_error("device %d out of range (0..%d)", device, n - 1); return DTHIP_EINVAL; }
  DTHIP_CHECK_HIP(hipSetDevice(device));
  dthip_ctx* ctx = new dthip_ctx();
  ctx->device = device;
  if (stream) { ctx->stream = static_cast<hipStream_t>(stream); ctx->own_stream = false; }
  else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx; set_error("hipStreamCreate failed"); return DTHIP_EDEVICE;
    }
    ctx->own_stream = true;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
  if (const char* e = getenv("DTHIP_AGG_PATH")) ctx->agg_path = atoi(e) >= 0 && atoi(e) <= 2 ? atoi(e) : 0;
  if (const char* e = getenv("DTHIP_BUCKET_VARIANT")) ctx->bucket_variant = atoi(e);
  if (const char* e = getenv("DTHIP_SORT_PATH")) ctx->sort_path = atoi(e) >= 0 && atoi(e) <= 2 ? atoi(e) : 0;
  if (const char* e = getenv("DTHIP_MSD_MIN_ROWS")) ctx->msd_min_rows = atoll(e);
  if (const char* e = getenv("DTHIP_FILTER_PATH")) ctx->filter_path = atoi(e) == 0 ? 0 : 1;
  if (const char* e = getenv("DTHIP_SMALL_PATH")) ctx->small_path = std::min(2, std::max(0, atoi(e)));
  if (const char* e = getenv("DTHIP_NONA_GUESS")) ctx->nona_guess = atoi(e) == 0 ? 0 : 1;
  if (const char* e = getenv("DTHIP_MSD_BUCKET_ROWS")) { const int v = atoi(e); if (v >= 1 && v <= 4096) ctx->msd_bucket_rows = v; }
  if (const char* e = getenv("DTHIP_GUARD")) { const int g = atoi(e); ctx->guard = (g >= 1 && g <= 3) ? g : 0; if (ctx->guard) guard_install_handler(); }
  (void)hipEventCreate(&ctx->t0);
  (void)hipEventCreate(&ctx->t1);
  *out = ctx;
  return DTHIP_OK;
}

int dthip_destroy(dthip_ctx* ctx) {
  if (!ctx) return DTHIP_OK;
  (void)hipSetDevice(ctx->device);
  (void)dthip_comm_destroy(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  prof_flush(ctx);
  dev_trim(ctx);
  for (auto& kv : ctx->live) (void)hipFree(kv.first);
  while (!ctx->guarded.empty()) guard_free(ctx, ctx->guarded.begin()->first, true);
  guard_trim(ctx, true);
  if (ctx->guard)
    fprintf(stderr, "[dthip guard] context closed: %lld guarded buffers, %lld synchronised launches, no fault\n",
            (long long)ctx->guard_allocs, (long long)ctx->guard_launches);
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->host_words) (void)hipHostFree(ctx->host_words);   // the mapped words of the small-table path: freed here and only here
  ctx->host_words = nullptr;
  ctx->host_words_dev = nullptr;
  if (ctx->t0) (void)hipEventDestroy(ctx->t0);
  if (ctx->t1) (void)hipEventDestroy(ctx->t1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return DTHIP_OK;
}

int dthip_use_stream(dthip_ctx* ctx, void* stream) {
  if (!ctx) { set_error("null context"); return DTHIP_EINVAL; }
  DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
  DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  prof_flush(ctx);
  dev_trim(ctx);                    // cached blocks are only safe to recycle in the stream they were used on
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = static_cast<hipStream_t>(stream);     // NULL = the device's default (legacy) stream
  ctx->own_stream = false;
  return DTHIP_OK;
}

int dthip_sync(dthip_ctx* ctx) {
  if (!ctx) { set_error("null context"); return DTHIP_EINVAL; }
  DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return DTHIP_OK;
}

int dthip_trim(dthip_ctx* ctx) { return ctx ? dev_trim(ctx) : DTHIP_EINVAL; }

int dthip_set_option(dthip_ctx* ctx, const char* name, int64_t value) {
  if (!ctx || !name) { set_error("null argument"); return DTHIP_EINVAL; }
  if (!strcmp(name, "agg_path")) {
    if (value < 0 || value > 2) { set_error("agg_path must be 0 (auto), 1 (sort) or 2 (bucket)"); return DTHIP_EINVAL; }
    ctx->agg_path = (int)value;
    return DTHIP_OK;
  }
  if (!strcmp(name, "bucket_variant")) { ctx->bucket_variant = (int)value; return DTHIP_OK; }
  if (!strcmp(name, "guard")) {
    if (value < 0 || value > 3) { set_error("guard must be 0 (off), 1 (buffer ends on an unmapped page), 2 (buffer starts on one) or 3 (launch tracing only)"); return DTHIP_EINVAL; }
    dev_trim(ctx);                      // cached blocks of the other flavour are not handed out again
    ctx->guard = (int)value;
    if (value) guard_install_handler();
    return DTHIP_OK;
  }
  if (!strcmp(name, "spec_min_rows")) { ctx->spec_min_rows = value; return DTHIP_OK; }
  if (!strcmp(name, "tl_level2")) {
    if (value < 0 || value > 2) { set_error("tl_level2 must be 0, 1 or 2"); return DTHIP_EINVAL; }
    ctx->tl_level2 = (int)value;
    return DTHIP_OK;
  }
  if (!strcmp(name, "filter_rows_fused")) {
    if (value < 0 || value > 1) { set_error("filter_rows_fused must be 0 or 1"); return DTHIP_EINVAL; }
    ctx->filter_rows_fused = (int)value;
    return DTHIP_OK;
  }
  if (!strcmp(name, "sort_path")) {
    if (value < 0 || value > 2) { set_error("sort_path must be 0 (auto), 1 (LSD passes only) or 2 (MSD levels whenever they apply)"); return DTHIP_EINVAL; }
    ctx->sort_path = (int)value;
    return DTHIP_OK;
  }
  if (!strcmp(name, "msd_min_rows")) { ctx->msd_min_rows = value; return DTHIP_OK; }
  if (!strcmp(name, "filter_path")) {
    if (value < 0 || value > 1) { set_error("filter_path must be 0 (one pass) or 1 (count pass + write pass)"); return DTHIP_EINVAL; }
    ctx->filter_path = (int)value;
    return DTHIP_OK;
  }
  if (!strcmp(name, "small_path")) {
    if (value < 0 || value > 2) { set_error("small_path must be 0, 1 or 2"); return DTHIP_EINVAL; }
    ctx->small_path = (int)value;
    return DTHIP_OK;
  }
  if (!strcmp(name, "nona_guess")) {
    if (value < 0 || value > 1) { set_error("nona_guess must be 0 or 1"); return DTHIP_EINVAL; }
    ctx->nona_guess = (int)value;
    return DTHIP_OK;
  }
  if (!strcmp(name, "msd_bucket_rows")) {
    if (value < 1 || value > 4096) { set_error("msd_bucket_rows must be in [1, 4096]"); return DTHIP_EINVAL; }
    ctx->msd_bucket_rows = (int)value;
    return DTHIP_OK;
  }
  if (!strcmp(name, "agg_offsets")) { ctx->agg_offsets = value != 0; return DTHIP_OK; }
  if (!strcmp(name, "f32_sum")) { ctx->f32_sum_ref = value != 0; return DTHIP_OK; }
  if (!strcmp(name, "join_table")) { ctx->join_table = value != 0; return DTHIP_OK; }
  if (!strcmp(name, "median_pairs")) { ctx->pairs_always = value != 0; return DTHIP_OK; }
  if (!strcmp(name, "hash_mode")) {
    if (value < 0 || value > 3) { set_error("hash_mode must be 0 (estimate), 1 (never), 2 (whenever it fits) or 3 (the same, exact-position partition)"); return DTHIP_EINVAL; }
    ctx->hash_mode = (int)value;
    return DTHIP_OK;
  }
  if (!strcmp(name, "cluster_mode")) {
    if (value < 0 || value > 2) { set_error("cluster_mode must be 0 (sample), 1 (never) or 2 (always)"); return DTHIP_EINVAL; }
    ctx->cluster_mode = (int)value;
    return DTHIP_OK;
  }
  set_error("unknown option '%s'", name);
  return DTHIP_EINVAL;
}

int dthip_malloc(dthip_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) { set_error("null argument"); return DTHIP_EINVAL; }
  DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
  return dev_alloc(ctx, bytes, dptr);
}
int dthip_free(dthip_ctx* ctx, void* dptr) {
  if (!ctx) return DTHIP_EINVAL;
  dev_release(ctx, dptr);
  return DTHIP_OK;
}
int dthip_memcpy_h2d(dthip_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return DTHIP_EINVAL;
  if (bytes == 0) return DTHIP_OK;
  DTHIP_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return DTHIP_OK;
}
int dthip_memcpy_d2h(dthip_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return DTHIP_EINVAL;
  if (bytes == 0) return DTHIP_OK;
  DTHIP_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return DTHIP_OK;
}

int dthip_host_register(dthip_ctx* ctx, void* ptr, size_t bytes) {
  if (!ctx || !ptr || !bytes) { set_error("null argument"); return DTHIP_EINVAL; }
  DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
  DTHIP_CHECK_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return DTHIP_OK;
}

int dthip_host_unregister(dthip_ctx* ctx, void* ptr) {
  if (!ctx || !ptr) { set_error("null argument"); return DTHIP_EINVAL; }
  DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
  DTHIP_CHECK_HIP(hipHostUnregister(ptr));
  return DTHIP_OK;
}

int dthip_timer_start(dthip_ctx* ctx) {
  if (!ctx) return DTHIP_EINVAL;
  DTHIP_CHECK_HIP(hipEventRecord(ctx->t0, ctx->stream));
  return DTHIP_OK;
}
int dthip_timer_stop(dthip_ctx* ctx, float* ms) {
  if (!ctx || !ms) return DTHIP_EINVAL;
  DTHIP_CHECK_HIP(hipEventRecord(ctx->t1, ctx->stream));
  DTHIP_CHECK_HIP(hipEventSynchronize(ctx->t1));
  DTHIP_CHECK_HIP(hipEventElapsedTime(ms, ctx->t0, ctx->t1));
  return DTHIP_OK;
}

int dthip_last_call_stats(const dthip_ctx* ctx, int64_t* out, int n) {
  if (!ctx || !out || n < 0) { set_error("null argument"); return DTHIP_EINVAL; }
  for (int i = 0; i < n; i++) out[i] = i < 5 ? ctx->call_stats[i] : 0;
  return DTHIP_OK;
}

int dthip_profile_enable(dthip_ctx* ctx, int on) {
  if (!ctx) return DTHIP_EINVAL;
  if (!on) prof_flush(ctx);
  ctx->prof = on != 0;
  return DTHIP_OK;
}
int dthip_profile_reset(dthip_ctx* ctx) {
  if (!ctx) return DTHIP_EINVAL;
  prof_flush(ctx);
  ctx->acc.clear();
  return DTHIP_OK;
}
int dthip_profile_get(dthip_ctx* ctx, const char* name, double* total_ms, int64_t* launches) {
  if (!ctx || !name) return DTHIP_EINVAL;
  DTHIP_TRY(prof_flush(ctx));
  double ms = 0; int64_t n = 0;
  for (auto& kv : ctx->acc) if (kv.first.find(name) != std::string::npos) { ms += kv.second.ms; n += kv.second.n; }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  return DTHIP_OK;
}
int dthip_profile_names(dthip_ctx* ctx, char* buf, size_t buflen) {
  if (!ctx || !buf || buflen == 0) return DTHIP_EINVAL;
  DTHIP_TRY(prof_flush(ctx));
  std::string s;
  for (auto& kv : ctx->acc) { s += kv.first; s += "\n"; }
  snprintf(buf, buflen, "%s", s.c_str());
  return DTHIP_OK;
}

}  // extern "C"
