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
#include "common.hpp"
#include "msd_plan.hpp"

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
#ifdef DTHIP_REPRO_UAF
    // `make uaf` only: the round-4 defect put back -- the mapped words of the small-table path freed
    // with the read-back buffer, their pointers left behind -- to show that it is what made ranks die of "Memory access
    // fault" on some boxes (profiles/r05_fault_hunt.txt).  Never in the product library.
    if (ctx->host_words) (void)hipHostFree(ctx->host_words);
#endif
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
static bool host_words(dthip_ctx* ctx) {
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
static int stage_in(dthip_ctx* ctx, Scratch& sc, const void* src, size_t bytes, int mem, const void** dev) {
  if (mem == DTHIP_DEVICE || src == nullptr) { *dev = src; return DTHIP_OK; }
  unsigned char* d = nullptr;
  DTHIP_TRY(sc.get<unsigned char>(bytes, &d));
  if (bytes) DTHIP_CHECK_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  *dev = d;
  return DTHIP_OK;
}

static int copy_out(dthip_ctx* ctx, void* dst, const void* dev_src, size_t bytes, int mem) {
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

// ---- sort planning ---------------------------------------------------------------
static int nbits_u64(unsigned long long v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

struct KeyPlan {
  int nkeys = 0;
  KeyColDev col[MAX_KEYCOLS];
  int nsig[MAX_KEYCOLS];
  // stages of consecutive keys whose packed width is <= 64 bits; stage 0 holds the most significant keys
  int nstages = 0;
  int stage_first[MAX_KEYCOLS], stage_last[MAX_KEYCOLS], stage_bits[MAX_KEYCOLS];
  bool speculative = false;   // integer key ranges are widened guesses from a sample (bucketed aggregation only)
};

constexpr uint32_t SPEC_SAMPLES = 1u << 17;
constexpr int DTHIP_RETRY_EXACT = 1;              // internal: a guessed key range was wrong, redo with the exact one

static void stype_int_limits(int st, long long* lo, long long* hi) {
  switch (st) {
    case DTHIP_BOOL: case DTHIP_INT8: *lo = INT8_MIN + 1; *hi = INT8_MAX; break;
    case DTHIP_INT16: *lo = INT16_MIN + 1; *hi = INT16_MAX; break;
    case DTHIP_INT32: *lo = (long long)INT32_MIN + 1; *hi = INT32_MAX; break;
    default: *lo = INT64_MIN + 1; *hi = INT64_MAX; break;
  }
}

// min/max of integer keys -> transform parameters (sort.cc:728-776), packing layout.
// speculative: the range of big integer columns is GUESSED from a sample and widened; only the
// bucketed aggregation may use such a plan, because its histogram pass verifies every row.
// tight (sort path): a guess whose 1/64 margin adds a significant bit to a key could cost a whole radix pass, which is
// more than the exact range scan it saves -- such a plan is made again with the exact range at once.
static int plan_keys(dthip_ctx* ctx, Scratch& sc, const dthip_col* keys_dev, int nkeys, int64_t n, int na_pos,
                     KeyPlan* plan, bool speculative = false, bool tight = false) {
  if (nkeys < 1 || nkeys > MAX_KEYCOLS) { set_error("number of key columns must be 1..%d", MAX_KEYCOLS); return DTHIP_EINVAL; }
  plan->nkeys = nkeys;
  plan->speculative = false;
  if (n < ctx->spec_min_rows) speculative = false;   // below this the exact range scan is cheap enough
  MinMax* d_mm = nullptr;
  DTHIP_TRY(sc.get<MinMax>(nkeys, &d_mm));
  bool any_int = false;
  for (int k = 0; k < nkeys; k++) {
    const int st = keys_dev[k].stype;
    if (stype_size(st) == 0) { set_error("unsupported key stype %d", st); return DTHIP_ENOTIMPL; }
    if (st >= DTHIP_INT8 && st <= DTHIP_INT64) {
      if (speculative) DTHIP_TRY(launch_minmax_sample(ctx, keys_dev[k].data, st, n, SPEC_SAMPLES, d_mm + k));
      else DTHIP_TRY(launch_minmax(ctx, keys_dev[k].data, st, n, d_mm + k));
      any_int = true;
    }
  }
  MinMax mm[MAX_KEYCOLS];
  if (any_int) DTHIP_TRY(read_back(ctx, mm, d_mm, sizeof(MinMax) * nkeys));
  if (speculative) {
    // a sample that met nothing but NAs (a sparse, mostly-NA key column) says nothing about the valid keys in the rows
    // it skipped: such a column gets its exact range (a plan that is not speculative is never verified)
    for (int k = 0; k < nkeys; k++) {
      const int st = keys_dev[k].stype;
      if (st >= DTHIP_INT8 && st <= DTHIP_INT64 && mm[k].nvalid == 0)
        return plan_keys(ctx, sc, keys_dev, nkeys, n, na_pos, plan, false, false);
    }
  }
  for (int k = 0; k < nkeys; k++) {
    KeyColDev& c = plan->col[k];
    const int st = keys_dev[k].stype;
    c.data = keys_dev[k].data;
    c.stype = st;
    c.desc = (keys_dev[k].flags & DTHIP_FLAG_DESCENDING) ? 1 : 0;
    c.shift = 0;
    if (st == DTHIP_BOOL) {
      c.edge = 0; c.inc = 0; c.na_repl = (na_pos == DTHIP_NA_LAST) ? 3 : 0; c.xmax = ~0ULL;
      plan->nsig[k] = 2;
    } else if (st == DTHIP_FLOAT32) {
      c.edge = 0; c.inc = 0; c.na_repl = (na_pos == DTHIP_NA_LAST) ? 0xFFFFFFFFULL : 0; c.xmax = ~0ULL;
      plan->nsig[k] = 32;
    } else if (st == DTHIP_FLOAT64) {
      c.edge = 0; c.inc = 0; c.na_repl = (na_pos == DTHIP_NA_LAST) ? 0xFFFFFFFFFFFFFFFFULL : 0; c.xmax = ~0ULL;
      plan->nsig[k] = 64;
    } else {
      long long mn = mm[k].mn, mx = mm[k].mx;
      if (mm[k].nvalid == 0) { mn = 0; mx = 0; }
      if (speculative && mm[k].nvalid > 0) {
        // widen the sampled range by 1/64 of its width (+64) on both sides, inside the stype's range
        long long tlo, thi;
        stype_int_limits(st, &tlo, &thi);
        const unsigned long long width = (unsigned long long)mx - (unsigned long long)mn;
        const int nb_sample = nbits_u64(width + 1ULL);
        const unsigned long long margin = width / 64 + 64;
        mn = ((unsigned long long)mn - (unsigned long long)tlo > margin) ? (long long)((unsigned long long)mn - margin) : tlo;
        mx = ((unsigned long long)thi - (unsigned long long)mx > margin) ? (long long)((unsigned long long)mx + margin) : thi;
        plan->speculative = true;
        if (tight && nbits_u64((unsigned long long)mx - (unsigned long long)mn + 1ULL) != nb_sample)
          return plan_keys(ctx, sc, keys_dev, nkeys, n, na_pos, plan, false, false);
      }
      const unsigned long long range1 = (unsigned long long)mx - (unsigned long long)mn + 1ULL;
      c.edge = c.desc ? (unsigned long long)mx : (unsigned long long)mn;
      c.inc = (na_pos == DTHIP_NA_LAST) ? 0 : 1;
      c.na_repl = (na_pos == DTHIP_NA_LAST) ? range1 : 0;
      c.xmax = range1 - 1ULL;              // valid keys: [inc, inc + range1 - 1] (range1 == 0: all 2^64 values, wraps to ~0)
      const int nb = nbits_u64(range1);
      plan->nsig[k] = nb ? nb : 64;
    }
  }
  // stages, built from the least significant key backwards
  int stages_rev_first[MAX_KEYCOLS], stages_rev_last[MAX_KEYCOLS], stages_rev_bits[MAX_KEYCOLS];
  int ns = 0;
  int k = nkeys - 1;
  while (k >= 0) {
    int bits = 0, last = k;
    while (k >= 0 && bits + plan->nsig[k] <= 64) { bits += plan->nsig[k]; k--; }
    stages_rev_first[ns] = k + 1; stages_rev_last[ns] = last; stages_rev_bits[ns] = bits;
    ns++;
  }
  plan->nstages = ns;
  for (int s = 0; s < ns; s++) {
    plan->stage_first[s] = stages_rev_first[ns - 1 - s];
    plan->stage_last[s] = stages_rev_last[ns - 1 - s];
    plan->stage_bits[s] = stages_rev_bits[ns - 1 - s];
    int sh = 0;
    for (int j = plan->stage_last[s]; j >= plan->stage_first[s]; j--) { plan->col[j].shift = sh; sh += plan->nsig[j]; }
  }
  return DTHIP_OK;
}

struct PaySpec {
  int n = 0;
  const void* in[MAX_PAYCOLS];
  int width[MAX_PAYCOLS];
  bool iota = false;            // column 0 is the row number
  // single int32 / int64 key whose column is wanted in sorted order: the last pass writes its ORIGINAL values here
  void* ukey_out = nullptr;
  // zeroed bitmap of n bits: the final MSD level marks the first row of every run of equal keys (SortOut::heads_done)
  unsigned long long* head_bitmap = nullptr;
};

struct SortOut {
  bool heads_done = false;      // PaySpec::head_bitmap was filled
  bool ukey_done = false;       // PaySpec::ukey_out was filled (then `keys` is NOT: the last pass wrote the original values instead)
  void* keys = nullptr;         // sorted packed keys (scratch-owned)
  int key64 = 0;
  void* pay[MAX_PAYCOLS];       // sorted payload columns (scratch-owned, or the input itself if nothing moved)
  int npasses_run = 0;
};

// ---- MSD levels (round 4) --------------------------------------------------------------------------------------------
// The reference sorts most-significant digit first and finishes small buckets with a cheap local sort
// (sort.cc:1206-1353 _radix_recurse, sort_insert.cc:95-141).  Same shape here for big inputs: two STABLE scatter levels
// over the top S1 + S2 bits (the LSD pass kernel with its digit at the top; the second level works inside the buckets
// of the first: ragged tiles that never span two of them, run positions from a scan segmented by parent bucket), then
// every final bucket (<= one radix tile) is ordered by the remaining <= 9 bits in LDS and written back over its own row
// range -- sequential writes, no histogram pass, no run positions.  Stability comes from the passes themselves (every
// level is a stable partition), so no row id has to travel.  Against three LSD passes: the last pass loses its write
// amplification (a (tile, digit) run of 16 rows shares its first and last 64-byte sector with the neighbouring tiles'
// runs: 1.74x the algorithmic bytes reach HBM, 4.55 ms per pass of C5; written in place: 3.1 ms) and one histogram pass.
static MsdPlan msd_plan(const dthip_ctx* ctx, int64_t n, int bits, int key64, uint32_t tile) {
  // measured (C5, 5e8 rows, 27 bits, MI355X, one box): levels 4.3 + 4.5 + final 4.4 ms (windows of whole buckets) and two
  // histogram passes against 3 x 4.7 ms of LSD passes and three: ~1 ms per call, more when the final level also writes the
  // original key column.  Below msd_min_rows the LSD passes are quick and the final buckets would be tiny.
  if (ctx->sort_path == 1 || key64 || (ctx->sort_path != 2 && n < ctx->msd_min_rows)) return MsdPlan();
  static const int rbmax = getenv("DTHIP_MSD_RBMAX") ? atoi(getenv("DTHIP_MSD_RBMAX")) : 9;
  return msd_split(n, bits, tile, ctx->msd_bucket_rows, rbmax);       // (host logic: csrc/msd_plan.hpp, tests/test_msd_plan.py)
}

// ---- windows of the final MSD level (whole buckets, together at most one tile of rows), planned on the device ----------
// Round 5: packed greedily per parent bucket (radix.hip msd_window_greedy_kernel; DTHIP_MSD_GREEDY=0: round 4's equal-step
// windows, kept for A/B).  ok = the windowed final level can run; maxsize = the largest final bucket either way.
struct WindowPlan { bool ok = false; uint32_t nwin = 0; const uint32_t* bounds = nullptr; const uint32_t* wfirst = nullptr;
                    int bits2 = 1; int pairs = 0; uint32_t maxsize = 0, span = 0, step = 0; };
static int plan_windows(dthip_ctx* ctx, Scratch& sc, const uint32_t* fstart, uint32_t nb1, uint32_t bins2, int64_t n, const uint32_t* d_max,
                        uint32_t tile, int maxw, int rb, WindowPlan* wp) {
  static const bool greedy = !(getenv("DTHIP_MSD_GREEDY") && atoi(getenv("DTHIP_MSD_GREEDY")) == 0);
  const uint32_t nbk = nb1 * bins2;
  // the (bucket, digit) counts of a window and their prefix live in the tile's exchange buffer: 2 x buckets x bins words
  uint32_t maxspan = 16;
  while (maxspan > 1 && (size_t)2 * maxspan * ((size_t)1 << rb) * 4 > (size_t)tile * maxw) maxspan >>= 1;
  const uint32_t nwmax = greedy ? (uint32_t)(2 * (n / tile)) + nb1 + 8 : (uint32_t)(n / (tile / 2)) + 2;
  uint32_t* wplan = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)3 * (nwmax + 2) + 4, &wplan));
  uint32_t* wbounds = wplan; uint32_t* wfirst = wplan + 2 * (nwmax + 2); uint32_t* winfo = wfirst + nwmax + 2;
  DTHIP_CHECK_HIP(hipMemsetAsync(winfo, 0, 4 * sizeof(uint32_t), ctx->stream));
  if (greedy) DTHIP_TRY(launch_msd_windows_greedy(ctx, fstart, nb1, bins2, tile, maxspan, nwmax, wbounds, wfirst, winfo));
  else DTHIP_TRY(launch_msd_windows(ctx, fstart, nbk, (uint32_t)n, d_max, tile, nwmax, wbounds, wfirst, winfo));
  uint32_t wi[4] = {0, 0, 0, 0};                     // equal-step: {windows, rows per step, largest span, -}; greedy: {~0 = infeasible, -, span, windows}
  DTHIP_TRY(read_back(ctx, wi, winfo, sizeof(wi)));
  DTHIP_TRY(read_back(ctx, &wp->maxsize, d_max, sizeof(uint32_t)));
  wp->bounds = wbounds; wp->wfirst = wfirst; wp->pairs = greedy ? 1 : 0; wp->span = wi[2];
  wp->nwin = greedy ? wi[3] : wi[0];
  wp->step = greedy ? 0 : wi[1];
  wp->bits2 = 1;
  while ((1u << wp->bits2) < wi[2]) wp->bits2++;
  wp->ok = wp->nwin > 0 && wp->nwin <= nwmax && wi[2] >= 1 && wi[2] <= maxspan && !(greedy && wi[0] == 0xFFFFFFFFu) &&
           (size_t)2 * ((size_t)1 << wp->bits2) * ((size_t)1 << rb) * 4 <= (size_t)tile * maxw;
  static const int win_env = getenv("DTHIP_MSD_WINDOWS") ? atoi(getenv("DTHIP_MSD_WINDOWS")) : 1;   // 0: one workgroup per bucket (A/B)
  if (win_env == 0) wp->ok = false;
  return DTHIP_OK;
}

// Stable sort of rows by one stage of packed keys, moving the payload columns along.
// `order` (nullable): the key columns are read through this ordering (later stages).
static int sort_stage(dthip_ctx* ctx, Scratch& sc, const KeyPlan& plan, int stage, int64_t n,
                      const int32_t* order, const PaySpec& pay, SortOut* out) {
  const int bits = plan.stage_bits[stage];
  const int key64 = bits > 32;
  const size_t ksz = key64 ? 8 : 4;
  out->key64 = key64;
  // digits of up to 9 bits (512 bins) whenever that saves a pass: 27 significant bits are 3 passes of 9, 63 bits 7 x 9.
  // (Round 1 kept 32-bit keys at 8 bits -- a 9-bit pass measured 2x slower there: its per-wave histograms left LDS
  // for one workgroup per CU only; they are 16-bit words now.)
  int npass = (bits + 7) / 8;
  if ((bits + 8) / 9 < npass) npass = (bits + 8) / 9;
  if (npass > MAX_PASSES) npass = MAX_PASSES;
  XformArgs xa;
  memset(&xa, 0, sizeof(xa));
  xa.ncols = plan.stage_last[stage] - plan.stage_first[stage] + 1;
  for (int j = 0; j < xa.ncols; j++) xa.cols[j] = plan.col[plan.stage_first[stage] + j];
  xa.n = (uint32_t)n;
  xa.order = order;
  xa.out64 = key64;
  xa.npass = npass;
  const uint32_t tile = radix_tile_items(key64, 8);
  const MsdPlan msd = msd_plan(ctx, n, bits, key64, tile);
  if (msd.ok) {
    // digits, least significant first: what the final level orders in LDS, then the two scatter levels (the LSD passes
    // can run the same layout, so giving up on the MSD levels after the histograms costs nothing)
    npass = 3;
    xa.npass = 3;
    xa.pbits[0] = msd.rb > 9 ? 9 : msd.rb; xa.pbits[1] = msd.s2; xa.pbits[2] = msd.s1;      // (a histogram row has 512 bins)
    xa.pshift[0] = 0; xa.pshift[1] = msd.rb; xa.pshift[2] = msd.rb + msd.s2;
  } else {
    const int base = bits / npass, rem = bits % npass;
    int sh = 0;
    for (int p = 0; p < npass; p++) { xa.pbits[p] = base + (p < rem ? 1 : 0); xa.pshift[p] = sh; sh += xa.pbits[p]; }
  }
  unsigned char* kA = nullptr;
  DTHIP_TRY(sc.get<unsigned char>((size_t)n * ksz, &kA));
  uint32_t* hist = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)2 * MAX_PASSES * HIST_STRIDE, &hist));
  uint32_t* base = hist + MAX_PASSES * HIST_STRIDE;
  DTHIP_CHECK_HIP(hipMemsetAsync(hist, 0, sizeof(uint32_t) * MAX_PASSES * HIST_STRIDE, ctx->stream));
  xa.out = kA;
  xa.hist = hist;
  // a plan made from a GUESSED key range (plan_keys: sampled min / max) is verified by this very pass: the word after
  // the last histogram row comes back non-zero when some key fell outside, and the caller plans again (exact range)
  static_assert(MAX_PASSES * 8 >= 64 + 8, "a histogram row stays free for the range check");
  xa.bad = plan.speculative ? hist + (size_t)npass * HIST_STRIDE : nullptr;
  DTHIP_TRY(launch_xform_hist(ctx, xa));
  // which passes actually permute anything?
  std::vector<uint32_t> hh((size_t)npass * HIST_STRIDE + 1);
  DTHIP_TRY(read_back(ctx, hh.data(), hist, hh.size() * sizeof(uint32_t)));
  if (plan.speculative && hh[(size_t)npass * HIST_STRIDE]) return DTHIP_RETRY_EXACT;
  int active[MAX_PASSES], nactive = 0;
  for (int p = 0; p < npass; p++) {
    bool constant = false;
    for (int d = 0; d < (1 << xa.pbits[p]); d++) if (hh[(size_t)p * HIST_STRIDE + d] == (uint32_t)n) constant = true;
    if (!constant) active[nactive++] = p;
  }
  out->npasses_run = nactive;
  for (int c = 0; c < pay.n; c++) out->pay[c] = const_cast<void*>(pay.in[c]);
  if (nactive == 0) {
    out->keys = kA;
    if (pay.iota) {
      int32_t* ri = nullptr;
      DTHIP_TRY(sc.get<int32_t>((size_t)n, &ri));
      DTHIP_TRY(launch_iota(ctx, ri, n));
      out->pay[0] = ri;
    }
    return DTHIP_OK;
  }
  DTHIP_TRY(launch_hist_scan(ctx, hist, base, npass));
  const uint32_t ntiles = (uint32_t)((n + tile - 1) / tile);
  // per-pass run positions: per-tile digit counts of the current key order -> P, gpre
  int maxbits = 0;
  for (int i = 0; i < nactive; i++) maxbits = std::max(maxbits, xa.pbits[active[i]]);
  BucketGeom hg;
  memset(&hg, 0, sizeof(hg));
  {
    const uint32_t gmax = (uint32_t)ctx->num_cus * 4;
    hg.ntiles = ntiles;
    hg.tpg = (ntiles + gmax - 1) / gmax; if (hg.tpg == 0) hg.tpg = 1;
    hg.G = (ntiles + hg.tpg - 1) / hg.tpg;
  }
  bool use_msd = msd.ok && nactive == 3;
  if (use_msd) {
    // The levels give up when a final bucket outgrows a tile, AFTER level 1 and two histogram passes.  The digit histograms
    // already on the host say when that is certain or likely, for nothing: rows can only land in (level-1 digit, level-2
    // digit) cells whose two marginal bins are non-empty, so fewer such cells than n / tile means an overflow for sure
    // (few distinct keys over a wide range); and if the two digits were independent the fullest cell would hold
    // max1 * max2 / n rows (a hot key, clustered keys).  Either way the LSD passes run at once.
    if (msd_overflow_expected(&hh[(size_t)2 * HIST_STRIDE], 1 << xa.pbits[2], &hh[(size_t)1 * HIST_STRIDE], 1 << xa.pbits[1], n, tile)) use_msd = false;
    if (getenv("DTHIP_MSD_DEBUG"))
      fprintf(stderr, "[dthip msd] n=%lld bits=%d digits %d+%d+%d -> %s\n", (long long)n, bits, msd.s1, msd.s2, msd.rb,
              use_msd ? "levels" : "LSD passes (overflow certain or likely)");
  } else if (getenv("DTHIP_MSD_DEBUG")) {
    fprintf(stderr, "[dthip msd] n=%lld bits=%d key64=%d: plan %s, active digits %d\n", (long long)n, bits, key64, msd.ok ? "ok" : "not applicable", nactive);
  }
  uint32_t* P = nullptr; uint32_t* gtot = nullptr; uint32_t* tot = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)(ntiles + (use_msd ? (2u << msd.s1) : 0u)) << maxbits, &P));
  DTHIP_TRY(sc.get<uint32_t>((size_t)hg.G << maxbits, &gtot));
  DTHIP_TRY(sc.get<uint32_t>((size_t)1 << maxbits, &tot));
  unsigned char* kB = nullptr;
  DTHIP_TRY(sc.get<unsigned char>((size_t)n * ksz, &kB));
  void* pbuf[2][MAX_PAYCOLS];
  for (int c = 0; c < pay.n; c++) {
    unsigned char* b0 = nullptr;
    DTHIP_TRY(sc.get<unsigned char>((size_t)n * pay.width[c], &b0));
    pbuf[0][c] = b0;
    pbuf[1][c] = nullptr;
    if (nactive > 1) {
      unsigned char* b1 = nullptr;
      DTHIP_TRY(sc.get<unsigned char>((size_t)n * pay.width[c], &b1));
      pbuf[1][c] = b1;
    }
  }
  out->ukey_done = false; out->heads_done = false;
  if (use_msd) {
    // ---- level 1: stable scatter by the top s1 bits (regular tiles) ------------------------------------------------
    const int p1 = 2, p2 = 1;
    const uint32_t nb1 = 1u << msd.s1, bins2 = 1u << msd.s2;
    hg.F = nb1;
    DTHIP_TRY(launch_radix_tile_hist(ctx, kA, key64, (uint32_t)n, xa.pshift[p1], xa.pbits[p1], ntiles, hg.tpg, hg.G, P, gtot));
    DTHIP_TRY(launch_bucket_gscan(ctx, hg, gtot, tot, base + (size_t)p1 * HIST_STRIDE, 1));
    RadixPass rp;
    memset(&rp, 0, sizeof(rp));
    rp.kin = kA; rp.kout = kB; rp.key64 = key64; rp.n = (uint32_t)n;
    rp.shift = xa.pshift[p1]; rp.bits = xa.pbits[p1];
    rp.P = P; rp.gpre = gtot; rp.tpg = hg.tpg;
    rp.iota = pay.iota ? 1 : 0;
    rp.pay.n = pay.n;
    for (int c = 0; c < pay.n; c++) { rp.pay.in[c] = pay.in[c]; rp.pay.out[c] = pbuf[0][c]; rp.pay.width[c] = pay.width[c]; }
    rp.label = "msd_level1_kernel";
    DTHIP_TRY(launch_radix_pass(ctx, rp));
    // ---- level 2: the same inside every level-1 bucket: ragged tiles, planned on the host from the level-1 histogram
    std::vector<uint32_t> tdesc, gdesc, gfirst;
    msd_level2_tiles(&hh[(size_t)p1 * HIST_STRIDE], nb1, tile, hg.tpg, &tdesc, &gdesc, &gfirst);
    const uint32_t ntiles2 = (uint32_t)(tdesc.size() / 4), G2 = (uint32_t)(gdesc.size() / 2);
    uint32_t* d_plan = nullptr;
    DTHIP_TRY(sc.get<uint32_t>(tdesc.size() + gdesc.size() + gfirst.size() + 4, &d_plan));
    uint32_t* d_tdesc = d_plan; uint32_t* d_gdesc = d_tdesc + tdesc.size(); uint32_t* d_gfirst = d_gdesc + gdesc.size();
    uint32_t* d_max = d_gfirst + gfirst.size();
    DTHIP_CHECK_HIP(hipMemcpyAsync(d_tdesc, tdesc.data(), tdesc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    DTHIP_CHECK_HIP(hipMemcpyAsync(d_gdesc, gdesc.data(), gdesc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    DTHIP_CHECK_HIP(hipMemcpyAsync(d_gfirst, gfirst.data(), gfirst.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    DTHIP_CHECK_HIP(hipMemsetAsync(d_max, 0, 4, ctx->stream));
    uint32_t* gtot2 = nullptr; uint32_t* fstart = nullptr;
    DTHIP_TRY(sc.get<uint32_t>((size_t)(G2 ? G2 : 1) * bins2, &gtot2));
    DTHIP_TRY(sc.get<uint32_t>((size_t)nb1 * bins2 + 1, &fstart));
    DTHIP_TRY(launch_radix_tile_hist(ctx, kB, key64, (uint32_t)n, xa.pshift[p2], xa.pbits[p2], ntiles2, hg.tpg, G2, P, gtot2, d_tdesc, d_gdesc));
    DTHIP_TRY(launch_msd_scan(ctx, gtot2, d_gfirst, base + (size_t)p1 * HIST_STRIDE, msd.s2, nb1, (uint32_t)n, fstart, d_max));
    // windows of the final level (whole buckets, together at most one tile of rows), planned on the device
    int maxw_w = 4;
    for (int c = 0; c < pay.n; c++) maxw_w = std::max(maxw_w, pay.width[c]);
    WindowPlan wp;
    DTHIP_TRY(plan_windows(ctx, sc, fstart, nb1, bins2, n, d_max, tile, maxw_w, msd.rb, &wp));
    const bool windows = wp.ok;
    const uint32_t maxsize = wp.maxsize;
    if (getenv("DTHIP_MSD_DEBUG"))
      fprintf(stderr, "[dthip msd] n=%lld s1=%d s2=%d rb=%d tiles2=%u groups2=%u largest bucket=%u windows=%u step=%u max buckets/window=%u -> %s\n",
              (long long)n, msd.s1, msd.s2, msd.rb, ntiles2, G2, maxsize, wp.nwin, wp.step, wp.span, windows ? "windows" : (maxsize <= tile ? "per bucket" : "LSD"));
    if (windows || maxsize <= tile) {
      rp.kin = kB; rp.kout = kA;
      rp.shift = xa.pshift[p2]; rp.bits = xa.pbits[p2];
      rp.P = P; rp.gpre = gtot2; rp.tpg = hg.tpg; rp.iota = 0;
      rp.ntiles = ntiles2; rp.tdesc = d_tdesc; rp.bounds = nullptr;
      for (int c = 0; c < pay.n; c++) { rp.pay.in[c] = pbuf[0][c]; rp.pay.out[c] = pbuf[1][c]; }
      rp.label = "msd_level2_kernel";
      DTHIP_TRY(launch_radix_pass(ctx, rp));
      // ---- final level: every bucket ordered by the remaining bits in LDS, written over its own rows
      rp.kin = kA; rp.kout = kB;
      rp.shift = 0; rp.bits = msd.rb;
      rp.P = nullptr; rp.gpre = nullptr;
      rp.ntiles = nb1 * bins2; rp.tdesc = nullptr; rp.bounds = fstart;
      static const int fb_env = getenv("DTHIP_MSD_FINAL_BLOCK") ? atoi(getenv("DTHIP_MSD_FINAL_BLOCK")) : 0;
      rp.block = (fb_env != 512 && maxsize <= tile / 2) ? 256 : 0;      // DTHIP_MSD_FINAL_BLOCK=512: A/B against the big workgroup
      if (windows) {
        // a bucket of ~2000 rows per workgroup leaves a CU with too few rows in flight (5.9 ms for C5's 5e8 rows); windows
        // of several whole buckets fill the tile (3.8 ms at ~6000 rows) at the price of a second ranking round in LDS
        rp.ntiles = wp.nwin; rp.bounds = wp.bounds; rp.wfirst = wp.wfirst; rp.block = 0;
        rp.bits2 = wp.bits2; rp.wpairs = wp.pairs;
#ifdef DTHIP_RP_EXPERIMENT
        if (getenv("DTHIP_MSD_R1ONLY")) rp.bits2 = 99;        // timing experiment: wrong results
        if (getenv("DTHIP_MSD_WIN_NOR2")) rp.wfirst = nullptr; // timing experiment: the one-round kernel over the real windows
#endif
      }
      for (int c = 0; c < pay.n; c++) { rp.pay.in[c] = pbuf[1][c]; rp.pay.out[c] = pbuf[0][c]; }
      if (pay.ukey_out) {
        const KeyColDev& kc = plan.col[plan.stage_first[stage]];
        rp.ukout = pay.ukey_out; rp.uk_stype = kc.stype; rp.uk_desc = kc.desc; rp.uk_bits = bits;
        rp.uk_edge = kc.edge; rp.uk_na_repl = kc.na_repl; rp.uk_inc = kc.inc;
        out->ukey_done = true;
      }
      if (pay.head_bitmap) { rp.headbits = reinterpret_cast<uint32_t*>(pay.head_bitmap); out->heads_done = true; }
      rp.label = "msd_final_kernel";
#ifdef DTHIP_RP_EXPERIMENT
      if (const char* fw = getenv("DTHIP_MSD_FAKE_WINDOW")) {
        // TIMING EXPERIMENT ONLY (wrong results): the final level over fixed windows of W rows instead of buckets
        const uint32_t W = (uint32_t)atoi(fw);
        std::vector<uint32_t> wb;
        for (uint64_t r = 0; r < (uint64_t)n; r += W) wb.push_back((uint32_t)r);
        wb.push_back((uint32_t)n);
        uint32_t* d_wb = nullptr;
        DTHIP_TRY(sc.get<uint32_t>(wb.size(), &d_wb));
        DTHIP_CHECK_HIP(hipMemcpyAsync(d_wb, wb.data(), wb.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        rp.bounds = d_wb; rp.ntiles = (uint32_t)wb.size() - 1; rp.block = 0;
      }
#endif
      DTHIP_TRY(launch_radix_pass(ctx, rp));
      out->keys = kB;
      for (int c = 0; c < pay.n; c++) out->pay[c] = pbuf[0][c];
      return DTHIP_OK;
    }
    // a final bucket does not fit a tile (heavy duplicates / clustered keys): the LSD passes below start over from kA
    // and the caller's payload columns, which level 1 only read
    if (msd.rb > 9) { set_error("MSD levels with a 10-bit final digit (experiment) cannot fall back"); return DTHIP_ENOTIMPL; }
  }
  unsigned char* kin = kA; unsigned char* kout = kB;
  out->ukey_done = false;
  for (int i = 0; i < nactive; i++) {
    const int p = active[i];
    hg.F = 1u << xa.pbits[p];
    DTHIP_TRY(launch_radix_tile_hist(ctx, kin, key64, (uint32_t)n, xa.pshift[p], xa.pbits[p], ntiles, hg.tpg, hg.G, P, gtot));
    DTHIP_TRY(launch_bucket_gscan(ctx, hg, gtot, tot, base + (size_t)p * HIST_STRIDE, 1));
    RadixPass rp;
    memset(&rp, 0, sizeof(rp));
    rp.kin = kin; rp.kout = kout; rp.key64 = key64; rp.n = (uint32_t)n;
    rp.shift = xa.pshift[p]; rp.bits = xa.pbits[p];
    rp.P = P; rp.gpre = gtot; rp.tpg = hg.tpg;
    rp.iota = (i == 0 && pay.iota) ? 1 : 0;
    // (measured on C5: in an LSD pass the 8-byte key values are scattered runs like every other column -- the last pass
    // got 1.05 ms slower and the group scan reads 8 instead of 4 bytes (+0.43), which eats the 1.52 ms of the untransform
    // pass; the final MSD level writes in place and keeps 0.7 ms of it.  DTHIP_FUSE_UKEY=2 forces it here for A/B runs)
    static const bool fuse_lsd = getenv("DTHIP_FUSE_UKEY") && atoi(getenv("DTHIP_FUSE_UKEY")) == 2;
    if (i == nactive - 1 && pay.ukey_out && fuse_lsd) {
      const KeyColDev& kc = plan.col[plan.stage_first[stage]];
      rp.ukout = pay.ukey_out; rp.uk_stype = kc.stype; rp.uk_desc = kc.desc; rp.uk_bits = bits;
      rp.uk_edge = kc.edge; rp.uk_na_repl = kc.na_repl; rp.uk_inc = kc.inc;
      out->ukey_done = true;
    }
    rp.pay.n = pay.n;
    for (int c = 0; c < pay.n; c++) {
      rp.pay.in[c] = (i == 0) ? pay.in[c] : pbuf[(i - 1) & 1][c];
      rp.pay.out[c] = pbuf[i & 1][c];
      rp.pay.width[c] = pay.width[c];
    }
    DTHIP_TRY(launch_radix_pass(ctx, rp));
    std::swap(kin, kout);
  }
  out->keys = kin;
  for (int c = 0; c < pay.n; c++) out->pay[c] = pbuf[(nactive - 1) & 1][c];
  return DTHIP_OK;
}

}  // namespace dthip

using namespace dthip;

static_assert(MAX_KEYCOLS == 8, "dthip_result::key is sized for MAX_KEYCOLS");

namespace dthip {

int result_alloc(dthip_ctx* ctx, dthip_result* r, size_t bytes, void** out) {
  DTHIP_TRY(dev_alloc(ctx, bytes, out));
  r->owned.push_back(*out);
  return DTHIP_OK;
}

static void result_adopt(Scratch& sc, dthip_result* r, void* p) {
  sc.disown(p);
  r->owned.push_back(p);
}

void result_destroy(dthip_ctx* ctx, dthip_result* r) {
  for (void* p : r->owned) dev_release(ctx, p);
  delete r;
}

// internal grouping state shared by groupby / groupby_agg / generic path
struct Grouping {
  int64_t n = 0, ngroups = 0;
  int32_t* rowindex = nullptr;          // scratch-owned (nullable)
  int32_t* offsets = nullptr;           // result-owned
  unsigned long long* bitmap = nullptr; // scratch-owned
  uint32_t* tile_first = nullptr;       // scratch-owned: index of first head per 2048-tile
  void* sorted_keys = nullptr; int key64 = 0;
  void* pay[MAX_PAYCOLS];
};

// the final MSD level can mark the group heads (its keys sit in LDS in sorted order): built, bit-exact on the forced
// suites, and measured a LOSS -- the level 4.75 -> 6.04 ms for a count_heads pass of 0.73 ms saved -- so it stays off
// AND out of the product build: with the head phase compiled in, the ordinary scatter variant of radix_pass_kernel spilled
// 60 instead of 24 VGPRs and C5's two scatter levels went from 4.3 + 4.9 to 5.7 + 6.0 ms although the phase never ran.
// `make -C datatable_amd/csrc heads` builds the flavour (-DDTHIP_RP_HEADS); there DTHIP_FUSE_HEADS=1 switches it on.
static bool fuse_heads_enabled() {
#ifdef DTHIP_RP_HEADS
  static const bool on = getenv("DTHIP_FUSE_HEADS") && atoi(getenv("DTHIP_FUSE_HEADS")) == 1;
  return on;
#else
  return false;
#endif
}

static int alloc_head_bitmap(dthip_ctx* ctx, Scratch& sc, int64_t n, unsigned long long** bitmap) {
  *bitmap = nullptr;
  if (!fuse_heads_enabled()) return DTHIP_OK;
  const size_t words = (size_t)((n + 63) / 64) + 1;
  DTHIP_TRY(sc.get<unsigned long long>(words, bitmap));
  DTHIP_CHECK_HIP(hipMemsetAsync(*bitmap, 0, words * 8, ctx->stream));
  return DTHIP_OK;
}

// ready: a head bitmap the sort itself filled (final MSD level) -- no pass over the keys
static int heads_to_offsets(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const void* keys, int key64,
                            const uint8_t* heads, int64_t n, Grouping* g, unsigned long long* ready = nullptr) {
  const uint32_t nt = (uint32_t)((n + SEG_TILE - 1) / SEG_TILE);
  uint32_t* tile_counts = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)nt + 4 + nt / 8192, &tile_counts));
  unsigned long long* bitmap = ready;
  if (!bitmap) DTHIP_TRY(sc.get<unsigned long long>((size_t)((n + 63) / 64) + 1, &bitmap));
  int64_t ng = 0;
  if (ready) DTHIP_TRY(launch_heads_from_bitmap(ctx, bitmap, n, tile_counts, tile_counts + nt, &ng));
  else
  DTHIP_TRY(launch_count_heads(ctx, keys, key64, heads, n, tile_counts, bitmap, tile_counts + nt, &ng));
  void* off = nullptr;
  DTHIP_TRY(result_alloc(ctx, res, sizeof(int32_t) * (size_t)(ng + 1), &off));
  DTHIP_TRY(launch_write_offsets(ctx, bitmap, n, tile_counts, ng, static_cast<int32_t*>(off)));
  g->n = n; g->ngroups = ng; g->offsets = static_cast<int32_t*>(off);
  g->bitmap = bitmap; g->tile_first = tile_counts;
  return DTHIP_OK;
}

// full group(): ordering + offsets (+ head bitmap) for any number of keys
static int group_core(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const dthip_col* keys_dev, int nkeys,
                      int64_t n, int na_pos, KeyPlan* plan, Grouping* g) {
  // integer key ranges of big columns are guessed from a sample first (saves the exact min / max scan: 0.8 ms per 1e9-row
  // int64 column); the key-transform pass of every stage verifies the guess, a wrong one costs one more round
  const int32_t* order = nullptr;
  SortOut so;
  unsigned long long* gc_bitmap = nullptr;
  for (int attempt = 0; attempt < 2; attempt++) {
    DTHIP_TRY(plan_keys(ctx, sc, keys_dev, nkeys, n, na_pos, plan, attempt == 0, true));
    order = nullptr;
    int rc = DTHIP_OK;
    for (int s = plan->nstages - 1; s >= 0; s--) {
      PaySpec ps;
      ps.n = 1; ps.width[0] = 4;
      if (order) { ps.in[0] = order; ps.iota = false; } else { ps.in[0] = nullptr; ps.iota = true; }
      if (plan->nstages == 1) {
        if (!gc_bitmap) DTHIP_TRY(alloc_head_bitmap(ctx, sc, n, &gc_bitmap));      // (null unless DTHIP_FUSE_HEADS=1)
        ps.head_bitmap = gc_bitmap;
      }
      rc = sort_stage(ctx, sc, *plan, s, n, order, ps, &so);
      if (rc != DTHIP_OK) break;
      order = static_cast<const int32_t*>(so.pay[0]);
    }
    if (rc == DTHIP_RETRY_EXACT && attempt == 0) { ctx->call_stats[0]++; continue; }
    if (rc == DTHIP_RETRY_EXACT) { set_error("group: exact key range violated"); return DTHIP_EDEVICE; }
    DTHIP_TRY(rc);
    break;
  }
  g->rowindex = const_cast<int32_t*>(order);
  g->sorted_keys = so.keys; g->key64 = so.key64;
  if (plan->nstages == 1) {
    DTHIP_TRY(heads_to_offsets(ctx, sc, res, so.keys, so.key64, nullptr, n, g, so.heads_done ? gc_bitmap : nullptr));
  } else {
    uint8_t* heads = nullptr;
    DTHIP_TRY(sc.get<uint8_t>((size_t)n, &heads));
    DTHIP_CHECK_HIP(hipMemsetAsync(heads, 0, (size_t)n, ctx->stream));
    for (int s = 0; s < plan->nstages; s++) {
      const int bits = plan->stage_bits[s];
      XformArgs xa;
      memset(&xa, 0, sizeof(xa));
      xa.ncols = plan->stage_last[s] - plan->stage_first[s] + 1;
      for (int j = 0; j < xa.ncols; j++) xa.cols[j] = plan->col[plan->stage_first[s] + j];
      xa.n = (uint32_t)n; xa.order = order; xa.out64 = bits > 32; xa.npass = 0;
      unsigned char* kk = nullptr;
      DTHIP_TRY(sc.get<unsigned char>((size_t)n * (xa.out64 ? 8 : 4), &kk));
      uint32_t* dummy = nullptr;
      DTHIP_TRY(sc.get<uint32_t>(16, &dummy));
      xa.out = kk; xa.hist = dummy;
      DTHIP_TRY(launch_xform_hist(ctx, xa));
      DTHIP_TRY(launch_mark_heads(ctx, kk, xa.out64, n, heads));
    }
    DTHIP_TRY(heads_to_offsets(ctx, sc, res, nullptr, 0, heads, n, g));
  }
  return DTHIP_OK;
}

static int check_common(dthip_ctx* ctx, int64_t nrows, int mem) {
  if (!ctx) { set_error("null context"); return DTHIP_EINVAL; }
  if (nrows < 0 || nrows > (int64_t)INT32_MAX) {
    set_error("nrows=%lld is outside [0, 2^31-1]: RowIndex and group offsets are int32", (long long)nrows);
    return DTHIP_EINVAL;
  }
  if (mem != DTHIP_HOST && mem != DTHIP_DEVICE) { set_error("bad mem space %d", mem); return DTHIP_EINVAL; }
  DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
  return DTHIP_OK;
}

// dthip_last_call_stats: the outermost query entry point starts the record, nested ones add to it
struct CallScope {
  dthip_ctx* c;
  explicit CallScope(dthip_ctx* ctx) : c(ctx) { if (c->call_depth++ == 0) memset(c->call_stats, 0, sizeof(c->call_stats)); }
  ~CallScope() { c->call_depth--; }
};

static int stage_cols(dthip_ctx* ctx, Scratch& sc, const dthip_col* cols, int ncols, int64_t nrows, int mem,
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

static int empty_result(dthip_ctx* ctx, dthip_result* res) {
  void* off = nullptr;
  DTHIP_TRY(result_alloc(ctx, res, sizeof(int32_t), &off));
  DTHIP_CHECK_HIP(hipMemsetAsync(off, 0, sizeof(int32_t), ctx->stream));
  res->offsets = static_cast<int32_t*>(off);
  res->nrows = 0; res->ngroups = 0;
  return DTHIP_OK;
}

static int reduce_outs_for(int op, void* dst, ReduceOuts* o) {
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

// ---- bucketed aggregation (bucket.hip): DT[:, aggs, by(keys)] without a sort -------------
// Accumulators each value column needs for the requested reducers.
constexpr int DTHIP_RETRY_NA = 3;                 // internal: a value column guessed NA-free holds an NA, aggregate again with valid counts

// guess_nona: the column is believed to hold no NA (sampled): its valid count IS the group size, so the per-column
// counter (one DS atomic per row) is dropped and the kernels verify the belief on every row instead (ACC_CHKNA)
static int acc_flags_for(const dthip_agg* aggs, int naggs, int col, int vstype, int colflags = 0, bool guess_nona = false) {
  int f = (colflags & DTHIP_FLAG_NONA) ? ACC_NONA : 0;
  const bool isf = stype_is_float(vstype);
  for (int a = 0; a < naggs; a++) {
    if (aggs[a].op == DTHIP_COUNT0 || aggs[a].col != col) continue;
    switch (aggs[a].op) {
      case DTHIP_SUM: f |= ACC_SUM; break;
      case DTHIP_MEAN: f |= ACC_VCNT | (isf ? ACC_SUM : ACC_FSUM); break;
      case DTHIP_MIN: f |= ACC_MIN | ACC_VCNT; break;
      case DTHIP_MAX: f |= ACC_MAX | ACC_VCNT; break;
      case DTHIP_COUNT: f |= ACC_VCNT; break;
      default: break;
    }
  }
  if (guess_nona && (f & ACC_VCNT) && !(f & ACC_NONA)) f = (f & ~ACC_VCNT) | ACC_CHKNA;
  return f;
}

static int floor_log2_sz(size_t v) { int b = -1; while (v) { b++; v >>= 1; } return b; }

constexpr size_t BUCKET_LDS_TABLE = 144 * 1024;   // LDS bytes one aggregation table may take
constexpr int BUCKET_MAX_R = 14;                  // slot keys are uint16
constexpr int BUCKET_MAX_D = 11;                  // <= 2048 buckets in one partition pass

static bool bucket_need_counts(const dthip_ctx* ctx, const dthip_agg* aggs, int naggs) {
  if (ctx->agg_offsets) return true;
  for (int a = 0; a < naggs; a++) if (aggs[a].op == DTHIP_COUNT0) return true;
  return false;
}

// Decides whether the bucket path applies; fills the slot-bit count r.
static bool bucket_eligible(const dthip_ctx* ctx, const KeyPlan& plan, const std::vector<dthip_col>& vd,
                            const std::vector<int>& used, const dthip_agg* aggs, int naggs, int64_t n, int* r_out, bool guess_nona = false) {
  if (ctx->agg_path == 1) return false;
  if (plan.nstages != 1) return false;
  const int B = plan.stage_bits[0];
  if (B > 32 || B < 1) return false;
  const int first_flag = (bucket_need_counts(ctx, aggs, naggs) || guess_nona) ? ACC_CNT : ACC_PRES;
  int r = BUCKET_MAX_R;
  bool first = true;
  auto fit = [&](int f) { int rc = BUCKET_MAX_R; while (rc > 0 && table_agg_lds_bytes(f, 1u << rc) > BUCKET_LDS_TABLE) rc--; return rc; };
  for (int c : used) {
    const int sz = stype_size(vd[c].stype);
    if (sz != 4 && sz != 8) return false;
    const int f = acc_flags_for(aggs, naggs, c, vd[c].stype, vd[c].flags, guess_nona) | (first ? first_flag : 0);
    first = false;
    r = std::min(r, fit(f));
  }
  if (first) r = std::min(r, fit(first_flag));
  if (r > B) r = B;
  if (B - r > BUCKET_MAX_D) return false;
  // the dense accumulator arrays have 2^B slots: only worth it when the key range is dense enough
  if (ctx->agg_path != 2 && (1ULL << B) > 16ULL * (unsigned long long)n + 4096ULL) return false;
  if (ctx->agg_path != 2 && n < 4096) return false;
  *r_out = r;
  return true;
}

static int bucket_groupby_agg(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const KeyPlan& plan,
                              const std::vector<dthip_col>& kd, const std::vector<dthip_col>& vd,
                              const std::vector<int>& used, const dthip_agg* aggs, int naggs, int64_t n, int r, bool guess_nona = false,
                              int r_counting = -1) {
  // r_counting: the slot bits the same query gets WITHOUT the NA-free guess.  When they equal r, a wrong guess repeats only
  // the aggregation over the rows already partitioned (the partition's output does not depend on the guess) instead of
  // the whole query (DTHIP_RETRY_NA): C3 with mean(), one planted NaN: 2.0x -> see DESIGN 6 "adversarial inputs"
  const int nkeys = plan.nkeys;
  const int B = plan.stage_bits[0];
  KeyXform kx;
  memset(&kx, 0, sizeof(kx));
  kx.ncols = nkeys;
  for (int k = 0; k < nkeys; k++) kx.cols[k] = plan.col[k];
  // vector key loads: up to 4 aligned key columns, all int64 or all int32, and aligned value columns
  int km = 0;
  if (nkeys <= 4) {
    bool all64 = true, all32 = true, aligned = true;
    for (int k = 0; k < nkeys; k++) {
      all64 &= kx.cols[k].stype == DTHIP_INT64;
      all32 &= kx.cols[k].stype == DTHIP_INT32;
      aligned &= (reinterpret_cast<uintptr_t>(kx.cols[k].data) & 15) == 0;
    }
    if (aligned && all64) km = 1;
    else if (aligned && all32) km = 2;
  }
  for (int c : used) if (reinterpret_cast<uintptr_t>(vd[c].data) & 15) km = 0;
  BucketGeom g;
  bucket_geometry(ctx, n, B, r, km, &g);
  const size_t nslots = (size_t)g.F * g.S;

  // want_offsets: group sizes are part of the result; need_cnt: rows per slot are COUNTED -- also when value columns are
  // guessed NA-free, whose valid counts the row counts then stand for
  const bool want_offsets = bucket_need_counts(ctx, aggs, naggs);
  bool need_cnt = want_offsets || guess_nona;
  int first_flag = need_cnt ? ACC_CNT : ACC_PRES;
  // --- partition (skipped when one table holds the whole key range) ---
  uint16_t* kpart = nullptr;
  std::vector<const void*> vsrc(vd.size(), nullptr);
  for (int c : used) vsrc[c] = vd[c].data;
  uint32_t* bbase = nullptr; WorkItem* items = nullptr; uint32_t* nitems = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)g.F + 5, &bbase));
  nitems = bbase + g.F + 1;
  uint32_t* d_bad = bbase + g.F + 2;
  uint32_t* d_clustered = bbase + g.F + 3;        // [2]
  // SMALL path (one table of <= SMALL_SLOTS slots: BASELINE C1, 1e6 rows / 100 groups, is bound by its ~17 launches, not by
  // bytes): the plan kernel also initialises every table, and one single-workgroup kernel turns the slot counts into the
  // group list, the offsets and the group count
  const bool small = ctx->small_path != 0 && g.d == 0 && nslots <= SMALL_SLOTS;
  FillList fills;
  fills.n = 0;
  auto fill = [&](void* p, size_t bytes, int byte) -> int {
    if (small && fills.n < 12 && (bytes & 3) == 0) {
      fills.p[fills.n] = static_cast<uint32_t*>(p); fills.words[fills.n] = (uint32_t)(bytes / 4); fills.val[fills.n] = byte ? 0xFFFFFFFFu : 0u;
      fills.n++;
      return DTHIP_OK;
    }
    DTHIP_CHECK_HIP(hipMemsetAsync(p, byte, bytes, ctx->stream));
    return DTHIP_OK;
  };
  DTHIP_TRY(fill(d_bad, sizeof(uint32_t), 0));
  uint32_t M;
  {
    // with fewer buckets than CUs (BASELINE C2: 32) the aggregation is bound by its DS atomics, one 1024-thread workgroup
    // per CU: many small parts even out the tail (measured on 1e8 rows x 4 columns: 512 parts 0.29 ms per column,
    // 1500 parts 0.26; C2 3.07 -> 2.89 ms); with >= 1024 buckets the parts are whole buckets anyway
    static const int part_div_env = getenv("DTHIP_PART_DIV") ? atoi(getenv("DTHIP_PART_DIV")) : 0;
    const int part_div = part_div_env > 0 ? part_div_env : (g.F < (uint32_t)ctx->num_cus ? 16 : 4);
    const uint64_t denom = std::max<uint64_t>(g.F, (uint64_t)ctx->num_cus * part_div);
    uint64_t m = (2 * (uint64_t)n + denom - 1) / denom;
    // a part must amortise the set-up and the flush of its LDS table (S slots): small tables allow small parts, so a
    // 1e6-row frame with 100 groups (BASELINE C1) still spreads over a few hundred workgroups instead of 16
    const uint64_t m_min = std::min<uint64_t>(65536, std::max<uint64_t>(4096, 16ull * g.S));
    if (m < m_min) m = m_min;
    m = (m + 7) & ~7ULL;
    M = (uint32_t)std::min<uint64_t>(m, 0x7FFFFFF8ULL);
  }
  const uint32_t max_items = g.F + (uint32_t)((uint64_t)n / M) + 1;
  DTHIP_TRY(sc.get<WorkItem>(max_items, &items));
  // sorted / clustered / constant keys? (decides which kernel variants run; one tiny read-back)
  bool clustered = ctx->cluster_mode == 2;
  bool even = ctx->bucket_variant == 3;           // rows spread evenly over the buckets (decides the tile-local layout)
  if (ctx->cluster_mode == 0 && n >= (1 << 20)) DTHIP_TRY(launch_bucket_cluster_sample(ctx, kx, n, g.r, d_clustered, &clustered, g.F, &even));
  int src = 1;
  // TILE-LOCAL layout: no histogram pass.  Every partition tile writes its rows, ordered by bucket, into its own row
  // range plus a 2-byte directory entry per bucket; the aggregation walks one short segment per tile.  Keys are read
  // once (16 B/row less HBM traffic for C3).  Random row order only: for sorted / clustered keys a bucket's rows sit in
  // few tiles and the exact-position layout (with its clustered kernel variants and row-range work items) is better.
  const uint16_t* dirT = nullptr; uint32_t dstride = 0;
  // Worth it when the segments are short and alike: >= 1024 buckets (<= 12 rows of a tile per bucket) and no hot bucket
  // (sampled).  Measured on 1e9 rows: C3 9.5 -> 9.1 ms, C4 11.0 -> 9.4; but 4 x float64 columns over 32 buckets 2.8 -> 3.9
  // and a heavily skewed key 10.6 -> 14.1, which therefore keep the exact-position layout.
  const bool tile_local = g.d > 0 && !clustered && ctx->bucket_variant != 2 && g.block == 1024 &&
                          ((n >= (1 << 22) && g.F >= 1024 && even) || ctx->bucket_variant == 3);   // variant 3: forced (tests)
  if (tile_local) {
    // round 6: 1024 x 16-row tiles (segments of 16 instead of 12 rows: fewer partly used sectors for the aggregation);
    // DTHIP_TL_ITEMS=12 keeps round 5's tiles (A/B)
    static const int tl_items = getenv("DTHIP_TL_ITEMS") ? atoi(getenv("DTHIP_TL_ITEMS")) : 16;
    if (tl_items == 16) {
      int maxw = 4;
      for (int c : used) maxw = std::max(maxw, stype_size(vd[c].stype));
      (void)bucket_tl16_geometry(ctx, n, maxw, &g);
    }
    uint16_t* dir = nullptr; uint16_t* dT = nullptr; uint32_t* tot = nullptr;
    dstride = (g.ntiles + 63u) & ~63u;
    DTHIP_TRY(sc.get<uint16_t>((size_t)g.ntiles * (g.F + 1) + 8, &dir));
    DTHIP_TRY(sc.get<uint16_t>((size_t)dstride * (g.F + 2) + 8, &dT));
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.F + 1, &tot));
    PayCols pc;
    memset(&pc, 0, sizeof(pc));
    DTHIP_TRY(sc.get<uint16_t>((size_t)g.ntiles * g.tile + 8, &kpart));
    for (int c : used) {
      unsigned char* vb = nullptr;
      const int w = stype_size(vd[c].stype);
      DTHIP_TRY(sc.get<unsigned char>((size_t)g.ntiles * g.tile * w + 64, &vb));
      pc.in[pc.n] = vd[c].data; pc.out[pc.n] = vb; pc.width[pc.n] = w; pc.n++;
      vsrc[c] = vb;
    }
    src = 2;
    DTHIP_TRY(launch_bucket_partition(ctx, kx, n, g, nullptr, nullptr, kpart, pc, false, dir, d_bad));
    DTHIP_TRY(launch_dir_prepare(ctx, dir, g.ntiles, g.F, dT, dstride, tot, M, items, nitems));
    dirT = dT;
  } else if (g.d > 0) {
    uint32_t* P = nullptr; uint32_t* gtot = nullptr; uint32_t* tot = nullptr;
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.ntiles * g.F, &P));
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.G * g.F, &gtot));
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.F, &tot));
    DTHIP_TRY(launch_bucket_hist(ctx, kx, n, g, P, gtot, d_bad, clustered));
    DTHIP_TRY(launch_bucket_gscan(ctx, g, gtot, tot, nullptr, 0));
    DTHIP_TRY(launch_bucket_plan(ctx, tot, g.F, 0, M, bbase, items, nitems));
    DTHIP_TRY(launch_bucket_gscan(ctx, g, gtot, tot, bbase, 1));
    PayCols pc;
    memset(&pc, 0, sizeof(pc));
    DTHIP_TRY(sc.get<uint16_t>((size_t)n + 8, &kpart));
    for (int c : used) {
      unsigned char* vb = nullptr;
      const int w = stype_size(vd[c].stype);
      DTHIP_TRY(sc.get<unsigned char>((size_t)n * w + 64, &vb));
      pc.in[pc.n] = vd[c].data; pc.out[pc.n] = vb; pc.width[pc.n] = w; pc.n++;
      vsrc[c] = vb;
    }
    src = 0;
    DTHIP_TRY(launch_bucket_partition(ctx, kx, n, g, P, gtot, kpart, pc, clustered));
  } else if (!small) {
    DTHIP_TRY(launch_bucket_plan(ctx, nullptr, 1, (uint32_t)n, M, bbase, items, nitems));
  }

  // --- dense accumulators + one aggregation launch per value column ---
  uint32_t* d_cnt = nullptr;      // rows per slot, or (no counts wanted) one presence bit per slot
  std::vector<AggTable> tabs(vd.size());
  std::vector<int> tflags(vd.size(), 0);
  int32_t* idx = nullptr;
  int64_t ng = 0;
  DTHIP_TRY(sc.get<int32_t>(std::min<size_t>(nslots, (size_t)n) + 1, &idx));
  for (int round = 0;; round++) {
  const size_t cnt_words = need_cnt ? nslots : (nslots + 31) / 32;
  DTHIP_TRY(sc.get<uint32_t>(cnt_words, &d_cnt));
  DTHIP_TRY(fill(d_cnt, cnt_words * 4, 0));
  for (auto& t : tabs) t = AggTable();
  bool first = true;
  for (int c : used) {              // tables first (all of them: the small path initialises them in ONE kernel) ...
    AggTable& t = tabs[c];
    int f = acc_flags_for(aggs, naggs, c, vd[c].stype, vd[c].flags, guess_nona);
    if (first) { f |= first_flag; if (need_cnt) t.cnt = d_cnt; else t.pres = d_cnt; }
    first = false;
    tflags[c] = f;
    if (f & ACC_SUM) { DTHIP_TRY(sc.get<unsigned long long>(nslots, &t.sum)); DTHIP_TRY(fill(t.sum, nslots * 8, 0)); }
    if (f & ACC_MIN) { DTHIP_TRY(sc.get<unsigned long long>(nslots, &t.mn)); DTHIP_TRY(fill(t.mn, nslots * 8, 0xFF)); }
    if (f & ACC_MAX) { DTHIP_TRY(sc.get<unsigned long long>(nslots, &t.mx)); DTHIP_TRY(fill(t.mx, nslots * 8, 0)); }
    if (f & ACC_FSUM) { DTHIP_TRY(sc.get<double>(nslots, &t.fsum)); DTHIP_TRY(fill(t.fsum, nslots * 8, 0)); }
    if (f & ACC_VCNT) { DTHIP_TRY(sc.get<uint32_t>(nslots, &t.vcnt)); DTHIP_TRY(fill(t.vcnt, nslots * 4, 0)); }
  }
  if (small) DTHIP_TRY(launch_bucket_plan(ctx, nullptr, 1, (uint32_t)n, M, bbase, items, nitems, &fills));
  first = true;
  for (int c : used) {              // ... then one aggregation launch per value column
    const AggTable& t = tabs[c];
    const int f = tflags[c];
    if (src == 2) {
      TableAggSegArgs sa;
      memset(&sa, 0, sizeof(sa));
      sa.items = items; sa.nitems = nitems; sa.max_items = max_items; sa.kpart = kpart; sa.val = vsrc[c]; sa.vstype = vd[c].stype;
      sa.dirT = dirT; sa.dstride = dstride; sa.tile_rows = g.tile; sa.S = g.S; sa.flags = f; sa.tab = t; sa.bad = d_bad;
      DTHIP_TRY(launch_table_agg_seg(ctx, sa));
      first = false;
      continue;
    }
    TableAggArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.items = items; ta.nitems = nitems; ta.max_items = max_items; ta.src = src;
    ta.kpart = kpart; ta.kx = kx; ta.val = vsrc[c]; ta.vstype = vd[c].stype; ta.S = g.S; ta.flags = f; ta.tab = t; ta.bad = d_bad; ta.clustered = clustered;
    DTHIP_TRY(launch_table_agg(ctx, ta));
    first = false;
  }
  if (first && src == 2) {
    TableAggSegArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.items = items; sa.nitems = nitems; sa.max_items = max_items; sa.kpart = kpart; sa.val = nullptr; sa.vstype = DTHIP_INT32;
    sa.dirT = dirT; sa.dstride = dstride; sa.tile_rows = g.tile; sa.S = g.S; sa.flags = first_flag;
    if (need_cnt) sa.tab.cnt = d_cnt; else sa.tab.pres = d_cnt;
    DTHIP_TRY(launch_table_agg_seg(ctx, sa));
  } else if (first) {   // no value column at all: row counts (or key presence) alone
    TableAggArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.items = items; ta.nitems = nitems; ta.max_items = max_items; ta.src = src;
    ta.kpart = kpart; ta.kx = kx; ta.val = nullptr; ta.vstype = DTHIP_INT32; ta.S = g.S; ta.flags = first_flag;
    if (need_cnt) ta.tab.cnt = d_cnt; else ta.tab.pres = d_cnt;
    ta.bad = d_bad; ta.clustered = clustered;
    DTHIP_TRY(launch_table_agg(ctx, ta));
  }

  // --- groups = non-empty slots in slot order ---
  PredArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.data = d_cnt; pa.stype = DTHIP_INT32; pa.cmp = DTHIP_GT; pa.ci = 0; pa.is_mask = need_cnt ? 0 : 2;
  ng = 0;
  if (small) {
    void* off = nullptr;
    if (want_offsets) DTHIP_TRY(result_alloc(ctx, res, sizeof(int32_t) * (nslots + 2), &off));
    SmallGroupsArgs ga;
    ga.cnt = d_cnt; ga.bits = need_cnt ? 0 : 1; ga.nslots = (uint32_t)nslots; ga.idx = idx;
    ga.off = static_cast<uint32_t*>(off); ga.bad = d_bad;
    uint32_t w[2] = {0, 0};
    if (ctx->small_path == 2 && host_words(ctx)) {
      // the kernel writes its two words straight into mapped host memory: no copy command, one stream wait
      ga.out = ctx->host_words_dev;
      DTHIP_TRY(launch_small_groups(ctx, ga));
      DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      w[0] = reinterpret_cast<volatile uint32_t*>(ctx->host_words)[0];
      w[1] = reinterpret_cast<volatile uint32_t*>(ctx->host_words)[1];
    } else {
      ga.out = d_clustered;                        // its two words were read before the partition and are free now
      DTHIP_TRY(launch_small_groups(ctx, ga));
      DTHIP_TRY(read_back(ctx, w, d_clustered, sizeof(w)));
    }
    if (plan.speculative && (w[1] & 1u)) return DTHIP_RETRY_EXACT;
    if (w[1] & 2u) return DTHIP_RETRY_NA;
    ng = w[0];
    res->offsets = static_cast<int32_t*>(off);
  } else {
    DTHIP_TRY(launch_compact(ctx, pa, (int64_t)nslots, idx, &ng));
    if (plan.speculative || guess_nona) {
      uint32_t bad = 0;
      DTHIP_TRY(read_back(ctx, &bad, d_bad, sizeof(bad)));
      if (plan.speculative && (bad & 1u)) return DTHIP_RETRY_EXACT;
      if (bad & 2u) {
        if (round == 0 && guess_nona && r_counting == r) {
          // the NA-free guess was wrong, the partitioned rows are still right: aggregate them once more, counting
          ctx->call_stats[1]++;
          guess_nona = false;
          need_cnt = want_offsets; first_flag = need_cnt ? ACC_CNT : ACC_PRES;
          DTHIP_CHECK_HIP(hipMemsetAsync(d_bad, 0, sizeof(uint32_t), ctx->stream));
          continue;
        }
        return DTHIP_RETRY_NA;
      }
    }
  }
  break;
  }
  res->nrows = n; res->ngroups = ng;
  if (want_offsets && !small) {
    // offsets = exclusive scan of the group sizes (Groupby offsets, groupby.h:54-91)
    void* off = nullptr;
    DTHIP_TRY(result_alloc(ctx, res, sizeof(int32_t) * ((size_t)ng + 2 + (size_t)ng / 8192 + 1), &off));
    DTHIP_TRY(launch_gather(ctx, d_cnt, DTHIP_INT32, idx, ng, off));
    DTHIP_TRY(launch_scan_tiles(ctx, static_cast<uint32_t*>(off), (uint32_t)ng, static_cast<uint32_t*>(off) + ng));
    res->offsets = static_cast<int32_t*>(off);
  }
  // group-key columns: the slot index is the packed transformed key
  for (int k = 0; k < nkeys; k++) {
    void* kp = nullptr;
    DTHIP_TRY(result_alloc(ctx, res, (size_t)ng * stype_size(kd[k].stype), &kp));
    res->key[k] = kp;
    DTHIP_TRY(launch_untransform_keys(ctx, idx, 0, nullptr, ng, plan.col[k], plan.nsig[k], kp));
  }
  for (int a = 0; a < naggs; a++) {
    void* ap = nullptr;
    DTHIP_TRY(result_alloc(ctx, res, (size_t)ng * stype_size(res->agg_stype[a]), &ap));
    res->agg[a] = ap;
  }
  for (int c : used) {
    TableFinArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.idx = idx; fa.ng = (uint32_t)ng; fa.tab = tabs[c]; fa.vstype = vd[c].stype;
    if (tflags[c] & ACC_CHKNA) fa.tab.vcnt = d_cnt;        // verified NA-free: the valid count of a group is its size
    std::vector<std::pair<int, int>> dups;
    int first_of_op[6] = {-1, -1, -1, -1, -1, -1};
    for (int a = 0; a < naggs; a++) {
      if (aggs[a].op == DTHIP_COUNT0 || aggs[a].col != c) continue;
      if (first_of_op[aggs[a].op] >= 0) { dups.push_back({a, first_of_op[aggs[a].op]}); continue; }
      first_of_op[aggs[a].op] = a;
      switch (aggs[a].op) {
        case DTHIP_SUM: fa.o_sum = res->agg[a]; break;
        case DTHIP_MEAN: fa.o_mean = res->agg[a]; break;
        case DTHIP_MIN: fa.o_min = res->agg[a]; break;
        case DTHIP_MAX: fa.o_max = res->agg[a]; break;
        default: fa.o_count = static_cast<int64_t*>(res->agg[a]); break;
      }
    }
    DTHIP_TRY(launch_table_finalize(ctx, fa));
    for (auto& d : dups)
      DTHIP_CHECK_HIP(hipMemcpyAsync(res->agg[d.first], res->agg[d.second], (size_t)ng * stype_size(res->agg_stype[d.first]),
                                     hipMemcpyDeviceToDevice, ctx->stream));
  }
  for (int a = 0; a < naggs; a++)
    if (aggs[a].op == DTHIP_COUNT0) DTHIP_TRY(launch_count0(ctx, res->offsets, ng, static_cast<int64_t*>(res->agg[a])));
  return DTHIP_OK;
}


// a path gave up after it had started to fill `res` (the hash combiner's pass i > 0 on a table overflow): its buffers go
// back to the cache AND every pointer into them is cleared, so the path that takes over cannot hand out a dangling one
static void drop_partial_result(dthip_ctx* ctx, dthip_result* res) {
  for (void* p : res->owned) dev_release(ctx, p);
  res->owned.clear();
  res->offsets = nullptr; res->rowindex = nullptr; res->ngroups = 0;
  for (auto& k : res->key) k = nullptr;
  for (auto& a : res->agg) a = nullptr;
}

// ---- hash combiner for sparse keys (bucket.hip): partial groups + merge ------------------------
constexpr int HASH_PK_BITS = 24, HASH_R = 13;          // pseudo key: 2048 buckets by hash

// distinct-key estimate from a strided sample of m rows: group the sample with the ordinary path,
// invert  u = N (1 - exp(-m / N))  (u distinct keys among m draws from N equally likely keys)
static int estimate_distinct(dthip_ctx* ctx, const std::vector<dthip_col>& kd, int nkeys, int64_t n, int na_pos, double* est) {
  const int64_t m = std::min<int64_t>(n, 1 << 21);
  Scratch sc(ctx);
  int32_t* ri = nullptr;
  DTHIP_TRY(sc.get<int32_t>((size_t)m, &ri));
  DTHIP_TRY(launch_sample_rows(ctx, ri, m, n));
  std::vector<dthip_col> sk(nkeys);
  for (int k = 0; k < nkeys; k++) {
    unsigned char* b = nullptr;
    DTHIP_TRY(sc.get<unsigned char>((size_t)m * stype_size(kd[k].stype), &b));
    DTHIP_TRY(launch_gather(ctx, kd[k].data, kd[k].stype, ri, m, b));
    sk[k] = kd[k];
    sk[k].data = b;
  }
  dthip_result* r = nullptr;
  DTHIP_TRY(dthip_groupby(ctx, sk.data(), nkeys, m, na_pos, DTHIP_DEVICE, 0, &r));
  const double u = (double)dthip_result_ngroups(r);
  dthip_result_free(ctx, r);
  if (m == n) { *est = u; return DTHIP_OK; }
  if (u > 0.97 * (double)m) { *est = 1e300; return DTHIP_OK; }     // (nearly) all distinct in the sample
  double lo = u, hi = 1e15;
  for (int it = 0; it < 200; it++) {
    const double mid = 0.5 * (lo + hi);
    const double f = mid * (1.0 - exp(-(double)m / mid));
    if (f < u) lo = mid; else hi = mid;
  }
  *est = hi;
  return DTHIP_OK;
}

static int hash_groupby_agg(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const KeyPlan& plan,
                            const std::vector<dthip_col>& kd, const std::vector<dthip_col>& vd,
                            const std::vector<int>& used, const dthip_agg* aggs, int naggs, int64_t n, int na_pos) {
  const int nkeys = plan.nkeys;
  if (ctx->hash_mode == 1 || ctx->in_merge || ctx->agg_path == 1) return DTHIP_NOT_APPLICABLE;
  // Several value columns (round 2): the rows are partitioned ONCE with every value column as payload; each column
  // then gets its own pass of LDS hash tables over the partitioned (key, value) rows and its own merge.  Every merge
  // orders the same set of keys, so the per-column results line up group by group.
  if (plan.nstages != 1 || (int)used.size() > MAX_PAYCOLS - 1) return DTHIP_NOT_APPLICABLE;
  for (int c : used) {
    if (vd[c].flags & DTHIP_FLAG_NONA) return DTHIP_NOT_APPLICABLE;
    if (stype_size(vd[c].stype) != 4 && stype_size(vd[c].stype) != 8) return DTHIP_NOT_APPLICABLE;
  }
  if (ctx->hash_mode != 2 && n < (1 << 22)) return DTHIP_NOT_APPLICABLE;
  const bool need_cnt = bucket_need_counts(ctx, aggs, naggs);
  const uint32_t F = 1u << (HASH_PK_BITS - HASH_R);
  double est = 0;
  DTHIP_TRY(estimate_distinct(ctx, kd, nkeys, n, na_pos, &est));
  // Passes of hash tables: (payload column, accumulator set).  A column's accumulators share one pass when an entry
  // (8-byte key + accumulators) is small enough for 2048 tables of load <= 0.75 to hold the estimated distinct keys;
  // otherwise they are split into {sum / mean / count}, {min}, {max} passes over the same partitioned rows.
  struct HPass { int col; int slot; int flags; uint32_t C; };
  auto table_entries = [](int flags) {
    const size_t entry = hash_agg_entry_bytes(flags);
    uint32_t C = (uint32_t)((158 * 1024) / entry) - 1;                // the whole LDS of a CU for one table ...
    for (;; C--) {                                                     // ... with a prime number of entries (double hashing)
      bool prime = C % 2 != 0;
      for (uint32_t q = 3; prime && q * q <= C; q += 2) prime = C % q != 0;
      if (prime) break;
    }
    return C;
  };
  auto fits = [&](int flags) { return est * 1.05 <= 0.75 * (double)F * (double)table_entries(flags); };
  std::vector<HPass> passes;
  std::vector<int> agg_pass(naggs, 0);           // which pass computes aggregate a (count() rides with pass 0)
  if (used.empty()) {
    passes.push_back(HPass{-1, -1, need_cnt ? ACC_CNT : 0, 0});
  } else {
    for (size_t i = 0; i < used.size(); i++) {
      const int c = used[i];
      const int fl = acc_flags_for(aggs, naggs, c, vd[c].stype);
      const int first = (need_cnt && passes.empty()) ? ACC_CNT : 0;
      if (fits(fl | first)) {
        for (int a = 0; a < naggs; a++) if (aggs[a].op != DTHIP_COUNT0 && aggs[a].col == c) agg_pass[a] = (int)passes.size();
        passes.push_back(HPass{c, (int)i, fl | first, 0});
        continue;
      }
      // split: {sum / mean / count}, {min}, {max}; min and max keep the valid count their NA rule needs
      bool want[3] = {false, false, false};
      for (int a = 0; a < naggs; a++) {
        if (aggs[a].op == DTHIP_COUNT0 || aggs[a].col != c) continue;
        want[aggs[a].op == DTHIP_MIN ? 1 : aggs[a].op == DTHIP_MAX ? 2 : 0] = true;
      }
      const int parts[3] = {fl & (ACC_SUM | ACC_FSUM | ACC_VCNT), ACC_MIN | ACC_VCNT, ACC_MAX | ACC_VCNT};
      for (int q = 0; q < 3; q++) {
        if (!want[q]) continue;
        const int f2 = parts[q] | ((need_cnt && passes.empty()) ? ACC_CNT : 0);
        if (!fits(f2)) return DTHIP_NOT_APPLICABLE;
        for (int a = 0; a < naggs; a++) {
          if (aggs[a].op == DTHIP_COUNT0 || aggs[a].col != c) continue;
          if ((aggs[a].op == DTHIP_MIN ? 1 : aggs[a].op == DTHIP_MAX ? 2 : 0) == q) agg_pass[a] = (int)passes.size();
        }
        passes.push_back(HPass{c, (int)i, f2, 0});
      }
    }
  }
  for (auto& hp : passes) hp.C = table_entries(hp.flags);
  if (!fits(passes[0].flags)) return DTHIP_NOT_APPLICABLE;
  const int ncolpass = (int)passes.size();
  KeyXform kx;
  memset(&kx, 0, sizeof(kx));
  kx.ncols = nkeys;
  for (int k = 0; k < nkeys; k++) kx.cols[k] = plan.col[k];
  // one int64 key: the raw key IS a usable 64-bit image (round 3): no packed-key array is written (8 of the 20 bytes
  // per row hash_xform moved), the key column itself is payload 0 of the partition, and the partial groups' keys come
  // out typed already
  static const bool raw_ok = !(getenv("DTHIP_HASH_RAW") && atoi(getenv("DTHIP_HASH_RAW")) == 0);
  const bool raw_key = raw_ok && nkeys == 1 && kd[0].stype == DTHIP_INT64;
  // round 6: with a raw key the histogram and partition kernels HASH THE KEY COLUMN ON THE FLY (a two-multiply 24-bit hash,
  // keyxform.hpp hash_pk24) instead of reading a pseudo-key array that a pass of its own wrote: 12 of the 80 bytes per row
  // and one sweep less (DTHIP_HASH_FUSED=0: the pseudo-key pass of rounds 3-5, A/B)
  static const bool fused_ok = !(getenv("DTHIP_HASH_FUSED") && atoi(getenv("DTHIP_HASH_FUSED")) == 0);
  const bool fused_pk = raw_key && fused_ok;
  unsigned long long* xs = nullptr; int32_t* pk = nullptr;
  if (!fused_pk) DTHIP_TRY(sc.get<int32_t>((size_t)n + 4, &pk));
  if (fused_pk) {
  } else if (raw_key) {
    DTHIP_TRY(launch_hash_pk_raw(ctx, kd[0].data, n, pk));
  } else {
    DTHIP_TRY(sc.get<unsigned long long>((size_t)n + 2, &xs));
    DTHIP_TRY(launch_hash_xform(ctx, kx, n, xs, pk));
  }
  // the bucket machinery, driven by the pseudo key pk in [0, 2^24)
  KeyXform pkx;
  memset(&pkx, 0, sizeof(pkx));
  pkx.ncols = 1;
  pkx.cols[0].data = pk; pkx.cols[0].stype = DTHIP_INT32; pkx.cols[0].desc = 0; pkx.cols[0].edge = 0;
  pkx.cols[0].na_repl = 0; pkx.cols[0].inc = 0; pkx.cols[0].xmax = ~0ULL; pkx.cols[0].shift = 0;
  int km = 2;
  if (fused_pk) { pkx.cols[0].data = kd[0].data; pkx.cols[0].stype = DTHIP_KEY_HASH64; km = 1; }
  for (int c : used) if (reinterpret_cast<uintptr_t>(vd[c].data) & 15) km = 0;
  if (raw_key && (reinterpret_cast<uintptr_t>(kd[0].data) & 15)) km = 0;
  BucketGeom g;
  bucket_geometry(ctx, n, HASH_PK_BITS, HASH_R, km, &g);
  uint32_t* bbase = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)g.F + 8, &bbase));
  uint32_t* nitems = bbase + g.F + 1;
  uint32_t* d_bad = bbase + g.F + 2;
  uint32_t* d_outn = bbase + g.F + 3;
  uint32_t* d_ovf = bbase + g.F + 4;
  DTHIP_CHECK_HIP(hipMemsetAsync(bbase + g.F + 1, 0, 7 * sizeof(uint32_t), ctx->stream));
  uint32_t M;
  {
    const uint64_t denom = std::max<uint64_t>(g.F, (uint64_t)ctx->num_cus * 4);
    uint64_t m = (2 * (uint64_t)n + denom - 1) / denom;
    // a part must amortise the set-up and the flush of its LDS table (S slots): small tables allow small parts, so a
    // 1e6-row frame with 100 groups (BASELINE C1) still spreads over a few hundred workgroups instead of 16
    const uint64_t m_min = std::min<uint64_t>(65536, std::max<uint64_t>(4096, 16ull * g.S));
    if (m < m_min) m = m_min;
    m = (m + 7) & ~7ULL;
    M = (uint32_t)std::min<uint64_t>(m, 0x7FFFFFF8ULL);
  }
  const uint32_t max_items = g.F + (uint32_t)((uint64_t)n / M) + 1;
  WorkItem* items = nullptr;
  DTHIP_TRY(sc.get<WorkItem>(max_items, &items));
  const size_t part_rows = (size_t)n;
  uint16_t* kslot = nullptr;         // not written: the packed key itself travels as payload 0
  unsigned long long* xs_part = nullptr;
  DTHIP_TRY(sc.get<unsigned long long>(part_rows + 8, &xs_part));
  PayCols pc;
  memset(&pc, 0, sizeof(pc));
  pc.in[0] = raw_key ? kd[0].data : static_cast<const void*>(xs); pc.out[0] = xs_part; pc.width[0] = 8; pc.n = 1;
  std::vector<unsigned char*> v_part(std::max<size_t>(used.size(), 1), nullptr);
  for (size_t i = 0; i < used.size(); i++) {
    const int w = stype_size(vd[used[i]].stype);
    DTHIP_TRY(sc.get<unsigned char>(part_rows * w + 64, &v_part[i]));
    pc.in[pc.n] = vd[used[i]].data; pc.out[pc.n] = v_part[i]; pc.width[pc.n] = w; pc.n++;
  }
  {
    uint32_t* P = nullptr; uint32_t* gtot = nullptr; uint32_t* tot = nullptr;
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.ntiles * g.F, &P));
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.G * g.F, &gtot));
    DTHIP_TRY(sc.get<uint32_t>((size_t)g.F, &tot));
    DTHIP_TRY(launch_bucket_hist(ctx, pkx, n, g, P, gtot, d_bad, false));
    DTHIP_TRY(launch_bucket_gscan(ctx, g, gtot, tot, nullptr, 0));
    DTHIP_TRY(launch_bucket_plan(ctx, tot, g.F, 0, M, bbase, items, nitems));
    DTHIP_TRY(launch_bucket_gscan(ctx, g, gtot, tot, bbase, 1));
    DTHIP_TRY(launch_bucket_partition(ctx, pkx, n, g, P, gtot, kslot, pc, false));
  }

  int64_t ng_all = -1;
  for (int i = 0; i < ncolpass; i++) {
    Scratch sci(ctx);                        // this pass's partial groups
    const int c0 = passes[i].col;
    const int vst = c0 >= 0 ? vd[c0].stype : DTHIP_INT32;
    const int flags = passes[i].flags;
    const uint32_t C = passes[i].C;
    DTHIP_CHECK_HIP(hipMemsetAsync(d_outn, 0, 2 * sizeof(uint32_t), ctx->stream));
    // partial groups
    const size_t out_cap = std::min<size_t>((size_t)n, (size_t)max_items * (C + 1));
    HashAggArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.items = items; ha.nitems = nitems; ha.max_items = max_items; ha.xs = xs_part; ha.val = c0 >= 0 ? v_part[passes[i].slot] : nullptr; ha.vstype = vst;
    ha.C = C; ha.flags = flags; ha.out_n = d_outn; ha.out_cap = (uint32_t)out_cap; ha.overflow = d_ovf;
    DTHIP_TRY(sci.get<unsigned long long>(out_cap, &ha.o_key));
    if (flags & ACC_CNT) DTHIP_TRY(sci.get<uint32_t>(out_cap, &ha.o_tab.cnt));
    if (flags & ACC_VCNT) DTHIP_TRY(sci.get<uint32_t>(out_cap, &ha.o_tab.vcnt));
    if (flags & ACC_SUM) DTHIP_TRY(sci.get<unsigned long long>(out_cap, &ha.o_tab.sum));
    if (flags & ACC_MIN) DTHIP_TRY(sci.get<unsigned long long>(out_cap, &ha.o_tab.mn));
    if (flags & ACC_MAX) DTHIP_TRY(sci.get<unsigned long long>(out_cap, &ha.o_tab.mx));
    if (flags & ACC_FSUM) DTHIP_TRY(sci.get<double>(out_cap, &ha.o_tab.fsum));
    DTHIP_TRY(launch_hash_agg(ctx, ha));
    uint32_t hn[2] = {0, 0};
    DTHIP_TRY(read_back(ctx, hn, d_outn, sizeof(hn)));       // {number of partial groups, overflow bits}
    if (hn[1]) { ctx->call_stats[2]++; return DTHIP_NOT_APPLICABLE; }      // a table filled up: the sort path takes over
    const int64_t np = hn[0];

    // typed columns of the partial groups
    std::vector<dthip_col> k2(nkeys);
    for (int k = 0; k < nkeys; k++) {
      k2[k] = kd[k];
      if (raw_key) { k2[k].data = ha.o_key; continue; }
      unsigned char* bb = nullptr;
      DTHIP_TRY(sci.get<unsigned char>((size_t)np * stype_size(kd[k].stype) + 16, &bb));
      DTHIP_TRY(launch_untransform_keys(ctx, ha.o_key, 1, nullptr, np, plan.col[k], plan.nsig[k], bb));
      k2[k].data = bb;
    }
    const bool isf = stype_is_float(vst);
    PartialColsArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.tab = ha.o_tab; pa.n = (uint32_t)np; pa.vstype = vst;
    std::vector<dthip_col> v2;
    std::vector<dthip_agg> a2;
    int iSUM = -1, iFSUM = -1, iMIN = -1, iMAX = -1, iVCNT = -1, iCNT = -1;
    // partial SUMS are merged with DTHIP_FLAG_NONA: a partial that is NaN (inf - inf) or wrapped to INT64_MIN is a value
    auto add_col = [&](void* data, int st, int op) { v2.push_back(dthip_col{data, st, op == DTHIP_SUM ? DTHIP_FLAG_NONA : 0}); a2.push_back(dthip_agg{op, (int32_t)v2.size() - 1}); return (int)a2.size() - 1; };
    if (flags & ACC_SUM) { DTHIP_TRY(sci.get<unsigned long long>((size_t)np + 2, &pa.o_sum)); iSUM = add_col(pa.o_sum, isf ? DTHIP_FLOAT64 : DTHIP_INT64, DTHIP_SUM); }
    if (flags & ACC_FSUM) { DTHIP_TRY(sci.get<double>((size_t)np + 2, &pa.o_fsum)); iFSUM = add_col(pa.o_fsum, DTHIP_FLOAT64, DTHIP_SUM); }
    if (flags & ACC_MIN) { unsigned char* bb = nullptr; DTHIP_TRY(sci.get<unsigned char>((size_t)np * 8 + 16, &bb)); pa.o_min = bb; iMIN = add_col(bb, vst, DTHIP_MIN); }
    if (flags & ACC_MAX) { unsigned char* bb = nullptr; DTHIP_TRY(sci.get<unsigned char>((size_t)np * 8 + 16, &bb)); pa.o_max = bb; iMAX = add_col(bb, vst, DTHIP_MAX); }
    if (flags & ACC_VCNT) { DTHIP_TRY(sci.get<int64_t>((size_t)np + 2, &pa.o_vcnt)); iVCNT = add_col(pa.o_vcnt, DTHIP_INT64, DTHIP_SUM); }
    if (flags & ACC_CNT) { DTHIP_TRY(sci.get<int64_t>((size_t)np + 2, &pa.o_cnt)); iCNT = add_col(pa.o_cnt, DTHIP_INT64, DTHIP_SUM); }
    DTHIP_TRY(launch_partial_columns(ctx, pa));

    // merge: the ordinary path on the partial groups (few rows), keys in their own stypes and flags
    dthip_result* r2 = nullptr;
    const int saved_off = ctx->agg_offsets;
    ctx->in_merge = true; ctx->agg_offsets = 0;
    int rc = dthip_groupby_agg(ctx, k2.data(), nkeys, v2.empty() ? nullptr : v2.data(), (int)v2.size(),
                               a2.empty() ? nullptr : a2.data(), (int)a2.size(), np, na_pos, DTHIP_DEVICE, &r2);
    ctx->in_merge = false; ctx->agg_offsets = saved_off;
    if (rc != DTHIP_OK) return rc;
    const int64_t ng = dthip_result_ngroups(r2);
    if (ng_all >= 0 && ng != ng_all) { dthip_result_free(ctx, r2); set_error("hash combiner: columns disagree on the number of groups"); return DTHIP_EDEVICE; }
    ng_all = ng;
    res->nrows = n; res->ngroups = ng;
    do {
      if (i == 0) {
        for (int k = 0; k < nkeys && rc == DTHIP_OK; k++) {
          void* kp = nullptr;
          const size_t bytes = (size_t)ng * stype_size(kd[k].stype);
          if ((rc = result_alloc(ctx, res, bytes, &kp)) != DTHIP_OK) break;
          res->key[k] = kp;
          if (bytes && hipMemcpyAsync(kp, dthip_result_key(r2, k), bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { set_error("D2D copy failed"); rc = DTHIP_EDEVICE; }
        }
        if (rc != DTHIP_OK) break;
        if (need_cnt) {
          void* off = nullptr;
          if ((rc = result_alloc(ctx, res, sizeof(int32_t) * ((size_t)ng + 2 + (size_t)ng / 8192 + 1), &off)) != DTHIP_OK) break;
          if ((rc = launch_narrow_i64_u32(ctx, static_cast<const long long*>(dthip_result_agg(r2, iCNT)), ng, static_cast<uint32_t*>(off))) != DTHIP_OK) break;
          if ((rc = launch_scan_tiles(ctx, static_cast<uint32_t*>(off), (uint32_t)ng, static_cast<uint32_t*>(off) + ng)) != DTHIP_OK) break;
          res->offsets = static_cast<int32_t*>(off);
        }
      }
      for (int a = 0; a < naggs && rc == DTHIP_OK; a++) {
        // this pass fills the aggregates assigned to it; count() (no column) goes with the first pass
        const bool mine = aggs[a].op == DTHIP_COUNT0 ? i == 0 : agg_pass[a] == i;
        if (!mine) continue;
        void* ap = nullptr;
        const size_t bytes = (size_t)ng * stype_size(res->agg_stype[a]);
        if ((rc = result_alloc(ctx, res, bytes, &ap)) != DTHIP_OK) break;
        res->agg[a] = ap;
        if (ng == 0) continue;
        const void* src = nullptr;
        switch (aggs[a].op) {
          case DTHIP_SUM:
            if (vst == DTHIP_FLOAT32) rc = launch_cast_f64_f32(ctx, static_cast<const double*>(dthip_result_agg(r2, iSUM)), ng, static_cast<float*>(ap));
            else src = dthip_result_agg(r2, iSUM);
            break;
          case DTHIP_MEAN:
            rc = launch_mean_div(ctx, static_cast<const double*>(dthip_result_agg(r2, isf ? iSUM : iFSUM)),
                                 static_cast<const long long*>(dthip_result_agg(r2, iVCNT)), ng, ap, vst == DTHIP_FLOAT32);
            break;
          case DTHIP_MIN: src = dthip_result_agg(r2, iMIN); break;
          case DTHIP_MAX: src = dthip_result_agg(r2, iMAX); break;
          case DTHIP_COUNT: src = dthip_result_agg(r2, iVCNT); break;
          default: src = dthip_result_agg(r2, iCNT); break;      // COUNT0
        }
        if (rc == DTHIP_OK && src && hipMemcpyAsync(ap, src, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { set_error("D2D copy failed"); rc = DTHIP_EDEVICE; }
      }
    } while (0);
    dthip_result_free(ctx, r2);
    if (rc != DTHIP_OK) return rc;
  }
  return DTHIP_OK;
}

}  // namespace dthip

extern "C" {

int dthip_abi_version(void) { return DTHIP_ABI_VERSION; }
const char* dthip_build_id(void) {
  static const char id[] =
#include "build_id.inc"
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
    if (value < 0 || value > 2) { set_error("hash_mode must be 0 (estimate), 1 (never) or 2 (whenever it fits)"); return DTHIP_EINVAL; }
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
  for (int i = 0; i < n; i++) out[i] = i < 4 ? ctx->call_stats[i] : 0;
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

// ---------------------------------------------------------------------------------
int dthip_groupby(dthip_ctx* ctx, const dthip_col* keys, int nkeys, int64_t nrows, int na_pos, int mem,
                  int want_rowindex, dthip_result** out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  CallScope call_scope(ctx);
  if (!keys || !out) { set_error("null argument"); return DTHIP_EINVAL; }
  const bool remove_na = na_pos == DTHIP_NA_REMOVE;
  if (remove_na) na_pos = DTHIP_NA_FIRST;      // sorted first, then cut off the front (sort.cc:598-608)
  if (na_pos != DTHIP_NA_FIRST && na_pos != DTHIP_NA_LAST) { set_error("na_pos %d not implemented", na_pos); return DTHIP_ENOTIMPL; }
  dthip_result* res = new dthip_result();
  res->nkeys = nkeys;
  int rc = DTHIP_OK;
  {
    Scratch sc(ctx);
    std::vector<dthip_col> kd;
    rc = stage_cols(ctx, sc, keys, nkeys, nrows, mem, &kd);
    if (rc == DTHIP_OK) {
      if (nrows == 0) {
        rc = empty_result(ctx, res);   // Groupby::zero_groups(), sort.cc:1428-1431
      } else {
        KeyPlan plan; Grouping g;
        // (round 6, measured and dropped: this call on the TILE-LOCAL levels of dthip_filter_groupby_rows with a predicate every
        // row passes -- no transform / tile-histogram passes, sequential writes -- took 6.6 + 5.4 + 6.2 ms for 1e9 rows against
        // 3.0 + 1.8 + 3.9 + 4.4 + 4.0 here: the gathers cost more than the passes they save, profiles/r06_groupby_tl_ab.txt)
        rc = group_core(ctx, sc, res, kd.data(), nkeys, nrows, na_pos, &plan, &g);
        if (rc == DTHIP_OK) {
          res->nrows = nrows; res->ngroups = g.ngroups; res->offsets = g.offsets;
          if (want_rowindex) {
            // the ordering may alias nothing user-owned here: it is always a scratch buffer
            result_adopt(sc, res, g.rowindex);
            res->rowindex = g.rowindex;
          }
          if (remove_na) {
            // SortContext::get_result_rowindex (sort.cc:598-608) cuts `nacount` rows off the front of the
            // NA-first ordering, where nacount is taken from the column sorted LAST (the context's current
            // column after continue_sort).  With one key that is exactly its NA group; with several keys it
            // is whatever leads the ordering -- reproduced as is.
            int32_t* scratch_idx = nullptr;
            int64_t skip = 0;
            rc = sc.get<int32_t>((size_t)nrows, &scratch_idx);
            if (rc == DTHIP_OK) {
              PredArgs p{kd[nkeys - 1].data, kd[nkeys - 1].stype, DTHIP_ISNA, 0.0, 0, 0};
              rc = launch_compact(ctx, p, nrows, scratch_idx, &skip);
            }
            if (rc == DTHIP_OK && skip > 0) {
              int32_t g0 = 0;              // groups that lie entirely inside the cut
              void* d_g0 = nullptr;
              rc = result_alloc(ctx, res, sizeof(int32_t) * ((size_t)g.ngroups + 2), &d_g0);
              int32_t* off2 = static_cast<int32_t*>(d_g0);
              if (rc == DTHIP_OK) rc = launch_offsets_drop_rows(ctx, g.offsets, g.ngroups, (int32_t)skip, off2, off2 + g.ngroups + 1);
              if (rc == DTHIP_OK) rc = read_back(ctx, &g0, off2 + g.ngroups + 1, sizeof(int32_t));
              if (rc == DTHIP_OK) {
                res->offsets = off2;
                res->ngroups = g.ngroups - g0;
                res->nrows = nrows - skip;
                if (res->rowindex) res->rowindex += skip;
              }
            }
          }
        }
      }
    }
  }
  if (rc != DTHIP_OK) { result_destroy(ctx, res); return rc; }
  *out = res;
  return DTHIP_OK;
}

int dthip_groupby_rows(dthip_ctx* ctx, const dthip_col* keys, int nkeys, const dthip_col* cols, int ncols,
                       int64_t nrows, int na_pos, int mem, int want_rowindex, dthip_result** out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  CallScope call_scope(ctx);
  if (!keys || !out || (ncols > 0 && !cols) || ncols < 0) { set_error("null argument"); return DTHIP_EINVAL; }
  if (na_pos != DTHIP_NA_FIRST && na_pos != DTHIP_NA_LAST) { set_error("na_pos %d not implemented", na_pos); return DTHIP_ENOTIMPL; }
  dthip_result* res = new dthip_result();
  res->nkeys = nkeys;
  res->col.assign(ncols, nullptr); res->col_stype.assign(ncols, 0);
  for (int c = 0; c < ncols; c++) res->col_stype[c] = cols[c].stype;
  int rc = DTHIP_OK;
  do {
    Scratch sc(ctx);
    std::vector<dthip_col> kd, cd;
    if ((rc = stage_cols(ctx, sc, keys, nkeys, nrows, mem, &kd)) != DTHIP_OK) break;
    if ((rc = stage_cols(ctx, sc, cols, ncols, nrows, mem, &cd)) != DTHIP_OK) break;
    if (nrows == 0) { rc = empty_result(ctx, res); break; }
    KeyPlan plan; Grouping g;
    if ((rc = plan_keys(ctx, sc, kd.data(), nkeys, nrows, na_pos, &plan, true, true)) != DTHIP_OK) break;   // guessed key range: verified below
    bool ride = plan.nstages == 1 && ncols + (want_rowindex ? 1 : 0) <= MAX_PAYCOLS && ncols > 0;
    for (int c = 0; c < ncols; c++) if (stype_size(cd[c].stype) < 4) ride = false;
    PaySpec ps;
    SortOut so;
    std::vector<int> slot(ncols, -1), is_key(ncols, -1);
    if (ride) {
      // the columns (and the row ids) ride through the radix passes: streaming reads and run-wise
      // writes instead of one random gather per column through the finished RowIndex
      ps.n = 0;
      if (want_rowindex) { ps.in[0] = nullptr; ps.width[0] = 4; ps.iota = true; ps.n = 1; }
      // a requested column that IS a key column does not ride along: the sorted packed keys are
      // turned back into it afterwards (streaming), which saves its bytes in every pass
      for (int c = 0; c < ncols; c++) {
        for (int k = 0; k < nkeys; k++)
          if (cols[c].data == keys[k].data && cols[c].stype == keys[k].stype) is_key[c] = k;
        if (is_key[c] >= 0) continue;
        slot[c] = ps.n;
        ps.in[ps.n] = cd[c].data; ps.width[ps.n] = stype_size(cd[c].stype); ps.n++;
      }
      // one int32 / int64 key that is also a wanted column: the last pass writes its original values (no untransform pass)
      static const bool fuse_ukey = !(getenv("DTHIP_FUSE_UKEY") && atoi(getenv("DTHIP_FUSE_UKEY")) == 0);
      int ukc = -1;
      if (fuse_ukey && nkeys == 1 && (kd[0].stype == DTHIP_INT64 || kd[0].stype == DTHIP_INT32))
        for (int c = 0; c < ncols; c++) if (is_key[c] == 0) { ukc = c; break; }
      if (ukc >= 0) {
        void* q = nullptr;
        if ((rc = result_alloc(ctx, res, (size_t)nrows * stype_size(kd[0].stype), &q)) != DTHIP_OK) break;
        ps.ukey_out = q;
      }
      if ((rc = alloc_head_bitmap(ctx, sc, nrows, &ps.head_bitmap)) != DTHIP_OK) break;
      rc = sort_stage(ctx, sc, plan, 0, nrows, nullptr, ps, &so);
      if (rc == DTHIP_RETRY_EXACT) {
        ctx->call_stats[0]++;
        // the sampled key range did not hold, so real keys lie OUTSIDE it: the exact range is usually WIDER, and with
        // several keys the packed width may now exceed 64 bits (two stages) -- then the columns cannot ride
        if ((rc = plan_keys(ctx, sc, kd.data(), nkeys, nrows, na_pos, &plan)) != DTHIP_OK) break;
        if (plan.nstages != 1) { ride = false; rc = DTHIP_OK; }
        else rc = sort_stage(ctx, sc, plan, 0, nrows, nullptr, ps, &so);
      }
      if (rc != DTHIP_OK) break;
    }
    if (ride) {
      if (so.heads_done) {
        if ((rc = heads_to_offsets(ctx, sc, res, nullptr, 0, nullptr, nrows, &g, ps.head_bitmap)) != DTHIP_OK) break;
      } else if (so.ukey_done) {
        // groups = runs of equal ORIGINAL key values (the transform is a bijection, NA <-> NA)
        if ((rc = heads_to_offsets(ctx, sc, res, ps.ukey_out, kd[0].stype == DTHIP_INT64, nullptr, nrows, &g)) != DTHIP_OK) break;
      } else
      if ((rc = heads_to_offsets(ctx, sc, res, so.keys, so.key64, nullptr, nrows, &g)) != DTHIP_OK) break;
      if (want_rowindex) { result_adopt(sc, res, so.pay[0]); res->rowindex = static_cast<int32_t*>(so.pay[0]); }
      bool ukey_used = false;
      for (int c = 0; c < ncols && rc == DTHIP_OK; c++) {
        if (is_key[c] >= 0) {
          void* q = nullptr;
          if (so.ukey_done && ps.ukey_out && res->col[c] == nullptr && is_key[c] == 0) {
            // the first copy of the key column is the buffer the last pass filled; further copies are duplicated from it
            bool first = true;
            for (int c2 = 0; c2 < c; c2++) if (is_key[c2] == 0) first = false;
            if (first) { res->col[c] = ps.ukey_out; continue; }
            const size_t bytes = (size_t)nrows * stype_size(cd[c].stype);
            if ((rc = result_alloc(ctx, res, bytes, &q)) != DTHIP_OK) break;
            if (hipMemcpyAsync(q, ps.ukey_out, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { set_error("D2D copy failed"); rc = DTHIP_EDEVICE; break; }
            res->col[c] = q;
            continue;
          }
          if (so.ukey_done) { set_error("groupby_rows: internal: packed keys missing"); rc = DTHIP_EDEVICE; break; }
          if (ps.ukey_out && is_key[c] == 0 && !ukey_used) { q = ps.ukey_out; ukey_used = true; }      // the buffer set aside for the last pass
          else if ((rc = result_alloc(ctx, res, (size_t)nrows * stype_size(cd[c].stype), &q)) != DTHIP_OK) break;
          rc = launch_untransform_keys(ctx, so.keys, so.key64, nullptr, nrows, plan.col[is_key[c]], plan.nsig[is_key[c]], q);
          res->col[c] = q;
          continue;
        }
        void* p = so.pay[slot[c]];
        if (p == cd[c].data) {       // nothing moved (single group / already ordered passes skipped): copy
          void* q = nullptr;
          const size_t bytes = (size_t)nrows * stype_size(cd[c].stype);
          if ((rc = result_alloc(ctx, res, bytes, &q)) != DTHIP_OK) break;
          if (hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { set_error("D2D copy failed"); rc = DTHIP_EDEVICE; break; }
          res->col[c] = q;
        } else {
          result_adopt(sc, res, p);
          res->col[c] = p;
        }
      }
      if (rc != DTHIP_OK) break;
    } else {
      if ((rc = group_core(ctx, sc, res, kd.data(), nkeys, nrows, na_pos, &plan, &g)) != DTHIP_OK) break;
      for (int c = 0; c < ncols; c++) {
        void* q = nullptr;
        if ((rc = result_alloc(ctx, res, (size_t)nrows * stype_size(cd[c].stype), &q)) != DTHIP_OK) break;
        if ((rc = launch_gather(ctx, cd[c].data, cd[c].stype, g.rowindex, nrows, q)) != DTHIP_OK) break;
        res->col[c] = q;
      }
      if (rc != DTHIP_OK) break;
      if (want_rowindex) { result_adopt(sc, res, g.rowindex); res->rowindex = g.rowindex; }
    }
    res->nrows = nrows; res->ngroups = g.ngroups; res->offsets = g.offsets;
  } while (0);
  if (rc != DTHIP_OK) { result_destroy(ctx, res); return rc; }
  *out = res;
  return DTHIP_OK;
}

// ---- V = DT[f.x <cmp> c, :]; V[:, cols, by(key)] in one call ----------------------------------------------------------
// The fused route (tlsort.hip): ONE sweep over the unfiltered rows evaluates the predicate, transforms the key and orders
// every tile's passing rows by the top digit inside the tile's own row range (sequential writes + a 16-bit directory);
// level 2 collects every bucket's rows from those segments and scatters them to their final buckets, which the final
// level orders in LDS and writes in place (the last two as in sort_stage's MSD levels).  DTHIP_NOT_APPLICABLE: the query
// does not fit (the caller then runs filter_take + groupby_rows); DTHIP_RETRY_EXACT: a guessed key range was wrong.
static int filter_rows_fused(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const dthip_col& pred, int cmp, double cf, int64_t ci,
                             const dthip_col* keys_orig, const std::vector<dthip_col>& kd, const dthip_col* cols_orig,
                             const std::vector<dthip_col>& cd, int ncols, int64_t n, int na_pos, int want_rowindex, bool speculative) {
  if (ctx->sort_path == 1 || (ctx->sort_path != 2 && n < ctx->msd_min_rows)) return DTHIP_NOT_APPLICABLE;
  if (kd.size() != 1 || (kd[0].stype != DTHIP_INT32 && kd[0].stype != DTHIP_INT64)) return DTHIP_NOT_APPLICABLE;
  if (stype_size(pred.stype) != 8) return DTHIP_NOT_APPLICABLE;
  // riding columns (requested columns that are not the key column) and the row number: <= 2, an 8-byte one first
  std::vector<int> is_key(ncols, 0), slot(ncols, -1);
  int ride[2] = {-1, -1}, nride = 0;
  for (int c = 0; c < ncols; c++) {
    if (cols_orig[c].data == keys_orig[0].data && cols_orig[c].stype == keys_orig[0].stype) { is_key[c] = 1; continue; }
    for (int c2 = 0; c2 < c; c2++) if (!is_key[c2] && cols_orig[c2].data == cols_orig[c].data && cols_orig[c2].stype == cols_orig[c].stype) slot[c] = slot[c2];
    if (slot[c] >= 0) continue;
    if (nride == 2) return DTHIP_NOT_APPLICABLE;
    const int w = stype_size(cd[c].stype);
    if (w != 4 && w != 8) return DTHIP_NOT_APPLICABLE;
    slot[c] = nride; ride[nride++] = c;
  }
  if (nride == 2 && stype_size(cd[ride[0]].stype) == 4 && stype_size(cd[ride[1]].stype) == 8) {
    std::swap(ride[0], ride[1]);
    for (int c = 0; c < ncols; c++) if (slot[c] >= 0) slot[c] ^= 1;
  }
  int npay = nride, rid_slot = -1;
  int payw[2] = {nride > 0 ? stype_size(cd[ride[0]].stype) : 0, nride > 1 ? stype_size(cd[ride[1]].stype) : 0};
  if (want_rowindex) {
    if (npay == 2) return DTHIP_NOT_APPLICABLE;
    rid_slot = npay; payw[npay++] = 4;
  }
  if (npay == 0) return DTHIP_NOT_APPLICABLE;
  if (npay == 2 && payw[0] == 4) return DTHIP_NOT_APPLICABLE;              // (4, 4): not a variant of the gather pass
  KeyPlan plan;
  DTHIP_TRY(plan_keys(ctx, sc, kd.data(), 1, n, na_pos, &plan, speculative, true));
  if (plan.nstages != 1 || plan.stage_bits[0] > 32 || plan.stage_bits[0] < 3) return DTHIP_NOT_APPLICABLE;
  const int bits = plan.stage_bits[0];
  const uint32_t tile = tl_tile_rows();
  PredArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.data = pred.data; pa.stype = pred.stype; pa.cmp = cmp; pa.cf = cf; pa.ci = ci; pa.is_mask = 0;
  // passing rows, estimated from 65536 evenly spaced rows: sizes the digits (a final bucket should hold ~msd_bucket_rows rows)
  uint32_t* d_cnt = nullptr;
  DTHIP_TRY(sc.get<uint32_t>(1, &d_cnt));
  DTHIP_CHECK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(uint32_t), ctx->stream));
  const uint32_t nsamp = (uint32_t)std::min<int64_t>(n, 65536);
  DTHIP_TRY(launch_tl_pred_sample(ctx, pa, (uint32_t)n, nsamp, d_cnt));
  uint32_t scnt = 0;
  DTHIP_TRY(read_back(ctx, &scnt, d_cnt, sizeof(scnt)));
  const int64_t est = std::max<int64_t>(2, (int64_t)((double)n * ((double)scnt + 0.5) / (double)nsamp));
  static const int rbmax = getenv("DTHIP_MSD_RBMAX") ? atoi(getenv("DTHIP_MSD_RBMAX")) : 9;
  const MsdPlan msd = msd_split(est, bits, tile, ctx->msd_bucket_rows, rbmax > 9 ? 9 : rbmax);
  if (!msd.ok) return DTHIP_NOT_APPLICABLE;
  const uint32_t nb1 = 1u << msd.s1, bins2 = 1u << msd.s2;
  // Level 2 writes tile-locally as well (default) -- decided HERE, from the estimate, because level 1's output format
  // depends on it: (the gathering final level works on windows of <= 16 whole buckets whose (bucket, digit) counts fit the
  // exchange buffer: tiny final buckets -- tests forcing the levels onto small inputs -- take the scatter form)
  int maxw_p = 4;
  for (int q = 0; q < npay; q++) maxw_p = std::max(maxw_p, payw[q]);
  const int64_t win_buckets = std::min<int64_t>(16, (int64_t)tile * maxw_p / ((int64_t)8 << msd.rb));
  static const int tl2_env = getenv("DTHIP_TL_LEVEL2") ? atoi(getenv("DTHIP_TL_LEVEL2")) : -1;
  const int tl2_opt = tl2_env >= 0 ? tl2_env : ctx->tl_level2;
  const bool tl2 = tl2_opt == 2 || (tl2_opt == 1 && (est >> (msd.s1 + msd.s2)) * win_buckets * 4 >= (int64_t)tile * 5);
  // ---- level 1: filter + key transform + top digit, tile-local ------------------------------------------------------------
  // first-level tiles: 512 threads x 16 rows, two workgroups per CU -- or (DTHIP_TL_BLOCK=1024, A/B) 16384-row tiles, whose
  // segments are twice as long for the level that gathers them, one workgroup per CU
  static const int tl_block = (getenv("DTHIP_TL_BLOCK") && atoi(getenv("DTHIP_TL_BLOCK")) == 1024) ? 1024 : 512;
  const uint32_t T1 = (uint32_t)tl_block * 16u;
  const uint32_t ntiles1 = (uint32_t)((n + T1 - 1) / T1);
  const uint32_t ntb = (ntiles1 + 63) / 64, dstride = ntb * 64;
  // Level 1 writes RECORDS {key, 4-byte riding value, 8-byte riding value} when a tile-local level 2 will gather them (one
  // 16-byte piece per row instead of three places: level 2's over-fetch 3.2x -> ~1.4x); separate arrays otherwise
  static const bool aos_env = !(getenv("DTHIP_TL_RECORDS") && atoi(getenv("DTHIP_TL_RECORDS")) == 0);
  const bool use_rec = aos_env && tl2 && tl_block == 512 && !(npay == 2 && payw[1] == 8);
  uint32_t* k1 = nullptr; uint16_t* dir = nullptr; uint16_t* dirT = nullptr; uint32_t* cc = nullptr; uint32_t* tot = nullptr;
  unsigned char* rec = nullptr;
  if (use_rec) DTHIP_TRY(sc.get<unsigned char>((size_t)n * 16, &rec));
  else DTHIP_TRY(sc.get<uint32_t>((size_t)n, &k1));
  DTHIP_TRY(sc.get<uint16_t>((size_t)ntiles1 * (nb1 + 1), &dir));
  DTHIP_TRY(sc.get<uint16_t>((size_t)(nb1 + 1) * dstride, &dirT));
  DTHIP_TRY(sc.get<uint32_t>((size_t)nb1 * ntb, &cc));
  DTHIP_TRY(sc.get<uint32_t>((size_t)nb1 + 1, &tot));
  DTHIP_CHECK_HIP(hipMemsetAsync(tot + nb1, 0, sizeof(uint32_t), ctx->stream));
  void* l1[2] = {nullptr, nullptr};
  if (!use_rec)
    for (int q = 0; q < npay; q++) { unsigned char* b = nullptr; DTHIP_TRY(sc.get<unsigned char>((size_t)n * payw[q], &b)); l1[q] = b; }
  TL1Args ta;
  memset(&ta, 0, sizeof(ta));
  ta.pred = pa; ta.key = plan.col[0]; ta.key.shift = 0;
  ta.n = (uint32_t)n; ta.block = tl_block; ta.shift = msd.rb + msd.s2; ta.bits = msd.s1;
  ta.kout = k1; ta.dir = dir; ta.rowid = rid_slot >= 0 ? static_cast<uint32_t*>(l1[rid_slot]) : nullptr;
  ta.keepx = -1; ta.pay.n = nride;
  for (int q = 0; q < nride; q++) {
    ta.pay.in[q] = cd[ride[q]].data; ta.pay.out[q] = l1[q]; ta.pay.width[q] = payw[q];
    static const bool keepx_on = !(getenv("DTHIP_TL_KEEPX") && atoi(getenv("DTHIP_TL_KEEPX")) == 0);     // (A/B: re-read the predicate column instead)
    if (keepx_on && cd[ride[q]].data == pred.data && payw[q] == 8) ta.keepx = q;
  }
  ta.bad = plan.speculative ? tot + nb1 : nullptr;
  ta.rec = rec; ta.rec4 = -1; ta.rec8 = -1;
  if (use_rec)
    for (int q = 0; q < npay; q++) {
      if (payw[q] == 8) ta.rec8 = q;
      else ta.rec4 = (q == rid_slot) ? -2 : q;
    }
  DTHIP_TRY(launch_tl_level1(ctx, ta));
  DTHIP_TRY(launch_tl_directory(ctx, dir, ntiles1, nb1, dirT, dstride, cc, ntb, tot));
  std::vector<uint32_t> htot((size_t)nb1 + 1);
  DTHIP_TRY(read_back(ctx, htot.data(), tot, htot.size() * sizeof(uint32_t)));
  if (plan.speculative && htot[nb1]) return DTHIP_RETRY_EXACT;
  int64_t npass = 0;
  for (uint32_t b = 0; b < nb1; b++) npass += htot[b];
  if (npass == 0) { DTHIP_TRY(empty_result(ctx, res)); return DTHIP_OK; }
  // ---- level 2: ragged tiles inside the level-1 buckets, planned on the host (msd_plan.hpp), rows read through the directory
  BucketGeom hg;
  memset(&hg, 0, sizeof(hg));
  {
    const uint32_t gmax = (uint32_t)ctx->num_cus * 4, nt = (uint32_t)((npass + tile - 1) / tile);
    hg.tpg = (nt + gmax - 1) / gmax; if (hg.tpg == 0) hg.tpg = 1;
  }
  std::vector<uint32_t> tdesc, gdesc, gfirst, pstart((size_t)nb1 + 1, 0);
  msd_level2_tiles(htot.data(), nb1, tile, hg.tpg, &tdesc, &gdesc, &gfirst);
  for (uint32_t b = 0; b < nb1; b++) pstart[b + 1] = pstart[b] + htot[b];
  const uint32_t ntiles2 = (uint32_t)(tdesc.size() / 4), G2 = (uint32_t)(gdesc.size() / 2);
  uint32_t* d_plan = nullptr;
  DTHIP_TRY(sc.get<uint32_t>(tdesc.size() + gdesc.size() + gfirst.size() + pstart.size() + 4, &d_plan));
  uint32_t* d_tdesc = d_plan; uint32_t* d_gdesc = d_tdesc + tdesc.size(); uint32_t* d_gfirst = d_gdesc + gdesc.size();
  uint32_t* d_pstart = d_gfirst + gfirst.size(); uint32_t* d_max = d_pstart + pstart.size();
  DTHIP_CHECK_HIP(hipMemcpyAsync(d_tdesc, tdesc.data(), tdesc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  DTHIP_CHECK_HIP(hipMemcpyAsync(d_gdesc, gdesc.data(), gdesc.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  DTHIP_CHECK_HIP(hipMemcpyAsync(d_gfirst, gfirst.data(), gfirst.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  DTHIP_CHECK_HIP(hipMemcpyAsync(d_pstart, pstart.data(), pstart.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  DTHIP_CHECK_HIP(hipMemsetAsync(d_max, 0, 4, ctx->stream));
  DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));            // (the host vectors above go out of use only at the end; pageable copies)
  // Level 2, tile-local (tl2, decided above): no histogram pass, sequential writes, the final level gathers its buckets'
  // segments.  tl_level2 = 0: level 2 scatters to exact positions (a gathering histogram pass first) and the final level
  // runs in place, as in sort_stage -- kept for A/B runs and for tiny final buckets.

  uint32_t* P = nullptr; uint32_t* gtot2 = nullptr; uint32_t* fstart = nullptr;
  DTHIP_TRY(sc.get<uint32_t>((size_t)nb1 * bins2 + 1, &fstart));
  const uint32_t nbk = nb1 * bins2;
  unsigned char* kA = nullptr; unsigned char* kB = nullptr;
  DTHIP_TRY(sc.get<unsigned char>((size_t)npass * 4, &kA));
  DTHIP_TRY(sc.get<unsigned char>((size_t)npass * 4, &kB));
  void* pb[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  for (int q = 0; q < npay; q++)
    for (int h = 0; h < 2; h++) { unsigned char* b = nullptr; DTHIP_TRY(sc.get<unsigned char>((size_t)npass * payw[q], &b)); pb[h][q] = b; }
  RadixPass rp;
  memset(&rp, 0, sizeof(rp));
  rp.kin = k1; rp.kout = kA; rp.key64 = 0; rp.n = (uint32_t)npass;
  rp.shift = msd.rb; rp.bits = msd.s2; rp.tpg = hg.tpg; rp.iota = 0;
  rp.ntiles = ntiles2; rp.tdesc = d_tdesc;
  rp.pay.n = npay;
  for (int q = 0; q < npay; q++) { rp.pay.in[q] = l1[q]; rp.pay.out[q] = pb[0][q]; rp.pay.width[q] = payw[q]; }
  rp.g_dirT = dirT; rp.g_dstride = dstride; rp.g_cc = cc; rp.g_ntb = ntb; rp.g_ntiles1 = ntiles1; rp.g_T1 = T1; rp.g_pstart = d_pstart;
  rp.g_rec = rec;
  rp.label = "tl_level2_kernel";
  uint16_t* dirT2 = nullptr; uint32_t* d_pfirst = nullptr;
  const uint32_t dstride2 = ((ntiles2 + 63) / 64) * 64;
  if (tl2) {
    uint16_t* dir2 = nullptr;
    DTHIP_TRY(sc.get<uint16_t>((size_t)(ntiles2 ? ntiles2 : 1) * (bins2 + 1), &dir2));
    DTHIP_TRY(sc.get<uint16_t>((size_t)(bins2 + 1) * dstride2, &dirT2));
    DTHIP_TRY(sc.get<uint32_t>((size_t)nb1 + 1, &d_pfirst));
    std::vector<uint32_t> pfirst((size_t)nb1 + 1, ntiles2);           // first level-2 tile of every parent bucket
    for (uint32_t t = ntiles2; t-- > 0;) pfirst[tdesc[4 * (size_t)t + 3]] = t;
    for (uint32_t b = nb1; b-- > 0;) if (htot[b] == 0) pfirst[b] = pfirst[b + 1];
    DTHIP_CHECK_HIP(hipMemcpyAsync(d_pfirst, pfirst.data(), pfirst.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    DTHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    rp.tl_dir2 = dir2;
    DTHIP_TRY(launch_radix_pass(ctx, rp));
    DTHIP_TRY(launch_tl_final_plan(ctx, dir2, ntiles2, nb1, msd.s2, d_pfirst, d_pstart, dirT2, dstride2, fstart, d_max));
  } else {
    DTHIP_TRY(sc.get<uint32_t>((size_t)(ntiles2 ? ntiles2 : 1) * bins2, &P));
    DTHIP_TRY(sc.get<uint32_t>((size_t)(G2 ? G2 : 1) * bins2, &gtot2));
    TLGatherHistArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.keys = k1; ga.shift = msd.rb; ga.bits = msd.s2; ga.tdesc = d_tdesc; ga.gdesc = d_gdesc; ga.pstart = d_pstart;
    ga.dirT = dirT; ga.dstride = dstride; ga.cc = cc; ga.ntb = ntb; ga.ntiles1 = ntiles1; ga.T1 = T1; ga.P = P; ga.gtot = gtot2;
    DTHIP_TRY(launch_tl_gather_hist(ctx, ga, G2));
    DTHIP_TRY(launch_msd_scan(ctx, gtot2, d_gfirst, d_pstart, msd.s2, nb1, (uint32_t)npass, fstart, d_max));
  }
  int maxw_w = 4;
  for (int q = 0; q < npay; q++) maxw_w = std::max(maxw_w, payw[q]);
  WindowPlan wp;
  DTHIP_TRY(plan_windows(ctx, sc, fstart, nb1, bins2, npass, d_max, tile, maxw_w, msd.rb, &wp));
  const bool windows = wp.ok;
  const uint32_t maxsize = wp.maxsize;
  if (getenv("DTHIP_MSD_DEBUG"))
    fprintf(stderr, "[dthip fused] n=%lld est=%lld pass=%lld bits=%d s1=%d s2=%d rb=%d tiles1=%u tiles2=%u groups2=%u largest bucket=%u windows=%u max buckets/window=%u records=%d -> %s\n",
            (long long)n, (long long)est, (long long)npass, bits, msd.s1, msd.s2, msd.rb, ntiles1, ntiles2, G2, maxsize, wp.nwin, wp.span, use_rec ? 1 : 0,
            windows ? "windows" : (maxsize <= tile ? "per bucket" : "not applicable"));
  // a final bucket outgrows a tile (heavy duplicates) -- or, with the gathering final level, there are no windows
  if (!(windows || (!tl2 && maxsize <= tile))) { ctx->call_stats[2]++; return DTHIP_NOT_APPLICABLE; }
  if (!tl2) { rp.P = P; rp.gpre = gtot2; DTHIP_TRY(launch_radix_pass(ctx, rp)); }
  // ---- final level: every bucket (window of buckets) ordered by the remaining bits in LDS, written to its rows of the result
  rp.g_dirT = nullptr; rp.g_cc = nullptr; rp.g_pstart = nullptr; rp.tl_dir2 = nullptr; rp.g_rec = nullptr;
  rp.kin = kA; rp.kout = kB; rp.shift = 0; rp.bits = msd.rb; rp.P = nullptr; rp.gpre = nullptr;
  rp.ntiles = nbk; rp.tdesc = nullptr; rp.bounds = fstart;
  rp.block = (maxsize <= tile / 2) ? 256 : 0;
  if (tl2) {
    rp.tdesc = d_tdesc;
    rp.g2_dirT = dirT2; rp.g2_dstride = dstride2; rp.g2_pfirst = d_pfirst; rp.g2_fstart = fstart; rp.g2_s2bits = msd.s2; rp.g2_nbk = nbk;
  }
  if (windows) {
    rp.ntiles = wp.nwin; rp.bounds = wp.bounds; rp.wfirst = wp.wfirst; rp.block = 0;
    rp.bits2 = wp.bits2; rp.wpairs = wp.pairs;
  }
  for (int q = 0; q < npay; q++) { rp.pay.in[q] = pb[0][q]; rp.pay.out[q] = pb[1][q]; }
  void* ukey_out = nullptr;
  for (int c = 0; c < ncols; c++) if (is_key[c]) { DTHIP_TRY(result_alloc(ctx, res, (size_t)npass * stype_size(kd[0].stype), &ukey_out)); break; }
  if (ukey_out) {
    const KeyColDev& kc = plan.col[0];
    rp.ukout = ukey_out; rp.uk_stype = kc.stype; rp.uk_desc = kc.desc; rp.uk_bits = bits;
    rp.uk_edge = kc.edge; rp.uk_na_repl = kc.na_repl; rp.uk_inc = kc.inc;
  }
  rp.label = "msd_final_kernel";
  DTHIP_TRY(launch_radix_pass(ctx, rp));
  Grouping g;
  if (ukey_out) DTHIP_TRY(heads_to_offsets(ctx, sc, res, ukey_out, kd[0].stype == DTHIP_INT64, nullptr, npass, &g));
  else DTHIP_TRY(heads_to_offsets(ctx, sc, res, kB, 0, nullptr, npass, &g));
  bool first_key = true;
  for (int c = 0; c < ncols; c++) {
    if (is_key[c]) {
      if (first_key) { res->col[c] = ukey_out; first_key = false; continue; }
      void* q = nullptr;
      const size_t bytes = (size_t)npass * stype_size(kd[0].stype);
      DTHIP_TRY(result_alloc(ctx, res, bytes, &q));
      DTHIP_CHECK_HIP(hipMemcpyAsync(q, ukey_out, bytes, hipMemcpyDeviceToDevice, ctx->stream));
      res->col[c] = q;
      continue;
    }
    void* p = pb[1][slot[c]];
    bool dup = false;
    for (int c2 = 0; c2 < c; c2++) if (!is_key[c2] && slot[c2] == slot[c]) dup = true;
    if (!dup) { result_adopt(sc, res, p); res->col[c] = p; continue; }
    void* q = nullptr;
    const size_t bytes = (size_t)npass * stype_size(cd[c].stype);
    DTHIP_TRY(result_alloc(ctx, res, bytes, &q));
    DTHIP_CHECK_HIP(hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    res->col[c] = q;
  }
  if (rid_slot >= 0) { result_adopt(sc, res, pb[1][rid_slot]); res->rowindex = static_cast<int32_t*>(pb[1][rid_slot]); }
  res->nrows = npass; res->ngroups = g.ngroups; res->offsets = g.offsets;
  return DTHIP_OK;
}

int dthip_filter_groupby_rows(dthip_ctx* ctx, const dthip_col* pred, int cmp, double cf, int64_t ci, const dthip_col* keys, int nkeys,
                              const dthip_col* cols, int ncols, int64_t nrows, int na_pos, int mem, int want_rowindex,
                              dthip_result** out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  CallScope call_scope(ctx);
  if (!pred || !keys || !out || nkeys < 1 || nkeys > MAX_KEYCOLS || ncols < 0 || (ncols > 0 && !cols)) { set_error("bad filter_groupby_rows arguments"); return DTHIP_EINVAL; }
  if (cmp < DTHIP_GT || cmp > DTHIP_ISNA) { set_error("bad comparison %d", cmp); return DTHIP_EINVAL; }
  if (na_pos != DTHIP_NA_FIRST && na_pos != DTHIP_NA_LAST) { set_error("na_pos %d not implemented", na_pos); return DTHIP_ENOTIMPL; }
  if (!stype_size(pred->stype) || (nrows > 0 && !pred->data)) { set_error("unsupported predicate column"); return DTHIP_ENOTIMPL; }
  dthip_result* res = new dthip_result();
  res->nkeys = nkeys;
  res->col.assign(ncols, nullptr); res->col_stype.assign(ncols, 0);
  for (int c = 0; c < ncols; c++) res->col_stype[c] = cols[c].stype;
  int rc = DTHIP_OK;
  do {
    Scratch sc(ctx);
    std::vector<dthip_col> pd, kd, cd;
    if ((rc = stage_cols(ctx, sc, pred, 1, nrows, mem, &pd)) != DTHIP_OK) break;
    if ((rc = stage_cols(ctx, sc, keys, nkeys, nrows, mem, &kd)) != DTHIP_OK) break;
    if ((rc = stage_cols(ctx, sc, cols, ncols, nrows, mem, &cd)) != DTHIP_OK) break;
    if (nrows == 0) { rc = empty_result(ctx, res); break; }
    if (mem == DTHIP_HOST) {
      // staged copies of one host column are different device buffers: the same-column tests below compare the CALLER's
      // pointers, and a staged column that is also the predicate column is mapped back onto it
      for (int c = 0; c < ncols; c++) if (cols[c].data == pred->data && cols[c].stype == pred->stype) cd[c].data = pd[0].data;
    }
    // fused route first (twice at most: a guessed key range, then the exact one)
    rc = DTHIP_NOT_APPLICABLE;
    static const bool fused_on = !(getenv("DTHIP_FILTER_ROWS_FUSED") && atoi(getenv("DTHIP_FILTER_ROWS_FUSED")) == 0);
    if (fused_on && ctx->filter_rows_fused) {
      for (int attempt = 0; attempt < 2; attempt++) {
        Scratch fs(ctx);
        rc = filter_rows_fused(ctx, fs, res, pd[0], cmp, cf, ci, keys, kd, cols, cd, ncols, nrows, na_pos, want_rowindex, attempt == 0);
        if (rc == DTHIP_ENOMEM) {             // 16 B per row of records + double buffers did not fit: the two calls need less
          rc = DTHIP_NOT_APPLICABLE; ctx->call_stats[2]++;
          dev_trim(ctx);
          break;
        }
        if (rc != DTHIP_RETRY_EXACT) break;
        ctx->call_stats[0]++;
      }
      if (rc == DTHIP_RETRY_EXACT) { set_error("filter_groupby_rows: exact key range violated"); rc = DTHIP_EDEVICE; }
      if (rc == DTHIP_OK) ctx->call_stats[3] = 4;
    }
    if (rc != DTHIP_NOT_APPLICABLE) break;
    // ---- the two-call sequence: filter (RowIndex + the view's columns in one sweep), then the rows in grouped order ------
    rc = DTHIP_OK;
    for (void* p : res->owned) dev_release(ctx, p);          // (anything a fused attempt set aside before it gave up)
    res->owned.clear();
    res->col.assign(ncols, nullptr);
    std::vector<const void*> uniq;                          // distinct source columns: keys first, then the requested columns
    std::vector<int> uniq_st, kmap(nkeys, -1), cmap(ncols, -1);
    auto add = [&](const dthip_col& orig, const dthip_col& dev) -> int {
      for (size_t u = 0; u < uniq.size(); u++) if (uniq[u] == dev.data && uniq_st[u] == orig.stype) return (int)u;
      uniq.push_back(dev.data); uniq_st.push_back(orig.stype);
      return (int)uniq.size() - 1;
    };
    for (int k = 0; k < nkeys; k++) kmap[k] = add(keys[k], kd[k]);
    for (int c = 0; c < ncols; c++) {
      bool same_key = false;
      for (int k = 0; k < nkeys; k++) if (cols[c].data == keys[k].data && cols[c].stype == keys[k].stype) { cmap[c] = kmap[k]; same_key = true; break; }
      if (!same_key) cmap[c] = add(cols[c], cd[c]);
    }
    if (uniq.size() > 8) { set_error("filter_groupby_rows: more than 8 distinct columns"); rc = DTHIP_ENOTIMPL; break; }
    PredArgs p;
    memset(&p, 0, sizeof(p));
    p.data = pd[0].data; p.stype = pred->stype; p.cmp = cmp; p.cf = cf; p.ci = ci; p.is_mask = 0;
    TakeCols tc;
    memset(&tc, 0, sizeof(tc));
    tc.n = (int)uniq.size();
    std::vector<void*> fbuf(uniq.size(), nullptr);
    for (size_t u = 0; u < uniq.size() && rc == DTHIP_OK; u++) {
      const int w = stype_size(uniq_st[u]);
      unsigned char* b = nullptr;
      if ((rc = sc.get<unsigned char>((size_t)nrows * w, &b)) != DTHIP_OK) break;
      fbuf[u] = b; tc.in[u] = uniq[u]; tc.out[u] = b; tc.width[u] = w;
    }
    if (rc != DTHIP_OK) break;
    int32_t* fri = nullptr;
    if (want_rowindex && (rc = sc.get<int32_t>((size_t)nrows, &fri)) != DTHIP_OK) break;
    int64_t npass = 0;
    if ((rc = launch_compact_take(ctx, p, nrows, fri, tc, &npass)) != DTHIP_OK) break;
    if (npass == 0) { rc = empty_result(ctx, res); break; }
    std::vector<dthip_col> fk(nkeys), fc(ncols + (want_rowindex ? 1 : 0));
    for (int k = 0; k < nkeys; k++) { fk[k] = keys[k]; fk[k].data = fbuf[kmap[k]]; }
    for (int c = 0; c < ncols; c++) { fc[c] = cols[c]; fc[c].data = fbuf[cmap[c]]; }
    if (want_rowindex) { fc[ncols].data = fri; fc[ncols].stype = DTHIP_INT32; fc[ncols].flags = 0; }
    dthip_result* tmp = nullptr;
    if ((rc = dthip_groupby_rows(ctx, fk.data(), nkeys, fc.data(), (int)fc.size(), npass, na_pos, DTHIP_DEVICE, 0, &tmp)) != DTHIP_OK) break;
    res->owned = tmp->owned; tmp->owned.clear();
    for (int c = 0; c < ncols; c++) res->col[c] = tmp->col[c];
    if (want_rowindex) res->rowindex = static_cast<int32_t*>(tmp->col[ncols]);      // the COMPOSED RowIndex (rowindex_array.cc:258-269)
    res->nrows = tmp->nrows; res->ngroups = tmp->ngroups; res->offsets = tmp->offsets;
    delete tmp;
  } while (0);
  if (rc != DTHIP_OK) { result_destroy(ctx, res); return rc; }
  *out = res;
  return DTHIP_OK;
}

int dthip_groupby_agg(dthip_ctx* ctx, const dthip_col* keys, int nkeys, const dthip_col* values, int nvalues,
                      const dthip_agg* aggs, int naggs, int64_t nrows, int na_pos, int mem, dthip_result** out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  CallScope call_scope(ctx);
  if (!keys || !out || (naggs > 0 && !aggs) || (nvalues > 0 && !values)) { set_error("null argument"); return DTHIP_EINVAL; }
  if (na_pos != DTHIP_NA_FIRST && na_pos != DTHIP_NA_LAST) { set_error("na_pos %d not implemented", na_pos); return DTHIP_ENOTIMPL; }
  for (int a = 0; a < naggs; a++) {
    if (aggs[a].op < DTHIP_SUM || aggs[a].op > DTHIP_LAST) { set_error("bad reducer op %d", aggs[a].op); return DTHIP_EINVAL; }
    if (aggs[a].op != DTHIP_COUNT0 && (aggs[a].col < 0 || aggs[a].col >= nvalues)) {
      set_error("agg %d refers to value column %d of %d", a, aggs[a].col, nvalues); return DTHIP_EINVAL;
    }
  }
  dthip_result* res = new dthip_result();
  res->nkeys = nkeys; res->naggs = naggs;
  res->agg.assign(naggs, nullptr); res->agg_stype.assign(naggs, 0);
  for (int a = 0; a < naggs; a++)
    res->agg_stype[a] = dthip_reduce_out_stype(aggs[a].op, aggs[a].op == DTHIP_COUNT0 ? DTHIP_INT64 : values[aggs[a].col].stype);
  for (int k = 0; k < nkeys && k < MAX_KEYCOLS; k++) res->key_stype[k] = keys[k].stype;
  int rc = DTHIP_OK;
  do {
    Scratch sc(ctx);
    std::vector<dthip_col> kd, vd;
    if ((rc = stage_cols(ctx, sc, keys, nkeys, nrows, mem, &kd)) != DTHIP_OK) break;
    if ((rc = stage_cols(ctx, sc, values, nvalues, nrows, mem, &vd)) != DTHIP_OK) break;
    if (nrows == 0) { rc = empty_result(ctx, res); break; }
    // value columns actually referenced
    std::vector<int> used;
    for (int a = 0; a < naggs; a++)
      if (aggs[a].op != DTHIP_COUNT0 && std::find(used.begin(), used.end(), aggs[a].col) == used.end()) used.push_back(aggs[a].col);
    bool fused = (int)used.size() <= MAX_PAYCOLS;
    for (int c : used) if (stype_size(vd[c].stype) < 4) fused = false;
    for (int a = 0; a < naggs; a++) if (aggs[a].op == DTHIP_FIRST || aggs[a].op == DTHIP_LAST) fused = false;   // need the row order
    bool f32_seq = false;       // option "f32_sum": the reference's float32 accumulation needs the rows of a group in order
    for (int a = 0; a < naggs; a++)
      if (ctx->f32_sum_ref && aggs[a].op == DTHIP_SUM && vd[aggs[a].col].stype == DTHIP_FLOAT32) { f32_seq = true; fused = false; }
    KeyPlan plan; Grouping g;
    std::vector<const void*> sorted_val(nvalues, nullptr);
    const int32_t* gather_ri = nullptr;
    if (fused) {
      // first attempt: key ranges guessed from a sample (verified by the bucketed path); if that
      // path does not apply, or the guess was wrong, plan again with the exact ranges
      int slot_bits = 0;
      bool done = false, hash_tried = false;
      // value columns whose reducers need a valid count: guess from a sample that they hold no NA; the bucketed path then
      // drops their per-row counter and verifies the guess on every row (DTHIP_RETRY_NA: aggregate again, counting)
      bool guess_nona = false;
      if (ctx->nona_guess && nrows >= ((int64_t)1 << 20)) {
        uint32_t* d_na = nullptr;
        std::vector<int> cand;
        for (int c : used)
          if (!(vd[c].flags & DTHIP_FLAG_NONA) && (acc_flags_for(aggs, naggs, c, vd[c].stype) & ACC_VCNT)) cand.push_back(c);
        if (!cand.empty()) {
          if ((rc = sc.get<uint32_t>(1, &d_na)) != DTHIP_OK) break;
          if (hipMemsetAsync(d_na, 0, sizeof(uint32_t), ctx->stream) != hipSuccess) { set_error("memset failed"); rc = DTHIP_EDEVICE; break; }
          for (int c : cand)
            if ((rc = launch_value_na_sample(ctx, vd[c].data, vd[c].stype, nrows, d_na)) != DTHIP_OK) break;
          if (rc != DTHIP_OK) break;
          uint32_t seen = 0;
          if ((rc = read_back(ctx, &seen, d_na, sizeof(seen))) != DTHIP_OK) break;
          guess_nona = seen == 0;
        }
      }
      for (int attempt = (ctx->agg_path == 1 ? 1 : 0); attempt < 2 && !done; attempt++) {
        if ((rc = plan_keys(ctx, sc, kd.data(), nkeys, nrows, na_pos, &plan, attempt == 0)) != DTHIP_OK) break;
        if (attempt == 0 && !plan.speculative) attempt = 1;      // nothing was guessed: this IS the exact plan
        if (attempt == 0 && nkeys == 1 && kd[0].stype == DTHIP_INT64 && !(kd[0].flags & DTHIP_FLAG_DESCENDING) &&
            na_pos == DTHIP_NA_FIRST && plan.stage_bits[0] >= 36 && !hash_tried) {
          // one wide int64 key: the guessed range already rules the bucketed path out (the exact range is at most a
          // bit narrower), and the hash combiner needs no range at all -- x = key - (INT64_MIN + 1) + 1 covers every
          // valid key in 64 bits -- so the exact min/max scan of the whole column (1.6 ms per 1e9 rows) is skipped
          // unless the hash path turns the query down
          KeyPlan full = plan;
          full.speculative = false;
          full.col[0].edge = (unsigned long long)(INT64_MIN + 1); full.col[0].inc = 1; full.col[0].na_repl = 0;
          full.col[0].xmax = ~0ULL; full.col[0].shift = 0;
          full.nsig[0] = 64; full.nstages = 1; full.stage_first[0] = 0; full.stage_last[0] = 0; full.stage_bits[0] = 64;
          hash_tried = true;
          rc = hash_groupby_agg(ctx, sc, res, full, kd, vd, used, aggs, naggs, nrows, na_pos);
          if (rc == DTHIP_OK) { done = true; ctx->call_stats[3] = 3; break; }
          if (rc != DTHIP_NOT_APPLICABLE) break;
          rc = DTHIP_OK;
          drop_partial_result(ctx, res);
          continue;
        }
        if (bucket_eligible(ctx, plan, vd, used, aggs, naggs, nrows, &slot_bits, guess_nona)) {
          int sb_counting = -1;
          if (guess_nona && !bucket_eligible(ctx, plan, vd, used, aggs, naggs, nrows, &sb_counting, false)) sb_counting = -1;
          rc = bucket_groupby_agg(ctx, sc, res, plan, kd, vd, used, aggs, naggs, nrows, slot_bits, guess_nona, sb_counting);
          if (rc == DTHIP_RETRY_NA) {                    // same plan once more, with valid counts
            ctx->call_stats[1]++;
            guess_nona = false; rc = DTHIP_OK; drop_partial_result(ctx, res); attempt--; continue;
          }
          if (rc == DTHIP_RETRY_EXACT && attempt == 0) { ctx->call_stats[0]++; rc = DTHIP_OK; drop_partial_result(ctx, res); continue; }
          if (rc == DTHIP_RETRY_EXACT) { set_error("bucketed aggregation: exact key range violated"); rc = DTHIP_EDEVICE; }
          done = true;
          if (rc == DTHIP_OK) ctx->call_stats[3] = 2;
        }
      }
      if (rc != DTHIP_OK || done) break;
      // sparse keys (exact plan at this point): hash combiner + merge, when its tables are large enough
      if (!hash_tried) rc = hash_groupby_agg(ctx, sc, res, plan, kd, vd, used, aggs, naggs, nrows, na_pos);
      else rc = DTHIP_NOT_APPLICABLE;
      if (rc == DTHIP_OK) { ctx->call_stats[3] = 3; break; }
      if (rc != DTHIP_NOT_APPLICABLE) break;
      rc = DTHIP_OK;
      drop_partial_result(ctx, res);                       // nothing of a half-built attempt survives
      if (plan.nstages != 1) fused = false;
    }
    if (fused) {
      // values ride through the sort; the RowIndex is never materialised
      PaySpec ps;
      ps.n = (int)used.size();
      for (int i = 0; i < ps.n; i++) { ps.in[i] = vd[used[i]].data; ps.width[i] = stype_size(vd[used[i]].stype); }
      SortOut so;
      ctx->call_stats[3] = 1;
      if ((rc = sort_stage(ctx, sc, plan, 0, nrows, nullptr, ps, &so)) != DTHIP_OK) break;
      for (int i = 0; i < ps.n; i++) sorted_val[used[i]] = so.pay[i];
      g.sorted_keys = so.keys; g.key64 = so.key64;
      if ((rc = heads_to_offsets(ctx, sc, res, so.keys, so.key64, nullptr, nrows, &g)) != DTHIP_OK) break;
    } else {
      ctx->call_stats[3] = 1;
      if ((rc = group_core(ctx, sc, res, kd.data(), nkeys, nrows, na_pos, &plan, &g)) != DTHIP_OK) break;
      for (int c : used) sorted_val[c] = vd[c].data;
      gather_ri = g.rowindex;
    }
    res->nrows = nrows; res->ngroups = g.ngroups; res->offsets = g.offsets;
    const int64_t ng = g.ngroups;
    // group-key columns: value of each key at the first row of its group
    for (int k = 0; k < nkeys; k++) {
      void* kp = nullptr;
      if ((rc = result_alloc(ctx, res, (size_t)ng * stype_size(kd[k].stype), &kp)) != DTHIP_OK) break;
      res->key[k] = kp;
      if (fused) {
        rc = launch_untransform_keys(ctx, g.sorted_keys, g.key64, g.offsets, ng, plan.col[k], plan.nsig[k], kp);
      } else {
        int32_t* firstrow = nullptr;
        if ((rc = sc.get<int32_t>((size_t)ng, &firstrow)) != DTHIP_OK) break;
        if ((rc = launch_gather(ctx, g.rowindex, DTHIP_INT32, g.offsets, ng, firstrow)) != DTHIP_OK) break;
        rc = launch_gather(ctx, kd[k].data, kd[k].stype, firstrow, ng, kp);
      }
      if (rc != DTHIP_OK) break;
    }
    if (rc != DTHIP_OK) break;
    // aggregates
    for (int a = 0; a < naggs && rc == DTHIP_OK; a++) {
      void* ap = nullptr;
      rc = result_alloc(ctx, res, (size_t)ng * stype_size(res->agg_stype[a]), &ap);
      res->agg[a] = ap;
    }
    if (rc != DTHIP_OK) break;
    for (int c : used) {
      ReduceOuts ro;
      std::vector<std::pair<int, int>> dups;   // (agg index, first agg index with same op)
      int first_of_op[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
      bool any_seg = false;
      for (int a = 0; a < naggs; a++) {
        if (aggs[a].op == DTHIP_COUNT0 || aggs[a].col != c) continue;
        if (aggs[a].op == DTHIP_FIRST || aggs[a].op == DTHIP_LAST) {
          if ((rc = launch_firstlast(ctx, sorted_val[c], vd[c].stype, gather_ri, g.offsets, ng, aggs[a].op == DTHIP_LAST,
                                     res->agg[a])) != DTHIP_OK) break;
          continue;
        }
        if (first_of_op[aggs[a].op] >= 0) { dups.push_back({a, first_of_op[aggs[a].op]}); continue; }
        first_of_op[aggs[a].op] = a;
        any_seg = true;
        if ((rc = reduce_outs_for(aggs[a].op, res->agg[a], &ro)) != DTHIP_OK) break;
      }
      if (rc != DTHIP_OK) break;
      if (any_seg)
        rc = launch_reduce(ctx, sorted_val[c], vd[c].stype, gather_ri, reinterpret_cast<const uint8_t*>(g.bitmap),
                           g.tile_first, nrows, ro, (vd[c].flags & DTHIP_FLAG_NONA) ? 1 : 0);
      if (rc != DTHIP_OK) break;
      for (auto& d : dups) {
        if (hipMemcpyAsync(res->agg[d.first], res->agg[d.second], (size_t)ng * stype_size(res->agg_stype[d.first]),
                           hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { set_error("D2D copy failed"); rc = DTHIP_EDEVICE; break; }
      }
      if (rc != DTHIP_OK) break;
    }
    if (rc != DTHIP_OK) break;
    for (int a = 0; a < naggs; a++) {
      if (aggs[a].op != DTHIP_COUNT0) continue;
      if ((rc = launch_count0(ctx, g.offsets, ng, static_cast<int64_t*>(res->agg[a]))) != DTHIP_OK) break;
    }
    if (rc != DTHIP_OK) break;
    for (int a = 0; a < naggs && f32_seq; a++) {
      if (aggs[a].op != DTHIP_SUM || vd[aggs[a].col].stype != DTHIP_FLOAT32) continue;
      if ((rc = launch_sum_f32_seq(ctx, vd[aggs[a].col].data, gather_ri, g.offsets, ng, res->agg[a])) != DTHIP_OK) break;
    }
  } while (0);
  if (rc != DTHIP_OK) { result_destroy(ctx, res); return rc; }
  *out = res;
  return DTHIP_OK;
}

int64_t dthip_result_ngroups(const dthip_result* r) { return r ? r->ngroups : -1; }
int64_t dthip_result_nrows(const dthip_result* r) { return r ? r->nrows : -1; }
const int32_t* dthip_result_rowindex(const dthip_result* r) { return r ? r->rowindex : nullptr; }
const int32_t* dthip_result_offsets(const dthip_result* r) { return r ? r->offsets : nullptr; }
const void* dthip_result_key(const dthip_result* r, int k) { return (r && k >= 0 && k < r->nkeys) ? r->key[k] : nullptr; }
const void* dthip_result_agg(const dthip_result* r, int a) { return (r && a >= 0 && a < r->naggs) ? r->agg[a] : nullptr; }
const void* dthip_result_col(const dthip_result* r, int c) { return (r && c >= 0 && c < (int)r->col.size()) ? r->col[c] : nullptr; }
int dthip_result_copy_col(dthip_ctx* ctx, const dthip_result* r, int c, void* dst, int mem) {
  if (!ctx || !r || c < 0 || c >= (int)r->col.size()) { set_error("bad column index"); return DTHIP_EINVAL; }
  if (r->nrows == 0) return DTHIP_OK;
  return copy_out(ctx, dst, r->col[c], (size_t)r->nrows * stype_size(r->col_stype[c]), mem);
}
int dthip_result_agg_stype(const dthip_result* r, int a) { return (r && a >= 0 && a < r->naggs) ? r->agg_stype[a] : 0; }

int dthip_result_copy_rowindex(dthip_ctx* ctx, const dthip_result* r, int32_t* dst, int mem) {
  if (!ctx || !r) { set_error("null argument"); return DTHIP_EINVAL; }
  if (r->nrows == 0) return DTHIP_OK;
  if (!r->rowindex) { set_error("result holds no RowIndex (want_rowindex=0 or fused aggregation)"); return DTHIP_EINVAL; }
  return copy_out(ctx, dst, r->rowindex, sizeof(int32_t) * (size_t)r->nrows, mem);
}
int dthip_result_copy_offsets(dthip_ctx* ctx, const dthip_result* r, int32_t* dst, int mem) {
  if (!ctx || !r) { set_error("null argument"); return DTHIP_EINVAL; }
  if (!r->offsets) { set_error("result holds no group offsets (option agg_offsets=0 and no count() requested)"); return DTHIP_EINVAL; }
  return copy_out(ctx, dst, r->offsets, sizeof(int32_t) * (size_t)(r->ngroups + 1), mem);
}
int dthip_result_copy_key(dthip_ctx* ctx, const dthip_result* r, int k, void* dst, int mem) {
  if (!ctx || !r || k < 0 || k >= r->nkeys) { set_error("bad key index"); return DTHIP_EINVAL; }
  if (r->ngroups == 0) return DTHIP_OK;
  if (!r->key[k]) { set_error("result holds no group-key columns (use dthip_result_group_keys)"); return DTHIP_EINVAL; }
  return copy_out(ctx, dst, r->key[k], (size_t)r->ngroups * stype_size(r->key_stype[k]), mem);
}
int dthip_result_copy_agg(dthip_ctx* ctx, const dthip_result* r, int a, void* dst, int mem) {
  if (!ctx || !r || a < 0 || a >= r->naggs) { set_error("bad agg index"); return DTHIP_EINVAL; }
  if (r->ngroups == 0) return DTHIP_OK;
  return copy_out(ctx, dst, r->agg[a], (size_t)r->ngroups * stype_size(r->agg_stype[a]), mem);
}

int dthip_result_group_keys(dthip_ctx* ctx, const dthip_result* r, const dthip_col* key, int mem, void* dst) {
  if (!ctx || !r || !key) { set_error("null argument"); return DTHIP_EINVAL; }
  if (r->ngroups == 0) return DTHIP_OK;
  if (!dst) { set_error("null argument"); return DTHIP_EINVAL; }
  if (!r->rowindex) { set_error("result holds no RowIndex"); return DTHIP_EINVAL; }
  DTHIP_CHECK_HIP(hipSetDevice(ctx->device));
  const int sz = stype_size(key->stype);
  if (!sz) { set_error("unsupported stype %d", key->stype); return DTHIP_ENOTIMPL; }
  Scratch sc(ctx);
  const void* kd = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, key->data, (size_t)r->nrows * sz, mem, &kd));
  int32_t* firstrow = nullptr;
  DTHIP_TRY(sc.get<int32_t>((size_t)r->ngroups, &firstrow));
  DTHIP_TRY(launch_gather(ctx, r->rowindex, DTHIP_INT32, r->offsets, r->ngroups, firstrow));
  if (mem == DTHIP_DEVICE) return launch_gather(ctx, kd, key->stype, firstrow, r->ngroups, dst);
  unsigned char* tmp = nullptr;
  DTHIP_TRY(sc.get<unsigned char>((size_t)r->ngroups * sz, &tmp));
  DTHIP_TRY(launch_gather(ctx, kd, key->stype, firstrow, r->ngroups, tmp));
  return copy_out(ctx, dst, tmp, (size_t)r->ngroups * sz, mem);
}

int dthip_result_free(dthip_ctx* ctx, dthip_result* r) {
  if (!ctx) return DTHIP_EINVAL;
  if (r) result_destroy(ctx, r);
  return DTHIP_OK;
}

int dthip_reduce_out_stype(int op, int st) {
  switch (op) {
    case DTHIP_SUM: case DTHIP_PROD: return st == DTHIP_FLOAT32 ? DTHIP_FLOAT32 : st == DTHIP_FLOAT64 ? DTHIP_FLOAT64 : DTHIP_INT64;
    case DTHIP_MEAN: return st == DTHIP_FLOAT32 ? DTHIP_FLOAT32 : DTHIP_FLOAT64;
    case DTHIP_MIN: case DTHIP_MAX: case DTHIP_FIRST: case DTHIP_LAST: return st;
    case DTHIP_SD: case DTHIP_MEDIAN: return st == DTHIP_FLOAT32 ? DTHIP_FLOAT32 : DTHIP_FLOAT64;   // head_reduce_unary.cc:221-229,484-491
    default: return DTHIP_INT64;
  }
}

namespace dthip {

// value column of a reducer on the device: it may be longer than nrows when read through a
// RowIndex (the caller guarantees the indices fit); host staging copies max(index)+1 rows
static int stage_value_col(dthip_ctx* ctx, Scratch& sc, const dthip_col* value, const int32_t* rowindex, int64_t nrows,
                           int mem, const void** d_val) {
  const int sz = stype_size(value->stype);
  if (!sz) { set_error("unsupported stype %d", value->stype); return DTHIP_ENOTIMPL; }
  *d_val = value->data;
  if (mem == DTHIP_HOST) {
    int64_t vrows = nrows;
    if (rowindex) { vrows = 0; for (int64_t i = 0; i < nrows; i++) if (rowindex[i] >= vrows) vrows = (int64_t)rowindex[i] + 1; }
    DTHIP_TRY(stage_in(ctx, sc, value->data, (size_t)vrows * sz, mem, d_val));
  }
  return DTHIP_OK;
}

// head bitmap (1 bit per grouped position) + per-tile "heads before this tile" from the offsets
static int heads_from_offsets(dthip_ctx* ctx, Scratch& sc, const int32_t* d_off, int64_t ngroups, int64_t nrows,
                              unsigned long long** bitmap, uint32_t** tile_counts) {
  DTHIP_TRY(sc.get<unsigned long long>((size_t)((nrows + 63) / 64) + 1, bitmap));
  const uint32_t nt = (uint32_t)((nrows + SEG_TILE - 1) / SEG_TILE);
  DTHIP_TRY(sc.get<uint32_t>((size_t)nt + 4 + nt / 8192, tile_counts));
  return launch_bitmap_from_offsets(ctx, d_off, ngroups, nrows, *bitmap, *tile_counts, *tile_counts + nt);
}

// column as float64 in grouped order (NA -> NaN); a float64 column already in order is used as is
static int grouped_f64(dthip_ctx* ctx, Scratch& sc, const void* d_val, int stype, const int32_t* d_ri, int64_t nrows,
                       const double** out) {
  if (stype == DTHIP_FLOAT64 && !d_ri) { *out = static_cast<const double*>(d_val); return DTHIP_OK; }
  double* t = nullptr;
  DTHIP_TRY(sc.get<double>((size_t)nrows, &t));
  DTHIP_TRY(launch_gather_f64(ctx, d_val, stype, d_ri, nrows, t));
  *out = t;
  return DTHIP_OK;
}

// median / nunique: the distinct (group, value) pairs in (group, value) order with their row counts =
// the fused groupby-aggregate on keys (group id, value) with count().  It takes the sort-free bucketed /
// hash paths when the composite key is dense or has few distinct values, the sort path otherwise.  The
// reference sorts every group separately (Column::sort_grouped, head_reduce_unary.cc:442-444) or fills
// a std::set per group (:379-385).
static int median_nunique(dthip_ctx* ctx, Scratch& sc, int op, const void* d_val, int stype, const int32_t* d_ri,
                          const int32_t* d_off, int64_t ngroups, int64_t nrows, void* d_out) {
  int32_t* gid = nullptr;
  DTHIP_TRY(sc.get<int32_t>((size_t)nrows, &gid));
  DTHIP_TRY(launch_ungroup(ctx, d_off, ngroups, nrows, gid));
  const void* vg = d_val;
  if (d_ri) {
    unsigned char* t = nullptr;
    DTHIP_TRY(sc.get<unsigned char>((size_t)nrows * stype_size(stype), &t));
    DTHIP_TRY(launch_gather(ctx, d_val, stype, d_ri, nrows, t));
    vg = t;
  }
  dthip_col keys[2] = {{gid, DTHIP_INT32, 0}, {vg, stype, 0}};
  if (stype_is_float(stype) && !ctx->pairs_always) {
    // mostly-distinct values: order the rows by (group, value) and read them through that order
    dthip_result* r1 = nullptr;
    DTHIP_TRY(dthip_groupby(ctx, keys, 2, nrows, DTHIP_NA_FIRST, DTHIP_DEVICE, 1, &r1));
    int rc1;
    if (op == DTHIP_MEDIAN) rc1 = launch_median_sorted(ctx, vg, stype, r1->rowindex, d_off, ngroups, d_out);
    else rc1 = launch_nunique_sorted(ctx, vg, stype, gid, r1->rowindex, r1->offsets, r1->ngroups, ngroups, static_cast<int64_t*>(d_out));
    result_destroy(ctx, r1);
    return rc1;
  }
  const dthip_agg cnt{DTHIP_COUNT0, -1};
  dthip_result* r2 = nullptr;
  const int saved_off = ctx->agg_offsets;
  ctx->agg_offsets = 1;                                         // the pairs' row offsets are needed (median)
  int rc = dthip_groupby_agg(ctx, keys, 2, nullptr, 0, &cnt, 1, nrows, DTHIP_NA_FIRST, DTHIP_DEVICE, &r2);
  ctx->agg_offsets = saved_off;
  if (rc != DTHIP_OK) return rc;
  if (op == DTHIP_MEDIAN) rc = launch_median(ctx, r2->key[1], stype, r2->offsets, r2->ngroups, d_off, ngroups, d_out);
  else rc = launch_nunique(ctx, r2->key[1], stype, static_cast<const int32_t*>(r2->key[0]), r2->ngroups, ngroups, static_cast<int64_t*>(d_out));
  result_destroy(ctx, r2);
  return rc;
}

}  // namespace dthip

int dthip_reduce(dthip_ctx* ctx, int op, const dthip_col* value, const int32_t* rowindex, const int32_t* offsets,
                 int64_t ngroups, int64_t nrows, int mem, void* out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (ngroups < 0 || ngroups > nrows) { set_error("ngroups=%lld inconsistent with nrows=%lld", (long long)ngroups, (long long)nrows); return DTHIP_EINVAL; }
  if (ngroups == 0) return DTHIP_OK;
  if (!offsets || !out) { set_error("null argument"); return DTHIP_EINVAL; }
  if (op < DTHIP_SUM || op > DTHIP_COUNTNA) { set_error("bad reducer op %d", op); return DTHIP_EINVAL; }
  if (op != DTHIP_COUNT0 && (!value || !value->data)) { set_error("reducer needs a value column"); return DTHIP_EINVAL; }
  Scratch sc(ctx);
  const void* d_off = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, offsets, sizeof(int32_t) * (size_t)(ngroups + 1), mem, &d_off));
  const int32_t* off32 = static_cast<const int32_t*>(d_off);
  const int ost = dthip_reduce_out_stype(op, op == DTHIP_COUNT0 ? DTHIP_INT64 : value->stype);
  const size_t obytes = (size_t)ngroups * stype_size(ost);
  void* d_out = out;
  if (mem == DTHIP_HOST) {
    unsigned char* t = nullptr;
    DTHIP_TRY(sc.get<unsigned char>(obytes, &t));
    d_out = t;
  }
  if (op == DTHIP_COUNT0) {
    DTHIP_TRY(launch_count0(ctx, off32, ngroups, static_cast<int64_t*>(d_out)));
  } else {
    const void* d_ri = nullptr;
    DTHIP_TRY(stage_in(ctx, sc, rowindex, sizeof(int32_t) * (size_t)nrows, mem, &d_ri));
    const int32_t* ri32 = static_cast<const int32_t*>(d_ri);
    const void* d_val = nullptr;
    DTHIP_TRY(stage_value_col(ctx, sc, value, rowindex, nrows, mem, &d_val));
    if (op == DTHIP_FIRST || op == DTHIP_LAST) {
      DTHIP_TRY(launch_firstlast(ctx, d_val, value->stype, ri32, off32, ngroups, op == DTHIP_LAST, d_out));
    } else if (op == DTHIP_MEDIAN || op == DTHIP_NUNIQUE) {
      DTHIP_TRY(median_nunique(ctx, sc, op, d_val, value->stype, ri32, off32, ngroups, nrows, d_out));
    } else if (op == DTHIP_PROD && stype_is_float(value->stype)) {
      DTHIP_TRY(launch_prod_float_seq(ctx, d_val, value->stype, ri32, off32, ngroups, d_out));
    } else {
      unsigned long long* bitmap = nullptr;
      uint32_t* tile_counts = nullptr;
      DTHIP_TRY(heads_from_offsets(ctx, sc, off32, ngroups, nrows, &bitmap, &tile_counts));
      if (op == DTHIP_SD) {
        const double* xg = nullptr;
        DTHIP_TRY(grouped_f64(ctx, sc, d_val, value->stype, ri32, nrows, &xg));
        DTHIP_TRY(launch_moments(ctx, xg, nullptr, reinterpret_cast<const uint8_t*>(bitmap), tile_counts, nrows, 0, d_out,
                                 ost == DTHIP_FLOAT32, off32, ngroups));
      } else if (op == DTHIP_PROD) {
        DTHIP_TRY(launch_reduce_prod_int(ctx, d_val, value->stype, ri32, reinterpret_cast<const uint8_t*>(bitmap), tile_counts, nrows, d_out));
      } else if (op == DTHIP_COUNTNA) {
        ReduceOuts ro;
        DTHIP_TRY(reduce_outs_for(DTHIP_COUNT, d_out, &ro));
        DTHIP_TRY(launch_reduce(ctx, d_val, value->stype, ri32, reinterpret_cast<const uint8_t*>(bitmap), tile_counts, nrows, ro));
        DTHIP_TRY(launch_countna_from_count(ctx, off32, ngroups, static_cast<int64_t*>(d_out)));
      } else {
        ReduceOuts ro;
        DTHIP_TRY(reduce_outs_for(op, d_out, &ro));
        if (op == DTHIP_SUM && value->stype == DTHIP_FLOAT32 && ctx->f32_sum_ref)
          DTHIP_TRY(launch_sum_f32_seq(ctx, d_val, ri32, off32, ngroups, d_out));
        else
          DTHIP_TRY(launch_reduce(ctx, d_val, value->stype, ri32, reinterpret_cast<const uint8_t*>(bitmap), tile_counts, nrows, ro));
      }
    }
  }
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, obytes, mem));
  return DTHIP_OK;
}

int dthip_reduce2_out_stype(int stype_a, int stype_b) {
  return (stype_a == DTHIP_FLOAT32 && stype_b == DTHIP_FLOAT32) ? DTHIP_FLOAT32 : DTHIP_FLOAT64;
}

int dthip_reduce2(dthip_ctx* ctx, int op, const dthip_col* a, const dthip_col* b, const int32_t* rowindex,
                  const int32_t* offsets, int64_t ngroups, int64_t nrows, int mem, void* out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (ngroups < 0 || ngroups > nrows) { set_error("ngroups=%lld inconsistent with nrows=%lld", (long long)ngroups, (long long)nrows); return DTHIP_EINVAL; }
  if (ngroups == 0) return DTHIP_OK;
  if (!offsets || !out || !a || !b || !a->data || !b->data) { set_error("null argument"); return DTHIP_EINVAL; }
  if (op != DTHIP_COV && op != DTHIP_CORR) { set_error("bad binary reducer op %d", op); return DTHIP_EINVAL; }
  Scratch sc(ctx);
  const void *d_off = nullptr, *d_ri = nullptr, *d_a = nullptr, *d_b = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, offsets, sizeof(int32_t) * (size_t)(ngroups + 1), mem, &d_off));
  DTHIP_TRY(stage_in(ctx, sc, rowindex, sizeof(int32_t) * (size_t)nrows, mem, &d_ri));
  DTHIP_TRY(stage_value_col(ctx, sc, a, rowindex, nrows, mem, &d_a));
  DTHIP_TRY(stage_value_col(ctx, sc, b, rowindex, nrows, mem, &d_b));
  const int ost = dthip_reduce2_out_stype(a->stype, b->stype);
  const size_t obytes = (size_t)ngroups * stype_size(ost);
  void* d_out = out;
  if (mem == DTHIP_HOST) {
    unsigned char* t = nullptr;
    DTHIP_TRY(sc.get<unsigned char>(obytes, &t));
    d_out = t;
  }
  unsigned long long* bitmap = nullptr;
  uint32_t* tile_counts = nullptr;
  DTHIP_TRY(heads_from_offsets(ctx, sc, static_cast<const int32_t*>(d_off), ngroups, nrows, &bitmap, &tile_counts));
  const double *xg = nullptr, *yg = nullptr;
  DTHIP_TRY(grouped_f64(ctx, sc, d_a, a->stype, static_cast<const int32_t*>(d_ri), nrows, &xg));
  if (d_b == d_a && b->stype == a->stype) yg = xg;
  else DTHIP_TRY(grouped_f64(ctx, sc, d_b, b->stype, static_cast<const int32_t*>(d_ri), nrows, &yg));
  DTHIP_TRY(launch_moments(ctx, xg, yg, reinterpret_cast<const uint8_t*>(bitmap), tile_counts, nrows, op == DTHIP_COV ? 1 : 2,
                           d_out, ost == DTHIP_FLOAT32, static_cast<const int32_t*>(d_off), ngroups));
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, obytes, mem));
  return DTHIP_OK;
}

int dthip_cumulate_out_stype(int op, int st) {
  if (op == DTHIP_CUMCOUNT || op == DTHIP_NGROUP) return DTHIP_INT64;
  if (op == DTHIP_CUMSUM || op == DTHIP_CUMPROD) return st == DTHIP_FLOAT32 ? DTHIP_FLOAT32 : st == DTHIP_FLOAT64 ? DTHIP_FLOAT64 : DTHIP_INT64;
  return st;
}

int dthip_cumulate(dthip_ctx* ctx, int op, const dthip_col* value, const int32_t* rowindex, const int32_t* offsets,
                   int64_t ngroups, int64_t nrows, int reverse, int mem, void* out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (ngroups < 0 || ngroups > nrows) { set_error("ngroups=%lld inconsistent with nrows=%lld", (long long)ngroups, (long long)nrows); return DTHIP_EINVAL; }
  if (nrows == 0) return DTHIP_OK;
  if (!offsets || !out || ngroups == 0) { set_error("null argument"); return DTHIP_EINVAL; }
  if (op < DTHIP_CUMSUM || op > DTHIP_NGROUP) { set_error("bad cumulative op %d", op); return DTHIP_EINVAL; }
  const bool counting = op == DTHIP_CUMCOUNT || op == DTHIP_NGROUP;
  if (!counting && (!value || !value->data)) { set_error("cumulative op needs a value column"); return DTHIP_EINVAL; }
  Scratch sc(ctx);
  const void* d_off = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, offsets, sizeof(int32_t) * (size_t)(ngroups + 1), mem, &d_off));
  const int32_t* off32 = static_cast<const int32_t*>(d_off);
  const int ost = dthip_cumulate_out_stype(op, counting ? DTHIP_INT64 : value->stype);
  const size_t obytes = (size_t)nrows * stype_size(ost);
  if (!obytes) { set_error("unsupported stype"); return DTHIP_ENOTIMPL; }
  void* d_out = out;
  if (mem == DTHIP_HOST) {
    unsigned char* t = nullptr;
    DTHIP_TRY(sc.get<unsigned char>(obytes, &t));
    d_out = t;
  }
  if (counting) {
    DTHIP_TRY(launch_cumcount(ctx, off32, ngroups, nrows, op == DTHIP_NGROUP, reverse, static_cast<int64_t*>(d_out)));
  } else {
    const void *d_ri = nullptr, *d_val = nullptr;
    DTHIP_TRY(stage_in(ctx, sc, rowindex, sizeof(int32_t) * (size_t)nrows, mem, &d_ri));
    DTHIP_TRY(stage_value_col(ctx, sc, value, rowindex, nrows, mem, &d_val));
    unsigned long long* bitmap = nullptr;
    uint32_t* tile_counts = nullptr;
    DTHIP_TRY(heads_from_offsets(ctx, sc, off32, ngroups, nrows, &bitmap, &tile_counts));
    DTHIP_TRY(launch_cumulate(ctx, d_val, value->stype, static_cast<const int32_t*>(d_ri), reinterpret_cast<const uint8_t*>(bitmap),
                              nrows, op, reverse, d_out, ost));
  }
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, obytes, mem));
  return DTHIP_OK;
}

int dthip_setop(dthip_ctx* ctx, int op, const dthip_col* stacked, const int64_t* cumsizes, int nsources, int64_t nrows,
                int mem, int32_t* out_indices, int64_t* nout) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (!nout) { set_error("null argument"); return DTHIP_EINVAL; }
  *nout = 0;
  if (nrows == 0) return DTHIP_OK;
  if (!stacked || !stacked->data || !cumsizes || !out_indices || nsources < 1) { set_error("null argument"); return DTHIP_EINVAL; }
  if (op < DTHIP_UNION || op > DTHIP_SYMDIFF) { set_error("bad set function %d", op); return DTHIP_EINVAL; }
  if (cumsizes[nsources - 1] != nrows) { set_error("cumsizes[last]=%lld != nrows=%lld", (long long)cumsizes[nsources - 1], (long long)nrows); return DTHIP_EINVAL; }
  // group the stacked column: stable, so row ids ascend inside every group
  dthip_result* g = nullptr;
  DTHIP_TRY(dthip_groupby(ctx, stacked, 1, nrows, DTHIP_NA_FIRST, mem, 1, &g));
  int rc = DTHIP_OK;
  {
    Scratch sc(ctx);
    const int64_t ng = g->ngroups;
    std::vector<int32_t> cum32((size_t)nsources);
    for (int k = 0; k < nsources; k++) cum32[(size_t)k] = (int32_t)cumsizes[k];
    const void* d_cum = nullptr;
    int8_t* mask = nullptr; int32_t *gidx = nullptr, *first = nullptr;
    int32_t* d_out = out_indices;
    int64_t cnt = 0;
    rc = stage_in(ctx, sc, cum32.data(), sizeof(int32_t) * (size_t)nsources, DTHIP_HOST, &d_cum);
    if (rc == DTHIP_OK) rc = sc.get<int8_t>((size_t)ng, &mask);
    if (rc == DTHIP_OK) rc = sc.get<int32_t>((size_t)ng, &gidx);
    if (rc == DTHIP_OK) rc = sc.get<int32_t>((size_t)ng, &first);
    if (rc == DTHIP_OK && mem == DTHIP_HOST) rc = sc.get<int32_t>((size_t)ng, &d_out);
    if (rc == DTHIP_OK) rc = launch_setop_flags(ctx, g->rowindex, g->offsets, ng, op, static_cast<const int32_t*>(d_cum), nsources, mask);
    if (rc == DTHIP_OK) { PredArgs p{mask, DTHIP_BOOL, 0, 0.0, 0, 1}; rc = launch_compact(ctx, p, ng, gidx, &cnt); }
    if (rc == DTHIP_OK && cnt) rc = launch_gather(ctx, g->offsets, DTHIP_INT32, gidx, cnt, first);
    if (rc == DTHIP_OK && cnt) rc = launch_gather(ctx, g->rowindex, DTHIP_INT32, first, cnt, d_out);
    if (rc == DTHIP_OK && cnt && mem == DTHIP_HOST) rc = copy_out(ctx, out_indices, d_out, sizeof(int32_t) * (size_t)cnt, mem);
    if (rc == DTHIP_OK) *nout = cnt;
  }
  result_destroy(ctx, g);
  return rc;
}

int dthip_join_index(dthip_ctx* ctx, const dthip_col* xkeys, const dthip_col* jkeys, int nkeys, int64_t xrows, int64_t jrows,
                     int mem, int32_t* out) {
  DTHIP_TRY(check_common(ctx, xrows, mem));
  DTHIP_TRY(check_common(ctx, jrows, mem));
  if (xrows == 0) return DTHIP_OK;
  if (!xkeys || !jkeys || !out || nkeys < 1 || nkeys > MAX_KEYCOLS) { set_error("bad join arguments (nkeys=%d)", nkeys); return DTHIP_EINVAL; }
  for (int k = 0; k < nkeys; k++) {
    if (!stype_size(xkeys[k].stype) || !stype_size(jkeys[k].stype)) { set_error("join: unsupported key stype"); return DTHIP_ENOTIMPL; }
  }
  Scratch sc(ctx);
  std::vector<dthip_col> xd, jd;
  DTHIP_TRY(stage_cols(ctx, sc, xkeys, nkeys, xrows, mem, &xd));
  DTHIP_TRY(stage_cols(ctx, sc, jkeys, nkeys, jrows, mem, &jd));
  int32_t* d_out = out;
  if (mem == DTHIP_HOST) DTHIP_TRY(sc.get<int32_t>((size_t)xrows, &d_out));
  DTHIP_TRY(launch_join_index(ctx, xd.data(), jd.data(), nkeys, xrows, jrows, d_out));
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, sizeof(int32_t) * (size_t)xrows, mem));
  return DTHIP_OK;
}

int dthip_range_bucket(dthip_ctx* ctx, const dthip_col* key, int64_t nrows, const int64_t* bounds, int nbounds, int mem,
                       int8_t* out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (nrows == 0) return DTHIP_OK;
  if (!key || !key->data || !out || (nbounds > 0 && !bounds)) { set_error("null argument"); return DTHIP_EINVAL; }
  const int sz = stype_size(key->stype);
  if (!sz) { set_error("unsupported stype %d", key->stype); return DTHIP_ENOTIMPL; }
  Scratch sc(ctx);
  const void* d_key = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, key->data, (size_t)nrows * sz, mem, &d_key));
  int8_t* d_out = out;
  if (mem == DTHIP_HOST) DTHIP_TRY(sc.get<int8_t>((size_t)nrows, &d_out));
  long long b[15];
  for (int j = 0; j < nbounds && j < 15; j++) b[j] = (long long)bounds[j];
  DTHIP_TRY(launch_range_bucket(ctx, d_key, key->stype, nrows, b, nbounds, d_out));
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, (size_t)nrows, mem));
  return DTHIP_OK;
}

int dthip_ungroup(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t nrows, int mem, int32_t* out) {
  DTHIP_TRY(check_common(ctx, nrows, mem));
  if (ngroups < 0 || ngroups > nrows) { set_error("ngroups=%lld inconsistent with nrows=%lld", (long long)ngroups, (long long)nrows); return DTHIP_EINVAL; }
  if (nrows == 0) return DTHIP_OK;
  if (!offsets || !out || ngroups == 0) { set_error("null argument"); return DTHIP_EINVAL; }
  Scratch sc(ctx);
  const void* d_off = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, offsets, sizeof(int32_t) * (size_t)(ngroups + 1), mem, &d_off));
  int32_t* d_out = out;
  if (mem == DTHIP_HOST) DTHIP_TRY(sc.get<int32_t>((size_t)nrows, &d_out));
  DTHIP_TRY(launch_ungroup(ctx, static_cast<const int32_t*>(d_off), ngroups, nrows, d_out));
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, sizeof(int32_t) * (size_t)nrows, mem));
  return DTHIP_OK;
}

static int compact_common(dthip_ctx* ctx, const PredArgs& p0, size_t elem, int64_t n, int mem, int32_t* out, int64_t* nout) {
  DTHIP_TRY(check_common(ctx, n, mem));
  if (!nout || (n > 0 && (!p0.data || !out))) { set_error("null argument"); return DTHIP_EINVAL; }
  Scratch sc(ctx);
  PredArgs p = p0;
  DTHIP_TRY(stage_in(ctx, sc, p0.data, (size_t)n * elem, mem, &p.data));
  int32_t* d_out = out;
  if (mem == DTHIP_HOST) DTHIP_TRY(sc.get<int32_t>((size_t)n, &d_out));
  DTHIP_TRY(launch_compact(ctx, p, n, d_out, nout));
  if (mem == DTHIP_HOST) DTHIP_TRY(copy_out(ctx, out, d_out, sizeof(int32_t) * (size_t)*nout, mem));
  return DTHIP_OK;
}

int dthip_bool_to_rowindex(dthip_ctx* ctx, const int8_t* mask, int64_t n, int mem, int32_t* out, int64_t* nout) {
  PredArgs p;
  memset(&p, 0, sizeof(p));
  p.data = mask; p.stype = DTHIP_BOOL; p.is_mask = 1;
  return compact_common(ctx, p, 1, n, mem, out, nout);
}

int dthip_filter_cmp(dthip_ctx* ctx, const dthip_col* col, int64_t n, int cmp, double cf, int64_t ci, int mem,
                     int32_t* out, int64_t* nout) {
  if (!col) { set_error("null column"); return DTHIP_EINVAL; }
  if (cmp < DTHIP_GT || cmp > DTHIP_ISNA) { set_error("bad comparison %d", cmp); return DTHIP_EINVAL; }
  const int sz = stype_size(col->stype);
  if (!sz) { set_error("unsupported stype %d", col->stype); return DTHIP_ENOTIMPL; }
  PredArgs p;
  memset(&p, 0, sizeof(p));
  p.data = col->data; p.stype = col->stype; p.cmp = cmp; p.cf = cf; p.ci = ci; p.is_mask = 0;
  return compact_common(ctx, p, sz, n, mem, out, nout);
}

int dthip_filter_take(dthip_ctx* ctx, const dthip_col* col, int cmp, double cf, int64_t ci, const dthip_col* cols, int ncols,
                      int64_t n, int mem, int32_t* out_rowindex, void* const* out_cols, int64_t* nout) {
  DTHIP_TRY(check_common(ctx, n, mem));
  if (!col || !nout || ncols < 0 || ncols > 8 || (ncols > 0 && (!cols || !out_cols))) { set_error("bad filter_take arguments"); return DTHIP_EINVAL; }
  if (cmp < DTHIP_GT || cmp > DTHIP_ISNA) { set_error("bad comparison %d", cmp); return DTHIP_EINVAL; }
  *nout = 0;
  if (n == 0) return DTHIP_OK;
  const int sz = stype_size(col->stype);
  if (!sz || !col->data) { set_error("unsupported predicate column"); return DTHIP_ENOTIMPL; }
  Scratch sc(ctx);
  PredArgs p;
  memset(&p, 0, sizeof(p));
  p.stype = col->stype; p.cmp = cmp; p.cf = cf; p.ci = ci; p.is_mask = 0;
  DTHIP_TRY(stage_in(ctx, sc, col->data, (size_t)n * sz, mem, &p.data));
  TakeCols tc;
  memset(&tc, 0, sizeof(tc));
  tc.n = ncols;
  std::vector<void*> d_out((size_t)ncols, nullptr);
  for (int c = 0; c < ncols; c++) {
    const int w = stype_size(cols[c].stype);
    if (!w || !cols[c].data || !out_cols[c]) { set_error("filter_take: bad column %d", c); return DTHIP_EINVAL; }
    tc.width[c] = w;
    DTHIP_TRY(stage_in(ctx, sc, cols[c].data, (size_t)n * w, mem, &tc.in[c]));
    d_out[(size_t)c] = out_cols[c];
    if (mem == DTHIP_HOST) { unsigned char* t = nullptr; DTHIP_TRY(sc.get<unsigned char>((size_t)n * w, &t)); d_out[(size_t)c] = t; }
    tc.out[c] = d_out[(size_t)c];
  }
  int32_t* d_ri = out_rowindex;
  if (out_rowindex && mem == DTHIP_HOST) DTHIP_TRY(sc.get<int32_t>((size_t)n, &d_ri));
  DTHIP_TRY(launch_compact_take(ctx, p, n, d_ri, tc, nout));
  if (mem == DTHIP_HOST) {
    if (out_rowindex) DTHIP_TRY(copy_out(ctx, out_rowindex, d_ri, sizeof(int32_t) * (size_t)*nout, mem));
    for (int c = 0; c < ncols; c++) DTHIP_TRY(copy_out(ctx, out_cols[c], d_out[(size_t)c], (size_t)*nout * tc.width[c], mem));
  }
  return DTHIP_OK;
}

int dthip_gather(dthip_ctx* ctx, const dthip_col* col, const int32_t* rowindex, int64_t nout, int mem, void* out) {
  DTHIP_TRY(check_common(ctx, nout, mem));
  if (nout == 0) return DTHIP_OK;
  if (!col || !col->data || !rowindex || !out) { set_error("null argument"); return DTHIP_EINVAL; }
  const int sz = stype_size(col->stype);
  if (!sz) { set_error("unsupported stype %d", col->stype); return DTHIP_ENOTIMPL; }
  if (mem == DTHIP_DEVICE) return launch_gather(ctx, col->data, col->stype, rowindex, nout, out);
  Scratch sc(ctx);
  int64_t vrows = 0;
  for (int64_t i = 0; i < nout; i++) if (rowindex[i] >= vrows) vrows = (int64_t)rowindex[i] + 1;
  const void* d_val = nullptr; const void* d_ri = nullptr;
  DTHIP_TRY(stage_in(ctx, sc, col->data, (size_t)vrows * sz, mem, &d_val));
  DTHIP_TRY(stage_in(ctx, sc, rowindex, sizeof(int32_t) * (size_t)nout, mem, &d_ri));
  unsigned char* d_out = nullptr;
  DTHIP_TRY(sc.get<unsigned char>((size_t)nout * sz, &d_out));
  DTHIP_TRY(launch_gather(ctx, d_val, col->stype, static_cast<const int32_t*>(d_ri), nout, d_out));
  return copy_out(ctx, out, d_out, (size_t)nout * sz, mem);
}

}  // extern "C"
