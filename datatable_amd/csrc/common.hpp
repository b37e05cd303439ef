// common.hpp -- context, error handling, caching device allocator and launch
// accounting shared by the libdthip translation units.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/dthip.h"

namespace dthip {

void set_error(const char* fmt, ...);

#define DTHIP_CHECK_HIP(expr)                                                        \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      ::dthip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                         __FILE__, __LINE__);                                        \
      return (_e == hipErrorOutOfMemory) ? DTHIP_ENOMEM : DTHIP_EDEVICE;             \
    }                                                                                \
  } while (0)

#define DTHIP_TRY(expr)            \
  do {                             \
    int _rc = (expr);              \
    if (_rc != DTHIP_OK) return _rc; \
  } while (0)

inline int stype_size(int st) {
  switch (st) {
    case DTHIP_BOOL: case DTHIP_INT8: return 1;
    case DTHIP_INT16: return 2;
    case DTHIP_INT32: case DTHIP_FLOAT32: return 4;
    case DTHIP_INT64: case DTHIP_FLOAT64: return 8;
    default: return 0;
  }
}
inline bool stype_is_float(int st) { return st == DTHIP_FLOAT32 || st == DTHIP_FLOAT64; }

struct ProfRec { const char* name; hipEvent_t a, b; };
struct ProfAcc { double ms = 0; int64_t n = 0; };

}  // namespace dthip

struct dthip_comm;
struct dthip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // caching allocator: blocks are reused on the same stream, so returning a
  // block to the cache while kernels that use it are still queued is safe.
  std::multimap<size_t, void*> cache;
  std::unordered_map<void*, size_t> live;
  size_t cached_bytes = 0;
  // pinned host scratch for small synchronous read-backs
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  // timers / per-kernel accounting
  hipEvent_t t0 = nullptr, t1 = nullptr;
  bool prof = false;
  std::vector<dthip::ProfRec> pending;
  std::vector<hipEvent_t> event_pool;
  std::map<std::string, dthip::ProfAcc> acc;
  int num_cus = 256;
  // kernels whose dynamic-LDS limit was raised on THIS context's device (hipFuncSetAttribute is per device: a
  // process-wide flag would leave the devices 1..7 of a local communicator at the 64 KB default)
  std::unordered_set<const void*> lds_raised;
  int agg_path = 0;          // 0 auto, 1 sort path, 2 bucket path whenever eligible
  int bucket_variant = 0;    // partition tile geometry (experiments)
  int64_t spec_min_rows = 1 << 23;   // key ranges are guessed from a sample only at or above this many rows
  int hash_mode = 0;         // hash combiner for sparse keys: 0 decide from a distinct-count estimate, 1 never, 2 whenever it fits, 3 = 2 with the exact-position partition
  int pairs_always = 0;      // median / nunique of float columns through the distinct (group, value) pairs too (tests)
  int join_table = 1;        // dthip_join_index: direct key->row table for a dense single integer key (0: always search)
  bool in_merge = false;     // internal: the merge of partial groups must not take the hash path again
  int cluster_mode = 0;      // clustered-key kernel variants: 0 decide from a sample, 1 never, 2 always
  int agg_offsets = 1;       // dthip_groupby_agg results carry group offsets (= sizes) even when no count() asks for them
  int f32_sum_ref = 0;       // 1: sum(float32) accumulates in float32, row by row, like the reference (slow path)
  int small_path = 2;        // slot tables of <= SMALL_SLOTS slots (one table): fused init / group-list kernels (BASELINE C1 is launch-bound:
                             // 0.116 ms general sequence, 0.087 with 1, 0.085 with 2 = counts written to mapped host memory)
  uint32_t* host_words = nullptr;   // 16 words of mapped pinned memory the last kernel of a small call writes its counts into
  uint32_t* host_words_dev = nullptr;
  int nona_guess = 1;        // bucketed aggregation: value columns whose sample shows no NA are aggregated without a valid counter
                             // (one DS atomic per row and column less), every row verified; a wrong guess aggregates again
  int filter_path = 1;       // row filters: 1 count pass + write pass (default: 1.3 + 5.8 ms per 1e9 float64 rows with two 8-byte columns
                             // taken); 0 ONE pass, tile offsets by decoupled look-back (measured 8.6-9.2 ms: the look-back chain costs
                             // more than the second read of the predicate column; kept selectable)
  int tl_level2 = 1;         // fused filter -> group-rows route: 1 (default) the second level writes tile-locally too when the final buckets
                             // are expected to fill windows, 0 never (exact-position scatter after a gathering histogram), 2 always (tests)
  int filter_rows_fused = 1; // dthip_filter_groupby_rows: 1 (default) the fused route of tlsort.hip where it applies, 0 always filter_take + groupby_rows
  int sort_path = 0;         // 0: MSD levels (two scatter levels + final buckets ordered in LDS) from msd_min_rows rows on where their preconditions hold, else LSD passes; 1: LSD passes only; 2: MSD levels whenever their preconditions hold, whatever msd_min_rows says (A/B runs, tests)
  int64_t msd_min_rows = 1 << 26;    // below this the LSD passes are quick enough (and the final buckets would be tiny)
  int msd_bucket_rows = 2048;        // target size of a final bucket (sorted in LDS: at most one radix tile)
  // what the LAST query entry point did on this context (dthip_last_call_stats): [0] sweeps repeated because a key range
  // guessed from a sample was wrong, [1] because a value column guessed NA-free held an NA, [2] a route given up after it had
  // started (fused filter route -> two calls, hash tables full -> sort path), [3] the path that produced the result
  // (1 sort, 2 bucketed, 3 hash combiner, 4 fused filter route, 5 small table)
  int64_t call_stats[5] = {0, 0, 0, 0, 0};     // [4] rows with keys outside a guessed range that were LISTED and grouped apart (no second sweep)
  int call_depth = 0;        // query entry points call each other (fused route -> two calls, hash combiner -> merge): only the outermost resets
  // multi-GPU (comm.hip): the communicator this context is a rank of
  struct dthip_comm* comm = nullptr;
  int comm_rank = 0;
  // guard-page mode (env DTHIP_GUARD / option "guard"; a debugging flavour like the reference's ASan build,
  // ci/ext.py:318-327): every device buffer is its own virtual-memory mapping with unmapped pages on both sides and
  // its END (1) or START (2) flush against them, nothing is cached, every launch is synchronised, so a kernel that
  // reads or writes one 16-byte piece outside a buffer faults at once and the SIGABRT handler names it
  int guard = 0;
  struct GuardBlock { void* va; size_t va_bytes; void* map; size_t map_bytes; hipMemGenericAllocationHandle_t h; };
  std::unordered_map<void*, GuardBlock> guarded;
  std::vector<GuardBlock> guard_limbo, guard_vas;    // released but still mapped / unmapped but still reserved
  int64_t guard_allocs = 0, guard_launches = 0;
};

// device-resident result of a groupby (dthip.h: dthip_result); every buffer in `owned` goes back to the context's
// cache on dthip_result_free
struct dthip_result {
  int64_t nrows = 0, ngroups = 0;
  int32_t* rowindex = nullptr;
  int32_t* offsets = nullptr;
  int nkeys = 0;
  void* key[8] = {};
  int key_stype[8] = {};
  int naggs = 0;
  std::vector<void*> agg;
  std::vector<int> agg_stype;
  std::vector<void*> col;        // dthip_groupby_rows: columns permuted into grouped order
  std::vector<int> col_stype;
  std::vector<void*> owned;
};

namespace dthip {

int result_alloc(dthip_ctx* ctx, dthip_result* r, size_t bytes, void** out);
void result_destroy(dthip_ctx* ctx, dthip_result* r);

int dev_alloc(dthip_ctx* ctx, size_t bytes, void** out);
void dev_release(dthip_ctx* ctx, void* p);   // back to the cache
int dev_trim(dthip_ctx* ctx);                // cache -> hipFree
int prof_flush(dthip_ctx* ctx);
hipEvent_t prof_event(dthip_ctx* ctx);
void guard_before_launch(dthip_ctx* ctx, const char* kname);
int guard_after_launch(dthip_ctx* ctx, const char* kname);

// Scoped set of temporary device buffers returned to the cache on exit.
struct Scratch {
  dthip_ctx* ctx;
  std::vector<void*> bufs;
  explicit Scratch(dthip_ctx* c) : ctx(c) {}
  ~Scratch() { for (void* p : bufs) dev_release(ctx, p); }
  template <typename T>
  int get(size_t count, T** out) {
    void* p = nullptr;
    int rc = dev_alloc(ctx, (count ? count : 1) * sizeof(T), &p);
    if (rc != DTHIP_OK) return rc;
    bufs.push_back(p);
    *out = static_cast<T*>(p);
    return DTHIP_OK;
  }
  // give one buffer back early (stream-ordered, like the destructor)
  void release(void* p) {
    for (auto& b : bufs) if (b == p) { b = bufs.back(); bufs.pop_back(); dev_release(ctx, p); return; }
  }
  // hand a buffer over to a longer-lived owner
  void disown(void* p) {
    for (auto& b : bufs) if (b == p) { b = bufs.back(); bufs.pop_back(); return; }
  }
};

// Launch with optional per-kernel event accounting.
#define DTHIP_LAUNCH(ctx, kname, kernel, grid, block, shmem, ...)                           \
  do {                                                                                      \
    hipEvent_t _ea = nullptr, _eb = nullptr;                                                \
    if ((ctx)->guard) ::dthip::guard_before_launch(ctx, kname);                             \
    if ((ctx)->prof) { _ea = ::dthip::prof_event(ctx); _eb = ::dthip::prof_event(ctx);      \
                       (void)hipEventRecord(_ea, (ctx)->stream); }                                \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shmem), (ctx)->stream, __VA_ARGS__); \
    if ((ctx)->prof) { (void)hipEventRecord(_eb, (ctx)->stream);                                  \
                       (ctx)->pending.push_back({kname, _ea, _eb});                         \
                       if ((ctx)->pending.size() > 2048) ::dthip::prof_flush(ctx); }        \
    hipError_t _le = hipGetLastError();                                                     \
    if (_le != hipSuccess) {                                                                \
      ::dthip::set_error("launch of %s failed: %s", kname, hipGetErrorString(_le));         \
      return DTHIP_EDEVICE;                                                                 \
    }                                                                                       \
    if ((ctx)->guard) { int _grc = ::dthip::guard_after_launch(ctx, kname);                 \
                        if (_grc != DTHIP_OK) return _grc; }                                \
  } while (0)

// raise a kernel's dynamic-LDS limit once per (context = device, kernel)
inline int ensure_dyn_lds(dthip_ctx* ctx, const void* kfn, int bytes) {
  if (ctx->lds_raised.count(kfn)) return DTHIP_OK;
  DTHIP_CHECK_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  ctx->lds_raised.insert(kfn);
  return DTHIP_OK;
}

// small synchronous device->host read-back through pinned memory
int read_back(dthip_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);

// ---- kernels' host entry points (one per .hip file) ------------------------

// stats.hip: min / max / valid-count of an integer column (NumericStats::compute_minmax)
struct MinMax { long long mn, mx, nvalid; };
int launch_minmax(dthip_ctx* ctx, const void* data, int stype, int64_t n, MinMax* d_out);
// the same over nsamp evenly spaced 16-byte pieces of the column (plus its first and last rows):
// a GUESS of the key range that the histogram pass of the bucketed aggregation verifies
int launch_minmax_sample(dthip_ctx* ctx, const void* data, int stype, int64_t n, uint32_t nsamp, MinMax* d_out);

// radix.hip
constexpr int MAX_KEYCOLS = 8;
constexpr int MAX_PASSES = 10;
constexpr int HIST_STRIDE = 512;
// internal stype of a key column descriptor: an int64 column whose "transformed key" is a 24-bit HASH of the raw value
// (hash combiner, one raw int64 key: the partition kernels hash on the fly instead of reading a pseudo-key array)
constexpr int DTHIP_KEY_HASH64 = 100;
struct KeyColDev {
  const void* data;
  int stype;
  int desc;
  unsigned long long edge;     // min (ascending) or max (descending), as the column's unsigned image
  unsigned long long na_repl;  // transformed value of NA
  unsigned long long inc;      // 1 when NA is first, else 0
  unsigned long long xmax;     // valid (non-NA) keys transform to [inc, inc + xmax]; a non-NA key outside: the key range was a guess and is wrong
  int shift;                   // bit position of this key inside the packed key
};
struct XformArgs {
  KeyColDev cols[MAX_KEYCOLS];
  int ncols;
  uint32_t n;
  const int32_t* order;  // nullable: read column rows through this ordering
  void* out;             // uint32_t[n] or uint64_t[n]
  int out64;
  uint32_t* hist;        // [npass][HIST_STRIDE], zeroed by the caller
  uint32_t* bad;         // nullable: set when a transformed key exceeds its column's xmax (the key range was a guess)
  int npass;
  int pshift[MAX_PASSES];
  int pbits[MAX_PASSES];
};
int launch_xform_hist(dthip_ctx* ctx, const XformArgs& a);
int launch_hist_scan(dthip_ctx* ctx, const uint32_t* hist, uint32_t* base, int npass);

constexpr int MAX_PAYCOLS = 8;
struct PayCols {
  int n;
  const void* in[MAX_PAYCOLS];
  void* out[MAX_PAYCOLS];
  int width[MAX_PAYCOLS];  // 4 or 8 bytes
};
struct RadixPass {
  const void* kin; void* kout; int key64;
  uint32_t n; int shift; int bits;
  const uint32_t* P;             // [ntiles][1<<bits] from launch_radix_tile_hist
  const uint32_t* gpre;          // [G][1<<bits] global run starts from launch_bucket_gscan(phase 1)
  uint32_t tpg;
  int iota;                      // payload column 0 is the row number (not loaded)
  PayCols pay;
  // MSD levels: ragged tiles {first row, rows, group, -} of a level inside the buckets of the level above, or the
  // final level's buckets as boundaries (tile t = rows [bounds[t], bounds[t+1]), sorted in LDS, written in place)
  uint32_t ntiles;               // 0: regular tiles over n rows
  const uint32_t* tdesc;
  const uint32_t* bounds;
  const char* label;             // nullable: name of this launch in the per-kernel accounting (dthip_profile_*)
  int block;                     // 0 / 256: workgroup size of the final MSD level (256: buckets of <= 4096 rows)
  const uint32_t* wfirst;        // final MSD level over windows of whole buckets (two rounds in LDS): first bucket of every window
  int bits2;                     //   and the bits of the bucket number inside a window
  int wpairs;                    //   bounds = (start, end) pairs of greedily packed windows (launch_msd_windows_greedy)
  // last pass only: write the original values of ONE int32 / int64 key column here instead of the packed keys
  void* ukout; int uk_stype; int uk_desc; int uk_bits;
  unsigned long long uk_edge, uk_na_repl, uk_inc;
  // gather mode (tlsort.hip; g_dirT != null): the level ABOVE wrote its rows tile-locally (rows of digit b of tile t at
  // [t * g_T1 + dirT[b][t], t * g_T1 + dirT[b + 1][t]) of kin / pay.in), so tile t of THIS level -- tdesc[4 t ..] = {first
  // row of the bucket-ordered sequence, rows, histogram group, bucket} -- collects its rows from those segments
  // (radix_dev.hpp tl_build_src).  Needs tdesc, 4-byte keys, and every payload column prefetched (<= 2, widths 8 / 4).
  const uint16_t* g_dirT; uint32_t g_dstride; const uint32_t* g_cc; uint32_t g_ntb, g_ntiles1, g_T1;
  const uint32_t* g_pstart;      // [buckets]: first row of every bucket in the bucket-ordered sequence
  const void* g_rec;             // the level above wrote 16-byte RECORDS (TL1Args::rec) instead of kin / pay.in: one gather per row
  // ... and may write ITS rows tile-locally as well (tl_dir2 != null; no P / gpre, no histogram pass): tile t's rows, ordered
  // by the level's digit, over the tile's own rows [tdesc[4 t], + rows) of kout / pay.out, tl_dir2[t][d] = first place of
  // digit d inside the tile ([bins] = the tile's rows)
  uint16_t* tl_dir2;
  // final level over windows reading such a level (g2_dirT != null): bucket c = (parent b1 = c >> g2_s2bits, digit d2) is
  // the concatenation, over the parent's tiles [g2_pfirst[b1], g2_pfirst[b1 + 1]), of the segments
  // [tdesc[4 p] + dirT[d2][p], tdesc[4 p] + dirT[d2 + 1][p]); it goes to rows [g2_fstart[c], g2_fstart[c + 1]) of the output
  const uint16_t* g2_dirT; uint32_t g2_dstride; const uint32_t* g2_pfirst; const uint32_t* g2_fstart; int g2_s2bits; uint32_t g2_nbk;
};
uint32_t radix_tile_items(int key64, int maxpaywidth);
int launch_radix_tile_hist(dthip_ctx* ctx, const void* keys, int key64, uint32_t n, int shift, int bits,
                           uint32_t ntiles, uint32_t tpg, uint32_t G, uint32_t* P, uint32_t* gtot,
                           const uint32_t* tdesc = nullptr, const uint32_t* gdesc = nullptr);
int launch_msd_windows_greedy(dthip_ctx* ctx, const uint32_t* fstart, uint32_t nparents, uint32_t pb, uint32_t tile, uint32_t maxspan,
                              uint32_t nwmax, uint32_t* wbounds, uint32_t* wfirst, uint32_t* info);
int launch_msd_scan(dthip_ctx* ctx, uint32_t* gtot, const uint32_t* gfirst, const uint32_t* pstart, int bits, uint32_t nb,
                    uint32_t n, uint32_t* fstart, uint32_t* maxsize);
int launch_radix_pass(dthip_ctx* ctx, const RadixPass& p);

// group.hip: run heads of a sorted key sequence -> offsets, head bitmap, tile head counts
// (tiles of 2048 positions, shared with reduce.hip)
constexpr int SEG_TILE = 2048;
int launch_scan_tiles(dthip_ctx* ctx, uint32_t* counts, uint32_t m, uint32_t* total);
// tile_counts[ntiles] <- exclusive scan of heads per tile; bitmap (nullable) <- 1 bit per position;
// d_total <- ngroups; ngroups_host (nullable) <- synchronous read-back of d_total
int launch_count_heads(dthip_ctx* ctx, const void* keys, int key64, const uint8_t* heads, int64_t n,
                       uint32_t* tile_counts, unsigned long long* bitmap, uint32_t* d_total,
                       int64_t* ngroups_host);
int launch_count_heads_presorted(dthip_ctx* ctx, const void* keys, int stype, int64_t n, uint32_t* tile_counts,
                                 unsigned long long* bitmap, uint32_t* d_flags, int64_t* ngroups_host, bool* sorted_host);
int launch_sorted_sample(dthip_ctx* ctx, const void* keys, int stype, int64_t n, uint32_t* d_flag, bool* maybe_sorted);
int launch_write_offsets(dthip_ctx* ctx, const unsigned long long* bitmap, int64_t n,
                         const uint32_t* tile_base, int64_t ngroups, int32_t* offsets);
int launch_mark_heads(dthip_ctx* ctx, const void* keys, int key64, int64_t n, uint8_t* heads);
int launch_bitmap_from_offsets(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t n,
                               unsigned long long* bitmap, uint32_t* tile_counts, uint32_t* d_total);
int launch_iota(dthip_ctx* ctx, int32_t* out, int64_t n);
int launch_ungroup(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t n, int32_t* out);
int launch_offsets_drop_rows(dthip_ctx* ctx, const int32_t* in, int64_t ngroups, int32_t skip, int32_t* out, int32_t* g0_out);
int launch_untransform_keys(dthip_ctx* ctx, const void* sorted_keys, int key64, const int32_t* offsets,
                            int64_t ngroups, const KeyColDev& col, int bits, void* out);

// bucket.hip: dense-key-range aggregation without a sort.  Rows are partitioned ONCE by the
// top `d` bits of the packed transformed key into F = 2^d buckets (any order inside a
// bucket), then every bucket is aggregated into an LDS-resident table of S = 2^r slots
// addressed by the low r bits; non-empty slots, in slot order, ARE the groups in key order.
struct KeyXform { KeyColDev cols[MAX_KEYCOLS]; int ncols; };
struct BucketGeom {
  int B, r, d;            // significant key bits, slot bits, bucket bits
  uint32_t F, S;          // 1 << d, 1 << r
  uint32_t block, items;  // partition / histogram tile geometry
  uint32_t tile, ntiles;  // rows per tile, number of tiles
  uint32_t tpg, G;        // tiles per histogram group, number of groups
  int km;                 // key load mode: 0 generic, 1 one aligned int64 column, 2 one aligned int32 column
};
struct WorkItem { uint32_t bucket, begin, end, single; };
enum { ACC_CNT = 1, ACC_SUM = 2, ACC_MIN = 4, ACC_MAX = 8, ACC_VCNT = 16, ACC_FSUM = 32, ACC_PRES = 64,
       ACC_NONA = 128 /* DTHIP_FLAG_NONA: every bit pattern of the value column is a value */,
       ACC_CHKNA = 256 /* the column was GUESSED to hold no NA (no valid count kept in LDS): an NA row is skipped and counted in
                          AggTable::nacnt by a global atomic -- the valid count of a group is its size minus that */ };
// dense accumulator arrays of F*S slots (slot index == transformed key)
struct AggTable {
  uint32_t* cnt = nullptr;              // rows per slot
  unsigned long long* sum = nullptr;    // int64 sum (integer values) or float64 sum bits (float values)
  unsigned long long* mn = nullptr;     // order-preserving unsigned image of the minimum
  unsigned long long* mx = nullptr;
  uint32_t* vcnt = nullptr;             // non-NA rows per slot
  double* fsum = nullptr;               // float64 sum of integer values (mean)
  uint32_t* pres = nullptr;             // 1 bit per slot: some row has this key (when row counts are not wanted)
  uint32_t* nacnt = nullptr;            // ACC_CHKNA: NA rows per slot of a column GUESSED NA-free (global atomics, rare): valid = cnt - nacnt
};
void bucket_geometry(dthip_ctx* ctx, int64_t n, int B, int r, int km, BucketGeom* g);
bool bucket_tl16_geometry(dthip_ctx* ctx, int64_t n, int maxw, BucketGeom* g);
// *bad is set when a row's transformed key exceeds its column's xmax (such rows are counted as key 0)
int launch_bucket_hist(dthip_ctx* ctx, const KeyXform& kx, int64_t n, const BucketGeom& g, uint32_t* P, uint32_t* gtot,
                       uint32_t* bad, bool clustered);
// *clustered <- neighbouring rows mostly share a bucket (65536 sampled row pairs; synchronises); flag2: 2 words of scratch
int launch_bucket_cluster_sample(dthip_ctx* ctx, const KeyXform& kx, int64_t n, int r, uint32_t* flag2, bool* clustered,
                                 uint32_t F = 0, bool* even = nullptr);
// phase 0: tot[F] <- bucket sizes; phase 1: gtot <- bbase + exclusive prefix over groups (in place)
int launch_bucket_gscan(dthip_ctx* ctx, const BucketGeom& g, uint32_t* gtot, uint32_t* tot, const uint32_t* bbase, int phase);
// tot (nullable: one bucket of n_raw rows) -> bbase[F+1], work items of <= M rows, *nitems
// fills: (small path) buffers the plan kernel also initialises -- saves one launch per memset on latency-bound calls
struct FillList { int n; uint32_t* p[12]; uint32_t words[12]; uint32_t val[12]; };
// the general sequence: every table of a query initialised by ONE launch (BASELINE C2 issued 22 runtime memsets per query:
// 0.1 ms of its 2.5; round 6).  16-byte stores; buffers come from the allocator (256-byte aligned), sizes are whole words
struct BigFill { int n; uint32_t* p[24]; unsigned long long words[24]; uint32_t val[24]; };
int launch_fill_list(dthip_ctx* ctx, const BigFill& f);
int launch_bucket_plan(dthip_ctx* ctx, const uint32_t* tot, uint32_t F, uint32_t n_raw, uint32_t M,
                       uint32_t* bbase, WorkItem* items, uint32_t* nitems, const FillList* fills = nullptr);
// small slot tables (<= SMALL_SLOTS): non-empty slots -> idx, their row counts -> offsets (exclusive scan, total appended),
// out[0] = groups, out[1] = *bad -- ONE single-workgroup launch for what launch_compact + gather + scan do in seven
constexpr uint32_t SMALL_SLOTS = 8192;
struct SmallGroupsArgs { const uint32_t* cnt; int bits; uint32_t nslots; int32_t* idx; uint32_t* off; const uint32_t* bad; uint32_t* out; };
int launch_small_groups(dthip_ctx* ctx, const SmallGroupsArgs& a);
int launch_bucket_partition(dthip_ctx* ctx, const KeyXform& kx, int64_t n, const BucketGeom& g, const uint32_t* P,
                            const uint32_t* gpre, uint16_t* kout, const PayCols& pay, bool clustered, uint16_t* dir = nullptr,
                            uint32_t* bad = nullptr, uint32_t* ovf_rows = nullptr, uint32_t* ovf_n = nullptr, uint32_t ovf_cap = 0);
// out[pos + i] = in[i] + delta for i < count: one piece of a spliced offsets array
int launch_offsets_piece(dthip_ctx* ctx, const int32_t* in, int64_t count, int32_t delta, int32_t* out);
// tile-local layout (no histogram pass): directory transpose + bucket totals + work list, and the aggregation over it
int launch_dir_prepare(dthip_ctx* ctx, const uint16_t* dir, uint32_t ntiles, uint32_t F, uint16_t* dirT, uint32_t dstride,
                       uint32_t* tot, uint32_t M, WorkItem* items, uint32_t* nitems, uint32_t* nitems2 = nullptr);
struct TableAggSegArgs {
  const WorkItem* items; const uint32_t* nitems; const uint32_t* nitems2; uint32_t max_items;      // nitems2: seg_plan_kernel's two lists
  const uint16_t* kpart; const void* val; int vstype;
  const uint16_t* dirT; uint32_t dstride; uint32_t tile_rows;
  uint32_t S; int flags; AggTable tab;
  uint32_t* bad;
  bool all_long;            // few buckets: every segment is long (the instance without the short mode)
};
int launch_table_agg_seg(dthip_ctx* ctx, const TableAggSegArgs& a);
int launch_value_na_sample(dthip_ctx* ctx, const void* const* data, const int* stype, int ncols, int64_t n, uint32_t* flag);
struct TableAggArgs {
  const WorkItem* items; const uint32_t* nitems; uint32_t max_items;
  int src;                    // 0: kpart + val of the partitioned rows, 1: raw rows (kx + val)
  const uint16_t* kpart;      // slot keys of the partitioned rows (src 0)
  KeyXform kx;
  const void* val; int vstype; // value column in the same row order as the keys (null: row counts only)
  uint32_t S; int flags;
  AggTable tab;
  uint32_t* bad;              // raw mode: set when a key exceeds its column's xmax
  bool clustered;             // waves mostly address one slot: reduce in registers first
};
int launch_table_agg(dthip_ctx* ctx, const TableAggArgs& a);
size_t table_agg_slot_bytes(int flags);
size_t table_agg_lds_bytes(int flags, uint32_t S);
struct TableFinArgs {
  const int32_t* idx; uint32_t ng; AggTable tab; int vstype;
  void* o_sum; void* o_mean; void* o_min; void* o_max; int64_t* o_count;
};
int launch_table_finalize(dthip_ctx* ctx, const TableFinArgs& a);

// hash combiner (bucket.hip): partial groups of sparse keys
int launch_hash_xform(dthip_ctx* ctx, const KeyXform& kx, int64_t n, unsigned long long* xs, int32_t* pk);
int launch_hash_pk_raw(dthip_ctx* ctx, const void* key, int64_t n, int32_t* pk);
struct HashAggArgs {
  const WorkItem* items; const uint32_t* nitems; uint32_t max_items;
  const unsigned long long* xs;      // packed transformed keys of the partitioned rows
  const void* val; int vstype;       // value column of the partitioned rows (null: counts only)
  uint32_t C; int flags;             // table entries per workgroup, accumulators
  unsigned long long* o_key; AggTable o_tab; uint32_t* out_n; uint32_t out_cap; uint32_t* overflow;
};
size_t hash_agg_entry_bytes(int flags);
size_t hash_agg_queue_bytes();      // LDS the waves' pending-row queues take in front of the table
int launch_hash_agg(dthip_ctx* ctx, const HashAggArgs& a);
// the same over a tile-local partition: items = (bucket, tile range), segments through the transposed directory
int launch_hash_agg_seg(dthip_ctx* ctx, const HashAggArgs& a, const uint16_t* dirT, uint32_t dstride, uint32_t tile_rows, bool rec);
// tile-local hash partition of one int64 key + one 8-byte value column into 16-byte records (16384-row tiles) + directory
int launch_hash_partition_rec(dthip_ctx* ctx, const void* key, const void* val, int64_t n, int r, uint32_t F, uint32_t ntiles,
                              void* rec, uint16_t* dir);
struct PartialColsArgs {
  AggTable tab; uint32_t n; int vstype;
  unsigned long long* o_sum; double* o_fsum; void* o_min; void* o_max; int64_t* o_vcnt; int64_t* o_cnt;
};
int launch_partial_columns(dthip_ctx* ctx, const PartialColsArgs& a);
int launch_mean_div(dthip_ctx* ctx, const double* sum, const long long* cnt, int64_t n, void* out, int out_f32);
int launch_cast_f64_f32(dthip_ctx* ctx, const double* in, int64_t n, float* out);
int launch_sample_rows(dthip_ctx* ctx, int32_t* out, int64_t m, int64_t n);
int launch_narrow_i64_u32(dthip_ctx* ctx, const long long* in, int64_t n, uint32_t* out);

// reduce.hip
struct ReduceOuts {
  void* sum = nullptr;    // int64 (integer stypes) / float (f32) / double (f64)
  void* mean = nullptr;   // double, or float for f32 input
  void* mn = nullptr;     // input stype
  void* mx = nullptr;     // input stype
  int64_t* count = nullptr;   // valid (non-NA) rows
};
int launch_reduce(dthip_ctx* ctx, const void* values, int stype, const int32_t* rowindex,
                  const uint8_t* bitmap, const uint32_t* tile_first_head, int64_t nrows,
                  const ReduceOuts& outs, int nona = 0);
int launch_count0(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t* out);
int launch_reduce_prod_int(dthip_ctx* ctx, const void* values, int stype, const int32_t* rowindex,
                           const uint8_t* bitmap, const uint32_t* tile_first_head, int64_t nrows, void* out);
int launch_prod_float_seq(dthip_ctx* ctx, const void* values, int stype, const int32_t* ri, const int32_t* offsets, int64_t ngroups, void* out);
int launch_countna_from_count(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t* out);
int launch_sum_f32_seq(dthip_ctx* ctx, const void* values, const int32_t* ri, const int32_t* offsets, int64_t ngroups, void* out);

constexpr int DTHIP_NOT_APPLICABLE = 2;     // internal: this path does not fit, take the next one

// groupwise.hip: sd / cov / corr, cumulative operators, median / nunique
int launch_gather_f64(dthip_ctx* ctx, const void* data, int stype, const int32_t* ri, int64_t n, double* out);
int launch_moments(dthip_ctx* ctx, const double* x, const double* y, const uint8_t* bitmap, const uint32_t* tile_first_head,
                   int64_t n, int op /*0 sd, 1 cov, 2 corr*/, void* out, int out_f32, const int32_t* offsets, int64_t ngroups);
int launch_cumulate(dthip_ctx* ctx, const void* data, int stype, const int32_t* ri, const uint8_t* bitmap, int64_t n,
                    int op, int reverse, void* out, int ostype);
int launch_cumcount(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t n, int ngroup, int reverse, int64_t* out);
int launch_median(dthip_ctx* ctx, const void* pair_val, int stype, const int32_t* pair_off, int64_t npairs, const int32_t* offsets,
                  int64_t ngroups, void* out);
int launch_nunique(dthip_ctx* ctx, const void* pair_val, int stype, const int32_t* pair_gid, int64_t npairs, int64_t ngroups,
                   int64_t* out);

int launch_median_sorted(dthip_ctx* ctx, const void* vg, int stype, const int32_t* order, const int32_t* offsets, int64_t ngroups,
                         void* out);
int launch_nunique_sorted(dthip_ctx* ctx, const void* vg, int stype, const int32_t* gid, const int32_t* order,
                          const int32_t* run_offsets, int64_t nruns, int64_t ngroups, int64_t* out);

// setjoin.hip: set functions over a stacked column, natural-join index
int launch_setop_flags(dthip_ctx* ctx, const int32_t* ri, const int32_t* off, int64_t ngroups, int op, const int32_t* cum,
                       int nsrc, int8_t* mask);
int launch_join_index(dthip_ctx* ctx, const dthip_col* xkeys, const dthip_col* jkeys, int nkeys, int64_t xrows, int64_t jrows,
                      int32_t* out);

// rowindex.hip
struct PredArgs {
  const void* data; int stype; int cmp; double cf; long long ci; int is_mask;   // is_mask 2: data is a bitmap (uint32 words)
};
int launch_compact(dthip_ctx* ctx, const PredArgs& p, int64_t n, int32_t* out, int64_t* nout_host);
struct TakeCols { int n; const void* in[8]; void* out[8]; int width[8]; };
int launch_compact_take(dthip_ctx* ctx, const PredArgs& p, int64_t n, int32_t* out_ri, const TakeCols& tc, int64_t* nout_host);
int launch_gather(dthip_ctx* ctx, const void* data, int stype, const int32_t* ri, int64_t nout, void* out);
int launch_range_bucket(dthip_ctx* ctx, const void* keys, int stype, int64_t n, const long long* bounds, int nbounds,
                        int8_t* out);
int launch_firstlast(dthip_ctx* ctx, const void* data, int stype, const int32_t* ri, const int32_t* offsets,
                     int64_t ngroups, int last, void* out);

// tlsort.hip: filter + key transform + first sort level in one sweep, tile-local output (see the file's header)
struct TL1Args {
  PredArgs pred;            // 8-byte predicate column (float64 / int64) <cmp> scalar
  KeyColDev key;            // ONE int32 / int64 key column; the packed key is its transformed value (<= 32 bits)
  uint32_t n;               // unfiltered rows
  int block;                // threads per workgroup, 512 | 1024: tile = block * 16 rows
  int shift, bits;          // the level's digit of the transformed key
  uint32_t* kout;           // [n] transformed keys of the passing rows, tile t's at [t * tile, t * tile + count)
  uint16_t* dir;            // [ntiles][bins + 1] first place of every digit inside the tile; [bins] = the tile's count
  uint32_t* rowid;          // [n] original row numbers, same layout (nullable)
  PayCols pay;              // riding columns: in = the unfiltered columns, out = same layout as kout
  int keepx;                // index of the riding column that IS the predicate column (-1: none)
  uint32_t* bad;            // set when a passing row's key lies outside the (guessed) key range
  // RECORD output (rec != null; kout / rowid / pay.out unused): one 16-byte record per passing row, same tile-local places --
  // {transformed key, the 4-byte riding value (row number or 4-byte column; 0 if none), the 8-byte riding value (0 if none)}.
  // The level that gathers the segments then fetches ONE 16-byte piece per row (an 8-row segment = 128 contiguous bytes)
  // instead of 4 + 4 + 8 bytes in three places: 1.4x instead of 3.2x over-fetch.  rec4 / rec8: which riding column goes where
  // (index into pay, -1 = none, -2 = the row number)
  void* rec; int rec4, rec8;
};
uint32_t tl_tile_rows();
int launch_tl_pred_sample(dthip_ctx* ctx, const PredArgs& p, uint32_t n, uint32_t nsamp, uint32_t* count);   // *count += passing sample rows
int launch_tl_level1(dthip_ctx* ctx, const TL1Args& a);
// dir -> dirT[bins + 1][dstride], cc[bins][ntb] = rows of digit b before tile block tb (64 tiles per block), tot[bins]
int launch_tl_directory(dthip_ctx* ctx, const uint16_t* dir, uint32_t ntiles, uint32_t F, uint16_t* dirT, uint32_t dstride, uint32_t* cc,
                        uint32_t ntb, uint32_t* tot);
struct TLGatherHistArgs {
  const uint32_t* keys; int shift, bits;
  const uint32_t* tdesc; const uint32_t* gdesc; const uint32_t* pstart;
  const uint16_t* dirT; uint32_t dstride; const uint32_t* cc; uint32_t ntb, ntiles1, T1;
  uint32_t* P; uint32_t* gtot;
};
int launch_tl_gather_hist(dthip_ctx* ctx, const TLGatherHistArgs& a, uint32_t G);
// second tile-local level: dir2[ntiles2][bins2 + 1] -> dirT2[bins2 + 1][dstride2], fstart[nbk + 1] = first row of every final
// bucket (parent-major: c = b1 * bins2 + d2; pfirst[b1] = first level-2 tile of parent b1), *maxsize = largest final bucket
int launch_tl_final_plan(dthip_ctx* ctx, const uint16_t* dir2, uint32_t ntiles2, uint32_t nb1, int s2bits, const uint32_t* pfirst,
                         const uint32_t* pstart, uint16_t* dirT2, uint32_t dstride2, uint32_t* fstart, uint32_t* maxsize);

}  // namespace dthip
