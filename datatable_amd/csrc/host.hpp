// host.hpp -- what the host-side translation units of libdthip.so share (round 6: api.hip of rounds 1-5, 3000 lines, is now
//   api.hip   context, allocator, guard pages, options, memory / timer / profile entry points, argument staging
//   plan.hip  key planning (sort.cc:728-776, 917-934), the sort driver (LSD passes / MSD levels), group heads -> offsets
//   agg.hip   dthip_groupby_agg: bucketed aggregation, hash combiner, sort path
//   rows.hip  dthip_groupby, dthip_groupby_rows, dthip_filter_groupby_rows (fused tile-local route)
//   ops.hip   result accessors, dthip_reduce / reduce2 / cumulate, set functions, join, filters, gather
// ).  Host code only: no kernel is declared here (common.hpp has the launchers).
#pragma once
#include <cstring>
#include <vector>
#include "common.hpp"
#include "msd_plan.hpp"

namespace dthip {

constexpr uint32_t SPEC_SAMPLES = 1u << 17;
constexpr int DTHIP_RETRY_EXACT = 1;              // internal: a guessed key range was wrong, redo with the exact one

struct KeyPlan {
  int nkeys = 0;
  KeyColDev col[MAX_KEYCOLS];
  int nsig[MAX_KEYCOLS];
  // stages of consecutive keys whose packed width is <= 64 bits; stage 0 holds the most significant keys
  int nstages = 0;
  int stage_first[MAX_KEYCOLS], stage_last[MAX_KEYCOLS], stage_bits[MAX_KEYCOLS];
  bool speculative = false;   // integer key ranges are widened guesses from a sample (bucketed aggregation only)
};

struct PaySpec {
  int n = 0;
  const void* in[MAX_PAYCOLS];
  int width[MAX_PAYCOLS];
  bool iota = false;            // column 0 is the row number
  // single int32 / int64 key whose column is wanted in sorted order: the last pass writes its ORIGINAL values here
  void* ukey_out = nullptr;
};

struct SortOut {
  bool ukey_done = false;       // PaySpec::ukey_out was filled (then `keys` is NOT: the last pass wrote the original values instead)
  void* keys = nullptr;         // sorted packed keys (scratch-owned)
  int key64 = 0;
  void* pay[MAX_PAYCOLS];       // sorted payload columns (scratch-owned, or the input itself if nothing moved)
  int npasses_run = 0;
};

struct WindowPlan { bool ok = false; uint32_t nwin = 0; const uint32_t* bounds = nullptr; const uint32_t* wfirst = nullptr;
                    int bits2 = 1; int pairs = 0; uint32_t maxsize = 0, span = 0, step = 0; };

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

// dthip_last_call_stats: the outermost query entry point starts the record, nested ones add to it
struct CallScope {
  dthip_ctx* c;
  explicit CallScope(dthip_ctx* ctx) : c(ctx) { if (c->call_depth++ == 0) memset(c->call_stats, 0, sizeof(c->call_stats)); }
  ~CallScope() { c->call_depth--; }
};

// api.hip
int stage_in(dthip_ctx* ctx, Scratch& sc, const void* src, size_t bytes, int mem, const void** dev);
int copy_out(dthip_ctx* ctx, void* dst, const void* dev_src, size_t bytes, int mem);
bool host_words(dthip_ctx* ctx);
int check_common(dthip_ctx* ctx, int64_t nrows, int mem);
int stage_cols(dthip_ctx* ctx, Scratch& sc, const dthip_col* cols, int ncols, int64_t nrows, int mem, std::vector<dthip_col>* out);
int empty_result(dthip_ctx* ctx, dthip_result* res);
int reduce_outs_for(int op, void* dst, ReduceOuts* o);
void result_adopt(Scratch& sc, dthip_result* r, void* p);
// plan.hip
int plan_keys(dthip_ctx* ctx, Scratch& sc, const dthip_col* keys_dev, int nkeys, int64_t n, int na_pos,
              KeyPlan* plan, bool speculative = false, bool tight = false);
MsdPlan msd_plan(const dthip_ctx* ctx, int64_t n, int bits, int key64, uint32_t tile);
int plan_windows(dthip_ctx* ctx, Scratch& sc, const uint32_t* fstart, uint32_t nb1, uint32_t bins2, int64_t n, const uint32_t* d_max,
                 uint32_t tile, int maxw, int rb, WindowPlan* wp);
int sort_stage(dthip_ctx* ctx, Scratch& sc, const KeyPlan& plan, int stage, int64_t n,
               const int32_t* order, const PaySpec& pay, SortOut* out);
int heads_to_offsets(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const void* keys, int key64,
                     const uint8_t* heads, int64_t n, Grouping* g);
int group_core(dthip_ctx* ctx, Scratch& sc, dthip_result* res, const dthip_col* keys_dev, int nkeys,
               int64_t n, int na_pos, KeyPlan* plan, Grouping* g);

}  // namespace dthip
