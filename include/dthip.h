/*
 * dthip.h -- C ABI of libdthip.so: the MI355X (gfx950) implementation of
 * datatable's DT[i, j, by()] hot path.
 *
 * Every entry point replaces (or is what a binding of) one seam of the
 * reference, h2oai/datatable v1.2.0a (paths relative to the reference root):
 *
 *   dthip_groupby        <- RiGb group(columns, flags, na_pos)
 *                           src/core/sort.h:56-58, src/core/sort.cc:1411-1495
 *   dthip_reduce         <- FExpr_ReduceUnary::evaluate1(col, gby, is_grouped)
 *                           src/core/expr/fexpr_reduce_unary.h:42 and the
 *                           reducer columns src/core/column/sumprod.h:34-59,
 *                           mean.h:33-52, minmax.h:33-62, count.h:35-88
 *   dthip_groupby_agg    <- EvalContext::evaluate() for DT[:, {reducers}, by(keys)]
 *                           src/core/expr/eval_context.cc:144-172,249-288,473-516
 *                           (group + reducers fused; no RowIndex materialised)
 *   dthip_groupby_rows   <- DT[:, cols, by(keys)]: group() + ColumnImpl::_materialize_fw of the
 *                           selected columns, src/core/expr/eval_context.cc:497-508,
 *                           src/core/column/column_impl.cc:78-101
 *   dthip_ungroup        <- Groupby::ungroup_rowindex, src/core/groupby.cc:117-130
 *   dthip_bool_to_rowindex <- ArrayRowIndexImpl::init_from_boolean_column
 *                           src/core/rowindex_array.cc:130-170
 *   dthip_filter_cmp     <- DT[f.x <cmp> c, :] : comparison FExpr + the above
 *                           src/core/expr/fexpr_func.cc:61-73
 *   dthip_gather         <- ColumnImpl::_materialize_fw over an ArrayView
 *                           src/core/column/column_impl.cc:78-101, view.cc:140-145
 *
 * Conventions (mirroring src/datatable/include/datatable.h:32-116, api.cc:34-38):
 *   - plain C, no HIP/torch types in signatures; pointers + sizes only
 *   - stype codes are the reference's SType values (src/core/stype.h:41-62)
 *   - NA sentinels are the reference's (src/core/stype.h:186-197): INT*_MIN for
 *     int8/16/32/64 and bool8 (-128), NaN for float32/64; no validity bitmaps
 *   - RowIndex / group offsets are int32 (ARR32; src/core/sort.cc:464-465,
 *     groupby.cc:43-48); nrows > INT32_MAX is rejected with DTHIP_EINVAL
 *     (the reference silently overflows)
 *   - every function returns 0 on success or a negative DTHIP_E* code and
 *     records a message retrievable with dthip_last_error() (thread-local)
 *   - `mem` says where ALL data pointers of that call live: DTHIP_HOST
 *     (pageable/pinned host memory; the call stages through HBM) or
 *     DTHIP_DEVICE (HBM pointers of the context's device; nothing is copied)
 *   - a context owns one HIP stream; calls are asynchronous on that stream
 *     for DTHIP_DEVICE data except where a result size must be known
 *     (ngroups), thread-compatible but not thread-safe
 */
#ifndef DTHIP_H
#define DTHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTHIP_ABI_VERSION 7   /* 2: + dthip_reduce2, dthip_cumulate, dthip_setop, dthip_join_index, reducer ops 8-10
                                 3: + dthip_comm_*, dthip_sharded_groupby_* (multi-GPU inside the library), DTHIP_FLAG_NONA,
                                    dthip_host_register / dthip_host_unregister
                                 4: + dthip_comm_last_stats; options sort_path, msd_min_rows, msd_bucket_rows, filter_path
                                 5: + dthip_build_id, dthip_from_arrow (Arrow-layout columns: validity bitmap -> sentinels on the device),
                                    dthip_filter_groupby_rows (config 5 in one call); option filter_rows_fused
                                 6: + dthip_last_call_stats; reducers DTHIP_PROD, DTHIP_COUNTNA (dthip_reduce)
                                 7: + DTHIP_FILLNA (dthip_cumulate) */

/* error codes */
#define DTHIP_OK        0
#define DTHIP_EINVAL   -1   /* bad argument (maps to ValueError/TypeError) */
#define DTHIP_ENOTIMPL -2   /* unsupported stype/op (maps to NotImplementedError) */
#define DTHIP_ENOMEM   -3   /* device allocation failed (MemoryError) */
#define DTHIP_EDEVICE  -4   /* HIP runtime error / kernel fault (RuntimeError) */

/* reference SType codes, src/core/stype.h:41-62 */
enum dthip_stype {
  DTHIP_BOOL = 1, DTHIP_INT8 = 2, DTHIP_INT16 = 3, DTHIP_INT32 = 4,
  DTHIP_INT64 = 5, DTHIP_FLOAT32 = 6, DTHIP_FLOAT64 = 7
};

/* reducers */
enum dthip_op {
  DTHIP_SUM = 0,     /* sum(col): NA skipped, empty / all-NA group -> 0; int -> int64 (wraps), float64 -> float64.
                        DEVIATION (documented): float32 sums are ACCUMULATED IN float64 and rounded to
                        float32 once; the reference accumulates in float32 row by row (column/sumprod.h:48-55), so
                        its result carries its own rounding (~1e-7 x sum|v| per group) that this one does not:
                        the two agree to ~1e-4 relative on long groups, exactly on short ones.  Option "f32_sum" = 1
                        reproduces the reference's float32 accumulation bit for bit (dthip_set_option) */
  DTHIP_MEAN = 1, DTHIP_MIN = 2, DTHIP_MAX = 3,
  DTHIP_COUNT = 4,   /* count(col): non-NA rows per group   (count.h:35-58) */
  DTHIP_COUNT0 = 5,  /* count():    rows per group          (count.h:61-88) */
  DTHIP_FIRST = 6,   /* first(col): element of the group's first row, NA included */
  DTHIP_LAST = 7,    /* last(col)   (FirstLast_ColumnImpl, src/core/expr/head_reduce_unary.cc:116-160);
                        both need the row order, so dthip_groupby_agg takes the sort path for them */
  /* dthip_reduce only: */
  DTHIP_SD = 8,      /* sd(col): sample standard deviation, NA if < 2 valid rows or a non-finite value
                        (sd_reducer, src/core/expr/head_reduce_unary.cc:194-216) */
  DTHIP_MEDIAN = 9,  /* median(col) of the valid values (Median_ColumnImpl, head_reduce_unary.cc:424-470) */
  DTHIP_NUNIQUE = 10,/* nunique(col): distinct valid values, int64 (op_nunique, head_reduce_unary.cc:377-387) */
  DTHIP_PROD = 11,   /* prod(col): NA skipped, empty / all-NA group -> 1; output stypes as DTHIP_SUM (FExpr_SumProd<false>,
                        src/core/expr/fexpr_sumprod.cc:47-82,98-110 registers sum and prod from one template; column/sumprod.h:34-59).
                        Integers: wrapping int64 product, exact.  float32 / float64: multiplied in the reference's own row
                        order (one accumulator per group), bit for bit -- a product that overflows or underflows on the way
                        depends on the order, so it is never re-associated */
  DTHIP_COUNTNA = 12 /* countna(col): NA rows per group, int64 (CountUnary_ColumnImpl<T, true>, column/count.h:35-58,
                        fexpr_count.cc:35-90) */
};

/* binary group reducers, dthip_reduce2 (src/core/expr/head_reduce_binary.cc:113-198) */
enum dthip_op2 { DTHIP_COV = 0, DTHIP_CORR = 1 };

/* group-wise cumulative operators, dthip_cumulate (src/core/column/cumsumprod.h:52-92,
 * cumminmax.h:48-98, cumcountngroup.h:55-72) */
enum dthip_cumop {
  DTHIP_CUMSUM = 0, DTHIP_CUMPROD = 1, DTHIP_CUMMIN = 2, DTHIP_CUMMAX = 3,
  DTHIP_CUMCOUNT = 4,  /* cumcount(): row number inside the group, int64 (no value column) */
  DTHIP_NGROUP = 5,    /* ngroup():   group number, int64 (no value column) */
  DTHIP_FILLNA = 6     /* fillna(col, reverse): every NA takes the last valid value before it in its group (the next one
                          when reverse), NA while there is none; output stype = input stype (FExpr_FillNA::fill_rowindex,
                          src/core/expr/fexpr_fillna.cc:85-117; the reference builds a RowIndex and views the column
                          through it, the values are the same) */
};

enum dthip_mem { DTHIP_HOST = 0, DTHIP_DEVICE = 1 };

/* NaPosition, src/core/sort.h:46-50.  REMOVE (dthip_groupby; the reference's sort(na_position="remove")):
 * the rows are ordered NA first and as many rows as the LAST key column has NAs are cut off the front
 * (SortContext::get_result_rowindex, sort.cc:598-608, reads the NA count of the column sorted last): with
 * one key exactly its NA rows */
enum dthip_napos { DTHIP_NA_FIRST = 0, DTHIP_NA_LAST = 1, DTHIP_NA_REMOVE = 2 };

/* SortFlag bits, src/core/sort.h:36-44 (key columns) */
#define DTHIP_FLAG_DESCENDING 1
/* value columns of dthip_groupby_agg: the column holds NO NA -- INT*_MIN / NaN are ordinary values that take part
 * in sum / count (the library's own merges of partial sums use it: a partial int64 sum that wrapped to INT64_MIN, or
 * a partial float sum that is NaN because +inf and -inf met, must not be skipped like an NA; the reference has no
 * such step, its reducers see every row of a group in one loop, column/sumprod.h:34-59) */
#define DTHIP_FLAG_NONA 2

/* comparison codes for dthip_filter_cmp */
enum dthip_cmp { DTHIP_GT = 0, DTHIP_GE = 1, DTHIP_LT = 2, DTHIP_LE = 3, DTHIP_EQ = 4, DTHIP_NE = 5,
                 DTHIP_NOTNA = 6, DTHIP_ISNA = 7 /* the scalar is ignored: rows whose value is valid / NA */ };

/* one typed column buffer: contiguous T[nrows] with sentinel NAs
 * (SentinelFw_ColumnImpl<T>, src/core/column/sentinel_fw.cc:141-181) */
typedef struct dthip_col {
  const void* data;
  int32_t stype;   /* enum dthip_stype */
  int32_t flags;   /* DTHIP_FLAG_* */
} dthip_col;

/* one requested aggregate: op applied to values[col] (col ignored for COUNT0) */
typedef struct dthip_agg {
  int32_t op;      /* enum dthip_op */
  int32_t col;
} dthip_agg;

typedef struct dthip_ctx dthip_ctx;       /* device + stream + workspace */
typedef struct dthip_result dthip_result; /* device-resident result of a groupby */

/* ---- library / context -------------------------------------------------- */
int         dthip_abi_version(void);
/* 12 hex digits: a hash of the SOURCES of the library (every .hip / .hpp file of csrc and this header) taken by the build -- what a profile or
 * a counter record was taken on (profiles/pmc_traffic.json "build_id"; bench.py drops a traffic record of another build) */
const char* dthip_build_id(void);
const char* dthip_last_error(void);
int         dthip_device_count(void);
/* stream: a hipStream_t to launch on (e.g. the caller's current stream), or NULL
 * to let the context create its own non-blocking stream. */
int  dthip_init(int device, void* stream, dthip_ctx** out);
/* launch on the caller's stream from now on; NULL = the device's default (legacy) stream, which
 * dthip_init cannot express because NULL there means "create one".  The context never owns it. */
int  dthip_use_stream(dthip_ctx* ctx, void* stream);
int  dthip_destroy(dthip_ctx* ctx);
int  dthip_sync(dthip_ctx* ctx);
/* release cached workspace back to the driver */
int  dthip_trim(dthip_ctx* ctx);
/* tuning knobs (defaults are right for production):
 *   "agg_path"       0 = choose per call (default), 1 = always the sort path,
 *                    2 = the bucketed (sort-free) aggregation whenever its preconditions hold
 *   "bucket_variant" partition tile geometry of the bucketed aggregation (0 = default)
 *   "agg_offsets"    1 (default): dthip_groupby_agg results always carry the group offsets;
 *                    0: only when a count() aggregate asks for group sizes -- the reference's result
 *                    Frame of DT[:, sum(f.v), by(f.k)] holds keys and sums only, and not counting rows
 *                    lets the bucketed aggregation use twice as many table slots per bucket
 *   "sort_path"      0 (default): inputs of at least "msd_min_rows" (2^26) rows whose packed key has <= 32 bits are ordered by
 *                    MSD levels -- two stable scatter levels, then every final bucket (about "msd_bucket_rows" rows; windows of
 *                    whole buckets fill a tile) ordered in LDS and written in place -- and fall back to the LSD passes when
 *                    a final bucket would not fit a tile (heavy duplicates over a wide range); everything else takes stable
 *                    LSD radix passes; 1: LSD passes only; 2: the MSD levels whenever their other preconditions hold, without
 *                    the "msd_min_rows" test (A/B runs, tests).  Same results, bit for bit
 *   "filter_rows_fused" 1 (default): dthip_filter_groupby_rows takes its fused route where it applies; 0: always the
 *                    two-call sequence (A/B runs, tests).  Same results, bit for bit
 *   "tl_level2"      the fused route's second level: 1 (default) tile-local output as well (no histogram pass, sequential
 *                    writes; the final level gathers its buckets' segments) when the final buckets are expected to fill
 *                    windows, 0 never (exact-position scatter), 2 always (tests).  Same results, bit for bit
 *   "small_path"     2 (default): a groupby_agg whose keys fit ONE table of <= 8192 slots runs a launch-lean sequence (tables
 *                    initialised by the plan kernel; group list + offsets + count from one single-workgroup kernel, which
 *                    writes its counts into mapped host memory -- no copy command); 1: the counts are copied back instead;
 *                    0: the general sequence.  1e6 rows / 100 groups: 0.116 ms (0), 0.087 (1), 0.085 (2).  Same results
 *   "nona_guess"     1 (default): groupby_agg samples the value columns; a column whose sample holds no NA is aggregated
 *                    without its valid counter (the group size stands for it) while every row is checked -- an NA found
 *                    anyway makes the call aggregate once more, counting.  0: always count.  Results are identical.
 *   "filter_path"    1 (default): row filters count, then write (two reads of the predicate column); 0: one pass whose tile
 *                    offsets come from a decoupled look-back (measured slower on MI355X; kept for A/B runs)
 *   "f32_sum"        0 (default): sum(float32 column) accumulates in float64 and rounds once (the documented deviation at
 *                    DTHIP_SUM); 1: it accumulates in float32, the valid rows of a group added one by one in grouped row
 *                    order -- bit for bit the reference's SumProd_ColumnImpl<float> (column/sumprod.h:48-55); one thread
 *                    per group through the RowIndex, a reproduction switch, not a fast path.  Single-GPU calls only:
 *                    dthip_sharded_groupby_agg* return DTHIP_ENOTIMPL for sum(float32) while it is set (partial sums of
 *                    row shards cannot reproduce the row order)
 *   "median_pairs"   0 (default): median / nunique of FLOAT columns order the rows by (group, value); integer
 *                    columns go through the distinct (group, value) pairs of a fused count() aggregation,
 *                    which is sort-free for categorical data; 1 = floats too
 *   "join_table"     1 (default): dthip_join_index looks a dense single integer key up in a direct
 *                    key -> row table instead of binary-searching J; 0 = always search
 *   "hash_mode"      0 (default): sparse key ranges (too wide for the bucketed aggregation) are combined
 *                    through LDS hash tables when a sampled distinct-count estimate says they fit, and the
 *                    partial groups are merged by the sort path; 1 = never; 2 = whenever the query shape allows;
 *                    3 = like 2, but the rows are partitioned by histogram + exact scatter positions (rounds 2-5) even
 *                    where the tile-local partition of round 6 (one aligned int64 key) applies -- tests and A/B runs
 *   "cluster_mode"   0 (default): a sample of neighbouring rows decides whether the bucketed aggregation
 *                    runs its variants for sorted / clustered / constant keys (a wave that addresses one
 *                    bucket or slot is counted / reduced in registers first); 1 = never, 2 = always
 *   "guard"          debugging flavour (also env DTHIP_GUARD=1|2 at dthip_init; the counterpart of the reference's ASan build,
 *                    ci/ext.py:318-327): every device buffer the library allocates (scratch, results, staged host
 *                    columns, dthip_malloc) becomes its own virtual-memory mapping between two unmapped pages, with its
 *                    END (1) or START (2) flush against them, nothing is cached and every kernel launch is synchronised:
 *                    a kernel that touches even one 16-byte piece outside a buffer faults at once and is named on stderr;
 *                    3 = only the synchronised, named launches (ordinary allocator), e.g. under a profiler
 *   "spec_min_rows"  from this many rows on, integer key ranges are first guessed from a sample
 *                    and verified by the bucketed aggregation (default 2^23) */
int  dthip_set_option(dthip_ctx* ctx, const char* name, int64_t value);

/* device memory helpers so a host-language binding needs no HIP runtime */
int  dthip_malloc(dthip_ctx* ctx, size_t bytes, void** dptr);
int  dthip_free(dthip_ctx* ctx, void* dptr);
int  dthip_memcpy_h2d(dthip_ctx* ctx, void* dst, const void* src, size_t bytes);
int  dthip_memcpy_d2h(dthip_ctx* ctx, void* dst, const void* src, size_t bytes);

/* DTHIP_HOST data: pageable buffers are staged by the HIP runtime through its own pinned bounce buffers (a CPU copy per
 * byte); a buffer registered here (hipHostRegister: page-locked and mapped once, ~0.2 s per GB) is DMA-read directly,
 * which can pay when the same column buffers are queried repeatedly.  Measured on MI355X / PCIe Gen5 x16 (bench.py
 * host_mode): 53.8 GB/s of input pageable, 54.5 GB/s registered -- either way the mode is bounded by the link, not by HBM. */
int  dthip_host_register(dthip_ctx* ctx, void* ptr, size_t bytes);
int  dthip_host_unregister(dthip_ctx* ctx, void* ptr);

/* stream timers (HIP events on the context's stream) */
int  dthip_timer_start(dthip_ctx* ctx);
int  dthip_timer_stop(dthip_ctx* ctx, float* elapsed_ms);   /* synchronises */
/* What the LAST query call on this context (dthip_groupby / _groupby_rows / _groupby_agg / _filter_groupby_rows) did besides
 * its result -- the default path GUESSES key ranges and NA-freeness from samples and verifies every row, so a wrong guess
 * costs a second sweep that nothing else reports:
 *   out[0] sweeps repeated because a guessed integer key range was violated by some row (an outlier key),
 *   out[1] aggregations repeated because a value column guessed NA-free held an NA (rounds 4-5; always 0 since round 6: such
 *          rows are skipped and counted apart, the valid count of a group is its size minus them -- no second aggregation),
 *   out[2] routes given up after they had started (fused filter route -> two calls; hash tables full -> sort path),
 *   out[3] the path that produced the result: 1 sort path, 2 bucketed aggregation, 3 hash combiner, 4 fused filter route,
 *          6 rows found in key order already (dthip_groupby_agg: heads of the raw key column, no grouping pass),
 *   out[4] rows whose key lay outside a guessed range and were LISTED, grouped apart and spliced in (at most 65536 of them:
 *          no second sweep; more than that counts as a wrong guess, out[0]).
 * n = number of values wanted (<= 5).  The reference has no counterpart (its min / max scan is exact, stats.cc:601-640). */
int  dthip_last_call_stats(const dthip_ctx* ctx, int64_t* out, int n);
/* per-kernel accounting: when enabled every launch of the library's kernels is
 * bracketed by HIP events; dthip_profile_get() synchronises and returns the
 * accumulated time and launch count of kernels whose name contains `name`. */
int  dthip_profile_enable(dthip_ctx* ctx, int on);
int  dthip_profile_reset(dthip_ctx* ctx);
int  dthip_profile_get(dthip_ctx* ctx, const char* name, double* total_ms, int64_t* launches);
/* names (newline separated) of all kernels seen since the last reset */
int  dthip_profile_names(dthip_ctx* ctx, char* buf, size_t buflen);

/* ---- S-grp: group() ------------------------------------------------------ */
/* Stable sort of rows by keys[0..nkeys) (keys[0] most significant), NA first
 * unless na_pos says otherwise, and the run-length grouping of equal keys.
 * The result holds, on the device: rowindex int32[nrows] (only if
 * want_rowindex) and offsets int32[ngroups+1]. */
int  dthip_groupby(dthip_ctx* ctx, const dthip_col* keys, int nkeys, int64_t nrows,
                   int na_pos, int mem, int want_rowindex, dthip_result** out);

/* ---- fused DT[:, aggs, by(keys)] ----------------------------------------- */
/* Groups by keys and evaluates aggs without materialising the RowIndex.
 * The result holds offsets, one group-key column per key (value of the key in
 * each group, stype of the key) and one column per agg, typed by
 * dthip_reduce_out_stype(). */
int  dthip_groupby_agg(dthip_ctx* ctx, const dthip_col* keys, int nkeys,
                       const dthip_col* values, int nvalues,
                       const dthip_agg* aggs, int naggs,
                       int64_t nrows, int na_pos, int mem, dthip_result** out);

/* ---- DT[:, cols, by(keys)]: rows in grouped order, materialised ------------- */
/* group() followed by the gather of `cols` through the RowIndex
 * (EvalContext::evaluate_select + ColumnImpl::_materialize_fw,
 * src/core/expr/eval_context.cc:497-508, src/core/column/column_impl.cc:78-101;
 * view.cc:140-145), fused: the columns ride through the sort, nothing is
 * gathered at random.  The result holds offsets, the RowIndex (if wanted) and
 * cols[c] permuted into grouped order (dthip_result_col).  Pass the key columns
 * among `cols` to get the by-columns of the result Frame. */
int  dthip_groupby_rows(dthip_ctx* ctx, const dthip_col* keys, int nkeys,
                        const dthip_col* cols, int ncols, int64_t nrows,
                        int na_pos, int mem, int want_rowindex, dthip_result** out);

/* ---- result accessors ---------------------------------------------------- */
int64_t dthip_result_ngroups(const dthip_result* r);
int64_t dthip_result_nrows(const dthip_result* r);
/* device pointers owned by the result (NULL if absent) */
const int32_t* dthip_result_rowindex(const dthip_result* r);
const int32_t* dthip_result_offsets(const dthip_result* r);
const void*    dthip_result_key(const dthip_result* r, int k);
const void*    dthip_result_agg(const dthip_result* r, int a);
const void*    dthip_result_col(const dthip_result* r, int c);   /* dthip_groupby_rows */
int            dthip_result_agg_stype(const dthip_result* r, int a);
/* copy out (dst in `mem` space, sized by the caller from ngroups/nrows) */
int  dthip_result_copy_rowindex(dthip_ctx* ctx, const dthip_result* r, int32_t* dst, int mem);
int  dthip_result_copy_offsets(dthip_ctx* ctx, const dthip_result* r, int32_t* dst, int mem);
int  dthip_result_copy_key(dthip_ctx* ctx, const dthip_result* r, int k, void* dst, int mem);
int  dthip_result_copy_agg(dthip_ctx* ctx, const dthip_result* r, int a, void* dst, int mem);
int  dthip_result_copy_col(dthip_ctx* ctx, const dthip_result* r, int c, void* dst, int mem);
/* fill the by-column k of a groupby (not _agg) result: key[rowindex[offsets[g]]]
 * (EvalContext::update_groupby_columns, eval_context.cc:473-485) */
int  dthip_result_group_keys(dthip_ctx* ctx, const dthip_result* r, const dthip_col* key,
                             int mem, void* dst);
int  dthip_result_free(dthip_ctx* ctx, dthip_result* r);

/* ---- S-red: per-group reducers over an existing grouping ------------------ */
/* output stype of op applied to a column of stype `stype`
 * (fexpr_sumprod.cc:47-66, fexpr_mean.cc:45-74, fexpr_minmax.cc:47-68, fexpr_count.cc) */
int  dthip_reduce_out_stype(int op, int stype);
/* value[rowindex[i]] for i in [offsets[g], offsets[g+1]) reduced per group.
 * rowindex may be NULL (identity: the column is already in grouped order).
 * value may be NULL for DTHIP_COUNT0.  out: T_out[ngroups], NA as sentinel. */
int  dthip_reduce(dthip_ctx* ctx, int op, const dthip_col* value,
                  const int32_t* rowindex, const int32_t* offsets,
                  int64_t ngroups, int64_t nrows, int mem, void* out);

/* cov / corr of two columns per group, over the rows where BOTH values are valid
 * (cov_reducer / corr_reducer, src/core/expr/head_reduce_binary.cc:113-135,167-198).
 * Output stype: float32 when both inputs are float32, else float64 (:47-51,150-155);
 * cov: NA for < 2 valid pairs; corr: NA unless var(a)*var(b) > 0.  out: T_out[ngroups]. */
int  dthip_reduce2_out_stype(int stype_a, int stype_b);
int  dthip_reduce2(dthip_ctx* ctx, int op /* enum dthip_op2 */, const dthip_col* a, const dthip_col* b,
                   const int32_t* rowindex, const int32_t* offsets,
                   int64_t ngroups, int64_t nrows, int mem, void* out);

/* cumsum / cumprod / cummin / cummax inside every group, in grouped row order (or from the
 * group's last row backwards when reverse != 0), and cumcount() / ngroup()
 * (FExpr_CumSumProd::evaluate1, src/core/expr/fexpr_cumsumprod.cc:72-99: integers -> int64,
 * float32 stays float32; NA counts as 0 / 1.  FExpr_CumMinMax::evaluate1,
 * fexpr_cumminmax.cc:87-101: output stype = input stype, NA until the first valid row;
 * fillna(col, reverse): fexpr_fillna.cc:85-117, ABI v7).
 * out: T_out[nrows], row i of the result is grouped position i (value[rowindex[i]]). */
int  dthip_cumulate_out_stype(int op /* enum dthip_cumop */, int stype);
int  dthip_cumulate(dthip_ctx* ctx, int op, const dthip_col* value,
                    const int32_t* rowindex, const int32_t* offsets,
                    int64_t ngroups, int64_t nrows, int reverse, int mem, void* out);

/* ---- set functions and natural join: the other callers of group() ------- */
/* dt.union / unique / intersect / setdiff / symdiff (src/core/set_funcs.cc:134-431).  `stacked` is the
 * sources' single columns concatenated (rbind, set_funcs.cc:118-124; one stype), cumsizes[k] = rows of
 * sources 0..k (host array).  out_indices (room for nrows): row ids into `stacked` of the result
 * elements, ascending by value, NA first -- the values are stacked[out_indices[i]]; *nout = their count. */
enum dthip_setfn { DTHIP_UNION = 0, DTHIP_INTERSECT = 1, DTHIP_SETDIFF = 2, DTHIP_SYMDIFF = 3 };
int  dthip_setop(dthip_ctx* ctx, int op, const dthip_col* stacked, const int64_t* cumsizes, int nsources,
                 int64_t nrows, int mem, int32_t* out_indices, int64_t* nout);

/* RowIndex natural_join(xdt, jdt) (src/core/frame/join.cc:386-446): jkeys are the key columns of a KEYED
 * frame (sorted ascending, unique: DataTable::set_key, src/core/frame/key.cc:74-133 = dthip_groupby_rows
 * + ngroups == nrows), xkeys the same-named columns of X, possibly of other stypes (comparators
 * FwCmp<TX,TJ>, join.cc:146-200,268-331).  out[i] = row of J matching X row i, or INT32_MIN. */
int  dthip_join_index(dthip_ctx* ctx, const dthip_col* xkeys, const dthip_col* jkeys, int nkeys,
                      int64_t xrows, int64_t jrows, int mem, int32_t* out);

/* Groupby::ungroup_rowindex (src/core/groupby.cc:117-130): out[i] = g for every i in
 * [offsets[g], offsets[g+1]) -- broadcasts one-value-per-group columns back to rows (GtoALL,
 * src/core/expr/workframe.cc:384-390) when composed with dthip_gather */
int  dthip_ungroup(dthip_ctx* ctx, const int32_t* offsets, int64_t ngroups, int64_t nrows,
                   int mem, int32_t* out);

/* Multi-GPU exchange of ROWS (row-returning queries; no counterpart in the single-process
 * reference): destination of every row in a range partition of an integer key column,
 * out[i] = number of boundaries <= key[i] (NA -> 0); at most 15 ascending boundaries.
 * dthip_groupby_rows on `out` then groups the rows by destination in their original order. */
int  dthip_range_bucket(dthip_ctx* ctx, const dthip_col* key, int64_t nrows,
                        const int64_t* bounds, int nbounds, int mem, int8_t* out);

/* ---- multi-GPU: one context per GPU, the exchange inside the library -------------------------
 * The reference is single-process; there is no seam to cite.  Rows are sharded by row block over the ranks
 * (rank r holds a contiguous block of the frame's rows); key-RANGE partitioning of the first key keeps the
 * global group order equal to the concatenation of the ranks' results in rank order -- the order the reference
 * returns (sort.cc:1411-1495: groups ascending, NA first / last).  Splitters come from an all-gathered 4096-bin
 * histogram of the first key, so skewed keys are spread evenly.  datatable_amd/csrc/comm.hip. */
#define DTHIP_COMM_ID_BYTES 128
/* rank 0 creates the id (ncclGetUniqueId) and hands its bytes to every rank by whatever means the application has
 * (a file, MPI, a TCP store, ...); loads librccl.so on first use */
int  dthip_comm_unique_id(void* id_out /* DTHIP_COMM_ID_BYTES */);
/* collective over all ranks: binds ctx (its device and its stream) as rank `rank` of a `world`-rank RCCL
 * communicator (ncclCommInitRank).  One rank per GPU, at most 127 ranks. */
int  dthip_comm_init(dthip_ctx* ctx, int rank, int world, const void* id);
/* binds `world` distinct contexts of ONE process (on any devices, also all on the same GPU) as ranks 0..world-1 of a
 * LOCAL communicator: the same phases, the exchange being device-to-device copies, no RCCL.  For logical shards
 * on one GPU and for tests; driven with the *_local entry points below (all ranks in one call). */
int  dthip_comm_init_local(dthip_ctx* const* ctxs, int world);
int  dthip_comm_destroy(dthip_ctx* ctx);
/* what the LAST sharded call on this context's communicator moved: out[0] bytes of the all-to-all-v that left this rank
   for other ranks (the xGMI traffic), out[1] bytes that stayed on it, out[2] rows sent, out[3] rows received,
   out[4] all-gather rounds; n = number of values wanted (<= 5).  With dthip_profile_enable the per-kernel accounting also
   carries the wall-clock phases "phase_local", "phase_allgather", "phase_plan", "phase_alltoallv", "phase_merge". */
int  dthip_comm_last_stats(const dthip_ctx* ctx, int64_t* out, int n);
int  dthip_comm_rank(const dthip_ctx* ctx);     /* -1 when ctx belongs to no communicator */
int  dthip_comm_world(const dthip_ctx* ctx);    /* 0 when ctx belongs to no communicator */

/* DT[:, aggs, by(keys)] over the sharded rows (collective: every rank calls it with its own rows and the same
 * query).  Local fused groupby-aggregate (the combiner) -> range-partitioned all-to-all-v of the partial groups
 * (ncclSend/ncclRecv in one group) -> merge on the owner.  The result of rank r holds the groups of the r-th key
 * range: group-key columns and one column per agg (no offsets; ask for count() to get group sizes).  Reducers:
 * sum / mean / min / max / count / count(); integer results and min/max are exact, float sums are re-associated
 * (<= 1e-6).  The first key must be ascending. */
int  dthip_sharded_groupby_agg(dthip_ctx* ctx, const dthip_col* keys, int nkeys,
                               const dthip_col* values, int nvalues, const dthip_agg* aggs, int naggs,
                               int64_t nrows_local, int na_pos, int mem, dthip_result** out);
/* the same for a local communicator: keys[r] / values[r] / nrows[r] are rank r's, outs[r] its result */
int  dthip_sharded_groupby_agg_local(dthip_ctx* const* ctxs, int world, const dthip_col* const* keys, int nkeys,
                                     const dthip_col* const* values, int nvalues, const dthip_agg* aggs, int naggs,
                                     const int64_t* nrows, int na_pos, int mem, dthip_result** outs);
/* DT[:, cols, by(keys)] -- rows in grouped order -- over the sharded rows: every row travels ONCE to the owner of
 * its key range (slabs arrive in source-rank order = global row order), one stable local grouping there.  The
 * result of rank r holds offsets, cols[c] in grouped order (dthip_result_col(c), c < ncols) and, as column
 * `ncols`, the GLOBAL row id (int64, row_offset + local row number) of every output row: the concatenation of the
 * ranks' row-id columns is the reference's RowIndex of the whole frame. */
int  dthip_sharded_groupby_rows(dthip_ctx* ctx, const dthip_col* keys, int nkeys, const dthip_col* cols, int ncols,
                                int64_t nrows_local, int64_t row_offset, int na_pos, int mem, dthip_result** out);
int  dthip_sharded_groupby_rows_local(dthip_ctx* const* ctxs, int world, const dthip_col* const* keys, int nkeys,
                                      const dthip_col* const* cols, int ncols, const int64_t* nrows,
                                      const int64_t* row_offsets, int na_pos, int mem, dthip_result** outs);

/* ---- RowIndex construction / application ---------------------------------- */
/* ascending ARR32 of rows whose mask is 1 and not NA; out has room for n */
int  dthip_bool_to_rowindex(dthip_ctx* ctx, const int8_t* mask, int64_t n, int mem,
                            int32_t* out, int64_t* nout);
/* rows where col[i] <cmp> scalar (NA compares false, except NE); scalar is cf
 * for float columns and ci for integer columns */
int  dthip_filter_cmp(dthip_ctx* ctx, const dthip_col* col, int64_t n, int cmp,
                      double cf, int64_t ci, int mem, int32_t* out, int64_t* nout);
/* DT[f.x <cmp> c, cols] materialised in one sweep: the ascending RowIndex of the passing rows (optional)
 * and cols[k] read through it (init_from_boolean_column + _materialize_fw of every column of the view,
 * src/core/rowindex_array.cc:130-170, src/core/column/column_impl.cc:78-101), the columns being read
 * next to the predicate column instead of gathered through the finished RowIndex.  out_cols[k] and
 * out_rowindex have room for n rows; *nout rows are written.  ncols <= 8. */
int  dthip_filter_take(dthip_ctx* ctx, const dthip_col* col, int cmp, double scalar_f, int64_t scalar_i,
                       const dthip_col* cols, int ncols, int64_t n, int mem,
                       int32_t* out_rowindex /* nullable */, void* const* out_cols, int64_t* nout);

/* An Arrow-layout fixed-width column -> the sentinel layout every other entry point takes, written to the DEVICE buffer
 * dst (nrows elements of `stype`, 16-byte aligned, e.g. from dthip_malloc; DTHIP_BOOL: one int8 per row).
 *   values    the Arrow data buffer: T[nrows]; for DTHIP_BOOL Arrow's bit-packed booleans (bit i = values[i/8] >> (i&7) & 1)
 *   validity  the Arrow validity bitmap, bit i set = row i valid (LSB first); NULL = no nulls
 *   mem       where values / validity live.  DTHIP_HOST: the two buffers cross PCIe as they are (a column without nulls
 *             is copied straight into dst) and one kernel writes NA sentinels at the invalid rows -- the reference's
 *             element-by-element CPU materialisation of an Arrow column never runs.  DTHIP_DEVICE: nothing is copied.
 * Replaces ArrowFw_ColumnImpl::_get / ArrowBool_ColumnImpl::get_element read by a materialising loop
 * (src/core/column/arrow_fw.cc:63-72, column/arrow_bool.cc, Column::from_arrow column_from_arrow.cc:40-59). */
int  dthip_from_arrow(dthip_ctx* ctx, const void* values, const uint8_t* validity, int64_t nrows, int stype, int mem,
                      void* dst);

/* V = DT[pred <cmp> scalar, :]; V[:, cols, by(keys)] -- BASELINE config 5's two statements (the reference refuses the
 * one-statement form DT[i, j, by()], fexpr_func.cc:76-78) -- in ONE call: the rows that pass, in grouped order.
 * Result: offsets, cols[c] of the passing rows in grouped order (dthip_result_col(c)), nrows = passing rows, and with
 * want_rowindex the COMPOSED RowIndex of filter and grouping (dthip_result_rowindex: ORIGINAL row numbers in grouped
 * order -- RowIndex composition ab*bc, rowindex_array.cc:258-269; sorted ascending it is the filter's own RowIndex).
 * Identical, bit for bit, to dthip_filter_take + dthip_groupby_rows over the filter's outputs (the route this call takes
 * itself when the fused one does not apply: several keys, keys wider than 32 bits, fewer than "msd_min_rows" rows, a
 * predicate column that is not 8 bytes wide, more than two riding columns; option "filter_rows_fused" = 0 forces it).
 * Fused route (csrc/tlsort.hip): one sweep over the unfiltered rows evaluates the predicate, transforms the key and
 * orders every 8192-row tile's passing rows by the top digit inside the tile's own row range -- no count pass, no
 * compaction, no key-transform pass, no histogram pass, sequential writes -- then two more levels finish the order.
 * Replaces init_from_boolean_column (rowindex_array.cc:130-170) + the view's columns (view.cc:140-145) + group()
 * (sort.cc:1411-1495) + _materialize_fw (column_impl.cc:78-101). */
int  dthip_filter_groupby_rows(dthip_ctx* ctx, const dthip_col* pred, int cmp, double scalar_f, int64_t scalar_i,
                               const dthip_col* keys, int nkeys, const dthip_col* cols, int ncols, int64_t nrows,
                               int na_pos, int mem, int want_rowindex, dthip_result** out);

/* out[i] = rowindex[i] < 0 ? NA : col[rowindex[i]] */
int  dthip_gather(dthip_ctx* ctx, const dthip_col* col, const int32_t* rowindex,
                  int64_t nout, int mem, void* out);

#ifdef __cplusplus
}
#endif
#endif /* DTHIP_H */
