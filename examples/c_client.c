/*
 * c_client.c -- libdthip.so used from plain C: no Python, no PyTorch, no HIP headers.
 *
 *   gcc -O2 -Iinclude examples/c_client.c -o /tmp/c_client -Ldatatable_amd -ldthip -Wl,-rpath,$PWD/datatable_amd -lm
 *   /tmp/c_client [nrows]
 *
 * What a compiled-language host (the reference's C++ core, or a cgo / JNI binding) does with the ABI:
 *   DT[:, [sum(f.v), count()], by(f.k)]          dthip_groupby_agg, host pointers in / host copies out
 *   V = DT[f.v > 0, :]; V[:, :, by(f.k)]          dthip_filter_cmp -> dthip_gather -> dthip_groupby_rows
 *   the same aggregate with device-resident data  dthip_malloc / dthip_memcpy_h2d / DTHIP_DEVICE, timed
 * Every result is checked against a scalar loop in this file; exit status 0 = all equal.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dthip.h"

#define CHECK(call)                                                                         \
  do {                                                                                      \
    int rc_ = (call);                                                                       \
    if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dthip_last_error()); return 1; } \
  } while (0)

static uint64_t rng_state = 88172645463325252ULL;
static uint64_t rng(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char** argv)
{
  const int64_t n = argc > 1 ? atoll(argv[1]) : 2000000;
  const int64_t ngmax = 5000;
  if (dthip_abi_version() != DTHIP_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
  if (dthip_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 1; }
  dthip_ctx* ctx = NULL;
  CHECK(dthip_init(0, NULL, &ctx));

  int64_t* k = malloc(sizeof(int64_t) * n);
  double* v = malloc(sizeof(double) * n);
  for (int64_t i = 0; i < n; i++) {
    k[i] = (int64_t)(rng() % ngmax) - 100;                 /* keys in [-100, 4900) */
    v[i] = (double)((int64_t)(rng() % 2001) - 1000) / 8.0; /* exactly representable: sums are exact */
    if (rng() % 97 == 0) k[i] = INT64_MIN;                 /* NA key */
    if (rng() % 89 == 0) v[i] = NAN;                       /* NA value */
  }

  /* expected, by a scalar loop: group NA first, then ascending keys */
  double* esum = calloc(ngmax + 1, sizeof(double));
  int64_t* ecnt = calloc(ngmax + 1, sizeof(int64_t));
  for (int64_t i = 0; i < n; i++) {
    const int64_t slot = k[i] == INT64_MIN ? 0 : k[i] + 100 + 1;
    ecnt[slot]++;
    if (!isnan(v[i])) esum[slot] += v[i];
  }

  /* ---- 1. fused groupby-aggregate, host pointers ------------------------------------------- */
  dthip_col key = {k, DTHIP_INT64, 0}, val = {v, DTHIP_FLOAT64, 0};
  dthip_agg aggs[2] = {{DTHIP_SUM, 0}, {DTHIP_COUNT0, -1}};
  dthip_result* r = NULL;
  CHECK(dthip_groupby_agg(ctx, &key, 1, &val, 1, aggs, 2, n, DTHIP_NA_FIRST, DTHIP_HOST, &r));
  const int64_t ng = dthip_result_ngroups(r);
  int64_t* gk = malloc(sizeof(int64_t) * ng);
  double* gs = malloc(sizeof(double) * ng);
  int64_t* gc = malloc(sizeof(int64_t) * ng);
  CHECK(dthip_result_copy_key(ctx, r, 0, gk, DTHIP_HOST));
  CHECK(dthip_result_copy_agg(ctx, r, 0, gs, DTHIP_HOST));
  CHECK(dthip_result_copy_agg(ctx, r, 1, gc, DTHIP_HOST));
  CHECK(dthip_result_free(ctx, r));
  int64_t g = 0, bad = 0;
  for (int64_t slot = 0; slot <= ngmax; slot++) {
    if (!ecnt[slot]) continue;
    const int64_t want_key = slot == 0 ? INT64_MIN : slot - 1 - 100;
    if (g >= ng || gk[g] != want_key || gc[g] != ecnt[slot] || gs[g] != esum[slot]) bad++;
    g++;
  }
  if (g != ng) bad++;
  printf("groupby_agg (host):   %lld rows -> %lld groups, %lld mismatches\n", (long long)n, (long long)ng, (long long)bad);

  /* ---- 2. filter -> RowIndex -> view gather -> rows in grouped order ---------------------- */
  int32_t* ri = malloc(sizeof(int32_t) * n);
  int64_t npass = 0;
  CHECK(dthip_filter_cmp(ctx, &val, n, DTHIP_GT, 0.0, 0, DTHIP_HOST, ri, &npass));
  int64_t* kv = malloc(sizeof(int64_t) * (npass ? npass : 1));
  double* vv = malloc(sizeof(double) * (npass ? npass : 1));
  CHECK(dthip_gather(ctx, &key, ri, npass, DTHIP_HOST, kv));
  CHECK(dthip_gather(ctx, &val, ri, npass, DTHIP_HOST, vv));
  dthip_col vkey = {kv, DTHIP_INT64, 0}, cols[2] = {{kv, DTHIP_INT64, 0}, {vv, DTHIP_FLOAT64, 0}};
  CHECK(dthip_groupby_rows(ctx, &vkey, 1, cols, 2, npass, DTHIP_NA_FIRST, DTHIP_HOST, 0, &r));
  int64_t* ok = malloc(sizeof(int64_t) * (npass ? npass : 1));
  double* ov = malloc(sizeof(double) * (npass ? npass : 1));
  CHECK(dthip_result_copy_col(ctx, r, 0, ok, DTHIP_HOST));
  CHECK(dthip_result_copy_col(ctx, r, 1, ov, DTHIP_HOST));
  const int64_t ngv = dthip_result_ngroups(r);
  CHECK(dthip_result_free(ctx, r));
  int64_t bad2 = 0, expect_pass = 0;
  for (int64_t i = 0; i < n; i++) expect_pass += v[i] > 0.0;
  if (npass != expect_pass) bad2++;
  for (int64_t i = 0; i < npass; i++) {
    if (!(ov[i] > 0.0)) bad2++;
    if (i && !(ok[i - 1] <= ok[i])) bad2++;                /* INT64_MIN (NA) sorts first */
  }
  printf("filter+groupby_rows:  %lld of %lld rows pass, %lld groups, %lld mismatches\n", (long long)npass, (long long)n,
         (long long)ngv, (long long)bad2);

  /* ---- 3. device-resident: upload once, aggregate, time with the library's stream timer ---- */
  void *dk = NULL, *dv = NULL;
  CHECK(dthip_malloc(ctx, sizeof(int64_t) * n, &dk));
  CHECK(dthip_malloc(ctx, sizeof(double) * n, &dv));
  CHECK(dthip_memcpy_h2d(ctx, dk, k, sizeof(int64_t) * n));
  CHECK(dthip_memcpy_h2d(ctx, dv, v, sizeof(double) * n));
  dthip_col dkey = {dk, DTHIP_INT64, 0}, dval = {dv, DTHIP_FLOAT64, 0};
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(dthip_timer_start(ctx));
    CHECK(dthip_groupby_agg(ctx, &dkey, 1, &dval, 1, aggs, 2, n, DTHIP_NA_FIRST, DTHIP_DEVICE, &r));
    CHECK(dthip_timer_stop(ctx, &ms));
    if (rep < 2) CHECK(dthip_result_free(ctx, r));
  }
  double* gs2 = malloc(sizeof(double) * ng);
  CHECK(dthip_result_copy_agg(ctx, r, 0, gs2, DTHIP_HOST));
  int64_t bad3 = dthip_result_ngroups(r) != ng;
  for (int64_t i = 0; i < ng && !bad3; i++) bad3 += gs2[i] != gs[i];
  CHECK(dthip_result_free(ctx, r));
  CHECK(dthip_free(ctx, dk));
  CHECK(dthip_free(ctx, dv));
  printf("groupby_agg (device): %.3f ms = %.3g rows/s, %lld mismatches\n", ms, (double)n / (ms * 1e-3), (long long)bad3);

  CHECK(dthip_destroy(ctx));
  return (bad || bad2 || bad3) ? 2 : 0;
}
