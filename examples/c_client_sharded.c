/*
 * c_client_sharded.c -- the multi-GPU entry points of libdthip.so from plain C (no Python, no PyTorch).
 *
 *   gcc -O2 -Iinclude examples/c_client_sharded.c -o /tmp/c_sharded -Ldatatable_amd -ldthip -Wl,-rpath,$PWD/datatable_amd -lm
 *
 *   one process per GPU over RCCL (start `world` of them, any launcher; the id travels through a file here):
 *       /tmp/c_sharded rccl <world> <rank> <idfile> [nrows_total]        rank r uses GPU r % device_count
 *   logical shards inside one process (no RCCL; also on one GPU):
 *       /tmp/c_sharded local <world> [nrows_total]
 *
 * The frame: nrows_total rows, key = i * 7919 % 100003 - 50 (NA every 1013th row), value = (i % 1001) / 8.0.
 * Rank r owns the row block [r n / world, (r+1) n / world).  Query: DT[:, [sum(f.v), count(), min(f.v)], by(f.k)].
 * Every rank checks ITS key range against a scalar loop over all rows; exit status 0 = all equal.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "dthip.h"

#define CHECK(call)                                                                         \
  do {                                                                                      \
    int rc_ = (call);                                                                       \
    if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dthip_last_error()); return 1; } \
  } while (0)

#define NKEYS 100003
static int64_t key_of(int64_t i) { return i % 1013 == 0 ? INT64_MIN : (i * 7919) % NKEYS - 50; }
static double val_of(int64_t i) { return (double)(i % 1001) / 8.0; }

static void fill(int64_t lo, int64_t hi, int64_t* k, double* v) {
  for (int64_t i = lo; i < hi; i++) { k[i - lo] = key_of(i); v[i - lo] = val_of(i); }
}

/* expected groups of the whole frame: slot 0 = NA, slot 1 + (key + 50) otherwise */
static double* esum; static int64_t* ecnt; static double* emin;
static void expect(int64_t n) {
  esum = calloc(NKEYS + 1, sizeof(double)); ecnt = calloc(NKEYS + 1, sizeof(int64_t)); emin = malloc((NKEYS + 1) * sizeof(double));
  for (int64_t s = 0; s <= NKEYS; s++) emin[s] = INFINITY;
  for (int64_t i = 0; i < n; i++) {
    const int64_t k = key_of(i), s = k == INT64_MIN ? 0 : k + 50 + 1;
    const double v = val_of(i);
    esum[s] += v; ecnt[s]++; if (v < emin[s]) emin[s] = v;
  }
}

/* compares one rank's result with the expected groups; returns mismatches, adds the groups it saw to *seen */
static int64_t verify(dthip_ctx* ctx, dthip_result* r, int64_t* seen, int64_t* first_key, int64_t* last_key) {
  const int64_t ng = dthip_result_ngroups(r);
  int64_t* gk = malloc(sizeof(int64_t) * (ng + 1)); double* gs = malloc(sizeof(double) * (ng + 1));
  int64_t* gc = malloc(sizeof(int64_t) * (ng + 1)); double* gm = malloc(sizeof(double) * (ng + 1));
  if (ng) {
    if (dthip_result_copy_key(ctx, r, 0, gk, DTHIP_HOST) || dthip_result_copy_agg(ctx, r, 0, gs, DTHIP_HOST) ||
        dthip_result_copy_agg(ctx, r, 1, gc, DTHIP_HOST) || dthip_result_copy_agg(ctx, r, 2, gm, DTHIP_HOST)) return -1;
  }
  int64_t bad = 0;
  for (int64_t g = 0; g < ng; g++) {
    const int64_t s = gk[g] == INT64_MIN ? 0 : gk[g] + 50 + 1;
    if (g > 0 && !(gk[g] > gk[g - 1])) bad++;                       /* ascending, NA (INT64_MIN) first */
    if (s < 0 || s > NKEYS || gc[g] != ecnt[s] || gs[g] != esum[s] || gm[g] != emin[s]) bad++;   /* eighths: sums are exact */
  }
  *seen += ng;
  *first_key = ng ? gk[0] : INT64_MAX; *last_key = ng ? gk[ng - 1] : INT64_MIN;
  free(gk); free(gs); free(gc); free(gm);
  return bad;
}

int main(int argc, char** argv)
{
  if (argc < 3) { fprintf(stderr, "usage: %s rccl <world> <rank> <idfile> [nrows] | local <world> [nrows]\n", argv[0]); return 2; }
  if (dthip_abi_version() != DTHIP_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
  const int ndev = dthip_device_count();
  if (ndev < 1) { fprintf(stderr, "no GPU\n"); return 1; }
  const int world = atoi(argv[2]);
  dthip_agg aggs[3] = {{DTHIP_SUM, 0}, {DTHIP_COUNT0, -1}, {DTHIP_MIN, 0}};
  int64_t nexpected = 0;

  if (!strcmp(argv[1], "local")) {
    const int64_t n = argc > 3 ? atoll(argv[3]) : 3000000;
    expect(n);
    for (int64_t s = 0; s <= NKEYS; s++) nexpected += ecnt[s] > 0;
    dthip_ctx** ctxs = malloc(sizeof(*ctxs) * world);
    const dthip_col** keys = malloc(sizeof(*keys) * world); const dthip_col** vals = malloc(sizeof(*vals) * world);
    int64_t* nrows = malloc(sizeof(int64_t) * world);
    dthip_result** outs = calloc(world, sizeof(*outs));
    for (int r = 0; r < world; r++) {
      CHECK(dthip_init(r % ndev, NULL, &ctxs[r]));
      const int64_t lo = r * n / world, hi = (r + 1) * n / world;
      int64_t* k = malloc(sizeof(int64_t) * (hi - lo + 1)); double* v = malloc(sizeof(double) * (hi - lo + 1));
      fill(lo, hi, k, v);
      dthip_col* kc = malloc(sizeof(dthip_col)); dthip_col* vc = malloc(sizeof(dthip_col));
      kc->data = k; kc->stype = DTHIP_INT64; kc->flags = 0; vc->data = v; vc->stype = DTHIP_FLOAT64; vc->flags = 0;
      keys[r] = kc; vals[r] = vc; nrows[r] = hi - lo;
    }
    CHECK(dthip_comm_init_local(ctxs, world));
    CHECK(dthip_sharded_groupby_agg_local(ctxs, world, keys, 1, vals, 1, aggs, 3, nrows, DTHIP_NA_FIRST, DTHIP_HOST, outs));
    int64_t seen = 0, bad = 0, prev_last = INT64_MIN; int first = 1;
    for (int r = 0; r < world; r++) {
      int64_t fk, lk;
      const int64_t b = verify(ctxs[r], outs[r], &seen, &fk, &lk);
      if (b < 0) { fprintf(stderr, "copy failed: %s\n", dthip_last_error()); return 1; }
      bad += b;
      if (fk <= lk) { if (!first && !(fk > prev_last)) bad++; prev_last = lk; first = 0; }     /* key ranges ascend with the rank */
      printf("rank %d of %d: %lld groups\n", r, world, (long long)dthip_result_ngroups(outs[r]));
      CHECK(dthip_result_free(ctxs[r], outs[r]));
    }
    if (seen != nexpected) bad++;
    printf("local communicator, %d shards, %lld rows, %lld groups: %lld mismatches\n", world, (long long)n, (long long)seen, (long long)bad);
    for (int r = 0; r < world; r++) CHECK(dthip_destroy(ctxs[r]));
    return bad ? 1 : 0;
  }

  if (strcmp(argv[1], "rccl") || argc < 5) { fprintf(stderr, "bad mode\n"); return 2; }
  const int rank = atoi(argv[3]);
  const char* idfile = argv[4];
  const int64_t n = argc > 5 ? atoll(argv[5]) : 3000000;
  unsigned char id[DTHIP_COMM_ID_BYTES];
  if (rank == 0) {
    CHECK(dthip_comm_unique_id(id));
    char tmp[4096]; snprintf(tmp, sizeof(tmp), "%s.tmp", idfile);
    FILE* f = fopen(tmp, "wb"); if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) { perror("idfile"); return 1; }
    fclose(f); rename(tmp, idfile);
  } else {
    FILE* f = NULL;
    for (int t = 0; t < 600 && !(f = fopen(idfile, "rb")); t++) usleep(100000);
    if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "rank %d: no communicator id in %s\n", rank, idfile); return 1; }
    fclose(f);
  }
  dthip_ctx* ctx = NULL;
  CHECK(dthip_init(rank % ndev, NULL, &ctx));
  CHECK(dthip_comm_init(ctx, rank, world, id));
  const int64_t lo = rank * n / world, hi = (rank + 1) * n / world;
  int64_t* k = malloc(sizeof(int64_t) * (hi - lo + 1)); double* v = malloc(sizeof(double) * (hi - lo + 1));
  fill(lo, hi, k, v);
  dthip_col kc = {k, DTHIP_INT64, 0}, vc = {v, DTHIP_FLOAT64, 0};
  dthip_result* out = NULL;
  CHECK(dthip_sharded_groupby_agg(ctx, &kc, 1, &vc, 1, aggs, 3, hi - lo, DTHIP_NA_FIRST, DTHIP_HOST, &out));
  expect(n);
  int64_t seen = 0, fk, lk;
  const int64_t bad = verify(ctx, out, &seen, &fk, &lk);
  if (bad < 0) { fprintf(stderr, "copy failed: %s\n", dthip_last_error()); return 1; }
  printf("rccl rank %d of %d: %lld rows in, %lld groups out (keys %lld..%lld): %lld mismatches\n", rank, world, (long long)(hi - lo),
         (long long)seen, (long long)fk, (long long)lk, (long long)bad);
  CHECK(dthip_result_free(ctx, out));
  CHECK(dthip_destroy(ctx));
  return bad ? 1 : 0;
}
