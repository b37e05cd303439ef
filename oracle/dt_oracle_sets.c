/*
 * dt_oracle_sets.c -- CPU restatement of the set functions and the natural join, the callers of
 * group() next to the DT[i, j, by()] hot path.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as dt_oracle.c).
 *
 * Parity pinning: checked against outputs of the unmodified reference stored in
 * tests/golden/sets_join_cases.npz (generator tests/golden/make_sets_golden.py, incl. the vectors of
 * tests/test-sets.py:129-207 and tests/test-join.py:33-70,169-182,270-279).
 *
 * What is restated (paths relative to /root/reference/src/core):
 *   union / unique       set_funcs.cc:134-177    first row of every group of the stacked column
 *   intersect            set_funcs.cc:222-277    groups holding a row of every source
 *   setdiff              set_funcs.cc:302-334    groups whose rows all come from the first source
 *   symdiff              set_funcs.cc:358-431    groups present in an odd number of sources
 *   natural_join         frame/join.cc:386-446   per X row a binary search of the keyed (sorted,
 *                        unique) J frame; comparator FwCmp<TX,TJ> frame/join.cc:146-200: the X value
 *                        is converted to J's type, an X value J's type cannot represent never
 *                        matches, NA matches NA
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

typedef struct { const void* data; int32_t stype; int32_t flags; } dto_col;
enum { ST_BOOL = 1, ST_INT8 = 2, ST_INT16 = 3, ST_INT32 = 4, ST_INT64 = 5, ST_FLOAT32 = 6, ST_FLOAT64 = 7 };
enum { SET_UNION = 0, SET_INTERSECT = 1, SET_SETDIFF = 2, SET_SYMDIFF = 3 };

int dto_group(const dto_col* keys, int nkeys, int64_t n, int na_pos, int32_t* ri, int32_t* offsets, int64_t* ngroups);

/* out: row indices into the stacked column, one per result element, in ascending value order
 * (NA first); returns the count.  cumsizes[k] = rows of sources 0..k. */
int64_t dto_setop(int op, const dto_col* col, const int64_t* cumsizes, int nsrc, int64_t nrows, int32_t* out)
{
  if (nrows == 0) return 0;
  int32_t* ri = malloc(sizeof(int32_t) * (size_t)nrows);
  int32_t* off = malloc(sizeof(int32_t) * (size_t)(nrows + 1));
  int64_t ng = 0;
  dto_group(col, 1, nrows, 0, ri, off, &ng);
  const int32_t n1 = (int32_t)cumsizes[0];
  int64_t j = 0;
  for (int64_t g = 0; g < ng; g++) {
    const int32_t off0 = off[g], off1 = off[g + 1];
    const int32_t x = ri[off0], y = ri[off1 - 1];
    int take = 0;
    if (op == SET_UNION || nsrc <= 1) take = 1;                                 /* set_funcs.cc:166-172 */
    else if (op == SET_SETDIFF) take = (x < n1 && y < n1);                      /* :324-330 */
    else if (nsrc == 2) {
      if (op == SET_INTERSECT) take = (x < n1 && y >= n1);                      /* :244-253 */
      else take = ((x < n1) == (y < n1));                                       /* :383-389 */
    } else {
      int32_t ii = off0; int kk = 0, all = 1;                                   /* :255-272, :393-409 */
      for (int k = 0; k < nsrc; k++) {
        const int32_t nk = (int32_t)cumsizes[k];
        if (ii >= off1 || ri[ii] >= nk) { all = 0; continue; }
        kk++;
        while (ii < off1 && ri[ii] < nk) ii++;
      }
      take = (op == SET_INTERSECT) ? all : (kk & 1);
    }
    if (take) out[j++] = x;
  }
  free(ri); free(off);
  return j;
}

/* ---- natural join -------------------------------------------------------------------------- */
static int is_float(int st) { return st == ST_FLOAT32 || st == ST_FLOAT64; }

static int get_i64(const dto_col* c, int64_t j, int64_t* out) {
  switch (c->stype) {
    case ST_BOOL: case ST_INT8: { int8_t v = ((const int8_t*)c->data)[j]; *out = v; return v != INT8_MIN; }
    case ST_INT16: { int16_t v = ((const int16_t*)c->data)[j]; *out = v; return v != INT16_MIN; }
    case ST_INT32: { int32_t v = ((const int32_t*)c->data)[j]; *out = v; return v != INT32_MIN; }
    default: { int64_t v = ((const int64_t*)c->data)[j]; *out = v; return v != INT64_MIN; }
  }
}
static int get_f64(const dto_col* c, int64_t j, double* out) {
  if (c->stype == ST_FLOAT64) { *out = ((const double*)c->data)[j]; return !isnan(*out); }
  float v = ((const float*)c->data)[j]; *out = (double)v; return !isnan(v);
}
static void int_limits(int st, int64_t* lo, int64_t* hi) {
  switch (st) {
    case ST_BOOL: case ST_INT8: *lo = INT8_MIN; *hi = INT8_MAX; break;
    case ST_INT16: *lo = INT16_MIN; *hi = INT16_MAX; break;
    case ST_INT32: *lo = INT32_MIN; *hi = INT32_MAX; break;
    default: *lo = INT64_MIN; *hi = INT64_MAX; break;
  }
}

/* the X value of one key column converted to J's domain (FwCmp::set_xrow, join.cc:171-200) */
typedef struct { int valid; int nomatch; int64_t i; double d; float f; } xval;

static xval set_xrow(const dto_col* xc, const dto_col* jc, int64_t row) {
  xval v = {0, 0, 0, 0.0, 0.0f};
  if (is_float(xc->stype)) {
    double x; v.valid = get_f64(xc, row, &x);
    if (!v.valid) return v;
    if (is_float(jc->stype)) { v.d = x; v.f = (float)x; }
    else {
      int64_t lo, hi; int_limits(jc->stype, &lo, &hi);
      if (!(x >= (double)lo && x <= (double)hi && x < 9223372036854775808.0) || (double)(int64_t)x != x) { v.nomatch = 1; return v; }
      v.i = (int64_t)x;
      if (v.i < lo || v.i > hi) v.nomatch = 1;
    }
  } else {
    int64_t x; v.valid = get_i64(xc, row, &x);
    if (!v.valid) return v;
    if (is_float(jc->stype)) { v.d = (double)x; v.f = (float)x; }
    else {
      int64_t lo, hi; int_limits(jc->stype, &lo, &hi);
      if (x < lo || x > hi) { v.nomatch = 1; return v; }
      v.i = x;
    }
  }
  return v;
}

/* FwCmp::cmp_jrow (join.cc:159-168): sign of (J value - X value); NA sorts first and equals NA */
static int cmp_jrow(const dto_col* jc, int64_t row, const xval* x) {
  int jvalid; int r;
  if (jc->stype == ST_FLOAT64) { double jv; jvalid = get_f64(jc, row, &jv); r = (jv > x->d) - (jv < x->d); }
  else if (jc->stype == ST_FLOAT32) { float jv = ((const float*)jc->data)[row]; jvalid = !isnan(jv); r = (jv > x->f) - (jv < x->f); }
  else { int64_t jv; jvalid = get_i64(jc, row, &jv); r = (jv > x->i) - (jv < x->i); }
  if (jvalid && x->valid) return r;
  return jvalid - x->valid;
}

#define MAXK 8
int dto_join_index(const dto_col* xcols, const dto_col* jcols, int nkeys, int64_t xrows, int64_t jrows, int32_t* out)
{
  if (nkeys > MAXK) return -1;
  for (int64_t i = 0; i < xrows; i++) {
    if (jrows == 0) { out[i] = INT32_MIN; continue; }                            /* join.cc:412-417 */
    xval xv[MAXK]; int bad = 0;
    for (int k = 0; k < nkeys; k++) { xv[k] = set_xrow(&xcols[k], &jcols[k], i); bad |= xv[k].nomatch; }
    if (bad) { out[i] = INT32_MIN; continue; }
    int64_t start = 0, end = jrows - 1, found = -1;                              /* binsearch, join.cc:368-380 */
    while (start < end) {
      int64_t mid = (start + end) >> 1;
      int r = 0;
      for (int k = 0; k < nkeys && !r; k++) r = cmp_jrow(&jcols[k], mid, &xv[k]);
      if (r > 0) end = mid; else if (r < 0) start = mid + 1; else { found = mid; break; }
    }
    if (found < 0) {
      int r = 0;
      for (int k = 0; k < nkeys && !r; k++) r = cmp_jrow(&jcols[k], start, &xv[k]);
      if (r == 0) found = start;
    }
    out[i] = found < 0 ? INT32_MIN : (int32_t)found;
  }
  return 0;
}
