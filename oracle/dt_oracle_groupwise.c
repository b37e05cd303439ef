/*
 * dt_oracle_groupwise.c -- CPU restatement of the group-wise operators that share the
 * Groupby of the DT[i, j, by()] hot path.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as dt_oracle.c: only tests/, smoke() and bench.py's
 * cpu_baseline may load this; the product is libdthip.so).
 *
 * Parity pinning: checked against outputs of the unmodified reference stored in
 * tests/golden/groupwise_cases.npz (generator tests/golden/make_groupwise_golden.py, which
 * also carries the reference's own vectors from tests/test-reduce.py:561-800,901-944,
 * tests/dt/test-cumsum.py, test-cumprod.py, test-cumminmax.py, test-cumcountngroup.py,
 * test-nunique.py).  See tests/test_oracle_groupwise.py.
 *
 * What is restated (paths relative to /root/reference/src/core); every loop is the
 * reference's sequential per-group loop, so float results are bit-identical:
 *   sd        expr/head_reduce_unary.cc:194-216   Welford, NA skipped, count<=1 or NaN m2 -> NA
 *   median    expr/head_reduce_unary.cc:424-470   values sorted inside the group (NA first),
 *                                                 middle / mean of the two middles in U
 *   nunique   expr/head_reduce_unary.cc:377-387   std::set<T> of the valid values
 *   cov/corr  expr/head_reduce_binary.cc:113-135,167-198  pairwise-valid Welford in T
 *   cumsum/cumprod   column/cumsumprod.h:52-92    NA -> 0 / 1, running op, optional reverse
 *   cummin/cummax    column/cumminmax.h:48-98     NA until the first valid, then running
 *   fillna(reverse)  expr/fexpr_fillna.cc:85-117  last valid value seen in the group (next one when reverse)
 *   cumcount/ngroup  column/cumcountngroup.h:55-72
 *   output stypes    expr/fexpr_cumsumprod.cc:72-99, fexpr_cumminmax.cc:87-101,
 *                    head_reduce_unary.cc:221-229,484-491, head_reduce_binary.cc:47-51
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct { const void* data; int32_t stype; int32_t flags; } dto_col;
enum { ST_BOOL = 1, ST_INT8 = 2, ST_INT16 = 3, ST_INT32 = 4, ST_INT64 = 5, ST_FLOAT32 = 6, ST_FLOAT64 = 7 };
enum { OP_SD = 8, OP_MEDIAN = 9, OP_NUNIQUE = 10 };
enum { OP2_COV = 0, OP2_CORR = 1 };
enum { CUM_SUM = 0, CUM_PROD = 1, CUM_MIN = 2, CUM_MAX = 3, CUM_COUNT = 4, CUM_NGROUP = 5, CUM_FILLNA = 6 };

static int is_float(int st) { return st == ST_FLOAT32 || st == ST_FLOAT64; }

/* SentinelFw get_element (column/sentinel_fw.cc:141-181): value + validity */
static int get_int(const dto_col* c, int64_t j, int64_t* out) {
  if (j < 0) return 0;
  switch (c->stype) {
    case ST_BOOL: case ST_INT8: { int8_t v = ((const int8_t*)c->data)[j]; *out = v; return v != INT8_MIN; }
    case ST_INT16: { int16_t v = ((const int16_t*)c->data)[j]; *out = v; return v != INT16_MIN; }
    case ST_INT32: { int32_t v = ((const int32_t*)c->data)[j]; *out = v; return v != INT32_MIN; }
    default: { int64_t v = ((const int64_t*)c->data)[j]; *out = v; return v != INT64_MIN; }
  }
}
static int get_f64(const dto_col* c, int64_t j, double* out) {
  if (j < 0) return 0;
  if (c->stype == ST_FLOAT64) { *out = ((const double*)c->data)[j]; return !isnan(*out); }
  if (c->stype == ST_FLOAT32) { float v = ((const float*)c->data)[j]; *out = (double)v; return !isnan(v); }
  int64_t iv; int ok = get_int(c, j, &iv); *out = (double)iv; return ok;
}
static int get_f32(const dto_col* c, int64_t j, float* out) {
  double d; int ok = get_f64(c, j, &d); *out = (float)d; return ok;
}

int dto_reducex_out_stype(int op, int st) {
  if (op == OP_NUNIQUE) return ST_INT64;
  return st == ST_FLOAT32 ? ST_FLOAT32 : ST_FLOAT64;           /* sd, median: U = float for float32 else double */
}

static void put_fx(void* out, int ost, int64_t g, double v, int valid) {
  if (ost == ST_FLOAT32) ((float*)out)[g] = valid ? (float)v : NAN;
  else ((double*)out)[g] = valid ? v : NAN;
}

typedef struct { double d; int64_t i; int na; } sortel;
static int cmp_f(const void* a, const void* b) {
  const sortel* x = a; const sortel* y = b;
  if (x->na != y->na) return y->na - x->na;       /* NA first */
  return (x->d > y->d) - (x->d < y->d);
}
static int cmp_i(const void* a, const void* b) {
  const sortel* x = a; const sortel* y = b;
  if (x->na != y->na) return y->na - x->na;
  return (x->i > y->i) - (x->i < y->i);
}

int dto_reducex(int op, const dto_col* col, const int32_t* ri, const int32_t* offsets, int64_t ng, void* out)
{
  const int st = col->stype;
  const int isf = is_float(st);
  const int ost = dto_reducex_out_stype(op, st);
  for (int64_t g = 0; g < ng; g++) {
    const int64_t i0 = offsets[g], i1 = offsets[g + 1];
    if (op == OP_SD) {                                            /* head_reduce_unary.cc:194-216 */
      double mean = 0, m2 = 0; int64_t count = 0;
      for (int64_t p = i0; p < i1; p++) {
        double v;
        if (!get_f64(col, ri ? ri[p] : p, &v)) continue;
        count++;
        double t1 = v - mean;
        mean += t1 / (double)count;
        double t2 = v - mean;
        m2 += t1 * t2;
      }
      if (count <= 1 || isnan(m2)) put_fx(out, ost, g, 0, 0);
      else put_fx(out, ost, g, m2 >= 0 ? sqrt(m2 / (double)(count - 1)) : 0.0, 1);
      continue;
    }
    /* median / nunique: sort the group's values, NA first */
    const int64_t m = i1 - i0;
    sortel* e = malloc(sizeof(sortel) * (size_t)(m ? m : 1));
    for (int64_t p = i0; p < i1; p++) {
      sortel* s = &e[p - i0];
      s->d = 0; s->i = 0;
      if (isf) s->na = !get_f64(col, ri ? ri[p] : p, &s->d);
      else s->na = !get_int(col, ri ? ri[p] : p, &s->i);
    }
    qsort(e, (size_t)m, sizeof(sortel), isf ? cmp_f : cmp_i);
    int64_t a = 0;
    while (a < m && e[a].na) a++;
    if (op == OP_NUNIQUE) {                                       /* head_reduce_unary.cc:377-387 */
      int64_t nu = 0;
      for (int64_t q = a; q < m; q++) {
        if (q == a) nu++;
        else if (isf ? (e[q].d != e[q - 1].d) : (e[q].i != e[q - 1].i)) nu++;
      }
      ((int64_t*)out)[g] = nu;
    } else {                                                      /* head_reduce_unary.cc:446-466 */
      if (a == m) { put_fx(out, ost, g, 0, 0); }
      else {
        const int64_t j = (a + m) / 2;
        if ((m - a) & 1) {
          if (ost == ST_FLOAT32) ((float*)out)[g] = (float)e[j].d;
          else ((double*)out)[g] = isf ? e[j].d : (double)e[j].i;
        } else {
          if (ost == ST_FLOAT32) ((float*)out)[g] = ((float)e[j].d + (float)e[j - 1].d) / 2;
          else if (isf) ((double*)out)[g] = (e[j].d + e[j - 1].d) / 2;
          else ((double*)out)[g] = ((double)e[j].i + (double)e[j - 1].i) / 2;
        }
      }
    }
    free(e);
  }
  return 0;
}

int dto_reduce2_out_stype(int sta, int stb) { return (sta == ST_FLOAT32 && stb == ST_FLOAT32) ? ST_FLOAT32 : ST_FLOAT64; }

/* head_reduce_binary.cc:113-135 (cov), :167-198 (corr); T = float only when both inputs are float32 */
int dto_reduce2(int op, const dto_col* ca, const dto_col* cb, const int32_t* ri, const int32_t* offsets,
                int64_t ng, void* out)
{
  const int f32 = dto_reduce2_out_stype(ca->stype, cb->stype) == ST_FLOAT32;
  for (int64_t g = 0; g < ng; g++) {
    const int64_t i0 = offsets[g], i1 = offsets[g + 1];
    if (f32) {
      float mean1 = 0, mean2 = 0, var1 = 0, var2 = 0, cov = 0; int64_t n = 0;
      for (int64_t p = i0; p < i1; p++) {
        float v1, v2;
        const int64_t j = ri ? ri[p] : p;
        const int ok1 = get_f32(ca, j, &v1), ok2 = get_f32(cb, j, &v2);
        if (!(ok1 && ok2)) continue;
        n++;
        float d1 = v1 - mean1, d2 = v2 - mean2;
        mean1 += d1 / (float)n; mean2 += d2 / (float)n;
        float t1 = v1 - mean1, t2 = v2 - mean2;
        cov += t1 * d2; var1 += t1 * d1; var2 += t2 * d2;
      }
      if (op == OP2_COV) ((float*)out)[g] = n > 1 ? cov / (float)(n - 1) : NAN;
      else { float vv = var1 * var2; ((float*)out)[g] = (n > 1 && vv > 0) ? cov / sqrtf(vv) : NAN; }
    } else {
      double mean1 = 0, mean2 = 0, var1 = 0, var2 = 0, cov = 0; int64_t n = 0;
      for (int64_t p = i0; p < i1; p++) {
        double v1, v2;
        const int64_t j = ri ? ri[p] : p;
        const int ok1 = get_f64(ca, j, &v1), ok2 = get_f64(cb, j, &v2);
        if (!(ok1 && ok2)) continue;
        n++;
        double d1 = v1 - mean1, d2 = v2 - mean2;
        mean1 += d1 / (double)n; mean2 += d2 / (double)n;
        double t1 = v1 - mean1, t2 = v2 - mean2;
        cov += t1 * d2; var1 += t1 * d1; var2 += t2 * d2;
      }
      if (op == OP2_COV) ((double*)out)[g] = n > 1 ? cov / (double)(n - 1) : NAN;
      else { double vv = var1 * var2; ((double*)out)[g] = (n > 1 && vv > 0) ? cov / sqrt(vv) : NAN; }
    }
  }
  return 0;
}

int dto_cumulate_out_stype(int op, int st) {
  if (op == CUM_COUNT || op == CUM_NGROUP) return ST_INT64;
  if (op == CUM_SUM || op == CUM_PROD) return st == ST_FLOAT32 ? ST_FLOAT32 : st == ST_FLOAT64 ? ST_FLOAT64 : ST_INT64;
  return st;
}

static void put_int(void* out, int st, int64_t p, int64_t v, int valid) {
  switch (st) {
    case ST_BOOL: case ST_INT8: ((int8_t*)out)[p] = valid ? (int8_t)v : INT8_MIN; break;
    case ST_INT16: ((int16_t*)out)[p] = valid ? (int16_t)v : INT16_MIN; break;
    case ST_INT32: ((int32_t*)out)[p] = valid ? (int32_t)v : INT32_MIN; break;
    default: ((int64_t*)out)[p] = valid ? v : INT64_MIN; break;
  }
}

/* out has offsets[ng] elements, in grouped order */
int dto_cumulate(int op, const dto_col* col, const int32_t* ri, const int32_t* offsets, int64_t ng, int reverse,
                 void* out)
{
  for (int64_t g = 0; g < ng; g++) {
    const int64_t i0 = offsets[g], i1 = offsets[g + 1];
    if (op == CUM_COUNT || op == CUM_NGROUP) {                    /* cumcountngroup.h:55-72 */
      for (int64_t p = i0; p < i1; p++)
        ((int64_t*)out)[p] = op == CUM_COUNT ? (reverse ? i1 - p - 1 : p - i0) : (reverse ? ng - g - 1 : g);
      continue;
    }
    const int st = col->stype;
    const int64_t step = reverse ? -1 : 1;
    const int64_t first = reverse ? i1 - 1 : i0, stop = reverse ? i0 - 1 : i1;
    if (op == CUM_SUM || op == CUM_PROD) {                        /* cumsumprod.h:52-92 */
      if (st == ST_FLOAT64) {
        double acc = 0;
        for (int64_t p = first; p != stop; p += step) {
          double v; int ok = get_f64(col, ri ? ri[p] : p, &v);
          double x = ok ? v : (op == CUM_SUM ? 0.0 : 1.0);
          acc = (p == first) ? x : (op == CUM_SUM ? acc + x : acc * x);
          ((double*)out)[p] = acc;
        }
      } else if (st == ST_FLOAT32) {
        float acc = 0;
        for (int64_t p = first; p != stop; p += step) {
          float v; int ok = get_f32(col, ri ? ri[p] : p, &v);
          float x = ok ? v : (op == CUM_SUM ? 0.0f : 1.0f);
          acc = (p == first) ? x : (op == CUM_SUM ? acc + x : acc * x);
          ((float*)out)[p] = acc;
        }
      } else {
        uint64_t acc = 0;                                         /* int64, wraps */
        for (int64_t p = first; p != stop; p += step) {
          int64_t v; int ok = get_int(col, ri ? ri[p] : p, &v);
          uint64_t x = ok ? (uint64_t)v : (op == CUM_SUM ? 0u : 1u);
          acc = (p == first) ? x : (op == CUM_SUM ? acc + x : acc * x);
          ((int64_t*)out)[p] = (int64_t)acc;
        }
      }
    } else if (op == CUM_FILLNA) {                                /* fexpr_fillna.cc:85-117 (fill_rowindex) */
      /* the reference builds a RowIndex: every row points at the last valid row seen so far in its group (walking
       * backwards when reverse), a leading run of NAs at the group's first row walked -- i.e. at an NA; the result column
       * is the view through it.  Here: the value carried along */
      if (is_float(st)) {
        double acc = 0; int have = 0;
        for (int64_t p = first; p != stop; p += step) {
          double v; int ok = get_f64(col, ri ? ri[p] : p, &v);
          if (ok) { acc = v; have = 1; }
          if (st == ST_FLOAT32) ((float*)out)[p] = have ? (float)acc : NAN;
          else ((double*)out)[p] = have ? acc : NAN;
        }
      } else {
        int64_t acc = 0; int have = 0;
        for (int64_t p = first; p != stop; p += step) {
          int64_t v; int ok = get_int(col, ri ? ri[p] : p, &v);
          if (ok) { acc = v; have = 1; }
          put_int(out, st, p, acc, have);
        }
      }
    } else {                                                      /* cumminmax.h:48-98 */
      const int mn = op == CUM_MIN;
      if (is_float(st)) {
        double acc = 0; int have = 0;
        for (int64_t p = first; p != stop; p += step) {
          double v; int ok = get_f64(col, ri ? ri[p] : p, &v);
          if (ok) {
            if (mn) acc = (have && acc < v) ? acc : v; else acc = (have && acc > v) ? acc : v;
            have = 1;
          }
          if (st == ST_FLOAT32) ((float*)out)[p] = have ? (float)acc : NAN;
          else ((double*)out)[p] = have ? acc : NAN;
        }
      } else {
        int64_t acc = 0; int have = 0;
        for (int64_t p = first; p != stop; p += step) {
          int64_t v; int ok = get_int(col, ri ? ri[p] : p, &v);
          if (ok) {
            if (mn) acc = (have && acc < v) ? acc : v; else acc = (have && acc > v) ? acc : v;
            have = 1;
          }
          put_int(out, st, p, acc, have);
        }
      }
    }
  }
  return 0;
}
