// pred.hpp -- the row predicate `col <cmp> scalar` (or a boolean mask) evaluated on the device, shared by rowindex.hip
// (filters -> RowIndex) and tlsort.hip (the filter fused into the first sort level).
#pragma once
#include "common.hpp"

namespace dthip {

// comparison of a column element with a scalar; NA compares false (NE: true),
// like the reference's comparison FExprs feed init_from_boolean_column
__device__ __forceinline__ bool pred_at(const PredArgs& p, uint32_t i) {
  if (p.is_mask == 2) return (static_cast<const uint32_t*>(p.data)[i >> 5] >> (i & 31)) & 1u;
  if (p.is_mask) {
    const int8_t v = static_cast<const int8_t*>(p.data)[i];
    return v != INT8_MIN && v != 0;
  }
  bool na; double fv = 0; long long iv = 0; bool isf = false;
  switch (p.stype) {
    case DTHIP_BOOL: case DTHIP_INT8: { int8_t v = static_cast<const int8_t*>(p.data)[i]; na = v == INT8_MIN; iv = v; break; }
    case DTHIP_INT16: { int16_t v = static_cast<const int16_t*>(p.data)[i]; na = v == INT16_MIN; iv = v; break; }
    case DTHIP_INT32: { int32_t v = static_cast<const int32_t*>(p.data)[i]; na = v == INT32_MIN; iv = v; break; }
    case DTHIP_INT64: { long long v = static_cast<const long long*>(p.data)[i]; na = v == INT64_MIN; iv = v; break; }
    case DTHIP_FLOAT32: { float v = static_cast<const float*>(p.data)[i]; na = v != v; fv = v; isf = true; break; }
    default: { double v = static_cast<const double*>(p.data)[i]; na = v != v; fv = v; isf = true; break; }
  }
  if (p.cmp == DTHIP_NOTNA) return !na;
  if (p.cmp == DTHIP_ISNA) return na;
  if (isf) {
    switch (p.cmp) {
      case DTHIP_GT: return !na && fv > p.cf;   case DTHIP_GE: return !na && fv >= p.cf;
      case DTHIP_LT: return !na && fv < p.cf;   case DTHIP_LE: return !na && fv <= p.cf;
      case DTHIP_EQ: return !na && fv == p.cf;  default: return na || fv != p.cf;
    }
  }
  switch (p.cmp) {
    case DTHIP_GT: return !na && iv > p.ci;   case DTHIP_GE: return !na && iv >= p.ci;
    case DTHIP_LT: return !na && iv < p.ci;   case DTHIP_LE: return !na && iv <= p.ci;
    case DTHIP_EQ: return !na && iv == p.ci;  default: return na || iv != p.ci;
  }
}

// the predicate on an already loaded element of a float64 / int64 column
__device__ __forceinline__ bool pred_val8(const PredArgs& p, unsigned long long bitsv) {
  if (p.stype == DTHIP_FLOAT64) {
    const double d = __longlong_as_double((long long)bitsv);
    const bool na = d != d;
    switch (p.cmp) {
      case DTHIP_GT: return !na && d > p.cf;   case DTHIP_GE: return !na && d >= p.cf;
      case DTHIP_LT: return !na && d < p.cf;   case DTHIP_LE: return !na && d <= p.cf;
      case DTHIP_EQ: return !na && d == p.cf;  case DTHIP_NE: return na || d != p.cf;
      case DTHIP_NOTNA: return !na;            default: return na;
    }
  }
  const long long iv = (long long)bitsv;
  const bool na = iv == INT64_MIN;
  switch (p.cmp) {
    case DTHIP_GT: return !na && iv > p.ci;   case DTHIP_GE: return !na && iv >= p.ci;
    case DTHIP_LT: return !na && iv < p.ci;   case DTHIP_LE: return !na && iv <= p.ci;
    case DTHIP_EQ: return !na && iv == p.ci;  case DTHIP_NE: return na || iv != p.ci;
    case DTHIP_NOTNA: return !na;             default: return na;
  }
}
}  // namespace dthip
