// keyxform.hpp -- the reference's key transforms, evaluated on the fly from the raw
// column buffers (shared by the radix sort and the bucketed aggregation kernels).
#pragma once
#include "common.hpp"

namespace dthip {

// ---------------------------------------------------------------------------
// key transform: column value -> unsigned key (sort.cc:689-720 _initB,
// :728-776 _initI, :808-845 _initF), evaluated on the fly from the raw column
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long xform_key(const KeyColDev& c, uint32_t row) {
  typedef unsigned long long u64;
  switch (c.stype) {
    case DTHIP_BOOL: {
      const uint8_t t = static_cast<const uint8_t*>(c.data)[row];
      if (t == 128) return c.na_repl;
      return c.desc ? (u64)(uint8_t)((uint8_t)(128 - t) >> 6) : (u64)(uint8_t)(t + 1);
    }
    case DTHIP_INT8: {
      const int8_t v = static_cast<const int8_t*>(c.data)[row];
      if (v == INT8_MIN) return c.na_repl;
      const u64 u = (u64)(long long)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_INT16: {
      const int16_t v = static_cast<const int16_t*>(c.data)[row];
      if (v == INT16_MIN) return c.na_repl;
      const u64 u = (u64)(long long)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_INT32: {
      const int32_t v = static_cast<const int32_t*>(c.data)[row];
      if (v == INT32_MIN) return c.na_repl;
      const u64 u = (u64)(long long)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_INT64: {
      const long long v = static_cast<const long long*>(c.data)[row];
      if (v == INT64_MIN) return c.na_repl;
      const u64 u = (u64)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_FLOAT32: {
      const uint32_t t = static_cast<const uint32_t*>(c.data)[row];
      if ((t & 0x7F800000u) == 0x7F800000u && (t & 0x007FFFFFu) != 0) return c.na_repl;
      return c.desc ? (u64)(uint32_t)(t ^ (0x7FFFFFFFu & ((t >> 31) - 1u)))
                    : (u64)(uint32_t)(t ^ (0x80000000u | (0u - (t >> 31))));
    }
    default: {  // FLOAT64
      const u64 t = static_cast<const u64*>(c.data)[row];
      if ((t & 0x7FF0000000000000ULL) == 0x7FF0000000000000ULL && (t & 0x000FFFFFFFFFFFFFULL) != 0)
        return c.na_repl;
      return c.desc ? t ^ (0x7FFFFFFFFFFFFFFFULL & ((t >> 63) - 1ULL))
                    : t ^ (0x8000000000000000ULL | (0ULL - (t >> 63)));
    }
  }
}

// packed transformed key of a row: every key column of the stage OR-ed at its bit position.
// Fully unrolled with constant column indices so that the column descriptors are read from
// the kernel-argument segment (a run-time index would copy the whole array to scratch).
__device__ __forceinline__ unsigned long long packed_key(const KeyColDev* cols, int ncols, uint32_t row) {
  unsigned long long k = 0;
#pragma unroll
  for (int j = 0; j < MAX_KEYCOLS; j++)
    if (j < ncols) k |= xform_key(cols[j], row) << cols[j].shift;
  return k;
}

// the same, verifying every column's transformed key against its xmax (speculative key ranges):
// an out-of-range key makes the whole packed key 0 and sets `bad`
__device__ __forceinline__ unsigned long long packed_key_checked(const KeyColDev* cols, int ncols, uint32_t row, bool& bad) {
  unsigned long long k = 0;
  bool b = false;
#pragma unroll
  for (int j = 0; j < MAX_KEYCOLS; j++) {
    if (j < ncols) {
      const unsigned long long x = xform_key(cols[j], row);
      b |= x > cols[j].xmax;
      k |= x << cols[j].shift;
    }
  }
  bad |= b;
  return b ? 0ULL : k;
}

}  // namespace dthip
