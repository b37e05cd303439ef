// keyxform.hpp -- the reference's key transforms, evaluated on the fly from the raw
// column buffers (shared by the radix sort and the bucketed aggregation kernels).
#pragma once
#include "common.hpp"

namespace dthip {

// ---------------------------------------------------------------------------
// key transform: column value -> unsigned key (sort.cc:689-720 _initB,
// :728-776 _initI, :808-845 _initF), evaluated on the fly from the raw column
// ---------------------------------------------------------------------------
// `na` <- the row holds NA (its transformed value is c.na_repl, which a VALID key just outside a guessed range can
// collide with: range checks must look at non-NA rows only, see xform_in_range)
// 24-bit pseudo key of a raw 64-bit key: the partition of the hash combiner splits on its top bits.  Two 32-bit multiplies
// (the two halves by different odd constants, summed): the kernels that evaluate it run one 12288-row tile per workgroup
// between barriers, where the two 64-bit multiplies of a splitmix round sat on the critical path (round 2: 8.8 -> 12+ ms);
// the hash tables inside a bucket use their own, full-strength hash, so this one only has to spread keys over buckets
__device__ __forceinline__ uint32_t hash_pk24(unsigned long long v) {
  const uint32_t h = (uint32_t)v * 0x9E3779B1u + (uint32_t)(v >> 32) * 0x85EBCA77u;
  return h >> 8;
}

__device__ __forceinline__ unsigned long long xform_key_na(const KeyColDev& c, uint32_t row, bool& na) {
  typedef unsigned long long u64;
  na = true;
  switch (c.stype) {
    case DTHIP_KEY_HASH64: {
      na = false;                                     // (an NA key is one more raw value here)
      return (u64)hash_pk24(static_cast<const u64*>(c.data)[row]);
    }
    case DTHIP_BOOL: {
      const uint8_t t = static_cast<const uint8_t*>(c.data)[row];
      if (t == 128) return c.na_repl;
      na = false;
      return c.desc ? (u64)(uint8_t)((uint8_t)(128 - t) >> 6) : (u64)(uint8_t)(t + 1);
    }
    case DTHIP_INT8: {
      const int8_t v = static_cast<const int8_t*>(c.data)[row];
      if (v == INT8_MIN) return c.na_repl;
      na = false;
      const u64 u = (u64)(long long)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_INT16: {
      const int16_t v = static_cast<const int16_t*>(c.data)[row];
      if (v == INT16_MIN) return c.na_repl;
      na = false;
      const u64 u = (u64)(long long)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_INT32: {
      const int32_t v = static_cast<const int32_t*>(c.data)[row];
      if (v == INT32_MIN) return c.na_repl;
      na = false;
      const u64 u = (u64)(long long)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_INT64: {
      const long long v = static_cast<const long long*>(c.data)[row];
      if (v == INT64_MIN) return c.na_repl;
      na = false;
      const u64 u = (u64)v;
      return c.desc ? c.edge - u + c.inc : u - c.edge + c.inc;
    }
    case DTHIP_FLOAT32: {
      const uint32_t t = static_cast<const uint32_t*>(c.data)[row];
      if ((t & 0x7F800000u) == 0x7F800000u && (t & 0x007FFFFFu) != 0) return c.na_repl;
      na = false;
      return c.desc ? (u64)(uint32_t)(t ^ (0x7FFFFFFFu & ((t >> 31) - 1u)))
                    : (u64)(uint32_t)(t ^ (0x80000000u | (0u - (t >> 31))));
    }
    default: {  // FLOAT64
      const u64 t = static_cast<const u64*>(c.data)[row];
      if ((t & 0x7FF0000000000000ULL) == 0x7FF0000000000000ULL && (t & 0x000FFFFFFFFFFFFFULL) != 0)
        return c.na_repl;
      na = false;
      return c.desc ? t ^ (0x7FFFFFFFFFFFFFFFULL & ((t >> 63) - 1ULL))
                    : t ^ (0x8000000000000000ULL | (0ULL - (t >> 63)));
    }
  }
}

__device__ __forceinline__ unsigned long long xform_key(const KeyColDev& c, uint32_t row) {
  bool na;
  return xform_key_na(c, row, na);
}

// Is the transformed value x of a NON-NA key inside the column's planned range?  Valid keys map to
// [inc, inc + xmax]; with a GUESSED range (plan_keys: sampled min / max) a key one below the guessed minimum
// would land on 0 (the NA-first code) and one above the guessed maximum on na_repl (NA last) -- both are
// caught here because the NA rows are excluded before the comparison (unsigned: x - inc wraps for x < inc).
__device__ __forceinline__ bool xform_in_range(const KeyColDev& c, unsigned long long x) { return x - c.inc <= c.xmax; }

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
      bool na;
      const unsigned long long x = xform_key_na(cols[j], row, na);
      b |= !na && !xform_in_range(cols[j], x);
      k |= x << cols[j].shift;
    }
  }
  bad |= b;
  return b ? 0ULL : k;
}

}  // namespace dthip
