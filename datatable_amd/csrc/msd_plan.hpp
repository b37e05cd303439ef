// msd_plan.hpp -- host-side planning of the MSD levels of the sort path (api.hip sort_stage): how the key bits are split
// into two scatter digits and the digit the final level orders in LDS, whether the digit histograms already rule the
// levels out, and the ragged tiles / histogram groups of level 2 inside the level-1 buckets.
// Plain C++ (no HIP): included by api.hip and compiled on its own by tests/test_msd_plan.py.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace dthip {

struct MsdPlan { bool ok = false; int s1 = 0, s2 = 0, rb = 0; };

// n rows, `bits` significant bits of the packed key, tile = rows of one radix tile, bucket_rows = wanted size of a final
// bucket, rbmax = widest digit the final level orders (9; 10 in an experiment).  The scatter levels take S = s1 + s2 top
// bits such that a final bucket holds about bucket_rows rows (never more than 56 % of a tile on average: skew needs room),
// the final level the remaining rb = bits - S bits.
inline MsdPlan msd_split(int64_t n, int bits, uint32_t tile, int bucket_rows, int rbmax) {
  MsdPlan m;
  if (n < 2) return m;
  int S = 2;
  while (S < 18 && (n >> S) > (int64_t)bucket_rows) S++;
  if ((n >> S) > (int64_t)(tile * 9 / 16)) return m;
  if (bits - S > rbmax || bits - S < 1) return m;
  m.s1 = (S + 1) / 2; m.s2 = S - m.s1; m.rb = bits - S;
  m.ok = m.s2 >= 1;
  return m;
}

// The levels give up when a final bucket outgrows a tile -- AFTER level 1 and two histogram passes.  The marginal
// histograms of the two scatter digits (h1: level 1, h2: level 2) say when that is certain or likely: rows can only land
// in cells whose two marginal bins are non-empty, so fewer such cells than n / tile means an overflow for sure (few
// distinct keys over a wide range); and if the two digits were independent the fullest cell would hold max1 * max2 / n
// rows (a hot key, clustered keys).
inline bool msd_overflow_expected(const uint32_t* h1, int bins1, const uint32_t* h2, int bins2, int64_t n, uint32_t tile) {
  uint64_t nz1 = 0, nz2 = 0, mx1 = 0, mx2 = 0;
  for (int d = 0; d < bins1; d++) { nz1 += h1[d] != 0; mx1 = std::max<uint64_t>(mx1, h1[d]); }
  for (int d = 0; d < bins2; d++) { nz2 += h2[d] != 0; mx2 = std::max<uint64_t>(mx2, h2[d]); }
  if (nz1 == 0 || nz2 == 0) return true;
  return (double)n / (double)(nz1 * nz2) > (double)tile || (double)mx1 * (double)mx2 / (double)n > (double)tile;
}

// Level 2 works inside every level-1 bucket: tiles never span two buckets.  sizes[b] = rows of level-1 bucket b (they lie
// one after the other from row 0).  tdesc gets 4 words per tile {first row, rows, histogram group, bucket}; gdesc 2 words
// per group {first tile, tiles} (<= tpg tiles of ONE bucket: a group's workgroup walks its tiles and hands every tile the
// digit counts of the group's earlier tiles); gfirst[b] = first group of bucket b (nb + 1 entries).  The first tile of a
// bucket is cut short by (first row mod 4) rows, so that every other tile of the bucket starts on a 16-byte boundary of a
// 4-byte key array and takes the vector-load path.
inline void msd_level2_tiles(const uint32_t* sizes, uint32_t nb, uint32_t tile, uint32_t tpg, std::vector<uint32_t>* tdesc,
                             std::vector<uint32_t>* gdesc, std::vector<uint32_t>* gfirst) {
  tdesc->clear(); gdesc->clear(); gfirst->assign((size_t)nb + 1, 0);
  uint32_t row = 0;
  for (uint32_t b = 0; b < nb; b++) {
    const uint32_t sz = sizes[b];
    (*gfirst)[b] = (uint32_t)(gdesc->size() / 2);
    uint32_t off = 0, t = 0;
    while (off < sz) {
      uint32_t len = (t == 0) ? tile - (row & 3u) : tile;
      if (len > sz - off) len = sz - off;
      if (t % tpg == 0) { gdesc->push_back((uint32_t)(tdesc->size() / 4)); gdesc->push_back(0); }
      (*gdesc)[gdesc->size() - 1]++;
      tdesc->push_back(row + off);
      tdesc->push_back(len);
      tdesc->push_back((uint32_t)(gdesc->size() / 2 - 1));
      tdesc->push_back(b);
      off += len; t++;
    }
    row += sz;
  }
  (*gfirst)[nb] = (uint32_t)(gdesc->size() / 2);
}

}  // namespace dthip
