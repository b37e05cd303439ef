// split_plan.hpp -- host-side planning of the multi-GPU key-range partition: global range of the
// order-preserving 64-bit key images, histogram bin width, and the world-1 splitters cut from the summed histogram.
// Plain C++ (no HIP): included by comm.hip and compiled on its own by tests/test_split_plan.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace dthip {

constexpr int SPLIT_BINS = 4096;

struct RangeAcc { unsigned long long lo, hi, nvalid; };   // lo starts at ~0, hi at 0 (valid images only)

struct GlobalRange { unsigned long long gmin, gmax, nvalid; int shift; };

// ranges[r] = rank r's (min image, max image, number of valid keys) -> the global range and the bin shift:
// bin(image) = (image - gmin) >> shift < SPLIT_BINS
inline GlobalRange reduce_key_ranges(const RangeAcc* ranges, int world) {
  GlobalRange g{~0ULL, 0ULL, 0ULL, 0};
  for (int r = 0; r < world; r++) {
    const RangeAcc& a = ranges[r];
    if (a.nvalid) { g.gmin = a.lo < g.gmin ? a.lo : g.gmin; g.gmax = a.hi > g.gmax ? a.hi : g.gmax; g.nvalid += a.nvalid; }
  }
  if (g.nvalid == 0) { g.gmin = 1; g.gmax = 1; }
  const unsigned long long width = g.gmax - g.gmin;
  int bits = 0;
  for (unsigned long long w = width; w; w >>= 1) bits++;
  g.shift = bits > 12 ? bits - 12 : 0;          // (width >> shift) < 4096
  return g;
}

// hist[r * SPLIT_BINS + b] = rank r's count of valid images in bin b.  bounds (world - 1, ascending): an image goes to
// rank  #{j : bounds[j] <= image};  NA images (0, or ~0 when NAs sort last) land on rank 0 / the last rank by themselves.
// Ranks < k receive at most k / world of the valid keys (a boundary is always a bin edge; one heavy bin cannot be cut).
inline void split_bounds(const unsigned long long* hist, int world, const GlobalRange& g, std::vector<unsigned long long>* bounds) {
  std::vector<unsigned long long> h(SPLIT_BINS, 0);
  for (int r = 0; r < world; r++)
    for (int b = 0; b < SPLIT_BINS; b++) h[b] += hist[(size_t)r * SPLIT_BINS + b];
  unsigned long long total = 0;
  for (int b = 0; b < SPLIT_BINS; b++) total += h[b];
  bounds->assign(world > 1 ? world - 1 : 0, ~0ULL);
  if (total == 0) return;
  unsigned long long cum = 0;
  int b = 0;
  for (int k = 1; k < world; k++) {
    const unsigned long long target = (unsigned long long)(((unsigned __int128)total * (unsigned)k) / (unsigned)world);
    while (b < SPLIT_BINS && cum + h[b] <= target) { cum += h[b]; b++; }
    // bins [0, b) hold <= target keys: boundary = start of bin b
    const unsigned __int128 edge = (unsigned __int128)g.gmin + ((unsigned __int128)(unsigned long long)b << g.shift);
    (*bounds)[k - 1] = edge >= (unsigned __int128)~0ULL ? ~0ULL - 1 : (unsigned long long)edge;
  }
}

}  // namespace dthip
