// split_plan.hpp -- host-side planning of the multi-GPU key-range partition: the world-1 splitters cut from
// every rank's quantile samples of the order-preserving 64-bit key images (aggregate path: exact local quantiles; rows
// path, round 5: a stratified random sample), the status header of every all-gathered blob; and the histogram splitters of
// rounds 2-4 (global range, bin width, 4096-bin histogram: two all-gathers) -- no longer on the product path, kept with
// their tests as the alternative whose balance is a hard guarantee (one bin) rather than a statistical one.
// Plain C++ (no HIP): included by comm.hip and compiled on its own by tests/test_split_plan.py.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <utility>
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

// ---- splitters from local quantiles (aggregates: every rank's partial groups ascend in the first key) --------------
constexpr int SPLIT_SAMPLES = 1024;

// samples[r * SPLIT_SAMPLES + i] = image at position floor(i * n[r] / SPLIT_SAMPLES) of rank r's ascending sequence
// (ignored when n[r] == 0).  Every sample stands for n[r] / SPLIT_SAMPLES elements; the boundary of rank k is the
// sample image at which the fair share of rank k - 1 (of what ranks 0..k-2 left) is reached.  All ranks hold the same samples
// and get the same boundaries.  Rank k's share exceeds its fair share by at most sum_r n[r] / SPLIT_SAMPLES elements plus
// the elements that share the boundary image (one key is never cut).
// (q samples per rank: SPLIT_SAMPLES exact local quantiles on the aggregate path, ROW_SAMPLES random-sample quantiles on the
// rows path)
inline void sample_bounds_q(const unsigned long long* samples, const long long* n, int world, int q, std::vector<unsigned long long>* bounds) {
  bounds->assign(world > 1 ? world - 1 : 0, ~0ULL);
  std::vector<std::pair<unsigned long long, double>> pts;
  double total = 0;
  for (int r = 0; r < world; r++) {
    if (n[r] <= 0) continue;
    const double w = (double)n[r] / q;
    for (int i = 0; i < q; i++) pts.emplace_back(samples[(size_t)r * q + i], w);
    total += (double)n[r];
  }
  if (pts.empty()) return;
  std::sort(pts.begin(), pts.end(), [](const std::pair<unsigned long long, double>& a, const std::pair<unsigned long long, double>& b) { return a.first < b.first; });
  // Runs of EQUAL images stay together (one key is never cut).  Round 6: a run heavier than a fair share -- the NA
  // group of a frame with many NAs, a hot key -- used to collapse the boundaries around it onto its image: the ranks before it
  // received nothing and its owner the run plus a fair share on top.  Now a heavy run gets a rank of its own (together with
  // the light keys just before it when that keeps the largest share smaller), and the light ranks split the LIGHT weight that
  // is left evenly among the light ranks that are left, wherever the heavy runs sit in the key order.
  struct Run { unsigned long long img; double w; bool heavy; };
  std::vector<Run> runs;
  for (size_t a = 0; a < pts.size();) {
    size_t e = a; double w = 0;
    while (e < pts.size() && pts[e].first == pts[a].first) { w += pts[e].second; e++; }
    runs.push_back(Run{pts[a].first, w, world > 1 && w > total / world});
    a = e;
  }
  std::vector<double> hw_after(runs.size() + 1, 0.0);        // weight / number of the heavy runs at and after position i
  std::vector<int> hn_after(runs.size() + 1, 0);
  for (size_t a = runs.size(); a-- > 0;) {
    hw_after[a] = hw_after[a + 1] + (runs[a].heavy ? runs[a].w : 0.0);
    hn_after[a] = hn_after[a + 1] + (runs[a].heavy ? 1 : 0);
  }
  double cum = 0;
  size_t i = 0;
  for (int k = 1; k < world; k++) {
    const double assigned = cum;
    const int light_ranks = std::max(1, (world - k + 1) - hn_after[i]);
    const double share = (total - cum - hw_after[i]) / light_ranks;       // a light rank's fair share of the light weight left
    const double target = cum + share;
    while (i < runs.size()) {
      const Run& r = runs[i];
      if (r.heavy) {
        // its own rank -- together with the light keys this rank already holds when that makes the largest share smaller
        // than closing this rank here would (the light ranks after it then are one more)
        const double before = cum - assigned, l_after = total - cum - hw_after[i];
        const int nl_nm = (world - k) - 1 - hn_after[i + 1], nl_m = nl_nm + 1;
        const double big = 4.0 * total + 1.0;
        const double max_nm = std::max(r.w, nl_nm >= 1 ? l_after / nl_nm : (l_after > 0 ? big : 0.0));
        const double max_m = std::max(before + r.w, nl_m >= 1 ? l_after / nl_m : (l_after > 0 ? big : 0.0));
        if (before <= 0 || max_m <= max_nm) { cum += r.w; i++; }
        break;
      }
      if (cum + r.w <= target) { cum += r.w; i++; continue; }
      if (cum <= assigned || (cum + r.w - target) < (target - cum)) { cum += r.w; i++; }
      break;
    }
    // runs [0, i) go to ranks < k: the boundary is the next run's image (everything below it goes to ranks < k)
    (*bounds)[k - 1] = i < runs.size() ? runs[i].img : ~0ULL;
  }
  for (int k = 1; k < world - 1; k++) if ((*bounds)[k] < (*bounds)[k - 1]) (*bounds)[k] = (*bounds)[k - 1];
}
inline void sample_bounds(const unsigned long long* samples, const long long* n, int world, std::vector<unsigned long long>* bounds) {
  sample_bounds_q(samples, n, world, SPLIT_SAMPLES, bounds);
}

// ---- splitters of the ROWS path (round 5): one all-gather instead of two ------------------------------------------------
// Rows are not ordered, so a rank cannot read off exact quantiles; rounds 2-4 therefore all-gathered the key range and then
// a 4096-bin histogram over it (two host-synchronous rounds).  A stratified pseudo-random sample does it in one: sample i
// of a rank is the row at  row_sample_pos(i, n)  -- inside the i-th of ROW_SAMPLES equal strata, at an offset hashed from i
// (a fixed stride would alias with periodic data) --, the ROW_SAMPLES images, sorted, are approximate local quantiles
// (standard error of a share ~ 1 / sqrt(world * ROW_SAMPLES): 0.6 % of the total at 8 ranks), and sample_bounds_q cuts the
// weighted union exactly as on the aggregate path.  The receive buffers are sized for fair share + 1/8 of the total before
// the counts are known; every rank sees the whole count matrix afterwards, so all ranks agree without a message when a
// share exceeds that bound (one key holding most rows: it is never cut) and the exact allocation + status round of rounds
// 2-4 comes back.
constexpr int ROW_SAMPLES = 4096;

inline unsigned long long row_sample_pos(unsigned i, unsigned long long n) {
  const unsigned long long lo = ((unsigned long long)i * n) / ROW_SAMPLES, hi = ((unsigned long long)(i + 1) * n) / ROW_SAMPLES;
  unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL;      // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
  const unsigned long long p = hi > lo ? lo + z % (hi - lo) : lo;
  return p < n ? p : (n ? n - 1 : 0);
}

inline long long rows_recv_bound(long long total, int world) { return total / world + total / 8 + 4096; }

// ---- what every all-gathered blob starts with: status agreement -------------------------------------------------------
struct ShardHdr { int32_t rc; uint32_t sig; long long n; };     // status of the rank so far, query signature, rows / partials

inline uint32_t fnv1a(uint32_t h, const void* p, size_t bytes) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < bytes; i++) { h ^= b[i]; h *= 16777619u; }
  return h;
}

// the first failing rank and its code (0 when none); *sig_ok = every rank sent the same query signature
inline int first_failure(const unsigned char* blobs, size_t stride, int world, int* rank, bool* sig_ok) {
  int rc = 0; *rank = -1; *sig_ok = true;
  ShardHdr h0; memcpy(&h0, blobs, sizeof(h0));
  for (int r = 0; r < world; r++) {
    ShardHdr h; memcpy(&h, blobs + (size_t)r * stride, sizeof(h));
    if (h.rc != 0 && rc == 0) { rc = h.rc; *rank = r; }
    if (h.sig != h0.sig) *sig_ok = false;
  }
  return rc;
}

}  // namespace dthip
