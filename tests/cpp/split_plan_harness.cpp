// CPU harness around datatable_amd/csrc/split_plan.hpp (the host-side planning of the multi-GPU key-range partition):
// does per rank what the device kernels do (range and 4096-bin histogram of the valid key images), then calls the
// same reduce_key_ranges / split_bounds the library calls.  Built by tests/test_split_plan.py with g++.
#include "../../datatable_amd/csrc/split_plan.hpp"

extern "C" int sp_bounds(const unsigned long long* imgs, const long long* cuts, int world, unsigned long long na_img,
                         unsigned long long* bounds, int* shift, unsigned long long* gmin) {
  using namespace dthip;
  std::vector<RangeAcc> ranges(world);
  for (int r = 0; r < world; r++) {
    RangeAcc a{~0ULL, 0ULL, 0ULL};
    for (long long i = cuts[r]; i < cuts[r + 1]; i++) {
      const unsigned long long v = imgs[i];
      if (v == na_img) continue;
      a.lo = v < a.lo ? v : a.lo; a.hi = v > a.hi ? v : a.hi; a.nvalid++;
    }
    ranges[r] = a;
  }
  const GlobalRange g = reduce_key_ranges(ranges.data(), world);
  std::vector<unsigned long long> hist((size_t)world * SPLIT_BINS, 0);
  if (g.nvalid)
    for (int r = 0; r < world; r++)
      for (long long i = cuts[r]; i < cuts[r + 1]; i++) {
        const unsigned long long v = imgs[i];
        if (v == na_img) continue;
        const unsigned long long bin = (v - g.gmin) >> g.shift;
        if (bin >= (unsigned long long)SPLIT_BINS) return -1;        // the bin shift must cover the whole range
        hist[(size_t)r * SPLIT_BINS + bin]++;
      }
  std::vector<unsigned long long> b;
  split_bounds(hist.data(), world, g, &b);
  for (size_t k = 0; k < b.size(); k++) bounds[k] = b[k];
  *shift = g.shift; *gmin = g.gmin;
  return 0;
}

// the aggregate path's splitters: every rank's slice imgs[cuts[r] .. cuts[r+1]) is ASCENDING; samples at the positions
// sample_image_kernel reads (floor(i * n / SPLIT_SAMPLES)), then the library's sample_bounds
extern "C" int sp_sample_bounds(const unsigned long long* imgs, const long long* cuts, int world, unsigned long long* bounds) {
  using namespace dthip;
  std::vector<unsigned long long> smp((size_t)world * SPLIT_SAMPLES, 0);
  std::vector<long long> n(world);
  for (int r = 0; r < world; r++) {
    n[r] = cuts[r + 1] - cuts[r];
    for (int i = 0; i < SPLIT_SAMPLES && n[r] > 0; i++)
      smp[(size_t)r * SPLIT_SAMPLES + i] = imgs[cuts[r] + (long long)(((unsigned long long)i * (unsigned long long)n[r]) / SPLIT_SAMPLES)];
  }
  std::vector<unsigned long long> b;
  sample_bounds(smp.data(), n.data(), world, &b);
  for (size_t k = 0; k < b.size(); k++) bounds[k] = b[k];
  return 0;
}

// status agreement over `world` blobs of `stride` bytes, each starting with a ShardHdr
extern "C" int sp_first_failure(const int* rcs, const unsigned* sigs, int world, int* rank, int* sig_ok) {
  using namespace dthip;
  const size_t stride = sizeof(ShardHdr) + 24;
  std::vector<unsigned char> blobs(stride * (size_t)world, 0xEE);
  for (int r = 0; r < world; r++) { const ShardHdr h{rcs[r], sigs[r], (long long)r}; memcpy(blobs.data() + stride * (size_t)r, &h, sizeof(h)); }
  bool ok = true;
  const int rc = first_failure(blobs.data(), stride, world, rank, &ok);
  *sig_ok = ok ? 1 : 0;
  return rc;
}

// sample_bounds on raw all-gathered samples (world x SPLIT_SAMPLES) -- what every rank of comm.hip computes after all-gather A
extern "C" int sp_bounds_from_samples(const unsigned long long* samples, const long long* n, int world, unsigned long long* bounds) {
  std::vector<unsigned long long> b;
  dthip::sample_bounds(samples, n, world, &b);
  for (size_t k = 0; k < b.size(); k++) bounds[k] = b[k];
  return dthip::SPLIT_SAMPLES;
}

// reduce_key_ranges + split_bounds on raw all-gathered ranges / histograms (the rows path after all-gathers A and B)
extern "C" int sp_range(const unsigned long long* ranges3, int world, unsigned long long* gmin, int* shift, unsigned long long* nvalid) {
  using namespace dthip;
  std::vector<RangeAcc> r(world);
  for (int k = 0; k < world; k++) r[k] = RangeAcc{ranges3[3 * k], ranges3[3 * k + 1], ranges3[3 * k + 2]};
  const GlobalRange g = reduce_key_ranges(r.data(), world);
  *gmin = g.gmin; *shift = g.shift; *nvalid = g.nvalid;
  return SPLIT_BINS;
}
extern "C" int sp_bounds_from_hist(const unsigned long long* ranges3, const unsigned long long* hist, int world, unsigned long long* bounds) {
  using namespace dthip;
  std::vector<RangeAcc> r(world);
  for (int k = 0; k < world; k++) r[k] = RangeAcc{ranges3[3 * k], ranges3[3 * k + 1], ranges3[3 * k + 2]};
  const GlobalRange g = reduce_key_ranges(r.data(), world);
  std::vector<unsigned long long> b;
  split_bounds(hist, world, g, &b);
  for (size_t k = 0; k < b.size(); k++) bounds[k] = b[k];
  return 0;
}

// the rows path (round 5): every rank's slice imgs[cuts[r] .. cuts[r+1]) in ROW order; samples at row_sample_pos, sorted
// per rank (what comm.hip's row_sample_kernel + the host's std::sort produce), then sample_bounds_q
extern "C" int sp_row_sample_bounds(const unsigned long long* imgs, const long long* cuts, int world, unsigned long long* bounds) {
  using namespace dthip;
  std::vector<unsigned long long> smp((size_t)world * ROW_SAMPLES, 0);
  std::vector<long long> n(world);
  for (int r = 0; r < world; r++) {
    n[r] = cuts[r + 1] - cuts[r];
    if (n[r] <= 0) continue;
    for (int i = 0; i < ROW_SAMPLES; i++) smp[(size_t)r * ROW_SAMPLES + i] = imgs[cuts[r] + (long long)row_sample_pos((unsigned)i, (unsigned long long)n[r])];
    std::sort(smp.begin() + (size_t)r * ROW_SAMPLES, smp.begin() + (size_t)(r + 1) * ROW_SAMPLES);
  }
  std::vector<unsigned long long> b;
  sample_bounds_q(smp.data(), n.data(), world, ROW_SAMPLES, &b);
  for (size_t k = 0; k < b.size(); k++) bounds[k] = b[k];
  return ROW_SAMPLES;
}
extern "C" unsigned long long sp_row_sample_pos(unsigned i, unsigned long long n) { return dthip::row_sample_pos(i, n); }
extern "C" int sp_bounds_from_row_samples(const unsigned long long* samples, const long long* n, int world, unsigned long long* bounds) {
  std::vector<unsigned long long> b;
  dthip::sample_bounds_q(samples, n, world, dthip::ROW_SAMPLES, &b);
  for (size_t k = 0; k < b.size(); k++) bounds[k] = b[k];
  return dthip::ROW_SAMPLES;
}
extern "C" long long sp_rows_recv_bound(long long total, int world) { return dthip::rows_recv_bound(total, world); }
