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
