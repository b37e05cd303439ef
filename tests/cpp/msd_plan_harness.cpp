// C entry points around datatable_amd/csrc/msd_plan.hpp for tests/test_msd_plan.py (host-only, g++).
#include "../../datatable_amd/csrc/msd_plan.hpp"

extern "C" {

int mp_split(long long n, int bits, unsigned tile, int bucket_rows, int rbmax, int* s1, int* s2, int* rb) {
  const dthip::MsdPlan m = dthip::msd_split(n, bits, tile, bucket_rows, rbmax);
  *s1 = m.s1; *s2 = m.s2; *rb = m.rb;
  return m.ok ? 1 : 0;
}

int mp_overflow(const unsigned* h1, int bins1, const unsigned* h2, int bins2, long long n, unsigned tile) {
  return dthip::msd_overflow_expected(h1, bins1, h2, bins2, n, tile) ? 1 : 0;
}

// returns the number of tiles; the arrays must hold 4 * (sum(sizes) / tile + 2 * nb + 2), 2 * the same, nb + 1 words
int mp_tiles(const unsigned* sizes, unsigned nb, unsigned tile, unsigned tpg, unsigned* tdesc, unsigned* gdesc, unsigned* gfirst, int* ngroups) {
  std::vector<uint32_t> t, g, f;
  dthip::msd_level2_tiles(sizes, nb, tile, tpg, &t, &g, &f);
  for (size_t i = 0; i < t.size(); i++) tdesc[i] = t[i];
  for (size_t i = 0; i < g.size(); i++) gdesc[i] = g[i];
  for (size_t i = 0; i < f.size(); i++) gfirst[i] = f[i];
  *ngroups = (int)(g.size() / 2);
  return (int)(t.size() / 4);
}

}
