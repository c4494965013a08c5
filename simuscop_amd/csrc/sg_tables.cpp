// sg_tables.cpp -- see sg_tables.h.  Compiled with -ffp-contract=off: the fp64 expressions below
// must round exactly like the reference's (lib/threadpool/ThreadPool.cpp:203-207).
#include "sg_tables.h"

namespace sg {

static inline double uniform_from_u32(uint32_t x, double start, double end) {
  double number = (double)x;
  return start + (end - start) * ((number - 0.0) / (4294967295.0 - 0.0 + 1.0));
}

template <class Pred>
static uint64_t count_true_prefix(Pred pred) {
  // pred(x) is monotone: true for x < cnt, false for x >= cnt.  Find cnt in [0, 2^32].
  uint64_t lo = 0, hi = 1ull << 32;  // invariant: pred true on [0,lo), false on [hi, 2^32)
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (pred((uint32_t)mid)) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

uint64_t count_le(double c) {
  return count_true_prefix([c](uint32_t x) { return uniform_from_u32(x, 2.2204e-16, 1) <= c; });
}
uint64_t count_unit_le(double c) {
  return count_true_prefix([c](uint32_t x) { return uniform_from_u32(x, 0, 1) <= c; });
}
uint64_t count_unit_lt(double c) {
  return count_true_prefix([c](uint32_t x) { return uniform_from_u32(x, 0, 1) < c; });
}

Row encode_row(const double* cdf, int ac) {
  Row r;
  if (ac <= 0) { r.k0 = 0; r.T.push_back(0xFFFFFFFFu); return r; }
  int k0 = -1;
  std::vector<uint64_t> cnt(ac);
  for (int k = 0; k < ac; k++) {
    cnt[k] = count_le(cdf[k]);
    if (k0 < 0 && cnt[k] >= 1) k0 = k;
  }
  if (k0 < 0) {  // nothing can ever satisfy r <= cdf[k]: randIndx falls through to ac-1
    r.k0 = (uint32_t)(ac - 1);
    r.T.push_back(0xFFFFFFFFu);
    return r;
  }
  r.k0 = (uint32_t)k0;
  for (int k = k0; k < ac; k++) {
    uint32_t t = (k == ac - 1) ? 0xFFFFFFFFu : (uint32_t)(cnt[k] - 1);
    r.T.push_back(t);
    if (t == 0xFFFFFFFFu) break;
  }
  return r;
}

CompactRow encode_compact_row(const double* cdf, int ac) {
  CompactRow r;
  Row full = encode_row(cdf, ac);
  for (size_t i = 0; i < full.T.size(); i++) {
    if (i > 0 && full.T[i] == full.T[i - 1]) continue;  // zero mass: never the first hit
    r.T.push_back(full.T[i]);
    r.sym.push_back((uint8_t)(full.k0 + i));
  }
  return r;
}

void encode_sub_row(const double* cdf4, uint32_t out[4]) {
  uint64_t cnt[4];
  int k0 = -1;
  for (int k = 0; k < 4; k++) {
    cnt[k] = count_le(cdf4[k]);
    if (k0 < 0 && cnt[k] >= 1) k0 = k;
  }
  if (k0 < 0) k0 = 3;
  for (int k = 0; k < 3; k++) out[k] = (k < k0) ? 0u : (uint32_t)(cnt[k] - 1);
  out[3] = (uint32_t)k0;
}

}  // namespace sg
