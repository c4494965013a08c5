// sg_tables.cpp -- see sg_tables.h.  Compiled with -ffp-contract=off: the fp64 expressions below
// must round exactly like the reference's (lib/threadpool/ThreadPool.cpp:203-207).
#include "sg_tables.h"

#include <stdexcept>

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

std::vector<uint64_t> row_masses(const double* cdf, int ac) {
  std::vector<uint64_t> n((size_t)ac, 0);
  uint64_t prev = 0;
  for (int k = 0; k + 1 < ac; k++) {
    uint64_t c = count_le(cdf[k]);
    if (c < prev) c = prev;
    n[k] = c - prev;
    prev = c;
  }
  if (ac > 0) n[ac - 1] = (1ull << 32) - prev;
  return n;
}

SubRow encode_sub_row_identity_first(const double* cdf4, int cd) {
  SubRow r;
  const std::vector<uint64_t> n = row_masses(cdf4, 4);
  r.order[0] = (uint8_t)cd;
  for (int k = 0, q = 1; k < 4; k++) if (k != cd) r.order[q++] = (uint8_t)k;
  uint64_t c = 0;
  r.j0 = 0;
  for (int i = 0; i < 3; i++) {
    c += n[r.order[i]];
    r.c[i] = c;
    if (c == 0) r.j0 = (uint32_t)i + 1;
    r.D[i] = c == 0 ? 0u : (uint32_t)(c - 1);
  }
  return r;
}

uint32_t symbols_with_mass(const std::vector<uint64_t>& masses) {
  uint32_t m = 0;
  for (uint64_t v : masses) m += v != 0;
  return m;
}

AliasRow build_alias_row(const std::vector<uint64_t>& masses, uint32_t lgW) {
  const uint32_t W = 1u << lgW;
  const uint64_t C = 1ull << (32 - lgW);
  std::vector<uint64_t> mass(W, 0), thr(W, C);
  std::vector<int> sym(W, -1);
  uint32_t m = 0;
  for (size_t k = 0; k < masses.size(); k++) {
    if (!masses[k]) continue;
    if (m >= W) throw std::runtime_error("build_alias_row: more symbols than columns");
    mass[m] = masses[k];
    sym[m] = (int)k;
    m++;
  }
  // Vose's construction on integers: a column short of C draws is topped up from one that has more than C
  std::vector<uint32_t> small, large;
  for (uint32_t c = 0; c < W; c++) (mass[c] < C ? small : large).push_back(c);
  std::vector<int> lo(sym), hi(sym);
  while (!small.empty() && !large.empty()) {
    const uint32_t s = small.back(); small.pop_back();
    const uint32_t g = large.back(); large.pop_back();
    thr[s] = mass[s];
    hi[s] = sym[g];
    mass[g] -= C - mass[s];
    (mass[g] < C ? small : large).push_back(g);
  }
  if (!small.empty()) throw std::runtime_error("build_alias_row: masses do not add up to 2^32");
  AliasRow r;
  r.thr.resize(W); r.lo.resize(W); r.hi.resize(W);
  for (uint32_t c = 0; c < W; c++) {
    if (thr[c] == C) { thr[c] = 0; hi[c] = lo[c]; }
    if (thr[c] == 0) lo[c] = hi[c];
    if (lo[c] < 0 || hi[c] < 0) throw std::runtime_error("build_alias_row: empty column left without a symbol");
    r.thr[c] = (uint32_t)thr[c]; r.lo[c] = (uint8_t)lo[c]; r.hi[c] = (uint8_t)hi[c];
  }
  return r;
}

}  // namespace sg
