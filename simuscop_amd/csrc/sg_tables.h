// sg_tables.h -- exact integer form of the reference's inverse-CDF sampling.
//
// The reference draws r = 2.2204e-16 + (1-2.2204e-16) * x/2^32 from ONE 32-bit mt19937 output x
// (lib/threadpool/ThreadPool.cpp:203-207) and returns the first k with r <= cdf[k], else ac-1
// (lib/mydefine/MyDefine.cpp:176-184).  r is monotone in x, so for every fp64 cdf value c the
// predicate `r <= c` is `x < count_le(c)` for an integer count_le(c) in [0, 2^32].  Tables are
// stored as rows  [k0, T0, T1, ...]  with
//     k0   = number of leading entries that can never be chosen (count 0),
//     T[i] = count_le(cdf[k0+i]) - 1      ("x <= T[i]"),  last kept entry forced to 0xFFFFFFFF,
// padded with 0xFFFFFFFF to a power-of-two width so that the lookup is a branch-free lower bound:
//     k = k0 + #{i : x > T[i]}.
// This reproduces randIndx for every one of the 2^32 possible draws (tests/test_tables.py).
#pragma once
#include <cstdint>
#include <vector>

namespace sg {

// #{x in [0,2^32) : ZERO + (1-ZERO)*x/2^32 <= c}
uint64_t count_le(double c);
// #{x : x/2^32 <= c}   (threadPool->randomDouble(0,1) <= c, Profile.cpp:1560-1561)
uint64_t count_unit_le(double c);
// #{x : x/2^32 < c}    (Profile.cpp:1569-1570)
uint64_t count_unit_lt(double c);

struct Row {
  uint32_t k0 = 0;
  std::vector<uint32_t> T;  // kept thresholds, T.back() == 0xFFFFFFFF
};
Row encode_row(const double* cdf, int ac);

// Compact row: entries that can never be chosen (zero probability mass, i.e. the same count as their
// predecessor) are dropped -- if x <= T[k] == T[k-1] then k-1 is found first -- and every kept entry
// remembers which original index it stands for:  k = sym[#{i : x > T[i]}].  Exact for all 2^32 draws.
struct CompactRow {
  std::vector<uint32_t> T;    // strictly increasing, T.back() == 0xFFFFFFFF
  std::vector<uint8_t> sym;   // original index of each kept entry
};
CompactRow encode_compact_row(const double* cdf, int ac);

// Substitution row (N = 4): {T0, T1, T2, k0};  k = max(k0, (x>T0)+(x>T1)+(x>T2)).
void encode_sub_row(const double* cdf4, uint32_t out[4]);

inline uint32_t pow2_at_least(uint32_t n) {
  uint32_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
inline uint32_t log2u(uint32_t p) {
  uint32_t l = 0;
  while ((1u << l) < p) l++;
  return l;
}

// Host mirror of the device lookup (used by tests and by the ABI self-check).
inline uint32_t row_lookup(const uint32_t* row, uint32_t lg, uint32_t x) {
  const uint32_t* T = row + 1;
  uint32_t pos = 0;
  for (uint32_t step = lg ? (1u << (lg - 1)) : 0; step; step >>= 1)
    if (x > T[pos + step - 1]) pos += step;
  return row[0] + pos;
}

}  // namespace sg
