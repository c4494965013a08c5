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

// ---- measure-preserving rearrangements used by the per-base sampling ----
// A CDF row of `ac` entries partitions the 2^32 draws: outcome k < ac-1 gets count_le(cdf[k]) - count_le(cdf[k-1])
// of them, outcome ac-1 the rest (randIndx's fall-through).  Any draw -> outcome map that keeps these counts samples
// the reference's distribution exactly.
std::vector<uint64_t> row_masses(const double* cdf, int ac);

// Substitution row, identity first: outcomes in the order o = [cd, the other base indexes ascending] (cd = the
// reference base), cumulative masses c0 <= c1 <= c2 of o[0..2];  j = #{i : x >= c_i},  k = o[j].
// Device form {D0, D1, D2}: j = max(j0, (x > D0) + (x > D1) + (x > D2)) with D_i = c_i - 1 and j0 = #{i : c_i == 0}.
struct SubRow {
  uint64_t c[3];
  uint8_t order[4];
  uint32_t D[3];
  uint32_t j0;
};
SubRow encode_sub_row_identity_first(const double* cdf4, int cd);

// Quality row as alias columns: W = 2^lgW columns of C = 2^32 / W draws; column col holds symbol lo for u < thr and hi
// for u >= thr (col = x >> (32 - lgW), u = x & (C - 1)); canonical: thr in [0, C), lo == hi when thr == 0.
struct AliasRow {
  std::vector<uint32_t> thr;
  std::vector<uint8_t> lo, hi;
};
uint32_t symbols_with_mass(const std::vector<uint64_t>& masses);
AliasRow build_alias_row(const std::vector<uint64_t>& masses, uint32_t lgW);  // throws std::runtime_error if masses do not fit

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
