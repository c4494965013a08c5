// oracle/philox.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; never included by the product).
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11), restated from the published algorithm.  Known-answer vectors (Random123 kat_vectors)
// are checked in tests/test_oracle_philox.py.
#pragma once
#include <cstdint>

namespace orc {

struct Philox4 {
  uint32_t v[4];
};

// Rounds of the per-base draws (KIND_BASE, two calls per eight sampled bases: the bulk of all calls).  Seven is the
// smallest round count of Philox4x32 that its authors report as passing BigCrush ("Crush-resistant", SC'11 section 5 /
// table 2; Random123's documented minimum); ten -- every other stream here -- is their default with a safety margin.
constexpr int kBaseRounds = 7;

static inline Philox4 philox4x32_r(int rounds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int r = 0; r < rounds; r++) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 out;
  out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
  return out;
}
static inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return philox4x32_r(10, c0, c1, c2, c3, k0, k1);
}

// Counter layout shared with the product (DESIGN.md "RNG addressing"):
//   key      = (seed_lo, seed_hi)
//   c3       = kind | (ctx24 << 8)
// host kinds : ctx24 = (popu_idx << 16) | chr_idx
// device kinds: ctx24 = (mate << 23) | batch_id
enum PhiloxKind : uint32_t {
  KIND_HAP = 1,    // c0 = segment ordinal in (popu,chr), c1 = draw index            -> v[0]
  KIND_GC = 2,     // c0 = window ordinal in segment, c1 = attempt, c2 = seg ordinal -> v[0]: cell and position of the quantile table
  KIND_PLAN = 3,   // c0 = window index in batch, c1 = attempt        -> [pos, isz, strand, -]
  KIND_INDEL = 4,  // c0 = pair slot, c1 = j/8, c2 = 0: 16-bit heads of the eight positions' 64-bit indel draws (word p/2);
                   //                            c2 = 1+q: 48-bit tails of positions 2q (words 0,1) and 2q+1 (words 2,3)
  KIND_AUX = 5,    // c0 = pair slot, c1 = j, c2 = blk                -> flat draw f=4*blk+lane: f=0 length, f>=1 inserted base f-1
  KIND_BASE = 6,   // (kBaseRounds rounds) c0 = pair slot, c1 = i/4, c2 = 0 heads / 1 tails; word i%4 belongs to output position i:
                   //   substitution draw = heads[31:16] << 16 | tails[31:16], quality draw = heads[15:0] << 16 | tails[15:0]
};

}  // namespace orc
