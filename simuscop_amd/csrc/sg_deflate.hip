// sg_deflate.hip -- device side of the block-gzip (BGZF) FASTQ sink (see sg_deflate.h).
//
// One workgroup of 512 lanes turns one 32 KB chunk of FASTQ text into one gzip member: lane = 64
// consecutive input bytes.  The lane -> byte map is aligned to the END of the chunk (a short last
// chunk leaves its leading lanes empty), so that the CRC-32 combine tree uses the same nine "advance by
// 64 * 2^k bytes" operators for every chunk length: leading zero bytes do not change a CRC register
// that starts at zero.
//
// Tokens.  The chunk sits in LDS next to a table of 8192 entries that every position's 8-byte gram is
// hashed into with ds_min: an entry ends up holding the FIRST position of (one of) its grams -- a choice
// that needs no sequential pass and is the same whatever order the lanes run in.  Each lane then walks
// its own 64 bytes greedily: at a position it may copy from the table's candidate (reads of one GC window
// overlap on the template, so most bases of a read were written earlier in the member) or from the
// previous byte (a run), whichever is longer by the rule of sg_deflate.h; a match ends at the lane's end,
// and a lane takes at most six.  The walk only visits positions that can start a match (two bit masks made by
// straight-line code over the lane's 64 bytes), so a wave spends a handful of iterations in it, not 64.  What it
// yields per lane is a 32-byte record: a bit mask of the match starts and six words (length, distance, bit offset
// inside the lane's own bit string).
//   gz_hist_kernel    512 chunks spread over the text: tokens -> histograms of the literal/length and distance symbols
//                     (the host builds the two Huffman codes from them)
//   gz_match_kernel   every chunk: tokens -> records, bits per lane, member size (offsets by the u32 -> u64 scan)
//   gz_encode_kernel  prefix (gzip header + block header); literal codes: one pass over the lane's 64 bytes, the
//                     bit position jumping over the matches; match codes: one pass over the lane's record; both
//                     ds_or into a zeroed staging area; end-of-block, CRC-32, ISIZE; one contiguous copy out
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sg_deflate.h"

namespace sg {

struct DevDeflate {
  const uint8_t* text;
  uint64_t bytes;
  uint32_t n_chunks;
  const uint32_t* code;       // [288] literal/length: reversed code | length << 16
  const uint32_t* len_tok;    // [68] length code + extra bits of a match length | bits << 24
  const uint32_t* dist_code;  // [32] reversed code | length << 16
  const uint32_t* prefix;     // member prefix words (BSIZE = 0)
  uint32_t prefix_words, prefix_bits;
  const uint32_t* crc_tab;    // [4][256]
  const uint32_t* crc_shift;  // [kGzLevels][8][16]
  uint32_t crc_init_full, crc_init_last;
  uint32_t* next;             // [2] chunk counters of the match / encode kernels (zeroed by the host)
  uint32_t* msize;            // [n_chunks] member bytes
  const uint64_t* moff;       // [n_chunks] member offsets
  uint4* rec;                 // [n_chunks * 512 * 2] token records
  uint32_t* lbits;            // [n_chunks * 512] bits of a lane's tokens | bytes its last match takes over from the next lanes << 10
  unsigned long long* hist;   // [320] literal/length counts, then distance counts at 288
  uint8_t* out;
  uint32_t min_run;           // shortest match taken at distance 1 (kGzMinRun)
  uint32_t min_copy;          // shortest copy taken from the table (kGzMinGramMatch)
};

constexpr uint32_t kGzTab = 1u << kGzHashBits;
// members sampled for the token histogram: every stride-th, at most kGzSamples of them
__host__ __device__ inline uint32_t gz_sample_stride(uint32_t n_chunks) { return (n_chunks + kGzSamples - 1u) / kGzSamples; }
constexpr uint32_t kGzTxtPad = 32;  // readable zero bytes past the chunk

// the lane's 64 input bytes as 16 words (bytes before the chunk start read as 0); returns the number of
// leading bytes that are not data
__device__ __forceinline__ uint32_t gz_load_lane(const DevDeflate& D, uint32_t chunk, uint32_t lane, uint32_t n, uint32_t (&w)[16]) {
  const int a = (int)n - (int)(kGzLaneBytes * (kGzThreads - lane));
  const uint8_t* base = D.text + (uint64_t)chunk * kGzChunk;
#pragma unroll
  for (int k = 0; k < 16; k++) w[k] = 0;
  if (a >= 0) {
    __builtin_memcpy(w, base + a, 64);
    return 0;
  }
  if (a + (int)kGzLaneBytes <= 0) return kGzLaneBytes;
  const uint32_t first = (uint32_t)(-a);
#pragma unroll 1
  for (uint32_t k = first; k < kGzLaneBytes; k++) {
    const uint32_t b = base[a + (int)k];
#pragma unroll
    for (int z = 0; z < 16; z++)
      if (z == (int)(k >> 2)) w[z] |= b << ((k & 3u) * 8u);
  }
  return first;
}

// exclusive scan over the workgroup's 512 lanes
__device__ __forceinline__ uint32_t gz_block_scan(uint32_t v, uint32_t* wave_tot, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d, 64);
    if ((int)lane >= d) incl += t;
  }
  if (lane == 63u) wave_tot[wv] = incl;
  __syncthreads();
  uint32_t base = 0, all = 0;
#pragma unroll
  for (uint32_t i = 0; i < kGzThreads / 64; i++) {
    const uint32_t t = wave_tot[i];
    if (i < wv) base += t;
    all += t;
  }
  *total = all;
  return base + incl - v;
}

__device__ __forceinline__ uint64_t gz_lds64(const uint8_t* p) {  // unaligned (ds_read_b64 takes any address on gfx950)
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

// table slot (top bits) and tag (low bits) of an 8-byte gram
__device__ __forceinline__ uint32_t gz_hash(uint32_t lo, uint32_t hi) { return (lo * 0x9E3779B1u + hi) * 0x85EBCA77u; }
constexpr uint32_t kGzTagBits = 32u - 15u;  // entry = position << 17 | tag

// number of leading bytes on which the text at a and at b agrees, at most `cap` (16 bytes per step)
__device__ __forceinline__ uint32_t gz_extend(const uint8_t* txt, uint32_t a, uint32_t b, uint32_t cap) {
  uint32_t l = 0;
#pragma unroll 1
  while (l < cap) {
    struct V { uint64_t lo, hi; } x, y;
    __builtin_memcpy(&x, txt + a + l, 16);
    __builtin_memcpy(&y, txt + b + l, 16);
    const uint64_t d0 = x.lo ^ y.lo, d1 = x.hi ^ y.hi;
    if (d0 | d1) {
      l += d0 ? (uint32_t)__builtin_ctzll(d0) >> 3 : 8u + ((uint32_t)__builtin_ctzll(d1) >> 3);
      break;
    }
    l += 16u;
  }
  return l < cap ? l : cap;
}

__device__ __forceinline__ uint32_t gz_dist_symbol(uint32_t d1, uint32_t* eb) {  // d1 = distance - 1
  if (d1 < 4u) { *eb = 0; return d1; }
  const uint32_t lg = 31u - (uint32_t)__builtin_clz(d1);
  *eb = lg - 1u;
  return 2u * lg + ((d1 >> (lg - 1u)) & 1u);
}
__device__ __forceinline__ uint32_t gz_len_symbol(uint32_t len) {  // 3..258 -> 257..285
  if (len == 258u) return 285u;
  const uint32_t l3 = len - 3u;
  if (l3 < 8u) return 257u + l3;
  const uint32_t lg = 31u - (uint32_t)__builtin_clz(l3);
  return 257u + 4u * lg - 4u + ((l3 >> (lg - 2u)) & 3u);
}

// The gram at the lane's byte k (static): low and high half, from the lane's words and the next lane's first two
#define GZ_GRAM(k, w, nx0, nx1, lo, hi)                                                                              \
  const uint32_t d_ = (k) >> 2, b_ = (k)&3u;                                                                         \
  const uint32_t e0_ = w[d_], e1_ = d_ + 1 < 16 ? w[d_ + 1 < 16 ? d_ + 1 : 0] : nx0,                                 \
                 e2_ = d_ + 2 < 16 ? w[d_ + 2 < 16 ? d_ + 2 : 0] : (d_ + 2 == 16 ? nx0 : nx1);                       \
  const uint32_t lo = b_ ? __builtin_amdgcn_alignbyte(e1_, e0_, b_) : e0_;                                           \
  const uint32_t hi = b_ ? __builtin_amdgcn_alignbyte(e2_, e1_, b_) : e1_;

// What a lane's walk yields: where its matches start, what they cover, their lengths / distances and token bits
struct GzLane {
  uint32_t first;                // leading bytes of the lane that are not data
  uint64_t starts, cover;        // bit k: a match starts at / covers the lane's byte k
  uint32_t mw[kGzLaneMatches];   // length - 3 | (distance - 1) << 6 | 1 << 31, in text order
  uint32_t w[16];                // the lane's 64 bytes
};

// Chunk -> LDS (text frame of 32 KB aligned to the chunk's end, first-occurrence table), then the lane's matches.
// Frame position of the lane's byte k: 64 * lane + k; data starts at kGzChunk - n.
//   1. the gram at every EVEN data position -> table slot by ds_min (entry = position << 17 | tag); a bit per position
//      whose five bytes from the previous one on are equal (a run of >= 4 can start there), another where nine are
//   2. at the positions 0 and 1 (mod 4): does the slot hold an earlier position with this gram's tag?  (Two residues
//      because only even positions are in the table: a copy at an odd distance is seen from odd positions.  A copy that
//      could start between two probed positions is found at the next one and grown backwards, so nothing is lost but
//      the copies shorter than eleven bytes that no probe falls into with eight bytes to go.)
//   3. the probed candidates in text order: verify the gram, extend forwards to the lane's end and backwards to the
//      previous match; then runs (straight from the bit mask) wherever no copy went, six matches per lane in all.
//      Positions inside a run of eight are left to the runs: their gram would be found, far away and no longer.
// FULL: a chunk of kGzChunk bytes (all but a text's last): every lane byte is data, no predicates in 1 and 2.
// (Grams that reach past the chunk's end take zero bytes from the padding; they sit in the chunk's last seven
// positions, where no match of eight bytes has room, so what they leave in the table is never used.)
template <bool FULL>
__device__ __forceinline__ void gz_tokens(const DevDeflate& D, uint32_t c, uint32_t n, uint8_t* txt, uint32_t* tab, GzLane& L) {
  uint32_t lane = threadIdx.x;
  // (opaque per call: otherwise the 64 "position << 17" constants of the unrolled loops below are hoisted out of the
  // kernels' chunk loops and live -- spilled -- across them)
  asm volatile("" : "+v"(lane));
  uint32_t (&w)[16] = L.w;
  uint32_t first = 0;
  if (FULL) __builtin_memcpy(w, D.text + (uint64_t)c * kGzChunk + 64u * lane, 64);
  else first = gz_load_lane(D, c, lane, n, w);
  L.first = first;
#pragma unroll
  for (int k = 0; k < 4; k++) ((uint4*)(txt + 64u * lane))[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
  if (lane < kGzTxtPad / 4u) ((uint32_t*)(txt + kGzChunk))[lane] = 0;
  for (uint32_t i = lane; i < kGzTab + 1u; i += kGzThreads) tab[i] = 0xFFFFFFFFu;  // [kGzTab]: where positions before the data go
  __syncthreads();
  const uint32_t nx0 = lane + 1u < kGzThreads ? ((const uint32_t*)(txt + 64u * (lane + 1u)))[0] : 0u;
  const uint32_t nx1 = lane + 1u < kGzThreads ? ((const uint32_t*)(txt + 64u * (lane + 1u)))[1] : 0u;
  const uint32_t pv = lane ? ((const uint32_t*)(txt + 64u * lane))[-1] : 0u;  // the four bytes before the lane
  const uint32_t q0 = kGzChunk - n;
  const int run_from = (int)q0 - (int)(64u * lane);  // a run needs a previous DATA byte: k > run_from
  uint32_t run_lo = 0, run_hi = 0, cand = 0, run8 = 0;  // cand, run8: bit 2 (k / 4) + (k & 1) for the probed k
#pragma unroll
  for (uint32_t k = 0; k < kGzLaneBytes; k++) {
    GZ_GRAM(k, w, nx0, nx1, lo, hi)
    if ((k & 1u) == 0u) {
      const uint32_t q = 64u * lane + k;
      const uint32_t h = gz_hash(lo, hi);
      const uint32_t slot = h >> (32u - kGzHashBits);
      atomicMin(&tab[FULL ? slot : (k >= first ? slot : kGzTab)], (q << kGzTagBits) | (h & ((1u << kGzTagBits) - 1u)));
    }
    // bytes k-1 .. k+2 against bytes k .. k+3
    uint32_t pm1;  // the word at byte k - 1
    if (k == 0) pm1 = __builtin_amdgcn_alignbyte(w[0], pv, 3);
    else {
      const uint32_t dk = (k - 1) >> 2, bk = (k - 1) & 3u;
      const uint32_t f0 = w[dk], f1 = dk + 1 < 16 ? w[dk + 1 < 16 ? dk + 1 : 0] : nx0;
      pm1 = bk ? __builtin_amdgcn_alignbyte(f1, f0, bk) : f0;
    }
    uint32_t rs = (uint32_t)(pm1 == lo);
    if (!FULL) rs &= (uint32_t)((int)k > run_from);
    else if (k == 0) rs &= (uint32_t)(lane != 0u);
    if (k < 32) run_lo |= rs << (k & 31u);
    else run_hi |= rs << (k & 31u);
    if ((k & 3u) < 2u) run8 |= (rs & (uint32_t)(lo == hi)) << (2u * (k >> 2) + (k & 1u));
    if ((k & 3u) == 3u) __builtin_amdgcn_sched_barrier(0);  // (64 independent chains: left alone the scheduler overlaps them all and spills)
  }
  __syncthreads();
  {
    uint32_t lane2 = lane, first2 = first;  // nothing of the first pass is worth keeping
    asm volatile("" : "+v"(lane2), "+v"(first2));
#pragma unroll
    for (uint32_t k = 0; k < kGzLaneBytes; k++) {
      if ((k & 3u) >= 2u) continue;
      GZ_GRAM(k, w, nx0, nx1, lo, hi)
      const uint32_t q = 64u * lane2 + k;
      const uint32_t h = gz_hash(lo, hi);
      const uint32_t e = tab[h >> (32u - kGzHashBits)];
      // same tag, earlier position: the entry is smaller than what this position's own would be
      uint32_t hit = (uint32_t)(e < ((q << kGzTagBits) | (h & ((1u << kGzTagBits) - 1u)))) & (uint32_t)(((e ^ h) & ((1u << kGzTagBits) - 1u)) == 0u);
      if (!FULL) hit &= (uint32_t)(k >= first2);
      cand |= hit << (2u * (k >> 2) + (k & 1u));
    }
  }
  const uint64_t runm = ((uint64_t)run_hi << 32) | run_lo;
  uint64_t starts = 0, cover = 0, gmask = 0, gq = 0;  // gq: distance - 1 of the copies in text order, fifteen bits each
  uint32_t gq2 = 0, nm = 0;
  // ---- copies ----
  {
    uint32_t todo = cand & ~run8;
    uint32_t lo_lim = first;  // where the previous match ended
#pragma unroll 1
    while (todo != 0u && nm < kGzLaneMatches) {
      const uint32_t bit = (uint32_t)__builtin_ctz(todo);
      const uint32_t k = 4u * (bit >> 1) + (bit & 1u);
      todo &= todo - 1u;
      if (k < lo_lim || k + kGzGram > kGzLaneBytes) continue;
      const uint32_t q = 64u * lane + k;
      const uint64_t own = gz_lds64(txt + q);
      const uint32_t h = gz_hash((uint32_t)own, (uint32_t)(own >> 32));
      const uint32_t cq = tab[h >> (32u - kGzHashBits)] >> kGzTagBits;
      if (cq >= q || gz_lds64(txt + cq) != own) continue;  // (a tag can collide)
      const uint32_t room = kGzLaneBytes - k;
      uint32_t len = 8u + (room > 8u ? gz_extend(txt, cq + 8u, q + 8u, room - 8u) : 0u);
      // backwards: the bytes before both, as far as they agree, down to the previous match and the start of the data
      uint32_t back = k - lo_lim;
      back = back < cq - q0 ? back : cq - q0;
      if (back) {
        const uint64_t x = gz_lds64(txt + q - 8u) ^ gz_lds64(txt + cq - 8u);  // (q >= 8 here: cq >= q0 + back, q > cq)
        const uint32_t same = x ? (uint32_t)__builtin_clzll(x) >> 3 : 8u;
        back = back < same ? back : same;
      }
      const uint32_t s0 = k - back;
      len += back;
      // A short copy from far away (five or six bits of length code, fourteen to sixteen of distance) costs more than
      // its bytes as literals where literals are cheap: quality lines, whose few symbols get two- to four-bit codes.
      // Base lines (four-bit literals) and names (seven-bit digits) gain from eight bytes on.  Told apart by the gram
      // itself: A/C/G/T bytes, or a '#' of a read name in it.
      if (len < D.min_copy) {
        // (three bytes of the gram stand for it: its first, fourth and last -- A/C/G/T by one shift of a bit set each)
        auto acgt = [](uint32_t c) { const uint32_t d = c - 65u; return d < 20u ? (0x80045u >> d) & 1u : 0u; };
        const uint32_t lo32 = (uint32_t)own, hi32 = (uint32_t)(own >> 32);
        const uint32_t votes = acgt(lo32 & 0xFFu) + acgt(lo32 >> 24) + acgt(hi32 >> 24);
        const uint32_t x = lo32 ^ 0x23232323u, y = hi32 ^ 0x23232323u;   // a '#' of a read name in the gram?
        const bool hash = ((((x - 0x01010101u) & ~x) | ((y - 0x01010101u) & ~y)) & 0x80808080u) != 0u;
        if (votes < 3u && !hash) {
          // the probes inside the span just seen would find the same short copy again: on to the first gram that leaves it
          const uint32_t past = s0 + len - 7u;
          const uint32_t below = 2u * (past >> 2) + ((past & 3u) < 2u ? (past & 3u) : 2u);
          todo &= below >= 32u ? 0u : ~((1u << below) - 1u);
          continue;
        }
      }
      if (nm < 4u) gq |= (uint64_t)(q - cq - 1u) << (15u * nm);
      else gq2 |= (q - cq - 1u) << (15u * (nm - 4u));
      nm++;
      const uint64_t span = (len == 64u ? ~0ull : ((1ull << len) - 1ull)) << s0;
      starts |= 1ull << s0;
      gmask |= 1ull << s0;
      cover |= span;
      lo_lim = s0 + len;
      const uint32_t below = 2u * (lo_lim >> 2) + ((lo_lim & 3u) < 2u ? (lo_lim & 3u) : 2u);  // probed positions before lo_lim
      todo &= below >= 32u ? 0u : ~((1u << below) - 1u);
    }
  }
  // ---- runs, where no copy went ----
  {
    uint64_t r = runm & ~cover & ~(first >= 64u ? ~0ull : (1ull << first) - 1ull);
#pragma unroll 1
    while (r != 0ull && nm < kGzLaneMatches) {
      const uint32_t s0 = (uint32_t)__builtin_ctzll(r);
      const uint64_t t = ~(runm >> s0), u = cover >> s0;
      const uint32_t ones = t ? (uint32_t)__builtin_ctzll(t) : 64u - s0;   // positions from s0 on whose next three bytes repeat
      const uint32_t room = u ? (uint32_t)__builtin_ctzll(u) : 64u - s0;   // up to the next copy / the lane's end
      uint32_t len = ones + 3u;
      len = len < room ? len : room;
      const uint64_t stretch = (ones >= 64u ? ~0ull : ((1ull << ones) - 1ull)) << s0;
      if (len >= D.min_run) {
        starts |= 1ull << s0;
        cover |= (len == 64u ? ~0ull : ((1ull << len) - 1ull)) << s0;
        nm++;
      }
      r &= ~stretch & ~cover;
    }
  }
  L.starts = starts;
  L.cover = cover;
  // ---- the matches in text order: length from the masks, distance from the queue (copies) or 1 (runs) ----
  {
    uint64_t st = starts;
#pragma unroll
    for (uint32_t j = 0; j < kGzLaneMatches; j++) {
      uint32_t v = 0;
      if (st != 0ull) {
        const uint32_t s0 = (uint32_t)__builtin_ctzll(st);
        st &= st - 1ull;
        const uint64_t stop = (~cover | st) >> s0;
        const uint32_t len = stop ? (uint32_t)__builtin_ctzll(stop) : 64u - s0;
        uint32_t d1 = 0;
        if ((gmask >> s0) & 1ull) {
          d1 = (uint32_t)gq & 0x7FFFu;
          gq = (gq >> 15) | ((uint64_t)(gq2 & 0x7FFFu) << 45);
          gq2 >>= 15;
        }
        v = (len - 3u) | (d1 << 6) | 0x80000000u;
      }
      L.mw[j] = v;
    }
  }
}

// One copy that runs through several lanes leaves as ONE token.  gz_tokens cuts every match at its lane's end (that is
// what makes the lanes independent); a read's 151 bases, copied from an earlier read of the same window, so became three
// or four matches of the same distance -- each with its own length and distance code, ~20 bits.  Inside every aligned
// group of four lanes (256 bytes, the most a group's copy can reach; the format allows 258): a lane's FIRST match, if it
// starts at the lane's first byte with the distance of the previous lane's LAST match, which ends at that lane's last
// byte, is ABSORBED -- it emits nothing, the earlier match grows by its length, and through a lane that one match covers
// whole the copy runs on.  Order-independent like the rest: every lane decides from its neighbours' descriptors.
//   returns: bit 0 this lane's first match is absorbed; bits 8.. bytes this lane's last match takes over from later lanes
//   scratch: 2 x kGzThreads words (the hash table's space: nobody reads the table any more)
__device__ __forceinline__ uint32_t gz_merge(const GzLane& L, uint32_t* scratch) {
  const uint32_t lane = threadIdx.x;
  const uint32_t nm = (uint32_t)__popcll(L.starts);
  uint32_t last = 0;
#pragma unroll
  for (uint32_t j = 0; j < kGzLaneMatches; j++) last = (j + 1u == nm) ? L.mw[j] : last;
  // descriptor: length | distance - 1 << 8 | 1 << 31, 0 = none
  const bool f_ok = L.first == 0u && (L.starts & 1ull) != 0ull;
  const bool e_ok = nm != 0u && (L.cover >> 63) != 0ull;
  const uint32_t fd = f_ok ? (((L.mw[0] & 63u) + 3u) | (((L.mw[0] >> 6) & 0x7FFFu) << 8) | 0x80000000u) : 0u;
  const uint32_t ed = e_ok ? (((last & 63u) + 3u) | (((last >> 6) & 0x7FFFu) << 8) | 0x80000000u) : 0u;
  __syncthreads();   // every lane is done with the table
  scratch[lane] = fd;
  scratch[kGzThreads + lane] = ed;
  __syncthreads();
  const uint32_t g = lane & 3u, base = lane - g;
  uint32_t F[4], E[4];
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++) { F[j] = scratch[base + j]; E[j] = scratch[kGzThreads + base + j]; }
  bool ab[4], whole[4];
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++) {
    ab[j] = j > 0u && (F[j] >> 31) && (E[j > 0u ? j - 1u : 0u] >> 31) && ((F[j] ^ E[j > 0u ? j - 1u : 0u]) & 0x7FFFFF00u) == 0u;
    whole[j] = (F[j] >> 31) && (F[j] & 0xFFu) == 64u;
  }
  uint32_t mine = 0, extra = 0;
  bool run = true;
#pragma unroll
  for (uint32_t j = 0; j < 4u; j++) {
    if (j == g) mine = ab[j] ? 1u : 0u;
    // lanes after this one: absorbed one after the other while the copy goes on
    if (j > g) {
      run = run && ab[j];
      if (run) extra += F[j] & 0xFFu;
      run = run && whole[j];
    }
  }
  // this lane's last match only grows if it is a match of its own: not a whole-lane match that was itself absorbed
  if (!(E[g] >> 31) || (mine && whole[g])) extra = 0;
  return mine | (extra << 8);
}

__global__ __launch_bounds__(kGzThreads, 4) void gz_hist_kernel(DevDeflate D) {
  extern __shared__ uint32_t gz_smem[];
  uint32_t* tab = gz_smem;                                  // [kGzTab]
  uint8_t* txt = (uint8_t*)(tab + kGzTab + 4);              // [kGzChunk + pad]
  uint32_t* h = (uint32_t*)(txt + kGzChunk + kGzTxtPad);    // [320]
  for (uint32_t i = threadIdx.x; i < 320u; i += kGzThreads) h[i] = 0;
  const uint32_t stride = gz_sample_stride(D.n_chunks);
  const uint32_t n_samples = (D.n_chunks + stride - 1u) / stride;
  for (uint32_t sidx = blockIdx.x; sidx < n_samples; sidx += gridDim.x) {
    const uint32_t c = sidx * stride;
    const uint64_t left = D.bytes - (uint64_t)c * kGzChunk;
    const uint32_t n = left < kGzChunk ? (uint32_t)left : kGzChunk;
    GzLane L;
    if (n == kGzChunk) gz_tokens<true>(D, c, n, txt, tab, L);
    else gz_tokens<false>(D, c, n, txt, tab, L);
    const uint32_t mg = gz_merge(L, tab);
    const uint32_t nm_h = (uint32_t)__popcll(L.starts);
#pragma unroll
    for (uint32_t j = 0; j < kGzLaneMatches; j++) {
      const uint32_t v = L.mw[j];
      if ((v >> 31) && !(j == 0u && (mg & 1u))) {
        uint32_t eb;
        atomicAdd(&h[gz_len_symbol((v & 63u) + 3u + (j + 1u == nm_h ? mg >> 8 : 0u))], 1u);
        atomicAdd(&h[288u + gz_dist_symbol((v >> 6) & 0x7FFFu, &eb)], 1u);
      }
    }
#pragma unroll
    for (uint32_t k = 0; k < kGzLaneBytes; k++)
      atomicAdd(&h[(k >= L.first && !((L.cover >> k) & 1ull)) ? (L.w[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu : 287u], 1u);  // [287]: unused slot
    if (threadIdx.x == 0) atomicAdd(&h[256], 1u);
    __syncthreads();
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 320u; i += kGzThreads)
    if (h[i]) atomicAdd(&D.hist[i], (unsigned long long)h[i]);
}

__global__ __launch_bounds__(kGzThreads, 4) void gz_match_kernel(DevDeflate D) {
  extern __shared__ uint32_t gz_smem[];
  uint32_t* tab = gz_smem;                                  // [kGzTab]
  uint8_t* txt = (uint8_t*)(tab + kGzTab + 4);              // [kGzChunk + pad]
  uint32_t* code = (uint32_t*)(txt + kGzChunk + kGzTxtPad); // [288] literal/length, [68] length tokens, [32] distance
  uint32_t* len_tok = code + 288;
  uint32_t* dist_code = len_tok + kGzLenTokens;
  uint32_t* wave_tot = dist_code + 32;                      // [8]
  uint8_t* cl8 = (uint8_t*)(wave_tot + 8);                  // [256] code length of every literal
  __shared__ uint32_t next_c;
  for (uint32_t i = threadIdx.x; i < 288u; i += kGzThreads) code[i] = D.code[i];
  for (uint32_t i = threadIdx.x; i < kGzLenTokens; i += kGzThreads) len_tok[i] = D.len_tok[i];
  for (uint32_t i = threadIdx.x; i < 32u; i += kGzThreads) dist_code[i] = D.dist_code[i];
  for (uint32_t i = threadIdx.x; i < 256u; i += kGzThreads) cl8[i] = (uint8_t)(D.code[i] >> 16);
  __syncthreads();
  for (uint32_t c = blockIdx.x; c < D.n_chunks;) {
    const uint64_t left = D.bytes - (uint64_t)c * kGzChunk;
    const uint32_t n = left < kGzChunk ? (uint32_t)left : kGzChunk;
    GzLane L;
    if (n == kGzChunk) gz_tokens<true>(D, c, n, txt, tab, L);
    else gz_tokens<false>(D, c, n, txt, tab, L);
    // copies that run on through the next lanes become one token (gz_merge)
    const uint32_t mg = gz_merge(L, tab);
    const uint32_t nm = (uint32_t)__popcll(L.starts);
    if (mg & 1u) L.mw[0] |= 0x7FFFu << 6;   // an absorbed match: no distance this large exists inside a member
    // bits of the matches, in order, one byte each
    uint64_t jumps = 0;
#pragma unroll
    for (uint32_t j = 0; j < kGzLaneMatches; j++) {
      const uint32_t v = L.mw[j];
      if ((v >> 31) && !(j == 0u && (mg & 1u))) {
        uint32_t eb;
        const uint32_t ds = gz_dist_symbol((v >> 6) & 0x7FFFu, &eb);
        const uint32_t len = (v & 63u) + 3u + (j + 1u == nm ? mg >> 8 : 0u);
        jumps |= (uint64_t)((len_tok[len] >> 24) + (dist_code[ds] >> 16) + eb) << (8u * j);
      }
    }
    // Code lengths of the lane's literals and the bit offset at which each match starts.  (Until round 3 one pass over the
    // 64 bytes carrying both along: ~1,500 of the kernel's ~4,400 instructions per lane.)  Now by words: the four code
    // lengths of a word from a byte table, packed; covered and non-data bytes masked out; v_sad_u8 sums a word's four
    // bytes, accumulating, which gives the literal bits up to every word boundary; a match start needs the sum up to its
    // word and the bytes of its word below it -- indexed by a lane-variable word number, so the packed lengths and the
    // prefix sums go through a private LDS record (25 words per lane: an odd stride, no bank conflicts; the text and the
    // hash table are dead by now, gz_merge's scratch words excepted).
    uint32_t* rec_lds = tab + 2u * kGzThreads + threadIdx.x * 25u;
    uint64_t lit = ~L.cover;                                               // bit k: byte k is a literal
    lit = L.first >= 64u ? 0ull : lit & ~((1ull << L.first) - 1ull);
    uint32_t run_sum = 0, prev_sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16u; i++) {
      const uint32_t wv_ = L.w[i];
      const uint32_t packed = (uint32_t)cl8[wv_ & 0xFFu] | ((uint32_t)cl8[(wv_ >> 8) & 0xFFu] << 8) | ((uint32_t)cl8[(wv_ >> 16) & 0xFFu] << 16) |
                              ((uint32_t)cl8[wv_ >> 24] << 24);
      const uint32_t m4 = (uint32_t)(lit >> (4u * i)) & 0xFu;
      const uint32_t keep = ((m4 * 0x00204081u) & 0x01010101u) * 0xFFu;    // four mask bits -> four mask bytes
      const uint32_t cw_ = packed & keep;
      rec_lds[i] = cw_;
      run_sum = __builtin_amdgcn_sad_u8(cw_, 0u, run_sum);                 // + the word's four bytes
      if (i & 1u) rec_lds[16u + (i >> 1)] = (prev_sum & 0xFFFFu) | (run_sum << 16);   // inclusive prefix of words i-1, i
      else prev_sum = run_sum;
    }
    uint32_t bits = 0;   // token bits of the matches before the current one
    uint32_t mw[kGzLaneMatches];
    {
      uint64_t st = L.starts;
#pragma unroll
      for (uint32_t j = 0; j < kGzLaneMatches; j++) {
        uint32_t off = 0;
        if (st != 0ull) {
          const uint32_t s0 = (uint32_t)__builtin_ctzll(st);
          st &= st - 1ull;
          const uint32_t wj = s0 >> 2, bj = s0 & 3u;
          const uint32_t below = rec_lds[wj] & ((1u << (8u * bj)) - 1u);
          const uint32_t pw = wj ? ((const uint16_t*)(rec_lds + 16))[wj - 1u] : 0u;
          off = pw + __builtin_amdgcn_sad_u8(below, 0u, 0u) + bits;
          bits += (uint32_t)(jumps >> (8u * j)) & 0xFFu;
        }
        mw[j] = L.mw[j] | ((off & 0x3FFu) << 21);
      }
    }
    bits += run_sum;
    // record: the start mask and six words (length - 3 | distance - 1 << 6 | bit offset in the lane << 21 | 1 << 31)
    uint4* rec = D.rec + ((size_t)c * kGzThreads + threadIdx.x) * 2;
    rec[0] = make_uint4((uint32_t)L.starts, (uint32_t)(L.starts >> 32), mw[0], mw[1]);
    rec[1] = make_uint4(mw[2], mw[3], mw[4], mw[5]);
    D.lbits[(size_t)c * kGzThreads + threadIdx.x] = bits | ((mg >> 8) << 10);
    uint32_t total;
    gz_block_scan(bits, wave_tot, &total);
    if (threadIdx.x == 0) {
      D.msize[c] = (D.prefix_bits + total + (code[256] >> 16) + 7u) / 8u + 8u;
      next_c = gridDim.x + atomicAdd(D.next, 1u);
    }
    __syncthreads();
    c = next_c;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kGzThreads) void gz_encode_kernel(DevDeflate D, uint32_t stage_words) {
  extern __shared__ uint32_t gz_smem[];
  uint32_t* stage = gz_smem;                       // [stage_words]
  uint32_t* code = stage + stage_words;            // [288]
  uint32_t* len_tok = code + 288;                  // [kGzLenTokens]
  uint32_t* dist_code = len_tok + kGzLenTokens;    // [32]
  uint32_t* crc_tab = dist_code + 32;              // [4][256]
  uint32_t* crc_shift = crc_tab + 1024;            // [kGzLevels][8][16]
  uint32_t* crcs = crc_shift + kGzLevels * 128;    // [kGzThreads]
  uint32_t* wave_tot = crcs + kGzThreads;          // [8]
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < 288u; i += kGzThreads) code[i] = D.code[i];
  for (uint32_t i = lane; i < kGzLenTokens; i += kGzThreads) len_tok[i] = D.len_tok[i];
  for (uint32_t i = lane; i < 32u; i += kGzThreads) dist_code[i] = D.dist_code[i];
  for (uint32_t i = lane; i < 1024; i += kGzThreads) crc_tab[i] = D.crc_tab[i];
  for (uint32_t i = lane; i < kGzLevels * 128; i += kGzThreads) crc_shift[i] = D.crc_shift[i];
  __syncthreads();
  __shared__ uint32_t next_c;
  // chunks beyond a workgroup's first come from a counter (CUs do not all run at the same speed)
  for (uint32_t c = blockIdx.x; c < D.n_chunks;) {
    const uint64_t left = D.bytes - (uint64_t)c * kGzChunk;
    const uint32_t n = left < kGzChunk ? (uint32_t)left : kGzChunk;
    const uint32_t msize = D.msize[c];
    const uint32_t out_words = (msize + 3u) / 4u;
    for (uint32_t i = lane; i < out_words; i += kGzThreads) stage[i] = 0;
    uint32_t w[16];
    const uint32_t first = gz_load_lane(D, c, lane, n, w);
    const uint4* rec = D.rec + ((size_t)c * kGzThreads + lane) * 2;
    const uint4 ra = rec[0], rb = rec[1];
    const uint32_t lb = D.lbits[(size_t)c * kGzThreads + lane];
    const uint32_t bits = lb & 0x3FFu, extra = lb >> 10;   // extra: bytes of the next lanes the lane's last match copies too
    uint32_t total;
    const uint32_t excl = gz_block_scan(bits, wave_tot, &total);  // its barrier also orders the zeroing above
    const uint32_t base = D.prefix_bits + excl;
    // ---- prefix, BSIZE ----
    if (lane < D.prefix_words) atomicOr(&stage[lane], D.prefix[lane]);
    if (lane == 0) atomicOr(&stage[4], (msize - 1u) & 0xFFFFu);  // bytes 16-17 of the member
    // ---- the lane's matches: codes at base + their recorded offset; the bytes they cover; their bit counts ----
    const uint64_t starts = ((uint64_t)ra.y << 32) | ra.x;
    uint64_t cover = 0, jumps = 0, st = starts;
    {
      const uint32_t mws[kGzLaneMatches] = {ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll
      for (uint32_t j = 0; j < kGzLaneMatches; j++) {
        const uint32_t v = mws[j];
        if (v >> 31) {
          const uint32_t len = (v & 63u) + 3u, d1 = (v >> 6) & 0x7FFFu, off = (v >> 21) & 0x3FFu;
          const uint32_t s = (uint32_t)__builtin_ctzll(st);
          st &= st - 1ull;
          cover |= (len == 64u ? ~0ull : ((1ull << len) - 1ull)) << s;
          if (d1 == 0x7FFFu) continue;   // absorbed by the previous lane's last match: covered, no token
          uint32_t eb;
          const uint32_t ds = gz_dist_symbol(d1, &eb);
          const uint32_t lt = len_tok[len + (st == 0ull ? extra : 0u)], dc = dist_code[ds];
          const uint32_t n1 = lt >> 24, n2 = dc >> 16;
          const uint64_t tok = (uint64_t)(lt & 0xFFFFFFu) | ((uint64_t)(dc & 0xFFFFu) << n1) | ((uint64_t)(d1 & ((1u << eb) - 1u)) << (n1 + n2));
          const uint32_t nb = n1 + n2 + eb;  // <= 48
          jumps |= (uint64_t)nb << (8u * j);
          const uint32_t pos = base + off, wi = pos >> 5, sh = pos & 31u;
          const uint64_t lo = tok << sh;
          atomicOr(&stage[wi], (uint32_t)lo);
          if (sh + nb > 32u) atomicOr(&stage[wi + 1u], (uint32_t)(lo >> 32));
          if (sh + nb > 64u) atomicOr(&stage[wi + 2u], (uint32_t)(tok >> (64u - sh)));
        }
      }
    }
    // ---- literal codes: every byte no match covers, the bit position stepping over the matches ----
    {
      uint32_t pos = base;
#pragma unroll
      for (uint32_t k = 0; k < kGzLaneBytes; k++) {
        if ((starts >> k) & 1ull) {
          pos += (uint32_t)jumps & 0xFFu;
          jumps >>= 8;
        }
        if (k >= first && !((cover >> k) & 1ull)) {
          const uint32_t e = code[(w[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu];
          const uint32_t cb = e & 0xFFFFu, nb = e >> 16, wi = pos >> 5, sh = pos & 31u;
          atomicOr(&stage[wi], cb << sh);
          if (sh + nb > 32u) atomicOr(&stage[wi + 1u], cb >> (32u - sh));
          pos += nb;
        }
      }
      if (lane == kGzThreads - 1u) {  // end-of-block after the chunk's last token
        const uint32_t e = code[256];
        const uint32_t cb = e & 0xFFFFu, nb = e >> 16, wi = pos >> 5, sh = pos & 31u;
        atomicOr(&stage[wi], cb << sh);
        if (sh + nb > 32u) atomicOr(&stage[wi + 1u], cb >> (32u - sh));
      }
    }
    // ---- CRC-32 of the chunk ----
    {
      uint32_t s = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        s ^= w[k];
        s = crc_tab[768 + (s & 0xFFu)] ^ crc_tab[512 + ((s >> 8) & 0xFFu)] ^ crc_tab[256 + ((s >> 16) & 0xFFu)] ^ crc_tab[s >> 24];
      }
      crcs[lane] = s;
      for (uint32_t k = 0; k < kGzLevels; k++) {
        __syncthreads();
        if ((lane & ((2u << k) - 1u)) == 0u) {
          const uint32_t x = crcs[lane], y = crcs[lane + (1u << k)];
          uint32_t r = y;  // x advanced over the bytes y covers: the operator nibble by nibble
#pragma unroll
          for (uint32_t i = 0; i < 8u; i++) r ^= crc_shift[k * 128u + i * 16u + ((x >> (4u * i)) & 15u)];
          crcs[lane] = r;
        }
      }
      if (lane == 0) {
        const uint32_t init = n == kGzChunk ? D.crc_init_full : D.crc_init_last;
        const uint64_t trailer = (uint64_t)(~(init ^ crcs[0])) | ((uint64_t)n << 32);
        const uint32_t tb = msize - 8u, tw = tb >> 2, sh = (tb & 3u) * 8u;
        atomicOr(&stage[tw], (uint32_t)(trailer << sh));
        atomicOr(&stage[tw + 1u], (uint32_t)((trailer << sh) >> 32));
        if (sh) atomicOr(&stage[tw + 2u], (uint32_t)(trailer >> (64u - sh)));
      }
    }
    __syncthreads();
    // ---- copy out ----
    uint8_t* dst = D.out + D.moff[c];
    const uint8_t* sb = (const uint8_t*)stage;
    for (uint32_t i = lane * 16u; i < msize; i += kGzThreads * 16u) {
      if (i + 16u <= msize) {
        uint4 v = *(const uint4*)(sb + i);
        __builtin_memcpy(dst + i, &v, 16);
      } else {
        for (uint32_t b = i; b < msize; b++) dst[b] = sb[b];
      }
    }
    if (lane == 0) next_c = gridDim.x + atomicAdd(D.next + 1, 1u);
    __syncthreads();
    c = next_c;
    __syncthreads();
  }
}

// ---- launchers -------------------------------------------------------------------------------------
static size_t gz_token_lds() { return ((size_t)kGzTab + 4 + 288 + kGzLenTokens + 32 + 8 + 64 + 64) * 4 + kGzChunk + kGzTxtPad; }
void launch_gz_hist(const void* d, uint32_t n_chunks, hipStream_t s) {
  if (!n_chunks) return;
  const size_t lds = gz_token_lds();
  (void)hipFuncSetAttribute((const void*)gz_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t stride = gz_sample_stride(n_chunks);
  const uint32_t grid = (n_chunks + stride - 1u) / stride;  // <= kGzSamples: one workgroup each, one round on 256 CUs
  hipLaunchKernelGGL(gz_hist_kernel, dim3(grid), dim3(kGzThreads), lds, s, *(const DevDeflate*)d);
}
uint32_t gz_stage_words(uint32_t prefix_bits) {  // worst case: 15 bits per byte (a match token is <= 48 bits for >= 4 bytes)
  return ((prefix_bits + 15u * kGzChunk + 15u + 7u) / 8u + 8u + 3u) / 4u + 4u;
}
void launch_gz_match(const void* d, uint32_t n_chunks, hipStream_t s) {
  if (!n_chunks) return;
  const size_t lds = gz_token_lds();
  (void)hipFuncSetAttribute((const void*)gz_match_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t grid = n_chunks < 512u ? n_chunks : 512u;  // two workgroups per CU fit in LDS
  hipLaunchKernelGGL(gz_match_kernel, dim3(grid), dim3(kGzThreads), lds, s, *(const DevDeflate*)d);
}
void launch_gz_encode(const void* d, uint32_t n_chunks, uint32_t prefix_bits, hipStream_t s) {
  if (!n_chunks) return;
  const uint32_t sw = gz_stage_words(prefix_bits);
  const size_t lds = ((size_t)sw + 288 + kGzLenTokens + 32 + 1024 + kGzLevels * 128 + kGzThreads + 8) * 4;
  (void)hipFuncSetAttribute((const void*)gz_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t grid = n_chunks < 512u ? n_chunks : 512u;  // two workgroups per CU fit in LDS
  hipLaunchKernelGGL(gz_encode_kernel, dim3(grid), dim3(kGzThreads), lds, s, *(const DevDeflate*)d, sw);
}

}  // namespace sg
