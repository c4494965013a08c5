// sg_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the read-sampling pass.
//
// Mapping: one lane = one window (plan) / one read (indel pass, emit).  A wave's 64 reads are
// consecutive fragment slots of the same 1 kbp window(s), so their haplotype bytes share a few
// cache lines and their FASTQ records form one contiguous ~21 KB output range.  All sampling is
// integer work on u32 thresholds (sg_tables.h); the only fp64 is the window-start draw, which must
// round exactly like the reference's `start+(end-start)*(x/2^32)` (ThreadPool.cpp:208-212).
//
//   plan_kernel      Segment::yieldReads draw loop   (lib/segment/Segment.cpp:735-762, 848)
//   namebase_kernel  per-segment fragCount numbering (Segment.cpp:732,763)
//   indel_kernel     Profile::predict indel pass     (lib/profile/Profile.cpp:1607-1634, 1556-1574)
//   scan_*           record offsets (replaces the 50 MB per-worker buffers + SeqWriter mutex,
//                    Segment.cpp:695-707,834-846; lib/seqwriter/SeqWriter.cpp:49-54)
//   emit_kernel      Profile::predict sampling loop  (Profile.cpp:1636-1700) + FASTQ formatting
//                    (Segment.cpp:803-832)
//   gc_kernel        calculateGCPercent              (lib/mydefine/MyDefine.cpp:279-303)
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "sg_device.h"

namespace sg {

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), one call = four 32-bit draws
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;  // one v_mad_u64_u32 each
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);  // three-input xor in one v_bitop3_b32
    c1 = lo1;
    c2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ uint32_t dev_ctx(uint32_t kind, uint32_t mate, uint32_t batch) {
  return kind | ((((mate & 1u) << 23) | (batch & 0xFFFFu)) << 8);
}

// lower bound over a {k0, T...} row (sg_tables.h)
__device__ __forceinline__ uint32_t row_search(const uint32_t* __restrict__ row, uint32_t lg, uint32_t x) {
  const uint32_t* T = row + 1;
  uint32_t pos = 0;
  for (uint32_t step = lg ? (1u << (lg - 1)) : 0; step; step >>= 1)
    if (x > T[pos + step - 1]) pos += step;
  return row[0] + pos;
}

// quality row {T0..T(W-1), sym bytes}: symbol of the first threshold >= x.  W is a multiple of 4 but
// not necessarily a power of two: probes past the row count as +infinity.
__device__ __forceinline__ uint32_t qual_lookup(const uint32_t* __restrict__ row, uint32_t w, uint32_t lg, uint32_t x) {
  uint32_t pos = 0;
  for (uint32_t step = lg ? (1u << (lg - 1)) : 0; step; step >>= 1) {
    const uint32_t i = pos + step - 1;
    if (i < w && x > row[i]) pos += step;
  }
  return ((const uint8_t*)(row + w))[pos];
}

__device__ __forceinline__ uint32_t aux_draw(const DevBatch& B, uint32_t slot, uint32_t j, uint32_t f, uint32_t mate) {
  uint32_t x[4];
  philox4x32_10(slot + B.slot_offset, j, f >> 2, dev_ctx(KIND_AUX, mate, B.batch_id), B.k0, B.k1, x);
  uint32_t l = f & 3;
  return l == 0 ? x[0] : l == 1 ? x[1] : l == 2 ? x[2] : x[3];
}

__device__ __forceinline__ uint32_t ndigits(uint32_t v) {
  return v < 10 ? 1 : v < 100 ? 2 : v < 1000 ? 3 : v < 10000 ? 4 : v < 100000 ? 5 : v < 1000000 ? 6
       : v < 10000000 ? 7 : v < 100000000 ? 8 : v < 1000000000 ? 9 : 10;
}

// ------------------------------------------------------------------------------------------------
// plan: PLAN_S lanes per window.  The reference draws (start, insert size) attempts one after the
// other and counts failures per window (fragment shorter than a read: Segment.cpp:753-762); attempt t
// of window w has the address (w, t).  Interior windows cannot fail -- every start leaves at least the
// largest insert size before the chain end -- so attempt t IS fragment t and the PLAN_S lanes take
// fragments t = j, j+PLAN_S, ... independently.  Windows near a chain end keep the sequential loop
// on lane 0.
// ------------------------------------------------------------------------------------------------
#define PLAN_S 8

__device__ __forceinline__ bool plan_attempt(const DevProfile& P, const DevBatch& B, const sg_window& win, uint64_t w,
                                             uint32_t attempt, uint64_t clen, uint32_t c3, PairRec& r, const uint32_t* isz_row) {
  uint32_t x[4];
  philox4x32_10((uint32_t)w + B.win_offset, attempt, 0, c3, B.k0, B.k1, x);
  // threadPool->randomInteger(spos, epos+1): (long)(start + (end-start)*(x/2^32)) in fp64
  const double frac = __dmul_rn((double)x[0], 1.0 / 4294967296.0);
  const double v = __dadd_rn((double)win.spos, __dmul_rn((double)win.len, frac));
  const uint32_t pos = (uint32_t)(long long)v;
  uint32_t isz;
  if (B.paired) isz = isz_row ? (uint32_t)P.isz_min + row_search(isz_row, P.isz_lg, x[1]) : (uint32_t)P.fixed_isz;
  else isz = win.len;
  const uint64_t avail = clen - (win.hap_base + pos);
  const uint32_t flen = avail < (uint64_t)isz ? (uint32_t)avail : isz;
  const uint32_t strand = B.paired ? 0u : (x[2] >> 31);  // randomInteger(0,2)
  r.win = (uint32_t)w; r.relpos = pos - win.spos; r.fl = flen | (strand << 31);
  return flen >= (uint32_t)P.L;
}

__global__ __launch_bounds__(256) void plan_kernel(DevProfile P, DevBatch B) {
  // the insert-size row (nine dependent probes per fragment) in LDS when it fits
  __shared__ uint32_t isz_lds[1025];
  const uint32_t* isz_row = P.isz_row;
  if (P.isz_row && P.isz_lg <= 10u) {
    for (uint32_t i = threadIdx.x; i < (1u << P.isz_lg) + 1u; i += blockDim.x) isz_lds[i] = P.isz_row[i];
    isz_row = isz_lds;
  }
  __syncthreads();
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t w = gid / PLAN_S;
  const uint32_t j = (uint32_t)(gid % PLAN_S);
  if (w >= B.n_windows) return;
  const sg_window win = B.windows[w];
  const int n = win.n_reads;
  const uint32_t planned = n <= 0 ? 0u : (B.paired ? ((uint32_t)n + 1u) / 2u : (uint32_t)n);
  const uint64_t clen = B.chain_len[win.chain];
  const uint32_t c3 = dev_ctx(KIND_PLAN, 0, B.batch_id);
  // can any attempt of this window fail?
  const uint64_t last_start = win.hap_base + win.spos + win.len - 1u;
  const uint64_t need = B.paired ? (uint64_t)P.isz_hi : (uint64_t)win.len;
  const uint32_t shortest = B.paired ? (uint32_t)P.isz_lo : win.len;
  const bool safe = last_start + need <= clen && shortest >= (uint32_t)P.L;
  if (safe) {
    for (uint32_t k = j; k < planned; k += PLAN_S) {
      PairRec r;
      plan_attempt(P, B, win, w, k, clen, c3, r, isz_row);
      r.k = k;
      B.pairs[win.slot_base + k] = r;
    }
    if (j == 0) B.win_actual[w] = planned;
    return;
  }
  if (j != 0) return;
  uint32_t done = 0, fail = 0, attempt = 0;
  while (done < planned) {
    PairRec r;
    if (!plan_attempt(P, B, win, w, attempt++, clen, c3, r, isz_row)) {
      if (++fail > 1000) break;
      continue;
    }
    r.k = done;
    B.pairs[win.slot_base + done] = r;
    done++;
  }
  for (uint32_t k = done; k < planned; k++) {
    PairRec r;
    r.win = (uint32_t)w; r.relpos = 0; r.fl = 0; r.k = k;
    B.pairs[win.slot_base + k] = r;
  }
  B.win_actual[w] = done;
}

// one wave per segment: exclusive scan of produced fragments over the segment's windows
__global__ __launch_bounds__(64) void namebase_kernel(DevBatch B) {
  const uint32_t s = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const uint32_t w0 = B.seg_first_window[s], w1 = B.seg_first_window[s + 1];
  uint32_t carry = 0;
  for (uint32_t base = w0; base < w1; base += 64) {
    uint32_t w = base + lane;
    uint32_t v = w < w1 ? B.win_actual[w] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t t = __shfl_up(incl, d);
      if ((int)lane >= d) incl += t;
    }
    if (w < w1) B.win_namebase[w] = carry + incl - v;
    carry += __shfl(incl, 63);
  }
  if (lane == 0) atomicAdd((unsigned long long*)&B.totals[2], (unsigned long long)carry);
}

// ------------------------------------------------------------------------------------------------
// indel pass: one lane per read
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void indel_kernel(DevProfile P, DevBatch B) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t m = blockIdx.y;
  if (t >= B.n_slots) return;
  const size_t idx = (size_t)m * B.n_slots + t;
  const PairRec rec = B.pairs[t];
  const uint32_t flen = rec.fl & 0x7FFFFFFFu;
  if (!flen) {
    B.rlen[idx] = 0; B.reclen[idx] = 0;
    B.meta[idx * 4] = make_uint4(0, 0, 0, 0);
    B.meta[idx * 4 + 1] = make_uint4(0, 0, 0, 0);
    return;
  }
  const int L = P.L;
  int j = 0, dl = 0;
  uint32_t nev = 0, first_ev = 0;
  uint32_t* ev = B.events + idx * SG_MAX_EVENTS;
  const uint32_t c3 = dev_ctx(KIND_INDEL, m, B.batch_id);
  // The insert test and the deletion test of a template position (Profile.cpp:1560-1570: two 32-bit draws,
  // P(ins) = cI / 2^32, P(del) = (1 - cI / 2^32) * cD / 2^32) are decided by ONE 64-bit uniform with exactly
  // that joint distribution:  ins iff x64 < A,  del iff A <= x64 < B,  A = cI * 2^32, B = A + (2^32 - cI) * cD.
  // x64 = head16 << 48 | tail48; the heads of eight positions are the halves of the four words of call (c, 0),
  // the tail (call (c, 1 + p/2)) is needed only when a head equals the head of A or of B (2 in 65536).  A call
  // ends after eight compares unless some lane of the wave has a head at or below B's (a quarter of a wave's
  // calls, almost always a real event); then only the flagged positions are walked.
  const uint64_t cI = (uint64_t)P.Tins + 1ull, A64 = cI << 32, B64 = A64 + ((1ull << 32) - cI) * (uint64_t)P.Cdel;
  const uint32_t hA = (uint32_t)(A64 >> 48), hB = (uint32_t)(B64 >> 48);
  for (int c = 0; 8 * c < L; c++) {
    if (8 * c + 7 < j) continue;  // all eight positions were consumed by a deletion
    uint32_t x[4];
    philox4x32_10(t + B.slot_offset, (uint32_t)c, 0, c3, B.k0, B.k1, x);
    uint32_t cand = 0;
#pragma unroll
    for (int p = 0; p < 8; p++) cand |= (uint32_t)(((x[p >> 1] >> (16 * (p & 1))) & 0xFFFFu) <= hB) << p;
    if (__ballot(cand != 0u) == 0ull) continue;
#pragma unroll
    for (int p = 0; p < 8; p++) {
      if (__ballot((cand >> p) & 1u) == 0ull) continue;
      const int jj = 8 * c + p;
      if (!((cand >> p) & 1u) || jj >= L || jj < j) continue;  // j: first position not covered by a deletion so far
      const uint32_t head = (x[p >> 1] >> (16 * (p & 1))) & 0xFFFFu;
      bool is_ins = head < hA, is_del = head > hA && head < hB;
      if (head == hA || head == hB) {
        uint32_t y[4];
        philox4x32_10(t + B.slot_offset, (uint32_t)c, 1u + (uint32_t)(p >> 1), c3, B.k0, B.k1, y);
        const uint64_t tail = (p & 1) ? ((uint64_t)(y[3] & 0xFFFFu) << 32) | y[2] : ((uint64_t)(y[1] & 0xFFFFu) << 32) | y[0];
        const uint64_t x64 = ((uint64_t)head << 48) | tail;
        is_ins = x64 < A64;
        is_del = !is_ins && x64 < B64;
      }
      if (is_ins) {
        uint32_t len = row_search(P.ins_row, P.ins_lg, aux_draw(B, t, (uint32_t)jj, 0, m));
        if (len > 0) {
          if (nev < SG_MAX_EVENTS) ev[nev] = ev_pack((uint32_t)jj, len, 0);
          if (nev == 0) first_ev = ev_pack((uint32_t)jj, len, 0);
          nev++;
          dl += (int)len;
        }
      } else if (is_del) {
        uint32_t len = row_search(P.del_row, P.del_lg, aux_draw(B, t, (uint32_t)jj, 0, m));
        if (len > 0) {
          uint32_t k = min((uint32_t)(L - jj), len);
          if (nev < SG_MAX_EVENTS) ev[nev] = ev_pack((uint32_t)jj, k, 1);
          if (nev == 0) first_ev = ev_pack((uint32_t)jj, k, 1);
          nev++;
          dl -= (int)k;
          j = jj + (int)k;
        }
      }
    }
  }
  if (L + dl < 50) { nev = 0; dl = 0; }  // Profile.cpp:1627-1634
  if (nev > SG_MAX_EVENTS) { atomicOr((unsigned long long*)&B.totals[3], 1ull); nev = 0; dl = 0; }
  const uint32_t np = (uint32_t)(L + dl);
  B.rlen[idx] = np | (nev << 16);
  const sg_window win = B.windows[rec.win];
  const uint32_t pos = win.spos + rec.relpos;
  const uint32_t namepos = pos % B.seg_size[win.seg];
  const uint32_t fragcount = B.win_namebase[rec.win] + rec.k + 1u;
  const uint32_t hdr = B.prefix_len + ndigits(namepos) + 1u + ndigits(fragcount) + (B.paired ? 2u : 0u) + 1u;
  B.reclen[idx] = hdr + 2u * np + 4u;
  // Per-read 64-byte row for the emit kernel: m0 = fragment offset + name fields, m1 = lengths,
  // then the header text "@popu#chr#pos%segsize#fragCount[/m]\n" (Segment.cpp:780,809,824) when it fits
  // 32 bytes; the read's first item lane stores it in front of the bases (emit_fast_kernel).
  const uint64_t foff = B.chain_off[win.chain] + win.hap_base + pos;
  const uint32_t rev = B.paired ? (m == 1u) : (rec.fl >> 31);
  B.meta[idx * 4] = make_uint4((uint32_t)foff, (uint32_t)(foff >> 32), namepos, fragcount);
  // ceil(2^32 / np) = floor((2^32-1)/np) + 1: bin = (i*bins*inv) >> 32 is exact while i*bins*np < 2^32
  // (sg_load_profile rejects profiles that could violate the bound)
  // m1.w: the event itself for single-event reads (handled inline by the emit kernel)
  B.meta[idx * 4 + 1] = make_uint4(flen | (rev << 31), np | (nev << 16) | (hdr << 22), 0xFFFFFFFFu / np + 1u, nev == 1u ? first_ev : 0u);
  if (hdr <= 32u) {
    uint32_t* hp = (uint32_t*)(B.meta + idx * 4 + 2);
    uint32_t w = 0, nb = 0;
    auto push = [&](uint32_t b) {
      w |= b << (nb * 8u);
      if (++nb == 4u) { *hp++ = w; w = 0; nb = 0; }
    };
    auto push_dec = [&](uint32_t v) {  // digits generated least-significant first into a byte queue
      const uint32_t nd = ndigits(v);
      uint64_t lo = 0;
      uint32_t hi = 0;
      for (uint32_t k = 0; k < nd; k++) {
        const uint32_t qd = v / 10u, d = '0' + (v - qd * 10u);
        if (k < 8u) lo = (lo << 8) | d; else hi = (hi << 8) | d;
        v = qd;
      }
      for (uint32_t k = 8; k < nd; k++) { push(hi & 0xFFu); hi >>= 8; }
      for (uint32_t k = 0; k < (nd < 8u ? nd : 8u); k++) { push((uint32_t)lo & 0xFFu); lo >>= 8; }
    };
    for (uint32_t i = 0; i < B.prefix_len; i++) push((B.prefix_w[i >> 2] >> (8u * (i & 3u))) & 0xFFu);  // hdr <= 32 implies prefix <= 16
    push_dec(namepos);
    push('#');
    push_dec(fragcount);
    if (B.paired) { push('/'); push('1' + m); }
    push('\n');
    if (nb) *hp = w;
  }
}

// ------------------------------------------------------------------------------------------------
// header (fallback): one lane per read, after the offset scan.  The fast emit kernel stores header
// texts of <= 32 bytes itself (from the read's row); this kernel serves the generic emit kernel and
// over-long headers.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void header_kernel(DevBatch B, uint32_t only_long) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t m = blockIdx.y;
  if (t >= B.n_slots) return;
  const size_t idx = (size_t)m * B.n_slots + t;
  const uint4 m1 = B.meta[idx * 4 + 1];
  if (!(m1.x & 0x7FFFFFFFu)) return;
  if (only_long && (m1.y >> 22) <= 32u) return;  // written by the emit kernel from the row
  const uint4 m0 = B.meta[idx * 4];
  const uint64_t ooff = B.recoff[idx];
  if (ooff + B.reclen[idx] > B.out_cap[m]) return;  // host re-checks totals before launching
  const uint32_t namepos = m0.z, fragcount = m0.w;
  if (!(B.diag & 4u)) {
    // byte stream -> unaligned dword stores (7 instead of 27 byte stores per record)
    uint8_t* hp = B.out[m] + ooff;
    uint32_t w = 0, nb = 0;
    auto push = [&](uint32_t b) {
      w |= b << (nb * 8u);
      if (++nb == 4u) { __builtin_memcpy(hp, &w, 4); hp += 4; w = 0; nb = 0; }
    };
    auto push_dec = [&](uint32_t v) {  // digits generated least-significant first into a byte queue
      const uint32_t nd = ndigits(v);
      uint64_t lo = 0;  // last (up to 8) digits, most significant in byte 0
      uint32_t hi = 0;  // leading digits of 9- and 10-digit numbers
      for (uint32_t k = 0; k < nd; k++) {
        const uint32_t qd = v / 10u, d = '0' + (v - qd * 10u);
        if (k < 8u) lo = (lo << 8) | d; else hi = (hi << 8) | d;
        v = qd;
      }
      for (uint32_t k = 8; k < nd; k++) { push(hi & 0xFFu); hi >>= 8; }
      for (uint32_t k = 0; k < (nd < 8u ? nd : 8u); k++) { push((uint32_t)lo & 0xFFu); lo >>= 8; }
    };
    if (B.prefix_len <= 16u) {
      for (uint32_t i = 0; i < B.prefix_len; i++) push((B.prefix_w[i >> 2] >> (8u * (i & 3u))) & 0xFFu);
    } else {
      for (uint32_t i = 0; i < B.prefix_len; i++) push(B.prefix[i]);
    }
    push_dec(namepos);
    push('#');
    push_dec(fragcount);
    if (B.paired) { push('/'); push('1' + m); }
    push('\n');
    for (uint32_t i = 0; i < nb; i++) hp[i] = (uint8_t)(w >> (8u * i));
  }
}

// ------------------------------------------------------------------------------------------------
// exclusive scan u32 -> u64 (three passes, 2048 items per block), one grid row per mate
// ------------------------------------------------------------------------------------------------
#define SCAN_ITEMS 8
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t* total) {
  __shared__ uint64_t wsum[SCAN_BLOCK / 64];
  const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint64_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(incl, d);
    if ((int)lane >= d) incl += t;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  uint64_t woff = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_BLOCK / 64; i++) {
    if (i < (int)wid) woff += wsum[i];
    tot += wsum[i];
  }
  __syncthreads();
  *total = tot;
  return woff + incl - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_reduce_kernel(const uint32_t* __restrict__ in, uint32_t n,
                                                                uint64_t* __restrict__ bsum, uint32_t nblk) {
  const uint32_t m = blockIdx.y;
  const uint32_t* src = in + (size_t)m * n;
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++)
    if (base + i < n) s += src[base + i];
  uint64_t tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) bsum[(size_t)m * nblk + blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void scan_sums_kernel(uint64_t* __restrict__ bsum, uint32_t nblk,
                                                        uint64_t* __restrict__ totals) {
  // one block per mate; sequential chunks of 1024 with a Hillis-Steele scan in LDS
  __shared__ uint64_t buf[1024];
  const uint32_t m = blockIdx.x;
  uint64_t* b = bsum + (size_t)m * nblk;
  uint64_t carry = 0;
  for (uint32_t base = 0; base < nblk; base += 1024) {
    uint32_t i = base + threadIdx.x;
    uint64_t v = i < nblk ? b[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      uint64_t t = threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    uint64_t incl = buf[threadIdx.x];
    if (i < nblk) b[i] = carry + incl - v;
    uint64_t chunk_total = buf[1023];
    __syncthreads();
    carry += chunk_total;
  }
  if (threadIdx.x == 0) totals[m] = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(const uint32_t* __restrict__ in, uint32_t n,
                                                               const uint64_t* __restrict__ bsum, uint32_t nblk,
                                                               uint64_t* __restrict__ out) {
  const uint32_t m = blockIdx.y;
  const uint32_t* src = in + (size_t)m * n;
  uint64_t* dst = out + (size_t)m * n;
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    v[i] = base + i < n ? src[base + i] : 0u;
    s += v[i];
  }
  uint64_t tot;
  uint64_t off = block_exclusive_scan(s, &tot) + bsum[(size_t)m * nblk + blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (base + i < n) dst[base + i] = off;
    off += v[i];
  }
}

// ------------------------------------------------------------------------------------------------
// emit: Profile::predict's sampling loop (Profile.cpp:1636-1700) + FASTQ formatting
// (Segment.cpp:803-832).
//
// Mapping (v3).  A wave owns G consecutive reads of one mate (one contiguous ~21 KB output range).
//   phase 0  lane = read: its 32-byte metadata row (written by indel_kernel / header_kernel) into LDS.
//   phase 1  lane = 8 consecutive bases.  The lane -> (read-in-iteration, item) map is FIXED:
//            TI = ceil((L+3)/8) items per read, RPI = 64/TI reads per wave iteration, so there is no
//            search, no division and no shuffle in the loop; read metadata comes from two LDS b128
//            reads.  Consecutive lanes write consecutive 8-byte runs: whole cache lines per store.
//            Reads grown by insertions beyond 8*TI-3 bases finish in a short clean-up loop.
// Haplotypes arrive pre-encoded (encode_kernel: A0 C1 T2 G3, N=4, other=5), thresholds live in LDS,
// sampling is branch-free integer compares.  v1 (lane per read) wrote each line in 32 partial
// stores (35.8 GB fabric writes for 4.2 GB of FASTQ); v2 fixed the traffic but spent ~1100
// instructions per 4 bases on bookkeeping (profiles/README.md).
// ------------------------------------------------------------------------------------------------
#define EMIT_THREADS 1024
#define EMIT_WAVES (EMIT_THREADS / 64)
#define META_ROW 32  // bytes of LDS metadata per read

// Orders one wave's LDS writes before its later LDS reads by other lanes (wave-private staging rows).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Source codes of a read that carries sequencing-indel events (rare path, kept OUT OF LINE and
// un-unrolled: inlined it pushed the kernel past the 64 KB instruction cache and cost 2x).
// Walks the events of Profile::predict's first loop (Profile.cpp:1607-1658) with a forward cursor and
// returns 13 nibbles, one per output position p = i0-5+q: base code 0..3 (template bases complemented
// for reverse reads), 4/5 = not in `bases`, bit 3 set = inserted base (already a profile code).
__device__ __noinline__ uint64_t slow_codes(const uint8_t* frag, uint32_t flen, uint32_t rev, const uint32_t* ev,
                                            uint32_t nev, uint32_t np, uint32_t i0, uint32_t K, uint32_t slot,
                                            uint32_t ctx_aux, uint32_t k0, uint32_t k1) {
  uint64_t out = 0;
  uint32_t e = 0;
  int shift = 0;  // output index - template index of the template bases after the consumed events
  uint32_t nextw = nev ? ev[0] : 0xFFFFFFFFu;
  const uint32_t revmask = rev ? 2u : 0u;
#pragma unroll 1
  for (int q = 0; q < 13; q++) {
    const int p = (int)i0 - 5 + q;
    uint32_t v = 4u;
    if (p >= 0 && (uint32_t)p < np && q >= 6 - (int)K) {
      // consume the events that end before p
#pragma unroll 1
      while (nextw != 0xFFFFFFFFu) {
        const int j = (int)(nextw & 0xFFFFu), len = (int)((nextw >> 16) & 0x7FFFu);
        const int pe = j + shift;  // output position of template base j / where it would have been
        if (nextw >> 31) { if (p < pe) break; shift -= len; }
        else { if (p <= pe + len) break; shift += len; }
        e++;
        nextw = e < nev ? ev[e] : 0xFFFFFFFFu;
      }
      bool inserted = false;
      if (nextw != 0xFFFFFFFFu && !(nextw >> 31)) {
        const int j = (int)(nextw & 0xFFFFu);
        const int pe = j + shift;
        if (p > pe) {  // inserted base number p-pe: randomInteger(0, N-1), never the last base (Profile.cpp:1564)
          const uint32_t f = (uint32_t)(p - pe);
          uint32_t x[4];
          philox4x32_10(slot, (uint32_t)j, f >> 2, ctx_aux, k0, k1, x);
          const uint32_t l = f & 3u;
          const uint32_t xv = l == 0 ? x[0] : l == 1 ? x[1] : l == 2 ? x[2] : x[3];
          v = 8u | __umulhi(xv, 3u);
          inserted = true;
        }
      }
      if (!inserted) {
        const uint32_t jt = (uint32_t)(p - shift);
        v = ((uint32_t)frag[rev ? flen - 1u - jt : jt] ^ revmask) & 7u;
      }
    }
    out |= (uint64_t)v << (4 * q);
  }
  return out;
}

// One item = output positions [8c, 8c+8) of one read: sample and store bases + qualities.
//   m0 = {frag_lo, frag_hi, out_lo, out_hi}   m1 = {flen | rev<<31, np | nev<<16 (6 bits) | hdr<<22, inv, -}
template <int KT, int QLG, bool SUB_LDS, bool QUAL_LDS>
__device__ __forceinline__ void emit_item(const DevProfile& P, const DevBatch& B, const uint4* lds_sub,
                                          const uint32_t* lds_qual, const uint4* gsub, uint32_t m, const uint4 m0,
                                          const uint4 m1, uint32_t slot, uint32_t c, bool active) {
  const uint32_t K = KT ? (uint32_t)KT : (uint32_t)P.kmer;
  const uint32_t bins = (uint32_t)P.bins;
  const uint32_t ctxmask = (1u << (2 * K)) - 1u;
  const uint32_t flen = m1.x & 0x7FFFFFFFu;
  const bool rev = (m1.x >> 31) != 0;
  const uint32_t np = m1.y & 0xFFFFu, nev = (B.diag & 32u) ? 0u : ((m1.y >> 16) & 0x3Fu), hdr = m1.y >> 22;
  const uint32_t inv = m1.z;
  const uint8_t* frag = B.chains + (((uint64_t)m0.y << 32) | m0.x);
  const uint32_t i0 = 8u * c;
  const uint32_t revmask = rev ? 2u : 0u;  // complement in code space: A0<->T2, C1<->G3 (Segment.cpp:81-103)

  // ---- source codes for positions i0-5 .. i0+7 (index q = p - i0 + 5); >= 4 means "not in bases" ----
  // Every lane loads the un-shifted template window: 16 encoded haplotype bytes; after the conditional
  // byte reversal position p sits at byte p-i0+5.  Unused slots (flen == 0: last partial group,
  // abandoned windows) read a harmless in-bounds address.
  uint32_t code[13];
  const uint8_t* src = flen == 0u ? B.chains + 128 : (rev ? frag + (int)flen - (int)i0 - 11 : frag + (int)i0 - 5);
  auto load_window = [&](const uint8_t* p16, uint32_t (&w)[4]) {
    __builtin_memcpy(w, p16, 16);
    if (rev) {
      const uint32_t t0 = __builtin_bswap32(w[3]), t1 = __builtin_bswap32(w[2]);
      const uint32_t t2 = __builtin_bswap32(w[1]), t3 = __builtin_bswap32(w[0]);
      w[0] = t0; w[1] = t1; w[2] = t2; w[3] = t3;
    }
  };
  {
    uint32_t w[4] = {0x00010203u + i0, 0x03020100u, 0x01000302u, 0x02030001u};
    if (!(B.diag & 2u)) load_window(src, w);
#pragma unroll
    for (int q = 0; q < 13; q++) code[q] = ((w[q >> 2] >> ((q & 3) * 8)) & 0xFFu) ^ revmask;
  }
  // Reads with exactly one sequencing indel (most event reads): past the event the template window is
  // the same window shifted by +-len; inserted bases are redrawn from their addressed Philox stream.
  if (__ballot(nev == 1u) != 0ull) {
    if (nev == 1u) {
      const uint32_t ew = m1.w;
      const int ej = (int)(ew & 0xFFFFu), elen = (int)((ew >> 16) & 0x7FFFu);
      const bool del = (ew >> 31) != 0;
      const int delta = del ? elen : -elen;  // template index shift past the event
      uint32_t w2[4];
      load_window(rev ? src - delta : src + delta, w2);
      const int first_shifted = del ? ej : ej + elen + 1;  // first output position that reads the shifted window
#pragma unroll
      for (int q = 0; q < 13; q++) {
        const int p = (int)i0 - 5 + q;
        const uint32_t c2 = ((w2[q >> 2] >> ((q & 3) * 8)) & 0xFFu) ^ revmask;
        if (p >= first_shifted) code[q] = c2;
      }
      if (!del) {
        // inserted run occupies output positions ej+1 .. ej+elen: randomInteger(0, N-1), never the last
        // base (Profile.cpp:1564); flat draw f = p - ej of stream (slot, ej)
#pragma unroll 1
        for (int q = 0; q < 13; q++) {
          const int p = (int)i0 - 5 + q;
          const bool ins = p > ej && p <= ej + elen && p >= 0;
          if (__ballot(ins) == 0ull) continue;
          if (ins) {
            const uint32_t v = 0x100u | __umulhi(aux_draw(B, slot, (uint32_t)ej, (uint32_t)(p - ej), m), 3u);
            // q is a loop variable: select through a mask instead of indexing registers dynamically
#pragma unroll
            for (int z = 0; z < 13; z++) if (z == q) code[z] = v;
          }
        }
      }
    }
  }
  if (__ballot(nev >= 2u) != 0ull) {
    if (nev >= 2u) {
      const uint32_t* ev = B.events + ((size_t)m * B.n_slots + slot) * SG_MAX_EVENTS;
      const uint64_t packed = slow_codes(frag, flen, rev ? 1u : 0u, ev, nev, np, i0, K, slot + B.slot_offset,
                                         dev_ctx(KIND_AUX, m, B.batch_id), B.k0, B.k1);
#pragma unroll
      for (int q = 0; q < 13; q++) {
        const uint32_t v = (uint32_t)(packed >> (4 * q)) & 0xFu;
        code[q] = (v & 8u) ? (0x100u | (v & 3u)) : v;
      }
    }
  }
  // natural index (A0 C1 T2 G3) -> profile base code; inserted bases (0x100 flag) already are profile codes
  const bool identity = P.remap_packed == 0xE4u;
#pragma unroll
  for (int q = 0; q < 13; q++) {
    uint32_t v = code[q];
    if (v & 0x100u) v &= 3u;
    else if (!identity && v < 4u) v = (P.remap_packed >> (2u * v)) & 3u;
    code[q] = v;
  }
  // ---- k-mer context of the K-1 positions before i0 (Profile::initKmers order, Profile.cpp:70-124) ----
  uint32_t ctxv = 0, vc = 0;
#pragma unroll
  for (int q = 0; q < 5; q++) {
    if (q >= 6 - (int)K && (int)i0 - 5 + q >= 0) {
      const bool valid = code[q] < 4u;
      ctxv = ((ctxv << 2) | (code[q] & 3u)) & ctxmask;
      vc = valid ? min(vc + 1u, K) : 0u;
    }
  }
  // ---- four Philox calls: [sub, qual] for 8 bases (counter = i/2) ----
  uint32_t x[16];
  const uint32_t c3b = dev_ctx(KIND_BASE, m, B.batch_id);
#pragma unroll
  for (int h = 0; h < 4; h++) {
    if (B.diag & 8u) { for (int z = 0; z < 4; z++) x[4 * h + z] = (slot * 2654435761u) ^ (c * 40503u + (4 * h + z) * 0x9E3779B9u); }
    else philox4x32_10(slot + B.slot_offset, 4u * c + (uint32_t)h, 0, c3b, B.k0, B.k1, x + 4 * h);
  }
  uint32_t sw[2] = {0, 0}, qw[2] = {0, 0};
#pragma unroll
  for (int h = 0; h < 8; h++) {
    const uint32_t i = i0 + (uint32_t)h;
    const uint32_t cd = code[5 + h];
    const bool valid = cd < 4u;
    ctxv = ((ctxv << 2) | (cd & 3u)) & ctxmask;
    vc = valid ? min(vc + 1u, K) : 0u;
    const uint32_t xs = x[2 * h], xq = x[2 * h + 1];
    const uint32_t bin = min(__umulhi(i * bins, inv), bins - 1u);  // i*binCount/n' (clamp only guards idle lanes)
    const uint32_t mlen = min(i + 1u, K);
    const uint32_t mmask = (1u << (2u * mlen)) - 1u;
    const bool ctx_ok = vc >= mlen && !(B.diag & 16u);
    // contexts with m real bases start at (4^m-4)/3 = (0x55555555 & (4^m-1)) - 1
    const uint32_t kidx = ctx_ok ? ((0x55555555u & mmask) - 1u) + (ctxv & mmask) : 0u;
    const uint4 row = SUB_LDS ? lds_sub[kidx * bins + bin] : gsub[(size_t)kidx * bins + bin];
    const uint32_t cnt = (xs > row.x) + (xs > row.y) + (xs > row.z);
    const uint32_t k = ctx_ok ? max(cnt, row.w) : cd;  // unknown context: the base is copied (Profile.cpp:1531-1533)
    const bool kvalid = k < 4u;
    const uint32_t kk = kvalid ? k : 0u;
    const uint32_t rowi = (((valid ? cd : 0u) * 4u + kk) * bins + bin) * P.qual_stride;
    const uint32_t* qrow = QUAL_LDS ? lds_qual + rowi : P.qual + rowi;
    uint32_t qi;
    if (B.diag & 16u) qi = 7u + (xq >> 29);
    else if (QLG == 3) {  // 8-wide row: three probes, then the symbol byte
      uint32_t pos = (xq > qrow[3]) ? 4u : 0u;
      pos += (xq > qrow[pos + 1]) ? 2u : 0u;
      pos += (xq > qrow[pos]) ? 1u : 0u;
      qi = (qrow[8 + (pos >> 2)] >> (8u * (pos & 3u))) & 0xFFu;
    } else {
      qi = qual_lookup(qrow, P.qual_w, P.qual_lg, xq);
    }
    uint32_t ch = kvalid ? ((P.bases_packed >> (8u * kk)) & 0xFFu) : (uint32_t)'N';
    uint32_t q = (uint32_t)P.min_qual + (kvalid ? qi : __umulhi(xq, 20u));  // getRandBaseQuality, Profile.cpp:1582-1584
    if (i >= np) {  // "\n+\n" after the bases, '\n' after the qualities
      ch = (i - np == 1u) ? '+' : '\n';
      q = '\n';
    }
    sw[h >> 2] |= ch << (8 * (h & 3));
    qw[h >> 2] |= q << (8 * (h & 3));
  }
  if (active && !(B.diag & 1u)) {
    uint8_t* so = B.out[m] + (((uint64_t)m0.w << 32) | m0.z) + hdr + i0;
    uint8_t* qo = so + np + 3u;
    const uint32_t ns = min(8u, np + 3u - i0);                  // bases + "\n+\n"
    const uint32_t nq = i0 <= np ? min(8u, np + 1u - i0) : 0u;  // qualities + '\n'
    if (ns == 8) __builtin_memcpy(so, sw, 8);
    else for (uint32_t b2 = 0; b2 < ns; b2++) so[b2] = (uint8_t)(sw[b2 >> 2] >> (8 * (b2 & 3)));
    if (nq == 8) __builtin_memcpy(qo, qw, 8);
    else for (uint32_t b2 = 0; b2 < nq; b2++) qo[b2] = (uint8_t)(qw[b2 >> 2] >> (8 * (b2 & 3)));
  }
}

template <int KT, int QLG, bool SUB_LDS, bool QUAL_LDS>
__global__ __launch_bounds__(EMIT_THREADS) void emit_kernel(DevProfile P, DevBatch B, uint32_t sub_rows, uint32_t qual_words,
                                                            uint32_t TI, uint32_t RPI) {
  extern __shared__ uint4 smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t m = blockIdx.y;
  const uint32_t tm = B.paired ? m : 0u;  // SE always samples from the mate-1 tables (Segment.cpp:770,777)
  // ---- LDS carve-up: [sub rows][qual words][per-wave metadata rows] ----
  uint4* lds_sub = smem;
  uint32_t* lds_qual = (uint32_t*)(smem + (SUB_LDS ? sub_rows : 0u));
  uint4* lds_meta_all = (uint4*)(lds_qual + (QUAL_LDS ? ((qual_words + 3u) & ~3u) : 0u));
  const uint4* gsub = P.sub + (size_t)tm * P.sub_mate_rows;
  if (SUB_LDS)
    for (uint32_t i = tid; i < sub_rows; i += EMIT_THREADS) lds_sub[i] = gsub[i];
  if (QUAL_LDS)
    for (uint32_t i = tid; i < qual_words; i += EMIT_THREADS) lds_qual[i] = P.qual[i];
  __syncthreads();
  uint4* meta_rows = lds_meta_all + (size_t)wv * 64 * (META_ROW / 16);

  const uint32_t G = RPI * (64u / RPI);             // reads per wave group
  const uint32_t ngroups = (B.n_slots + G - 1u) / G;
  const uint32_t sub = lane / TI, c_lane = lane - sub * TI;  // fixed lane -> (read in iteration, item)
  const bool lane_ok = sub < RPI;

  // first group by position, further ones from a per-mate counter (its own: this kernel also re-emits a
  // batch after emit_fast_kernel has run, see sg_result)
  uint32_t* next_group = (uint32_t*)(B.totals + 6) + m;
  for (uint32_t g = blockIdx.x * EMIT_WAVES + wv; g < ngroups;) {
    // ================= phase 0: lane = read: its 32-byte row (coalesced) into LDS =================
    const uint32_t t = g * G + lane;
    uint32_t items = 0;
    uint4 my0 = make_uint4(0, 0, 0, 0), my1 = make_uint4(0, 0, 0, 0);
    if (lane < G && t < B.n_slots) {
      const size_t idx = (size_t)m * B.n_slots + t;
      my0 = B.meta[idx * 4];
      my1 = B.meta[idx * 4 + 1];
      const uint64_t ooff = B.recoff[idx];
      my0.z = (uint32_t)ooff;
      my0.w = (uint32_t)(ooff >> 32);
      if (my1.x & 0x7FFFFFFFu) items = ((my1.y & 0xFFFFu) + 10u) / 8u;  // ceil((np + 3) / 8): bases + "\n+\n"
    }
    meta_rows[lane * 2] = my0;
    meta_rows[lane * 2 + 1] = my1;
    wave_lds_sync();

    // ================= phase 1: lane = 8 consecutive bases, fixed lane -> (read, item) map =================
    if (!(B.diag & 64u)) {
      // main steps: RPI reads per step through the fixed map; then the reads grown by insertions past
      // the TI items of the map (rare) finish one read at a time.  One call site keeps the loop small.
      const uint32_t nmain = (G + RPI - 1u) / RPI;
      unsigned long long more = __ballot(items > TI);
      uint32_t cb = TI;
      for (uint32_t step = 0;; step++) {
        uint32_t r, c;
        bool ok;
        if (step < nmain) {
          r = step * RPI + sub;
          c = c_lane;
          ok = lane_ok && r < G;
          if (!ok) r = step * RPI;
        } else {
          if (!more) break;
          r = (uint32_t)__builtin_ctzll(more);
          c = cb + lane;
          ok = true;
        }
        const uint4 m0 = meta_rows[r * 2], m1 = meta_rows[r * 2 + 1];
        const uint32_t np = m1.y & 0xFFFFu;
        const uint32_t nitems = (np + 10u) / 8u;
        if (step >= nmain) {
          cb += 64u;
          if (cb >= nitems) { more &= more - 1ull; cb = TI; }
        }
        const bool active = ok && (m1.x & 0x7FFFFFFFu) != 0u && c < nitems;
        emit_item<KT, QLG, SUB_LDS, QUAL_LDS>(P, B, lds_sub, lds_qual, gsub, m, m0, m1, g * G + r, active ? c : 0u, active);
      }
    }

    wave_lds_sync();  // the next group's phase 0 rewrites the metadata rows
    uint32_t nx = 0;
    if (lane == 0u) nx = atomicAdd(next_group, 1u);
    g = gridDim.x * EMIT_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
  }
}

// ------------------------------------------------------------------------------------------------
// emit, straight-line variant for the common profile shape: kmer == 3, tables in LDS (quality rows of
// <= 8 symbols whole, wider alphabets with the reference == called rows only: DIAG).  Same results as
// emit_kernel, far fewer instructions:
//   * the 13 source codes of an item are packed 2 bits each (natural order A0 C1 T2 G3) into one
//     word, so a k-mer context is ONE bit-field extract; the LDS copy of the substitution table is
//     permuted at staging time to that digit order (DevProfile::sub_perm), the quality rows to the
//     natural reference-base order;
//   * the 8-base block is straight-line and branch-free (the compiler can overlap the LDS reads of
//     different bases); first-of-read contexts ("XXb", "Xbb") use per-lane extract constants;
//   * a group's reads walk the steps as one item stream, 64 items per step, ordered by event class;
//   * what is per read rather than per item -- header text, the partial last item, "\n+\n" and '\n' --
//     is stored by a per-read pass once per group (the last item waits in the read's own LDS row);
//   * windows holding a non-ACGT base and reads with >= 2 sequencing indels (both rare; DIAG: items
//     with a substitution) go to a global queue for emit_slow_kernel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack4(uint32_t w) {  // four code bytes (0..3) -> 8 bits
  return (w | (w >> 6) | (w >> 12) | (w >> 18)) & 0xFFu;
}

template <bool PAIRED, bool DIAG, bool GUARD>
__device__ __forceinline__ bool fast_item(const DevProfile& P, const DevBatch& B, const uint4* lds_sub,
                                          const uint32_t* lds_qual, uint32_t m, const uint4 m0, const uint4 m1,
                                          uint32_t slot, uint32_t c, bool active, uint32_t hoff0, uint32_t hw0,
                                          uint32_t hk0, uint32_t hoff1, uint32_t hw1, uint32_t hk1, uint4* tail_row, int d0, uint32_t n_in,
                                          uint32_t ew_in, uint32_t tj_in) {
  const uint32_t bins = (uint32_t)P.bins;
  const uint32_t flen = m1.x & 0x7FFFFFFFu;
  const bool rev = PAIRED ? (m == 1u) : ((m1.x >> 31) != 0);
  // GUARD (item-stream map): an idle lane may be looking at a row whose read is finished; its fragment
  // offset, reciprocal and event word then hold the parked last item (see below), so an idle lane must
  // not follow them -- it reads a harmless in-bounds window and has no event.  With the fixed map an
  // idle lane only ever sees rows of reads of its own step, which are still intact when it loads them.
  // The caller has reduced the read's sequencing indels to what this item sees: d0 = template index minus
  // output index at the item's first position (events before it), n_in = events reaching into it (0, 1,
  // or 2 = too many for the two-window code), ew_in = that event with its OUTPUT position, tj_in = its
  // template position (the address of its draws).
  const uint32_t np = m1.y & 0xFFFFu, nev = (GUARD && !active) ? 0u : n_in, hdr = m1.y >> 22;
  const uint32_t inv = m1.z;
  const uint8_t* frag = B.chains + (((uint64_t)m0.y << 32) | m0.x);
  const uint32_t i0 = 8u * c;
  const bool no_frag = GUARD ? !active : flen == 0u;
  const uint8_t* src = no_frag ? B.chains + 128 : (rev ? frag + (int)flen - (int)i0 - 11 - d0 : frag + (int)i0 - 5 + d0);

  // 16 encoded bytes -> byte order by position (reverse reads), complement, validity, 2-bit pack
  auto window = [&](const uint8_t* p16, uint32_t& bad) -> uint32_t {
    uint32_t w[4];
    __builtin_memcpy(w, p16, 16);
    if (rev) {  // PAIRED: wave-uniform branch
      const uint32_t t0 = __builtin_bswap32(w[3]) ^ 0x02020202u, t1 = __builtin_bswap32(w[2]) ^ 0x02020202u;
      const uint32_t t2 = __builtin_bswap32(w[1]) ^ 0x02020202u, t3 = __builtin_bswap32(w[0]) ^ 0x02020202u;
      w[0] = t0; w[1] = t1; w[2] = t2; w[3] = t3;
    }
    // positions i0-2 .. i0+7 are bytes 3..12 (complementing an invalid code 4/5 gives 6/7: still >= 4)
    bad = (w[0] & 0xFC000000u) | ((w[1] | w[2]) & 0xFCFCFCFCu) | (w[3] & 0xFCu);
    return pack4(w[0]) | (pack4(w[1]) << 8) | (pack4(w[2]) << 16) | ((w[3] & 3u) << 24);
  };
  uint32_t bad;
  uint32_t cw = window(src, bad);

  // reads with exactly one sequencing indel: past the event the window is shifted by +-len
  if (__ballot(nev == 1u) != 0ull) {
    if (nev == 1u) {
      const uint32_t ew = ew_in;
      const int ej = (int)(ew & 0xFFFFu), elen = (int)((ew >> 16) & 0x7FFFu);
      const bool del = (ew >> 31) != 0;
      const int delta = del ? elen : -elen;
      uint32_t bad2;
      const uint32_t cw2 = window(rev ? src - delta : src + delta, bad2);
      bad |= bad2;
      const int first_shifted = del ? ej : ej + elen + 1;       // first output position reading the shifted window
      const int q0 = first_shifted - ((int)i0 - 5);               // its index in the item window
      const uint32_t keep = q0 <= 0 ? 0u : (q0 >= 13 ? 0xFFFFFFFFu : ((1u << (2 * q0)) - 1u));
      cw = (cw & keep) | (cw2 & ~keep);
      if (!del) {
        // inserted run = output positions ej+1 .. ej+elen: randomInteger(0, N-1), never the last base
        // (Profile.cpp:1564); flat draw f = p - ej of stream (slot, ej); codes are `bases` indexes
#pragma unroll 1
        for (int q = 3; q < 13; q++) {
          const int p = (int)i0 - 5 + q;
          const bool ins = p > ej && p <= ej + elen;
          if (__ballot(ins) == 0ull) continue;
          if (ins) {
            const uint32_t prof = __umulhi(aux_draw(B, slot, tj_in, (uint32_t)(p - ej), m), 3u);
            const uint32_t nat = (P.inv_remap_packed >> (2u * prof)) & 3u;
            cw = (cw & ~(3u << (2 * q))) | (nat << (2 * q));
          }
        }
      }
    }
  }
  // ---- four Philox calls: [sub, qual] for 8 bases (counter = i/2) ----
  uint32_t x[16];
  const uint32_t c3b = dev_ctx(KIND_BASE, m, B.batch_id);
#pragma unroll
  for (int h = 0; h < 4; h++) philox4x32_10(slot + B.slot_offset, 4u * c + (uint32_t)h, 0, c3b, B.k0, B.k1, x + 4 * h);

  uint32_t ksel[2] = {0, 0}, qw[2] = {0, 0};
  const uint32_t ib0 = __umul24(i0, bins);
  uint32_t offdiag = 0;
  const uint32_t* qrows[8];
#pragma unroll
  for (int h = 0; h < 8; h++) {
    const uint32_t xs = x[2 * h];
    // context id in packed digit order; first two bases of a read use the short-context blocks
    uint32_t kv;
    if (h == 0) kv = __builtin_amdgcn_ubfe(cw, hoff0, hw0) + hk0;
    else if (h == 1) kv = __builtin_amdgcn_ubfe(cw, hoff1, hw1) + hk1;
    else kv = ((cw >> (2 * h + 6)) & 63u) + 20u;
    // i*binCount/n'.  Idle lanes may compute a bin past the table: LDS reads beyond the allocation
    // return 0 and their results are never stored.
    const uint32_t bin = __umulhi(ib0 + (uint32_t)h * bins, inv);
    const uint4 row = lds_sub[__umul24(kv, bins) + bin];
    const uint32_t k = max((uint32_t)(xs > row.x) + (uint32_t)(xs > row.y) + (uint32_t)(xs > row.z), row.w);
    const uint32_t cd = (cw >> (2 * h + 10)) & 3u;
    ksel[h >> 2] |= k << (8 * (h & 3));
    if (DIAG) {
      // only the (reference base == called base) quality rows live in LDS; an item with a substitution
      // is finished by the generic code
      offdiag |= k ^ ((P.remap_packed >> (2u * cd)) & 3u);
      qrows[h] = lds_qual + __umul24(__umul24(cd, bins) + bin, P.qual_stride);
    } else {
      qrows[h] = lds_qual + __umul24(__umul24((cd << 2) | k, bins) + bin, P.qual_stride);
    }
  }
  if (!DIAG) {
#pragma unroll
    for (int h = 0; h < 8; h++) {  // 8-wide rows: three probes, then the symbol byte
      const uint32_t xq = x[2 * h + 1];
      const uint32_t* qrow = qrows[h];
      uint32_t pos = (xq > qrow[3]) ? 4u : 0u;
      pos += (xq > qrow[pos + 1]) ? 2u : 0u;
      pos += (xq > qrow[pos]) ? 1u : 0u;
      qw[h >> 2] |= (uint32_t)((const uint8_t*)(qrow + 8))[pos] << (8 * (h & 3));
    }
  } else {
    // rows of P.qual_w symbols: qual_lg probe rounds, the 8 bases' probes of a round are independent
    uint32_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t step = 1u << (P.qual_lg - 1u); step; step >>= 1) {
#pragma unroll
      for (int h = 0; h < 8; h++) {
        const uint32_t i = pos[h] + step - 1u;
        const uint32_t t = i < P.qual_w ? qrows[h][i] : 0xFFFFFFFFu;
        pos[h] += (x[2 * h + 1] > t) ? step : 0u;
      }
    }
#pragma unroll
    for (int h = 0; h < 8; h++) qw[h >> 2] |= (uint32_t)((const uint8_t*)(qrows[h] + P.qual_w))[pos[h]] << (8 * (h & 3));
  }
  const bool slow = active && (bad != 0u || nev >= 2u || offdiag != 0u);  // queued for the generic item code by the caller
  const bool go = active && !slow;
  // called base characters: byte select from `bases` by the packed codes, quality symbols -> ASCII
  uint32_t sw[2];
  sw[0] = __builtin_amdgcn_perm(0u, P.bases_packed, ksel[0]);
  sw[1] = __builtin_amdgcn_perm(0u, P.bases_packed, ksel[1]);
  qw[0] += 0x01010101u * (uint32_t)P.min_qual;
  qw[1] += 0x01010101u * (uint32_t)P.min_qual;
  // A whole item is two 8-byte stores.  The read's last, partial item (np % 8 bases) is parked in the
  // read's own LDS row instead: the per-read pass after the step loop merges it with the record
  // separators, so the byte-granular stores run once per read group rather than in every step.
  if (active) {
    if (i0 + 8u <= np) {
      if (go) {
        const uint64_t S = ((uint64_t)sw[1] << 32) | sw[0], Q = ((uint64_t)qw[1] << 32) | qw[0];
        uint8_t* so = B.out[m] + (((uint64_t)m0.w << 32) | m0.z) + hdr + i0;
        uint8_t* qo = so + np + 3u;
        __builtin_memcpy(so, &S, 8);
        __builtin_memcpy(qo, &Q, 8);
      }
    } else {
      // parked in the read's own LDS row, over the fields no lane needs once the last item has been
      // sampled (fragment offset, 2^32/n', event): base characters are never 0xFF
      tail_row[0].x = slow ? 0xFFFFFFFFu : sw[0];
      tail_row[0].y = sw[1];
      tail_row[1].z = qw[0];
      tail_row[1].w = qw[1];
    }
  }
  return slow;
}

// n (< 16) bytes of the 128-bit value (lo, hi) to q: 8/4/2/1-byte pieces
__device__ __forceinline__ void store_var(uint8_t* q, uint64_t lo, uint64_t hi, uint32_t n) {
  if (n & 8u) { __builtin_memcpy(q, &lo, 8); q += 8; lo = hi; }
  if (n & 4u) { const uint32_t v = (uint32_t)lo; __builtin_memcpy(q, &v, 4); q += 4; lo >>= 32; }
  if (n & 2u) { const uint16_t v = (uint16_t)lo; __builtin_memcpy(q, &v, 2); q += 2; lo >>= 16; }
  if (n & 1u) *q = (uint8_t)lo;
}

#define SLOW_CAP 128  // per-wave queue of items deferred to the generic code

template <bool PAIRED, bool DIAG, bool STREAM>
__global__ __launch_bounds__(EMIT_THREADS) void emit_fast_kernel(DevProfile P, DevBatch B, uint32_t sub_rows, uint32_t qual_words,
                                                                 uint32_t TI, uint32_t map_arg) {
  extern __shared__ uint4 smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t m = blockIdx.y;
  const uint32_t tm = PAIRED ? m : 0u;
  const uint32_t bins = (uint32_t)P.bins;
  uint4* lds_sub = smem;
  uint32_t* lds_qual = (uint32_t*)(smem + sub_rows);
  uint4* lds_meta_all = (uint4*)(lds_qual + ((qual_words + 3u) & ~3u));
  uint32_t* slow_all = (uint32_t*)(lds_meta_all + EMIT_WAVES * 64 * (META_ROW / 16));
  uint8_t* perm_all = (uint8_t*)(slow_all + EMIT_WAVES * SLOW_CAP);
  uint16_t* ovf_all = (uint16_t*)(perm_all + EMIT_WAVES * 64);  // items just past the stream map (read | item << 8)
  const uint4* gsub = P.sub + (size_t)tm * P.sub_mate_rows;
  // staging with the fast kernel's digit / base-order permutations
  for (uint32_t i = tid; i < sub_rows; i += EMIT_THREADS) {
    const uint32_t d = i / bins, b = i - d * bins;
    lds_sub[i] = gsub[(size_t)P.sub_perm[d] * bins + b];
  }
  const uint32_t qrow_words = 4u * bins * P.qual_stride;  // rows of one reference base
  if (!DIAG) {
    for (uint32_t i = tid; i < qual_words; i += EMIT_THREADS) {
      const uint32_t cdn = i / qrow_words, rest = i - cdn * qrow_words;
      lds_qual[i] = P.qual[((P.remap_packed >> (2u * cdn)) & 3u) * qrow_words + rest];
    }
  } else {
    // diagonal rows only: LDS row (cdn, bin) <- table row (ref = called = profile code of cdn, bin)
    const uint32_t drow_words = bins * P.qual_stride;
    for (uint32_t i = tid; i < qual_words; i += EMIT_THREADS) {  // qual_words = 4 * drow_words here
      const uint32_t cdn = i / drow_words, rest = i - cdn * drow_words;
      const uint32_t pc = (P.remap_packed >> (2u * cdn)) & 3u;
      lds_qual[i] = P.qual[pc * qrow_words + pc * drow_words + rest];
    }
  }
  __syncthreads();
  uint4* meta_rows = lds_meta_all + (size_t)wv * 64 * (META_ROW / 16);
  uint32_t* slow_list = slow_all + wv * SLOW_CAP;
  uint8_t* perm = perm_all + wv * 64;
  uint16_t* ovf = ovf_all + wv * 128;

  // Two lane -> (read, item) maps over the first TI items of a group's reads:
  //   STREAM  G = 63 reads; their items form one stream, 64 per step: lane l of step s does stream item
  //           i = 64 s + l = item i % TI of the (i / TI)-th read in step order; every lane busy whatever TI is;
  //   fixed   RPI = 64 / TI whole reads per step, lane = (read in step, item): constant per lane, cheaper per
  //           step, but 64 - RPI * TI lanes idle (7 at TI 19).
  // map_arg = ceil-reciprocal of TI (STREAM) or RPI (fixed).  launch_emit picks by measurement.
  const uint32_t inv_TI = map_arg, RPI = STREAM ? 1u : map_arg;
  // STREAM: 63 reads, so that TI steps (64 TI item slots) hold their 63 TI map items plus up to TI appended ones
  const uint32_t G = STREAM ? 63u : RPI * (64u / RPI);
  const uint32_t ngroups = (B.n_slots + G - 1u) / G;
  const uint32_t sub = lane / TI, c_lane = lane - sub * TI;
  const bool lane_ok = sub < RPI;
  // first item of a read: its first two bases have 1- and 2-base contexts ("XXb", "Xbb")
  const bool head_f = c_lane == 0u;
  const uint32_t hoff0 = head_f ? 10u : 6u, hw0 = head_f ? 2u : 6u, hk0 = head_f ? 0u : 20u;
  const uint32_t hoff1 = head_f ? 10u : 8u, hw1 = head_f ? 4u : 6u, hk1 = head_f ? 4u : 20u;
  // XCD-aware order: workgroups go to the 8 XCDs round-robin by their linear id, so the 16 workgroups of
  // one XCD (x % 8 equal; both mates) take 16 CONSECUTIVE runs of read groups each round -- neighbouring
  // reads overlap on the haplotype (30x coverage), and their lines are then fetched into one L2 instead of
  // eight.  (Measured on C2: FETCH_SIZE and time unchanged -- the haplotype bytes were already fetched about
  // once, the read-side traffic is the per-read rows -- so this is tidiness, not a speed-up.)
  const uint32_t bx = (gridDim.x & 7u) == 0u ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  // first group by position, further groups from a per-mate counter: waves that drew cheap groups (few
  // event reads) take more of them, so the grid drains evenly
  uint32_t* next_group = (uint32_t*)(B.totals + 5) + m;
  for (uint32_t g = bx * EMIT_WAVES + wv; g < ngroups;) {
    const uint32_t t = g * G + lane;
    uint32_t items = 0;
    uint4 my0 = make_uint4(0, 0, 0, 0), my1 = make_uint4(0, 0, 0, 0);
    if (lane < G && t < B.n_slots) {
      const size_t idx = (size_t)m * B.n_slots + t;
      my0 = B.meta[idx * 4];
      my1 = B.meta[idx * 4 + 1];
      const uint64_t ooff = B.recoff[idx];
      my0.z = (uint32_t)ooff;
      my0.w = (uint32_t)(ooff >> 32);
      if (my1.x & 0x7FFFFFFFu) items = ((my1.y & 0xFFFFu) + 7u) / 8u;  // separators: per-read pass below
    }
    meta_rows[lane * 2] = my0;
    meta_rows[lane * 2 + 1] = my1;
    wave_lds_sync();
    uint32_t nslow = 0;  // wave-uniform
    // Items the straight-line code cannot do (non-ACGT window, >= 2 indels) go to the batch's global
    // queue: one atomic per flush reserves the range; emit_slow_kernel runs the generic code on them
    // afterwards.  Keeping that code out of this kernel is worth ~14 % (SGPR spills, I-cache).
    auto flush_slow = [&]() {
      wave_lds_sync();
      uint32_t base = 0;
      if (lane == 0u) base = atomicAdd(B.slowq_count + m, nslow);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      for (uint32_t b0 = 0; b0 < nslow; b0 += 64u) {
        const uint32_t i = b0 + lane;
        if (i < nslow && base + i < B.slowq_cap) {
          const uint32_t e = slow_list[i];
          B.slowq[(size_t)m * B.slowq_cap + base + i] = make_uint2(g * G + (e & 0xFFu), e >> 8);
        }
      }
      if (lane == 0u && base + nslow > B.slowq_cap) atomicOr((unsigned long long*)(B.totals + 3), 2ull);
      nslow = 0;
      wave_lds_sync();
    };
    // Order of the group's reads through the step loop: reads without a sequencing indel first, then the
    // reads with one, then the reads with several (so that the two-window code and the event-list walk run
    // in the few steps that need them instead of whenever one of a step's reads has an event, 42 % of the
    // steps at XTen rates).
    const uint32_t nev_l = (my1.y >> 16) & 0x3Fu;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long in_group = G >= 64u ? ~0ull : ((1ull << G) - 1ull);
    const unsigned long long m_multi = __ballot(items > 0u && nev_l >= 2u);
    const unsigned long long m_one = __ballot(items > 0u && nev_l == 1u);
    const unsigned long long m_rest = in_group & ~(m_multi | m_one);
    const uint32_t n_rest = (uint32_t)__popcll(m_rest), n_one = (uint32_t)__popcll(m_one);
    const uint32_t n_fast = n_rest + n_one + (uint32_t)__popcll(m_multi);  // all of them walk the steps
    if (lane < G) {
      uint32_t pos;
      if ((m_rest >> lane) & 1ull) pos = (uint32_t)__popcll(m_rest & lt);
      else if ((m_one >> lane) & 1ull) pos = n_rest + (uint32_t)__popcll(m_one & lt);
      else pos = n_rest + n_one + (uint32_t)__popcll(m_multi & lt);
      perm[pos] = (uint8_t)lane;
    }
    // STREAM: TI = ceil(L / 8) exactly; a read that an insertion grew by one or two items appends them to
    // the end of the stream (they fill lanes of the last step that would idle anyway); only reads longer
    // than that take steps of their own below.
    uint32_t n_ovf = 0;
    unsigned long long more;
    if (STREAM) {
      const uint32_t extra = items > TI ? items - TI : 0u;
      const bool small = extra == 1u || extra == 2u;
      const unsigned long long b1 = __ballot(small), b2 = __ballot(extra == 2u);
      if (small) {
        const uint32_t o = (uint32_t)__popcll(b1 & lt) + (uint32_t)__popcll(b2 & lt);
        ovf[o] = (uint16_t)(lane | (TI << 8));
        if (extra == 2u) ovf[o + 1u] = (uint16_t)(lane | ((TI + 1u) << 8));
      }
      n_ovf = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2);
      more = __ballot(extra > 2u);
    } else {
      more = __ballot(items > TI);
    }
    wave_lds_sync();
    const uint32_t n_items = n_fast * TI, n_stream = n_items + n_ovf;
    const uint32_t nmain = STREAM ? (n_stream + 63u) / 64u : (n_fast + RPI - 1u) / RPI;
    uint32_t cb = TI;
    // one item per lane; windows with a non-ACGT base (or, DIAG, a substitution) are queued for the
    // generic code.  Two call sites -- the fixed map and the long-read tail -- so that the per-lane
    // context constants of the fixed map stay loop-invariant registers.
    auto run_item = [&](uint32_t r, uint32_t c, bool ok, const uint4 m0, const uint4 m1, uint32_t o0, uint32_t a0, uint32_t k0_,
                        uint32_t o1, uint32_t a1, uint32_t k1_, uint32_t c_idle) {
      const uint32_t nitems = ((m1.y & 0xFFFFu) + 7u) / 8u;
      const bool active = ok && (m1.x & 0x7FFFFFFFu) != 0u && c < nitems;
      // what the read's sequencing indels mean for this item (see fast_item)
      const uint32_t nev_r = active ? ((m1.y >> 16) & 0x3Fu) : 0u;
      int d0 = 0;
      uint32_t n_in = nev_r == 1u ? 1u : 0u, ew_in = m1.w, tj_in = m1.w & 0xFFFFu;
      if (__ballot(nev_r >= 2u) != 0ull) {
        if (nev_r >= 2u) {
          // walk the event list like slow_codes does: shift = output index - template index so far
          const uint32_t* ev = B.events + ((size_t)m * B.n_slots + (g * G + r)) * SG_MAX_EVENTS;
          const int p_lo = (int)(8u * c) - 2, p_hi = (int)(8u * c) + 7;  // positions whose source bases the item uses
          int shift = 0;
          n_in = 0;
          for (uint32_t k = 0; k < nev_r; k++) {
            const uint32_t w = ev[k];
            const int j = (int)(w & 0xFFFFu), len = (int)((w >> 16) & 0x7FFFu);
            const bool del = (w >> 31) != 0;
            const int pe = j + shift;  // output position of template base j
            if (del ? pe <= p_lo : pe + len < p_lo) { shift += del ? -len : len; continue; }  // wholly before the item
            if (del ? pe > p_hi : pe >= p_hi) break;                                              // this and the rest: after it
            if (n_in == 0u) { d0 = -shift; ew_in = (w & 0xFFFF0000u) | (uint32_t)pe; tj_in = (uint32_t)j; }
            n_in++;
            shift += del ? -len : len;
          }
          if (n_in == 0u) d0 = -shift;
        }
      }
      const bool slow = fast_item<PAIRED, DIAG, STREAM>(P, B, lds_sub, lds_qual, m, m0, m1, g * G + r, active ? c : c_idle, active, o0, a0, k0_,
                                          o1, a1, k1_, meta_rows + r * 2, d0, n_in, ew_in, tj_in);
      const unsigned long long sm = __ballot(slow);
      if (sm) {
        if (slow) slow_list[nslow + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull))] = r | (c << 8);
        nslow += (uint32_t)__popcll(sm);
      }
    };
    if (STREAM) {
      for (uint32_t step = 0; step < nmain; step++) {  // the item stream, 64 items per step
        if (nslow > SLOW_CAP - 64u) flush_slow();
        const uint32_t i = step * 64u + lane;
        const bool ok = i < n_stream, in_map = i < n_items;
        const uint32_t ri = __umul24(i, inv_TI) >> 20;  // i / TI (exact: i * TI < 2^20)
        const uint32_t e = ovf[(ok && !in_map) ? i - n_items : 0u];
        const uint32_t r = in_map ? perm[ri] : (ok ? (e & 0xFFu) : perm[0]);
        const uint32_t c = in_map ? i - __umul24(ri, TI) : (ok ? e >> 8 : 1u);
        const uint4 m0 = meta_rows[r * 2], m1 = meta_rows[r * 2 + 1];
        const bool head = c == 0u;
        run_item(r, c, ok, m0, m1, head ? 10u : 6u, head ? 2u : 6u, head ? 0u : 20u, head ? 10u : 8u, head ? 4u : 6u, head ? 4u : 20u, c);
      }
    } else {
      for (uint32_t step = 0; step < nmain; step++) {  // RPI whole reads per step
        if (nslow > SLOW_CAP - 64u) flush_slow();
        const uint32_t ri = step * RPI + sub;
        const bool ok = lane_ok && ri < n_fast;
        const uint32_t r = perm[ok ? ri : step * RPI];
        const uint4 m0 = meta_rows[r * 2], m1 = meta_rows[r * 2 + 1];
        run_item(r, c_lane, ok, m0, m1, hoff0, hw0, hk0, hoff1, hw1, hk1, c_lane);
      }
    }
    while (more) {  // items past the fixed map (reads grown by insertions, reads of more than 64 items): never a first item
      if (nslow > SLOW_CAP - 64u) flush_slow();
      const uint32_t r = (uint32_t)__builtin_ctzll(more);
      const uint32_t c = cb + lane;
      const uint4 m0 = meta_rows[r * 2], m1 = meta_rows[r * 2 + 1];
      const uint32_t nitems = ((m1.y & 0xFFFFu) + 7u) / 8u;
      cb += 64u;
      if (cb >= nitems) { more &= more - 1ull; cb = TI; }
      run_item(r, c, true, m0, m1, 6u, 6u, 20u, 8u, 6u, 20u, 1u);
    }
    if (nslow) flush_slow();
    // ---- per-read pass, lane = read: header text, last partial item, record separators ----
    // These byte-granular stores touch lines the steps above have just written from this wave, so
    // they merge in L2.  Reads whose last item went to the generic code (0xFFFFFFFF row) get only the
    // separators here; emit_slow_kernel writes the same separator bytes again (benign).
    wave_lds_sync();
    if (lane < G && t < B.n_slots) {
      const uint4 r0 = meta_rows[lane * 2], r1 = meta_rows[lane * 2 + 1];
      if (r1.x & 0x7FFFFFFFu) {
        const uint32_t np = r1.y & 0xFFFFu, hl = r1.y >> 22;
        uint8_t* rec = B.out[m] + (((uint64_t)r0.w << 32) | r0.z);
        if (hl <= 32u) {
          const uint4* hrow = B.meta + ((size_t)m * B.n_slots + t) * 4 + 2;
          const uint4 h0 = hrow[0], h1 = hrow[1];
          uint8_t* q = rec;
          uint32_t rem = hl;
          uint4 part = h0;
          if (hl >= 16u) { __builtin_memcpy(q, &h0, 16); q += 16; rem -= 16u; part = h1; }
          if (rem == 16u) __builtin_memcpy(q, &part, 16);
          else store_var(q, ((uint64_t)part.y << 32) | part.x, ((uint64_t)part.w << 32) | part.z, rem);
        }
        const uint4 tr = make_uint4(r0.x, r0.y, r1.z, r1.w);  // the parked last item (fast_item)
        const uint32_t d = (tr.x != 0xFFFFFFFFu) ? (np & 7u) : 0u;  // bases (and qualities) of the partial item
        const uint64_t S = ((uint64_t)tr.y << 32) | tr.x, Q = ((uint64_t)tr.w << 32) | tr.z;
        const uint64_t keep = (1ull << (8u * d)) - 1ull;
        uint8_t* so = rec + hl + (np - d);
        // d bases + "\n+\n" (3..10 bytes), d qualities + '\n' (1..8 bytes)
        const uint64_t s_lo = (S & keep) | (0x0A2B0Aull << (8u * d));
        const uint64_t s_hi = d > 5u ? (0x0A2B0Aull >> (8u * (8u - d))) : 0ull;
        store_var(so, s_lo, s_hi, d + 3u);
        store_var(so + np + 3u, (Q & keep) | (0x0Aull << (8u * d)), 0ull, d + 1u);
      }
    }
    wave_lds_sync();
    uint32_t nx = 0;
    if (lane == 0u) nx = atomicAdd(next_group, 1u);
    g = gridDim.x * EMIT_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
  }
}

// The queued items of emit_fast_kernel, one lane each, through the generic item code; the tables are
// staged in LDS like in emit_kernel when they fit (persistent workgroup per CU, grid-stride over the queue).
template <int KT, int QLG, bool SUB_LDS, bool QUAL_LDS>
__global__ __launch_bounds__(EMIT_THREADS) void emit_slow_kernel(DevProfile P, DevBatch B, uint32_t sub_rows, uint32_t qual_words) {
  extern __shared__ uint4 smem[];
  const uint32_t m = blockIdx.y, tid = threadIdx.x, lane = tid & 63u;
  const uint32_t tm = B.paired ? m : 0u;
  uint32_t n = B.slowq_count[m];
  if (n > B.slowq_cap) n = B.slowq_cap;  // overflow: the host reruns the batch through emit_kernel
  if ((blockIdx.x * EMIT_THREADS) >= n) return;  // nothing for this workgroup: skip the staging too
  uint4* lds_sub = smem;
  uint32_t* lds_qual = (uint32_t*)(smem + (SUB_LDS ? sub_rows : 0u));
  const uint4* gsub = P.sub + (size_t)tm * P.sub_mate_rows;
  if (SUB_LDS)
    for (uint32_t i = tid; i < sub_rows; i += EMIT_THREADS) lds_sub[i] = gsub[i];
  if (QUAL_LDS)
    for (uint32_t i = tid; i < qual_words; i += EMIT_THREADS) lds_qual[i] = P.qual[i];
  __syncthreads();
  const uint2* q = B.slowq + (size_t)m * B.slowq_cap;
  for (uint32_t b0 = (blockIdx.x * EMIT_THREADS + tid) & ~63u; b0 < n; b0 += gridDim.x * EMIT_THREADS) {
    const uint32_t i = b0 + lane;
    const bool act = i < n;
    const uint2 e = q[act ? i : b0];
    const size_t idx = (size_t)m * B.n_slots + e.x;
    uint4 m0 = B.meta[idx * 4];
    const uint4 m1 = B.meta[idx * 4 + 1];
    const uint64_t ooff = B.recoff[idx];
    m0.z = (uint32_t)ooff;
    m0.w = (uint32_t)(ooff >> 32);
    emit_item<KT, QLG, SUB_LDS, QUAL_LDS>(P, B, lds_sub, lds_qual, gsub, m, m0, m1, e.x, e.y, act);
  }
}

// ------------------------------------------------------------------------------------------------
// haplotype encoding: ASCII -> base code, in place (A0 C1 T2 G3, 'N' = 4, anything else = 5)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t encode4(uint32_t w) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t b = (w >> (8 * i)) & 0xFFu;
    const bool acgt = (b == 'A') | (b == 'C') | (b == 'G') | (b == 'T');
    const uint32_t v = acgt ? ((b >> 1) & 3u) : (b == 'N' ? 4u : 5u);
    o |= v << (8 * i);
  }
  return o;
}
__global__ __launch_bounds__(256) void encode_kernel(uint4* __restrict__ buf, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = buf[i];
    v.x = encode4(v.x); v.y = encode4(v.y); v.z = encode4(v.z); v.w = encode4(v.w);
    buf[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// GC% per window: one wave per window, 16 B per lane per step
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t count_eq_bytes(uint32_t w, uint32_t c) {
  uint32_t x = w ^ (c * 0x01010101u);
  uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  t = ~(t | x | 0x7F7F7F7Fu);  // 0x80 in every byte of x that is zero
  return __popc(t);
}

__global__ __launch_bounds__(256) void gc_kernel(const uint8_t* __restrict__ chains, const uint64_t* __restrict__ chain_off,
                                                 const sg_gc_window* __restrict__ wins, uint64_t n, int32_t* __restrict__ out) {
  const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (w >= n) return;
  const sg_gc_window win = wins[w];
  const uint8_t* p = chains + chain_off[win.chain] + win.start;
  uint32_t gc = 0, nn = 0;
  for (uint32_t b = lane * 16; b < win.len; b += 64 * 16) {
    uint32_t v[4];
    __builtin_memcpy(v, p + b, 16);
    const uint32_t rem = win.len - b;  // bytes of this 16-byte group inside the window
#pragma unroll
    for (int i = 0; i < 4; i++) {
      uint32_t x = v[i];
      const int left = (int)rem - 4 * i;
      if (left <= 0) x = 0;
      else if (left < 4) x &= (1u << (8 * left)) - 1u;  // masked-off bytes read as 0 = 'A': neither GC nor N
      gc += count_eq_bytes(x, 1u) + count_eq_bytes(x, 3u);  // encoded C, G
      nn += count_eq_bytes(x, 4u);                           // encoded N
    }
  }
#pragma unroll
  for (int d = 32; d; d >>= 1) {
    gc += __shfl_xor(gc, d);
    nn += __shfl_xor(nn, d);
  }
  if (lane == 0) out[w] = win.len == 0 ? 0 : (nn > 0 ? -1 : (int32_t)(100u * gc / win.len));
}

// ------------------------------------------------------------------------------------------------
// launchers (called from sg_api.cpp through plain C++ declarations)
// ------------------------------------------------------------------------------------------------
void launch_plan(const DevProfile& P, const DevBatch& B, hipStream_t s) {
  if (!B.n_windows) return;
  uint32_t grid = (uint32_t)((B.n_windows * PLAN_S + 255) / 256);
  hipLaunchKernelGGL(plan_kernel, dim3(grid), dim3(256), 0, s, P, B);
}
void launch_namebase(const DevBatch& B, hipStream_t s) {
  if (!B.n_segs) return;
  hipLaunchKernelGGL(namebase_kernel, dim3(B.n_segs), dim3(64), 0, s, B);
}
void launch_indel(const DevProfile& P, const DevBatch& B, hipStream_t s) {
  if (!B.n_slots) return;
  dim3 grid((B.n_slots + 255) / 256, B.paired ? 2 : 1);
  hipLaunchKernelGGL(indel_kernel, grid, dim3(256), 0, s, P, B);
}
bool emit_uses_fast_kernel(const DevProfile& P, const DevBatch& B);
void launch_header(const DevProfile& P, const DevBatch& B, hipStream_t s) {
  if (!B.n_slots) return;
  const bool fast = emit_uses_fast_kernel(P, B);
  // fast kernel: headers <= 32 bytes come from the rows; 24 bytes beyond the prefix is the longest tail
  if (fast && B.prefix_len + 24u <= 32u) return;
  dim3 grid((B.n_slots + 255) / 256, B.paired ? 2 : 1);
  hipLaunchKernelGGL(header_kernel, grid, dim3(256), 0, s, B, fast ? 1u : 0u);
}
uint32_t scan_blocks(uint32_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
void launch_scan(const DevBatch& B, uint64_t* bsum, hipStream_t s) {
  if (!B.n_slots) return;
  const uint32_t nm = B.paired ? 2 : 1;
  const uint32_t nblk = scan_blocks(B.n_slots);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblk, nm), dim3(SCAN_BLOCK), 0, s, B.reclen, B.n_slots, bsum, nblk);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(nm), dim3(1024), 0, s, bsum, nblk, B.totals);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nblk, nm), dim3(SCAN_BLOCK), 0, s, B.reclen, B.n_slots, bsum, nblk, B.recoff);
}
// exclusive scan of n u32 values into u64 offsets (one row); total -> *total
void launch_scan_u32(const uint32_t* in, uint32_t n, uint64_t* bsum, uint64_t* out, uint64_t* total, hipStream_t s) {
  if (!n) return;
  const uint32_t nblk = scan_blocks(n);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblk, 1), dim3(SCAN_BLOCK), 0, s, in, n, bsum, nblk);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, bsum, nblk, total);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nblk, 1), dim3(SCAN_BLOCK), 0, s, in, n, bsum, nblk, out);
}
// LDS budget of emit_kernel: tables that fit are staged, the rest is read through L2.
static const size_t kLdsBytes = 160 * 1024;
template <int KT, int QLG, bool SL, bool QL>
static void launch_emit_variant(const DevProfile& P, const DevBatch& B, dim3 grid, size_t lds, uint32_t sub_rows,
                                uint32_t qual_words, uint32_t TI, uint32_t RPI, hipStream_t s) {
  (void)hipFuncSetAttribute((const void*)emit_kernel<KT, QLG, SL, QL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((emit_kernel<KT, QLG, SL, QL>), grid, dim3(EMIT_THREADS), lds, s, P, B, sub_rows, qual_words, TI, RPI);
}
struct EmitLds { uint32_t sub_rows, qual_words, diag_words; size_t lds, lds_diag; bool sub_lds, qual_lds, diag_lds; };
static EmitLds emit_lds(const DevProfile& P) {
  EmitLds e;
  uint32_t kmer_count = 0;
  for (int m = 1, p = 1; m <= P.kmer; m++) { p *= 4; kmer_count += p; }
  e.sub_rows = kmer_count * (uint32_t)P.bins;
  e.qual_words = 16u * (uint32_t)P.bins * P.qual_stride;
  e.diag_words = 4u * (uint32_t)P.bins * P.qual_stride;
  const size_t fixed = (size_t)EMIT_WAVES * 64 * META_ROW + (size_t)EMIT_WAVES * SLOW_CAP * 4 + (size_t)EMIT_WAVES * 64 +
                       (size_t)EMIT_WAVES * 128 * 2;
  const size_t sub_b = (size_t)e.sub_rows * 16, qual_b = ((size_t)e.qual_words * 4 + 15) & ~(size_t)15;
  const size_t diag_b = ((size_t)e.diag_words * 4 + 15) & ~(size_t)15;
  e.sub_lds = fixed + sub_b <= kLdsBytes;
  e.qual_lds = e.sub_lds && fixed + sub_b + qual_b <= kLdsBytes;
  e.diag_lds = e.sub_lds && fixed + sub_b + diag_b <= kLdsBytes;
  e.lds = fixed + (e.sub_lds ? sub_b : 0) + (e.qual_lds ? qual_b : 0);
  e.lds_diag = fixed + sub_b + diag_b;
  return e;
}
// 0: generic kernel, 1: fast kernel with the whole 8-wide quality table in LDS, 2: fast kernel with the
// diagonal quality rows in LDS (wide quality alphabets)
static int emit_fast_mode(const DevProfile& P) {
  const EmitLds e = emit_lds(P);
  if (P.kmer != 3 || getenv("SG_DIAG") != nullptr) return 0;
  if (P.qual_w == 8 && e.sub_lds && e.qual_lds) return 1;
  if (P.qual_w <= 64 && e.diag_lds) return 2;
  return 0;
}
int emit_variant(const DevProfile& P) { return emit_fast_mode(P); }
bool emit_uses_fast_kernel(const DevProfile& P, const DevBatch& B) {
  (void)B;
  return emit_fast_mode(P) != 0;
}
void launch_emit(const DevProfile& P, const DevBatch& B, hipStream_t s, bool force_generic, hipEvent_t after_main) {
  if (!B.n_slots) {
    if (after_main) (void)hipEventRecord(after_main, s);
    return;
  }
  const EmitLds e = emit_lds(P);
  const uint32_t sub_rows = e.sub_rows, qual_words = e.qual_words;
  const size_t lds = e.lds;
  const bool sub_lds = e.sub_lds, qual_lds = e.qual_lds;
  // fixed lane map: TI items of 8 bases per read, RPI reads per wave iteration
  uint32_t TI = ((uint32_t)P.L + 3u + 7u) / 8u;
  if (TI > 64u) TI = 64u;  // longer reads finish in the clean-up loop
  const uint32_t RPI = 64u / TI;
  const uint32_t G = RPI * (64u / RPI);
  const uint32_t ngroups = (B.n_slots + G - 1u) / G;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const uint32_t nm = B.paired ? 2 : 1;
  uint32_t gx = (uint32_t)cus / nm;  // one 1024-thread workgroup per CU, persistent over read groups
  if (gx < 1) gx = 1;
  const uint32_t need = (ngroups + EMIT_WAVES - 1) / EMIT_WAVES;
  if (gx > need) gx = need;
  dim3 grid(gx, nm);
  const int mode = force_generic ? 0 : emit_fast_mode(P);
  // straight-line kernel, item-stream map by default: measured on the four shipped profiles it ties with the
  // fixed map on XTen (4.91 vs 4.97 ms) and wins clearly with the diagonal-row variant (7.1 vs 9.2 ms
  // HiSeq2500, 7.4 vs 10.5 ms HiSeq2000 / GAIIx).  SG_EMIT_MAP=fixed selects the other map (diagnostics).
  bool stream = true;
  if (const char* e = getenv("SG_EMIT_MAP")) stream = e[0] == 's';
  // fixed map: three bases of slack in TI (a read that outgrows the map costs a whole extra step with one
  // busy lane; at TI = ceil(L/8) that happened to 5 % of the XTen reads).  The stream map needs none: such
  // items are appended to the stream.
  uint32_t TIf = ((uint32_t)P.L + (stream ? 0u : 3u) + 7u) / 8u;
  if (TIf > 64u) TIf = 64u;
  const uint32_t RPIf = 64u / TIf;
  const uint32_t Gf = stream ? 63u : RPIf * (64u / RPIf);
  const uint32_t fneed = ((B.n_slots + Gf - 1u) / Gf + EMIT_WAVES - 1) / EMIT_WAVES;
  uint32_t fgx = (uint32_t)cus / nm;
  if (fgx < 1) fgx = 1;
  if (fgx > fneed) fgx = fneed;
  const dim3 fgrid(fgx, nm);
  const uint32_t map_arg = stream ? (1u << 20) / TIf + 1u : RPIf;
  auto launch_fast = [&](auto kern, size_t bytes, uint32_t qwords) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipLaunchKernelGGL(kern, fgrid, dim3(EMIT_THREADS), bytes, s, P, B, sub_rows, qwords, TIf, map_arg);
  };
  auto pick = [&](auto paired_t, auto diag_t, size_t bytes, uint32_t qwords) {
    constexpr bool PA = decltype(paired_t)::value, DG = decltype(diag_t)::value;
    if (stream) launch_fast(emit_fast_kernel<PA, DG, true>, bytes, qwords);
    else launch_fast(emit_fast_kernel<PA, DG, false>, bytes, qwords);
  };
  if (mode == 1) {
    if (B.paired) pick(std::true_type{}, std::false_type{}, lds, qual_words);
    else pick(std::false_type{}, std::false_type{}, lds, qual_words);
  } else if (mode == 2) {
    if (B.paired) pick(std::true_type{}, std::true_type{}, e.lds_diag, e.diag_words);
    else pick(std::false_type{}, std::true_type{}, e.lds_diag, e.diag_words);
  }
  if (mode != 0) {
    if (after_main) (void)hipEventRecord(after_main, s);
    // generic-code layout of the tables (no permutation): sub rows + the whole quality table when they fit
    const size_t slow_sub = sub_lds ? (size_t)sub_rows * 16 : 0;
    const size_t slow_qual = (sub_lds && slow_sub + (((size_t)qual_words * 4 + 15) & ~(size_t)15) <= kLdsBytes) ? (((size_t)qual_words * 4 + 15) & ~(size_t)15) : 0;
    const dim3 sgrid(gx, nm);
    auto launch_slow = [&](auto kern) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(slow_sub + slow_qual));
      hipLaunchKernelGGL(kern, sgrid, dim3(EMIT_THREADS), slow_sub + slow_qual, s, P, B, sub_rows, qual_words);
    };
    if (slow_sub && slow_qual && P.qual_w == 8) launch_slow(emit_slow_kernel<3, 3, true, true>);
    else if (slow_sub && slow_qual) launch_slow(emit_slow_kernel<0, 0, true, true>);
    else if (slow_sub) launch_slow(emit_slow_kernel<0, 0, true, false>);
    else launch_slow(emit_slow_kernel<0, 0, false, false>);
  } else if (P.kmer == 3 && P.qual_w == 8 && sub_lds && qual_lds) launch_emit_variant<3, 3, true, true>(P, B, grid, lds, sub_rows, qual_words, TI, RPI, s);
  else if (sub_lds && qual_lds) launch_emit_variant<0, 0, true, true>(P, B, grid, lds, sub_rows, qual_words, TI, RPI, s);
  else if (sub_lds) launch_emit_variant<0, 0, true, false>(P, B, grid, lds, sub_rows, qual_words, TI, RPI, s);
  else launch_emit_variant<0, 0, false, false>(P, B, grid, lds, sub_rows, qual_words, TI, RPI, s);
  if (mode == 0 && after_main) (void)hipEventRecord(after_main, s);
}
void launch_encode(uint8_t* buf, size_t bytes, hipStream_t s) {  // bytes is a multiple of 16
  if (!bytes) return;
  const size_t n16 = bytes / 16;
  uint32_t grid = (uint32_t)std::min<size_t>((n16 + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(encode_kernel, dim3(grid), dim3(256), 0, s, (uint4*)buf, n16);
}
void launch_gc(const uint8_t* chains, const uint64_t* chain_off, const sg_gc_window* wins, uint64_t n, int32_t* out, hipStream_t s) {
  if (!n) return;
  uint32_t grid = (uint32_t)((n + 3) / 4);
  hipLaunchKernelGGL(gc_kernel, dim3(grid), dim3(256), 0, s, chains, chain_off, wins, n, out);
}

}  // namespace sg
