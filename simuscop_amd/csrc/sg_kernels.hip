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
//   block_base_kernel  record offsets: in-block prefix by indel_kernel, block and segment bases here (replaces the
//                    50 MB per-worker buffers + SeqWriter mutex, Segment.cpp:695-707,834-846;
//                    lib/seqwriter/SeqWriter.cpp:49-54)
//   emit_fast_kernel, emit_slow_kernel, emit_kernel (generic) + header_kernel
//                    Profile::predict sampling loop  (Profile.cpp:1636-1700) + FASTQ formatting
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
template <int ROUNDS>
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
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
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  philox4x32<10>(c0, c1, c2, c3, k0, k1, out);
}
// the per-base draws (KIND_BASE): kBaseRounds rounds (sg_device.h)
__device__ __forceinline__ void philox_base(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  philox4x32<kBaseRounds>(c0, c1, c2, c3, k0, k1, out);
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
#define PLAN_S 16

__device__ __forceinline__ bool plan_attempt(const DevProfile& P, const DevBatch& B, const sg_window& win, uint64_t w,
                                             uint32_t attempt, uint64_t clen, uint32_t c3, PairRec& r, const uint32_t* isz_row,
                                             uint32_t seg_size, uint64_t chain_off) {
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
  r.win = (uint32_t)w; r.fl = flen | (strand << 31);
  r.namepos = pos % seg_size;
  r.foff = chain_off + win.hap_base + pos;
  r.pad = 0;
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
  // the pass's first kernel also clears its counters and results (nothing here uses them; one command less in the stream)
  if (blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < kTotalsBytes / 4u; i += blockDim.x) ((uint32_t*)B.totals)[i] = 0u;
  const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t w = gid / PLAN_S;
  const uint32_t j = (uint32_t)(gid % PLAN_S);
  if (w >= B.n_windows) return;
  const sg_window win = B.windows[w];
  const int n = win.n_reads;
  const uint32_t planned = n <= 0 ? 0u : (B.paired ? ((uint32_t)n + 1u) / 2u : (uint32_t)n);
  const uint64_t clen = B.chain_len[win.chain];
  const uint32_t seg_size = B.seg_size[win.seg];
  const uint64_t chain_off = B.chain_off[win.chain];
  const uint32_t c3 = dev_ctx(KIND_PLAN, 0, B.batch_id);
  // can any attempt of this window fail?
  const uint64_t last_start = win.hap_base + win.spos + win.len - 1u;
  const uint64_t need = B.paired ? (uint64_t)P.isz_hi : (uint64_t)win.len;
  const uint32_t shortest = B.paired ? (uint32_t)P.isz_lo : win.len;
  const bool safe = last_start + need <= clen && shortest >= (uint32_t)P.L;
  if (safe) {
    for (uint32_t k = j; k < planned; k += PLAN_S) {
      PairRec r;
      plan_attempt(P, B, win, w, k, clen, c3, r, isz_row, seg_size, chain_off);
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
    if (!plan_attempt(P, B, win, w, attempt++, clen, c3, r, isz_row, seg_size, chain_off)) {
      if (++fail > 1000) break;
      continue;
    }
    r.k = done;
    B.pairs[win.slot_base + done] = r;
    done++;
  }
  for (uint32_t k = done; k < planned; k++) {
    PairRec r;
    r.win = (uint32_t)w; r.namepos = 0; r.fl = 0; r.k = k; r.foff = 0; r.pad = 0;
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

// Orders one wave's LDS writes before its later LDS reads by other lanes (wave-private staging rows).  Wavefront scope:
// the LDS executes one wave's operations in issue order, so only the compiler has to be held back -- a workgroup-scope
// fence would also wait for every global store the wave has in flight (s_waitcnt vmcnt(0)), several times per read group.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Byte offset of read t's record in its mate's text: base of its segment of 2^seg_shift blocks + base of its block of
// 256 reads inside the segment + prefix inside the block (indel_kernel, block_base_kernel)
__device__ __forceinline__ uint64_t rec_offset(const DevBatch& B, uint32_t m, uint32_t t) {
  const uint32_t blk = t >> 8;
  const uint64_t* segbase = (const uint64_t*)((const uint8_t*)B.totals + kTotalsSegBase);
  return segbase[m * 16u + (blk >> B.seg_shift)] + B.blkbase[(size_t)m * ((B.n_slots + 255u) >> 8) + blk] +
         B.recloc[(size_t)m * B.n_slots + t];
}

// ------------------------------------------------------------------------------------------------
// indel pass: one lane per fragment
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void indel_kernel(DevProfile P, DevBatch B) {
  // lane = fragment, both of its reads: what the two share -- the PairRec, the window's name base, the digits of the
  // name -- is fetched and made once.
  // The reads' 48-byte rows leave through LDS: written by their lanes (16 bytes and single characters at a stride of 48:
  // straight to memory every store instruction touched 64 cache lines), stored by the wave as 3 KB in one piece; mate 2's
  // rows reuse the space (its rows differ from mate 1's in the middle 16 bytes and in one character).
  __shared__ uint4 row_lds[256 * 3];
  // reads with at least one indel candidate (16 % at XTen rates), as local index | mate << 8, and what their walk found:
  // {events, length change}, first event.  A wave in which every lane walks its own read's candidates pays for its
  // unluckiest lane with most lanes idle; the listed reads are walked 64 to a wave instead.
  __shared__ uint16_t cand[512];
  __shared__ uint32_t cand_n;
  __shared__ uint2 found[2][256];
  __shared__ uint32_t found_first[2][256];
  __shared__ uint32_t wave_len[2][4];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t t = blockIdx.x * blockDim.x + tid;
  const uint32_t nm = B.paired ? 2u : 1u;
  uint4* const wave_rows = row_lds + (tid & ~63u) * 3u;
  uint4* const my_row = wave_rows + lane * 3u;
  const uint32_t t_wave = t - lane;  // the wave's first slot
  // the table of the indel distances (see below), staged by the whole block
  __shared__ uint64_t gap_lds[256];
  const uint64_t* gap = P.gap_row;
  if ((uint32_t)P.L + 1u <= 256u) {  // (longer reads: from L2)
    for (uint32_t i = tid; i <= (uint32_t)P.L; i += blockDim.x) gap_lds[i] = P.gap_row[i];
    gap = gap_lds;
  }
  if (tid == 0u) cand_n = 0u;
  const bool in_batch = t < B.n_slots;
  PairRec rec = {};
  if (in_batch) rec = B.pairs[t];
  const uint32_t flen = rec.fl & 0x7FFFFFFFu;
  const bool live = flen != 0u;
  const uint32_t L = (uint32_t)P.L;
  // What the reads' rows need beyond the PairRec (plan_kernel put the window's part there), fetched before anything
  // else: the window's name base and the bad-block bits, gathers whose latency the Philox calls below cover.
  uint32_t fragcount = 0, touches_bad[2] = {0, 0};
  const uint64_t foff = rec.foff;
  const uint32_t rev1 = rec.fl >> 31;  // SE: the read's strand (PE: mate 1 forward, mate 2 reverse)
  if (live) {
    fragcount = B.win_namebase[rec.win] + rec.k + 1u;
    // A read's template is the first (forward) / last (reverse) L bases of the fragment; does it -- with the two context
    // bases before it and the slack of the emit kernel's last item -- touch a 64-base block holding a non-ACGT base?
    // Such reads go through the generic item code (the straight-line kernel reads 2-bit codes).
    for (uint32_t m = 0; m < nm; m++) {
      const bool rev = B.paired ? (m == 1u) : (rev1 != 0u);
      const uint64_t t0 = (rev ? foff + flen - L : foff) - 8u, t1 = t0 + L + 24u;  // inside the guard bytes
      uint32_t bad = 0;
      for (uint64_t b = t0 >> 6; b <= (t1 >> 6); b++) bad |= (B.chains_bad[b >> 4] >> (b & 15u)) & 1u;
      touches_bad[m] = bad;
    }
  }
  __syncthreads();  // gap_lds, cand_n
  // Sequencing indels by skipping ahead (DevProfile::gap_row): every template position is an indel candidate with
  // probability evB / 2^64, independently, so the distance to the next one is geometric and is drawn directly -- call
  // (slot, e, 0) of the read's e-th candidate: x = words (1, 0) against the table P(no candidate in k positions),
  // y = words (3, 2): an insertion iff floor(y evB / 2^64) < evA.  A read without indels (84 % at XTen rates) costs ONE
  // call and ONE compare (x < gap[L]); testing every position took 19 calls per 151-base read.
  for (uint32_t m = 0; m < nm; m++) {
    found[m][tid] = make_uint2(0u, 0u);
    found_first[m][tid] = 0u;
    bool has = false;
    if (live) {
      uint32_t x4[4];
      philox4x32_10(t + B.slot_offset, 0u, 0, dev_ctx(KIND_INDEL, m, B.batch_id), B.k0, B.k1, x4);
      const uint64_t x = ((uint64_t)x4[1] << 32) | x4[0];
      has = !(x < gap[L]);
    }
    const unsigned long long hm = __ballot(has);
    if (hm) {
      uint32_t base = 0;
      if (lane == (uint32_t)__builtin_ctzll(hm)) base = atomicAdd(&cand_n, (uint32_t)__popcll(hm));
      base = __shfl(base, __builtin_ctzll(hm), 64);
      if (has) cand[base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = (uint16_t)(tid | (m << 8));
    }
  }
  __syncthreads();
  {
    const uint32_t nc = cand_n;
#pragma unroll 1
    for (uint32_t i = tid; i < nc; i += blockDim.x) {
      const uint32_t e16 = cand[i], lt = e16 & 255u, m = e16 >> 8;
      const uint32_t tt = blockIdx.x * blockDim.x + lt;
      uint32_t* ev = B.events + ((size_t)m * B.n_slots + tt) * SG_MAX_EVENTS;
      const uint32_t c3 = dev_ctx(KIND_INDEL, m, B.batch_id);
      int j = 0, dl = 0;
      uint32_t nev = 0, first_ev = 0, e = 0;
#pragma unroll 1
      while (j < (int)L) {
        uint32_t x4[4];
        philox4x32_10(tt + B.slot_offset, e++, 0, c3, B.k0, B.k1, x4);
        const uint64_t x = ((uint64_t)x4[1] << 32) | x4[0], y = ((uint64_t)x4[3] << 32) | x4[2];
        const int room = (int)L - j;
        if (x < gap[room]) break;                 // no candidate before the read's end
        int lo = 0, hi = room - 1;                // g = #{k in [1, room): x < gap[k]} (gap decreases)
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (x < gap[mid]) lo = mid; else hi = mid - 1;
        }
        const int jj = j + lo;
        const bool is_ins = __umul64hi(y, P.evB) < P.evA;
        j = jj + 1;
        if (is_ins) {
          uint32_t len = row_search(P.ins_row, P.ins_lg, aux_draw(B, tt, (uint32_t)jj, 0, m));
          if (len > 0) {
            if (nev < SG_MAX_EVENTS) ev[nev] = ev_pack((uint32_t)jj, len, 0);
            if (nev == 0) first_ev = ev_pack((uint32_t)jj, len, 0);
            nev++;
            dl += (int)len;
          }
        } else {
          uint32_t len = row_search(P.del_row, P.del_lg, aux_draw(B, tt, (uint32_t)jj, 0, m));
          if (len > 0) {
            uint32_t k = min((uint32_t)((int)L - jj), len);
            if (nev < SG_MAX_EVENTS) ev[nev] = ev_pack((uint32_t)jj, k, 1);
            if (nev == 0) first_ev = ev_pack((uint32_t)jj, k, 1);
            nev++;
            dl -= (int)k;
            j = jj + (int)k;
          }
        }
      }
      found[m][lt] = make_uint2(nev, (uint32_t)dl);
      found_first[m][lt] = first_ev;
    }
  }
  __syncthreads();
  // ---- the reads' rows, their record lengths and the prefix of those inside the block ----
  // Per-read 48-byte row for the emit kernels: m0 = fragment offset + the two numbers of the read's name
  // "@popu#chr#pos%segsize#fragCount[/m]" (Segment.cpp:780,809,824), m1 = lengths, reciprocal, event, then the
  // name's own part "pos#count[/m]\n" as text when it fits 16 bytes (emit_fast_kernel stores it behind the batch's
  // constant prefix; this kernel has the VALU time for the digits, that one has not).
  const uint32_t namepos = rec.namepos;
  const uint32_t hdr = B.prefix_len + ndigits(namepos) + 1u + ndigits(fragcount) + (B.paired ? 2u : 0u) + 1u;
  uint32_t mate_at = 0;  // where the name's mate digit sits in the text
  my_row[0] = live ? make_uint4((uint32_t)foff, (uint32_t)(foff >> 32), namepos, fragcount) : make_uint4(0, 0, 0, 0);
  my_row[2] = make_uint4(0, 0, 0, 0);
  if (live && hdr - B.prefix_len <= 16u) {
    uint8_t* hb = (uint8_t*)(my_row + 2);
    uint32_t o = 0;
    auto put_dec = [&](uint32_t v) {  // the digits of a number from its last one backwards
      const uint32_t nd = ndigits(v);
      for (uint32_t k = nd; k-- > 0u;) {
        const uint32_t qd = v / 10u;
        hb[o + k] = (uint8_t)('0' + (v - qd * 10u));
        v = qd;
      }
      o += nd;
    };
    put_dec(namepos);
    hb[o++] = '#';
    put_dec(fragcount);
    if (B.paired) { hb[o++] = '/'; mate_at = o; hb[o++] = '1'; }
    hb[o] = '\n';
  }
  uint32_t rl[2] = {0, 0};
  for (uint32_t m = 0; m < nm; m++) {
    if (m == 1u) {
      wave_lds_sync();  // mate 1's rows have been read
      if (mate_at) ((uint8_t*)(my_row + 2))[mate_at] = '2';
    }
    uint4 mid = make_uint4(0, 0, 0, 0);
    if (live) {
      const uint2 f = found[m][tid];
      uint32_t nev = f.x;
      int dl = (int)f.y;
      if ((int)L + dl < 50) { nev = 0; dl = 0; }  // Profile.cpp:1627-1634
      if (nev > SG_MAX_EVENTS) { atomicOr((unsigned long long*)&B.totals[3], 1ull); nev = 0; dl = 0; }
      const uint32_t np = (uint32_t)((int)L + dl);
      rl[m] = hdr + 2u * np + 4u;
      const uint32_t rev = B.paired ? m : rev1;
      // ceil(2^32 / np) = floor((2^32-1)/np) + 1: bin = (i*bins*inv) >> 32 is exact while i*bins*np < 2^32
      // (sg_load_profile rejects profiles that could violate the bound)
      // m1.w: the event itself for single-event reads (handled inline by the emit kernel)
      mid = make_uint4(flen | (touches_bad[m] << 30) | (rev << 31), np | (nev << 16) | (hdr << 22), 0xFFFFFFFFu / np + 1u,
                       nev == 1u ? found_first[m][tid] : 0u);
    }
    my_row[1] = mid;
    if (t_wave < B.n_slots) {
      wave_lds_sync();
      uint4* dst = B.meta + ((size_t)m * B.n_slots + t_wave) * 3;
      const uint32_t n_pieces = min(B.n_slots - t_wave, 64u) * 3u;  // 16-byte pieces of the wave's rows that exist
#pragma unroll
      for (uint32_t i = 0; i < 3u; i++) {
        const uint32_t q = i * 64u + lane;
        if (q < n_pieces) dst[q] = wave_rows[q];
      }
    }
  }
  // Record offsets: exclusive prefix of the record lengths inside the block of 256 reads, here; the blocks' bases by
  // one small kernel afterwards (block_base_kernel).  offset = base[block] + prefix (rec_offset()).
  uint32_t incl[2];
  for (uint32_t m = 0; m < nm; m++) {
    uint32_t v = rl[m];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(v, d, 64);
      if ((int)lane >= d) v += up;
    }
    incl[m] = v;
    if (lane == 63u) wave_len[m][wv] = v;
  }
  __syncthreads();
  for (uint32_t m = 0; m < nm; m++) {
    uint32_t base = 0, all = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4u; i++) {
      const uint32_t w = wave_len[m][i];
      if (i < wv) base += w;
      all += w;
    }
    if (in_batch) B.recloc[(size_t)m * B.n_slots + t] = base + incl[m] - rl[m];
    if (tid == 0u) B.blkbase[(size_t)m * gridDim.x + blockIdx.x] = all;
  }
}

// ------------------------------------------------------------------------------------------------
// Read names "@popu#chr#pos%segsize#fragCount[/m]\n" (Segment.cpp:780,809,824).  emit_fast_kernel stores the names of
// its reads itself (phase 0: the batch's prefix from the kernel arguments + the read's part from its row) when the
// prefix fits 16 bytes; header_kernel (one lane per read, after the offset scan) serves the generic emit kernel and
// longer prefixes.
// ------------------------------------------------------------------------------------------------
// any name, byte stream -> unaligned dword stores (7 instead of 27 byte stores per record)
__device__ __attribute__((noinline)) void write_name(uint8_t* hp, const uint8_t* __restrict__ prefix, uint32_t prefix_len,
                                                     uint32_t namepos, uint32_t fragcount, uint32_t paired, uint32_t m) {
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
  for (uint32_t i = 0; i < prefix_len; i++) push(prefix[i]);
  push_dec(namepos);
  push('#');
  push_dec(fragcount);
  if (paired) { push('/'); push('1' + m); }
  push('\n');
  for (uint32_t i = 0; i < nb; i++) hp[i] = (uint8_t)(w >> (8u * i));
}

__global__ __launch_bounds__(256) void header_kernel(DevBatch B) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t m = blockIdx.y;
  if (t >= B.n_slots) return;
  const size_t idx = (size_t)m * B.n_slots + t;
  const uint4 m1 = B.meta[idx * 3 + 1];
  if (!(m1.x & 0x7FFFFFFFu)) return;
  const uint4 m0 = B.meta[idx * 3];
  const uint64_t ooff = rec_offset(B, m, t);
  if (ooff + ((m1.y >> 22) + 2u * (m1.y & 0xFFFFu) + 4u) > B.out_cap[m]) return;  // (the record's length: header + 2 lines + 4 line breaks and '+')
  if (!(B.diag & 4u)) write_name(B.out[m] + ooff, B.prefix, B.prefix_len, m0.z, m0.w, B.paired, m);
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
  // one block per row (mate): thread i owns a run of ceil(nblk / 1024) consecutive sums -- adds them up, the 1024 run
  // totals are scanned in LDS (one Hillis-Steele pass), then every thread writes the exclusive prefixes of its run
  __shared__ uint64_t buf[1024];
  const uint32_t m = blockIdx.x;
  uint64_t* b = bsum + (size_t)m * nblk;
  const uint32_t run = (nblk + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * run, hi = min(lo + run, nblk);
  uint64_t mine = 0;
  for (uint32_t i = lo; i < hi; i++) mine += b[i];
  buf[threadIdx.x] = mine;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const uint64_t t = threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
    __syncthreads();
    buf[threadIdx.x] += t;
    __syncthreads();
  }
  uint64_t carry = buf[threadIdx.x] - mine;
  for (uint32_t i = lo; i < hi; i++) {
    const uint64_t v = b[i];
    b[i] = carry;
    carry += v;
  }
  if (threadIdx.x == 1023u) totals[m] = buf[1023];
}

// Record offsets, middle level: the block sums indel_kernel left in blkbase, cut into <= 16 segments of 2^seg_shift
// blocks; workgroup j scans segment j in place (thread i owns a run of consecutive sums), the workgroup that arrives
// last turns the 16 segment totals into segment bases and the mate's text size (totals[m]).  One launch, 16 CUs per
// mate busy instead of one.
__global__ __launch_bounds__(1024) void block_base_kernel(DevBatch B) {
  __shared__ uint64_t buf[1024];
  __shared__ uint32_t ticket;
  const uint32_t m = blockIdx.y, j = blockIdx.x;
  const uint32_t nblk = (B.n_slots + 255u) >> 8;
  uint64_t* b = B.blkbase + (size_t)m * nblk;
  uint64_t* segbase = (uint64_t*)((uint8_t*)B.totals + kTotalsSegBase) + m * 16u;
  uint32_t* arrived = (uint32_t*)((uint8_t*)B.totals + kTotalsSegBase + 2 * 16 * 8) + m;
  const uint32_t s0 = min(j << B.seg_shift, nblk), s1 = min((j + 1u) << B.seg_shift, nblk);
  const uint32_t run = ((s1 - s0) + 1023u) / 1024u;
  const uint32_t lo = min(s0 + threadIdx.x * run, s1), hi = min(lo + run, s1);
  uint64_t mine = 0;
  for (uint32_t i = lo; i < hi; i++) mine += b[i];
  buf[threadIdx.x] = mine;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const uint64_t t = threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
    __syncthreads();
    buf[threadIdx.x] += t;
    __syncthreads();
  }
  uint64_t carry = buf[threadIdx.x] - mine;
  for (uint32_t i = lo; i < hi; i++) {
    const uint64_t v = b[i];
    b[i] = carry;
    carry += v;
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(&segbase[j], buf[1023], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    ticket = atomicAdd(arrived, 1u);
  }
  __syncthreads();
  if (ticket + 1u == gridDim.x && threadIdx.x == 0) {  // every segment total is in: bases, text size
    __threadfence();
    uint64_t acc = 0;
    for (uint32_t i = 0; i < gridDim.x; i++) {
      const uint64_t v = __hip_atomic_load(&segbase[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&segbase[i], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc += v;
    }
    B.totals[m] = acc;
  }
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

// A template byte of the chains (A0 C1 T2 G3, N = 4, X = 6, anything else = 5) as the sampling loop sees it: a base code in
// read orientation (reverse reads: complemented, A0 <-> T2, C1 <-> G3, Segment.cpp:81-103), 4 = unknown, 6 = a literal 'X'.
// The reference's k-mer trie holds the place-holder contexts "XXb" and "Xbb" of a read's first bases under the character
// 'X' (Profile.cpp:94-101), so a literal X in the genome -- on a forward read: the complement of anything but A/C/G/T is
// 'N' (Segment.cpp:99) -- is the same place holder in the middle of a read.  `x_known` false (--strict-bases): an X is an
// unknown base like any other.
__device__ __forceinline__ uint32_t template_code(uint32_t raw, bool rev, bool x_known) {
  return raw < 4u ? (rev ? raw ^ 2u : raw) : ((raw == 6u && !rev && x_known) ? 6u : 4u);
}

// Source codes of a read that carries sequencing-indel events (rare path, kept OUT OF LINE and
// un-unrolled: inlined it pushed the kernel past the 64 KB instruction cache and cost 2x).
// Walks the events of Profile::predict's first loop (Profile.cpp:1607-1658) with a forward cursor and
// returns 13 nibbles, one per output position p = i0-5+q: base code 0..3 (template bases complemented
// for reverse reads), 4/5 = not in `bases`, bit 3 set = inserted base (already a profile code).
__device__ __noinline__ uint64_t slow_codes(const uint8_t* frag, uint32_t flen, uint32_t rev, const uint32_t* ev,
                                            uint32_t nev, uint32_t np, uint32_t i0, uint32_t K, uint32_t slot,
                                            uint32_t ctx_aux, uint32_t k0, uint32_t k1, bool x_known) {
  uint64_t out = 0;
  uint32_t e = 0;
  int shift = 0;  // output index - template index of the template bases after the consumed events
  uint32_t nextw = nev ? ev[0] : 0xFFFFFFFFu;
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
        v = template_code(frag[rev ? flen - 1u - jt : jt], rev != 0u, x_known);
      }
    }
    out |= (uint64_t)v << (4 * q);
  }
  return out;
}

// One item = output positions [8c, 8c+8) of one read: sample and store bases + qualities.
//   m0 = {frag_lo, frag_hi, out_lo, out_hi}   m1 = {flen | rev<<31, np | nev<<16 (6 bits) | hdr<<22, inv, -}
template <int KT, bool SUB_LDS>
__device__ __forceinline__ void emit_item(const DevProfile& P, const DevBatch& B, const uint4* lds_sub, const uint4* gsub,
                                          uint32_t m, const uint4 m0, const uint4 m1, uint32_t slot, uint32_t c, bool active) {
  const uint32_t K = KT ? (uint32_t)KT : (uint32_t)P.kmer;
  const uint32_t bins = (uint32_t)P.bins;
  const uint32_t ctxmask = (1u << (2 * K)) - 1u;
  const uint32_t flen = m1.x & 0x3FFFFFFFu;  // bit 30: the read touches a non-ACGT block (straight-line kernel only)
  const bool rev = (m1.x >> 31) != 0;
  const uint32_t np = m1.y & 0xFFFFu, nev = (B.diag & 32u) ? 0u : ((m1.y >> 16) & 0x3Fu), hdr = m1.y >> 22;
  const uint32_t inv = m1.z;
  const uint8_t* frag = B.chains + (((uint64_t)m0.y << 32) | m0.x);
  const uint32_t i0 = 8u * c;
  const bool x_known = B.strict_bases == 0u;  // a literal X of the genome is the trie's place holder (template_code)

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
    for (int q = 0; q < 13; q++) code[q] = template_code((w[q >> 2] >> ((q & 3) * 8)) & 0xFFu, rev, x_known);
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
        const uint32_t c2 = template_code((w2[q >> 2] >> ((q & 3) * 8)) & 0xFFu, rev, x_known);
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
                                         dev_ctx(KIND_AUX, m, B.batch_id), B.k0, B.k1, x_known);
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
  // ---- k-mer context (Profile::initKmers order, Profile.cpp:70-124).  The reference looks the K characters ending at a
  // position up in a trie that holds b^K and the place-holder forms X^a b^(K-a) (Profile.cpp:94-101, 220-226); the
  // characters before a read's first base are 'X' (Profile.cpp:1660-1663), and so is a literal X of the genome.  A position
  // has a context iff its window is a run of X (xr of them) followed by vc >= 1 bases; the context then has min(vc, K)
  // bases.  State: vc = bases at the end of what was seen, xr = the X run right before them, xc = the X run at the end.
  uint32_t ctxv = 0, vc = 0, xr = 0, xc = 0;
  auto ctx_step = [&](uint32_t c) {
    const bool valid = c < 4u;
    ctxv = ((ctxv << 2) | (c & 3u)) & ctxmask;
    if (valid) { if (vc == 0u) xr = xc; vc = min(vc + 1u, K); xc = 0u; }
    else { vc = 0u; xr = 0u; xc = c == 6u ? min(xc + 1u, K) : 0u; }
  };
#pragma unroll
  for (int q = 0; q < 5; q++)
    if (q >= 6 - (int)K) ctx_step((int)i0 - 5 + q >= 0 ? code[q] : 6u);   // the K-1 positions before i0
  // ---- four Philox calls: heads and tails of output positions i0 .. i0+7 (call = i/4, c2 = 0 heads / 1 tails, word = i%4):
  //   substitution draw = heads[31:16] << 16 | tails[31:16],  quality draw = heads[15:0] << 16 | tails[15:0]
  uint32_t xh[8], xt[8];
  const uint32_t c3b = dev_ctx(KIND_BASE, m, B.batch_id);
#pragma unroll
  for (int g = 0; g < 2; g++) {
    if (B.diag & 8u) {
      for (int z = 0; z < 4; z++) { xh[4 * g + z] = (slot * 2654435761u) ^ (c * 40503u + (4 * g + z) * 0x9E3779B9u); xt[4 * g + z] = ~xh[4 * g + z]; }
    } else {
      philox_base(slot + B.slot_offset, 2u * c + (uint32_t)g, 0, c3b, B.k0, B.k1, xh + 4 * g);
      philox_base(slot + B.slot_offset, 2u * c + (uint32_t)g, 1, c3b, B.k0, B.k1, xt + 4 * g);
    }
  }
  const uint32_t lgW = P.lgW;
  uint32_t sw[2] = {0, 0}, qw[2] = {0, 0};
#pragma unroll
  for (int h = 0; h < 8; h++) {
    const uint32_t i = i0 + (uint32_t)h;
    const uint32_t cd = code[5 + h];
    const bool valid = cd < 4u;
    ctx_step(cd);
    const uint32_t xs = (xh[h] & 0xFFFF0000u) | (xt[h] >> 16), xq = (xh[h] << 16) | (xt[h] & 0xFFFFu);
    const uint32_t bin = min(__umulhi(i * bins, inv), bins - 1u);  // i*binCount/n' (clamp only guards idle lanes)
    const uint32_t mlen = max(vc, 1u);
    const uint32_t mmask = (1u << (2u * mlen)) - 1u;
    const bool ctx_ok = vc >= 1u && vc + xr >= K && !(B.diag & 16u);
    // contexts with m real bases start at (4^m-4)/3 = (0x55555555 & (4^m-1)) - 1
    const uint32_t kidx = ctx_ok ? ((0x55555555u & mmask) - 1u) + (ctxv & mmask) : 0u;
    const uint4 row = SUB_LDS ? lds_sub[kidx * bins + bin] : gsub[(size_t)kidx * bins + bin];
    // identity-first row: j = max(j0, #{xs > D_i}), called base = o_j (o_0 = the reference base)
    const uint32_t j = max((uint32_t)(xs > row.x) + (uint32_t)(xs > row.y) + (uint32_t)(xs > row.z), row.w & 3u);
    const uint32_t k = ctx_ok ? ((row.w >> (2u * j + 2u)) & 3u) : cd;  // unknown context: the base is copied (Profile.cpp:1531-1533)
    const bool kvalid = k < 4u;
    const uint32_t kk = kvalid ? k : 0u;
    // alias column of row (reference base, called base, bin)
    const uint2 e = P.alias[(((size_t)((valid ? cd : 0u) * 4u + kk) * bins + bin) << lgW) + (xq >> (32u - lgW))];
    const uint32_t u = xq & ((1u << (32u - lgW)) - 1u);
    const uint32_t qi = (B.diag & 16u) ? 7u + (xq >> 29) : (u < e.x ? (e.y & 0xFFu) : ((e.y >> 8) & 0xFFu));
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

// Which read groups a wave does.  The groups of a mate are cut into eight contiguous partitions, one per XCD (workgroups
// go to the XCDs round-robin by their linear id: x % 8; a grid that is not a multiple of eight gets one partition): the
// workgroups of an XCD work through ONE eighth of the haplotype and of the text, whose lines then live in that XCD's L2.
// Inside its partition a wave takes its first group by position and every further one from the partition's counter
// (CUs and waves do not run at one speed; with a static stride the grid drained 18 % late), then from the other
// partitions' counters.  One counter per partition,
// each in a cache line of its own, because one address serves ~85 M atomics per second: a single counter per mate was
// the ceiling of the whole kernel for 75-base reads (410 k groups per C2 pass = 4.8 ms whatever the groups cost).
struct GroupRuns {
  uint32_t* counters;        // this kernel's and mate's eight counters, 32 words apart
  uint32_t ngroups, P, part, tried, waves_per_part;
  uint32_t g;                // the wave's current group (uniform); valid while more()
  bool ok;
  __device__ __forceinline__ uint32_t lo(uint32_t p) const { return (uint32_t)((uint64_t)ngroups * p / P); }
  __device__ __forceinline__ GroupRuns(uint64_t* totals, uint32_t kernel, uint32_t mate, uint32_t n, uint32_t wv, uint32_t waves_per_wg) {
    ngroups = n;
    P = (gridDim.x & 7u) == 0u ? 8u : 1u;
    part = P == 8u ? blockIdx.x & 7u : 0u;
    tried = 0;
    waves_per_part = (gridDim.x / P) * waves_per_wg;
    counters = (uint32_t*)((uint8_t*)totals + 128) + (kernel * 2u + mate) * 8u * 32u;
    g = lo(part) + (P == 8u ? blockIdx.x >> 3 : blockIdx.x) * waves_per_wg + wv;
    ok = g < lo(part + 1u);
  }
  __device__ __forceinline__ bool more() const { return ok; }
  // the next group: from the own partition's counter; when that is exhausted, from the next partition's (the XCDs do not
  // finish together either), until all eight have been tried
  __device__ __forceinline__ void next(uint32_t lane) {
    for (;;) {
      uint32_t nx = 0;
      if (lane == 0u) nx = atomicAdd(counters + part * 32u, 1u);
      nx = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
      const uint32_t first_dyn = lo(part) + waves_per_part, end = lo(part + 1u);
      g = first_dyn + nx;
      if (first_dyn < end && nx < end - first_dyn) { ok = true; return; }
      part = part + 1u == P ? 0u : part + 1u;
      if (++tried >= P) { ok = false; return; }
    }
  }
};

template <int KT, bool SUB_LDS>
__global__ __launch_bounds__(EMIT_THREADS) void emit_kernel(DevProfile P, DevBatch B, uint32_t sub_rows, uint32_t TI, uint32_t RPI) {
  extern __shared__ uint4 smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t m = blockIdx.y;
  // launched before the host knew the text size (sg_api.cpp run_pass): a text that does not fit the buffers is left to
  // a second launch into larger ones
  if (B.totals[m] + 64u > B.out_cap[m]) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr((unsigned long long*)&B.totals[3], 4ull);
    return;
  }
  const uint32_t tm = B.paired ? m : 0u;  // SE always samples from the mate-1 tables (Segment.cpp:770,777)
  // ---- LDS carve-up: [sub rows][per-wave metadata rows]; the alias columns are read through L2 ----
  uint4* lds_sub = smem;
  uint4* lds_meta_all = smem + (SUB_LDS ? sub_rows : 0u);
  const uint4* gsub = P.sub + (size_t)tm * P.sub_mate_rows;
  if (SUB_LDS)
    for (uint32_t i = tid; i < sub_rows; i += EMIT_THREADS) lds_sub[i] = gsub[i];
  __syncthreads();
  uint4* meta_rows = lds_meta_all + (size_t)wv * 64 * (META_ROW / 16);

  const uint32_t G = RPI * (64u / RPI);             // reads per wave group
  const uint32_t ngroups = (B.n_slots + G - 1u) / G;
  const uint32_t sub = lane / TI, c_lane = lane - sub * TI;  // fixed lane -> (read in iteration, item)
  const bool lane_ok = sub < RPI;

  // groups from a counter of its own: this kernel also re-emits a batch after emit_fast_kernel has run (sg_result)
  GroupRuns runs(B.totals, 1u, m, ngroups, wv, EMIT_WAVES);
  for (; runs.more(); runs.next(lane)) {
    const uint32_t g = runs.g;
    // ================= phase 0: lane = read: its 32-byte row (coalesced) into LDS =================
    const uint32_t t = g * G + lane;
    uint32_t items = 0;
    uint4 my0 = make_uint4(0, 0, 0, 0), my1 = make_uint4(0, 0, 0, 0);
    if (lane < G && t < B.n_slots) {
      const size_t idx = (size_t)m * B.n_slots + t;
      my0 = B.meta[idx * 3];
      my1 = B.meta[idx * 3 + 1];
      const uint64_t ooff = rec_offset(B, m, t);
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
        emit_item<KT, SUB_LDS>(P, B, lds_sub, gsub, m, m0, m1, g * G + r, active ? c : 0u, active);
      }
    }

    wave_lds_sync();  // the next group's phase 0 rewrites the metadata rows
  }
}

// ------------------------------------------------------------------------------------------------
// emit, straight-line variant for the common profile shape (kmer == 3, tables in LDS).  Same results as
// emit_kernel, far fewer instructions.  Cost model (tools/valu_microbench.hip, 4 waves per SIMD): only
// add / sub / and / or / xor / right shift / mov issue at 2.3 cycles per wave instruction (v_bitop3_b32 2.8);
// compares, selects, left shifts, bit-field extracts, multiplies and every other three-operand integer op take
// 4.2.  The 8-base block is therefore written in subtractions, right shifts and three-input boolean ops:
//   * the 13 source codes of an item are packed 2 bits each (natural order A0 C1 T2 G3) into one word, so a
//     k-mer context is a shift and a mask; the tables are laid out for that digit order on the host;
//   * per base ONE 32-bit draw (the "heads" word): 12 bits decide "no substitution" against the context row's
//     keep count (identity-first rows: one subtraction, the sign is the answer), 20 bits pick an alias column of
//     the (reference == called) quality row and the side of its threshold (again a subtraction; the sign,
//     extended over the high bits, selects the symbol through a three-input boolean op);
//   * a base the heads cannot decide -- a substitution (0.35 % of the bases), a head equal to a threshold's --
//     sets a flag bit; flagged bases are redone exactly by the fix-up loop after the block (full rows from L2,
//     the "tails" Philox call only for a head on a threshold);
//   * table addresses come from a per-item look-up row (bin offsets of the item's eight positions; the short
//     contexts of a read's first two bases are table regions of their own, selected by that row);
//   * a group's reads walk three loops of steps, 64 items per step: the plain reads (no indel, the profile's own length)
//     their own stream, the items of one-indel reads that lie wholly before or behind the indel a list of their own
//     (the plain step with computed bins and a shifted template), everything else one general stream ordered by
//     event class;
//   * what is per read rather than per item: the name goes out at phase 0 (lane = read; prefix from the kernel
//     arguments, the read's own part from its row), the partial last item, "\n+\n" and '\n' are stored by a per-read
//     pass once per group (the last item waits in the read's own LDS row);
//   * windows holding a non-ACGT base and items two sequencing indels reach into (both rare) go to a global
//     queue for emit_slow_kernel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack4(uint32_t w) {  // four code bytes (0..3) -> 8 bits
  return (w | (w >> 6) | (w >> 12) | (w >> 18)) & 0xFFu;
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define FAST_CTX 192u  // context rows per bin: [0,64) three bases, [64,128) two bases (x4), [128,192) one base (x16)
#define LUT_ROW 12u    // dwords per item in the look-up row: so[0..7], qo[0..1], 2 unused

// v_bitop3_b32 truth tables, f(a, b, c) evaluated on a = 0xF0, b = 0xCC, c = 0xAA
#define BITOP_OR_AND 0xF8       // a | (b & c)
#define BITOP_XOR_AND 0x78      // a ^ (b & c)
#define BITOP_ANDN_OR 0xBA      // (a & ~b) | c

// "tails" word of output position i0 + h (KIND_BASE call (2c + h/4, 1)); rare, kept out of line
__device__ __noinline__ uint32_t tail_word(uint32_t slot, uint32_t c, uint32_t h, uint32_t c3b, uint32_t k0, uint32_t k1) {
  uint32_t y[4];
  philox_base(slot, 2u * c + (h >> 2), 1u, c3b, k0, k1, y);
  const uint32_t l = h & 3u;
  return l == 0u ? y[0] : l == 1u ? y[1] : l == 2u ? y[2] : y[3];
}

// one base of the 8-base block (sample8 below)
template <int H>
__device__ __forceinline__ void sample1(uint32_t col_sh, uint32_t col_mask, uint32_t cw, uint32_t wh, uint32_t so_h, uint32_t qo_h, uint32_t cd4_w,
                                        uint32_t& acc, uint32_t& sy) {
  // so_h, qo_h are absolute LDS byte addresses (the image's own address is part of them: the look-up rows hold it, the
  // arithmetic paths add it), read through integer-made LDS pointers: with a pointer base in the expression every
  // address cost an instruction more (an add of the base, a constant the compiler only learns at link time)
  typedef const __attribute__((address_space(3))) uint32_t* lds_word;
  // substitution: sure "none" iff the 16-bit head (high half) is below the row's keep count: d_s = keep_h - 1 - head >= 0
  const uint32_t kv4 = (cw >> (2 * H + 4)) & 0xFCu;   // the context's row word, as a byte offset
  const uint32_t keepm1 = *(lds_word)(uintptr_t)(so_h + kv4);
  const uint32_t d_s = keepm1 - (wh >> 16);
  // quality: alias column (col, cd) of the diagonal row; its low half holds col : thr_head, so the draw's low half minus
  // it is d_q = u_head - thr_head: negative -> lo, positive -> hi, zero -> the tail decides (fix-up)
  // (column * 16 by a right shift and a mask: both issue at the full rate, a bit-field extract and a left shift do not)
  const uint32_t col16 = (wh >> col_sh) & col_mask;
  const uint32_t qa = qo_h + ((cd4_w >> (8 * (H & 3))) & 0xFFu);
  const uint32_t e = *(lds_word)(uintptr_t)(qa + col16 + FAST_CTX * 4u);
  const uint32_t d_q = (wh & 0xFFFFu) - (e & 0xFFFFu);
  // flag (bits 17.. are copies of it): substitution possible (d_s < 0) or head on the threshold (d_q == 0)
  const uint32_t z = __builtin_amdgcn_bitop3_b32(d_q - 1u, d_q, d_s, BITOP_ANDN_OR);
  acc = __builtin_amdgcn_bitop3_b32(acc, z, 1u << (17 + H), BITOP_OR_AND);
  // symbol = hi ^ ((lo ^ hi) & (d_q < 0)), fields at bits 24.. and 16..: lands in byte 2 (the image holds the symbols as
  // characters: the profile's lowest quality character is added on the host)
  sy = __builtin_amdgcn_bitop3_b32(e >> 8, e, d_q, BITOP_XOR_AND);
}

// The 8-base block of the straight-line kernels: eight positions' substitution and quality decisions from their heads
// words.  Written for few instructions and, where there is a choice, for the ones that issue at the full rate (add, sub,
// and/or/xor, right shifts: 2.3 cycles against 4.2, tools/valu_microbench.hip): 16-bit fields that operand selects (SDWA)
// pick for free, byte permutes for packing, shift + mask instead of bit-field extract + left shift.
//   cw   13 source codes, 2 bits each (output positions i0-5 .. i0+7)      x[h]  heads word of position i0 + h
//   so[h] LDS byte address of position h's bin in the table image (+ the short-context region of a read's first two bases),
//   qo0/qo1 the plain bin addresses of positions 0 and 1
// Out: sw = the eight called characters (= the reference bases), qw = the eight quality characters, acc bit 17 + h = the
// head could not decide base h (possible substitution, or a head equal to a threshold's): the fix-up pass redoes it.
__device__ __forceinline__ void sample8(const uint32_t* code4, uint32_t lgW, uint32_t cw, const uint32_t (&x)[8],
                                        const uint32_t (&so)[8], uint32_t qo0, uint32_t qo1, uint32_t (&sw)[2], uint32_t (&qw)[2],
                                        uint32_t& acc_out) {
  // the items's eight reference codes as bytes code * 4 (alias column offset, and >> 2 the character selector)
  const uint32_t cd4[2] = {code4[(cw >> 10) & 0xFFu], code4[(cw >> 18) & 0xFFu]};
  const uint32_t col_sh = 12u - lgW, col_mask = ((1u << lgW) - 1u) << 4;   // (lgW <= 7)
  uint32_t acc = 0, sy[8];
  sample1<0>(col_sh, col_mask, cw, x[0], so[0], qo0, cd4[0], acc, sy[0]);
  sample1<1>(col_sh, col_mask, cw, x[1], so[1], qo1, cd4[0], acc, sy[1]);
  sample1<2>(col_sh, col_mask, cw, x[2], so[2], so[2], cd4[0], acc, sy[2]);
  sample1<3>(col_sh, col_mask, cw, x[3], so[3], so[3], cd4[0], acc, sy[3]);
  sample1<4>(col_sh, col_mask, cw, x[4], so[4], so[4], cd4[1], acc, sy[4]);
  sample1<5>(col_sh, col_mask, cw, x[5], so[5], so[5], cd4[1], acc, sy[5]);
  sample1<6>(col_sh, col_mask, cw, x[6], so[6], so[6], cd4[1], acc, sy[6]);
  sample1<7>(col_sh, col_mask, cw, x[7], so[7], so[7], cd4[1], acc, sy[7]);
  // byte 2 of four symbols words -> one word (selector bytes: 0-3 from the second operand, 4-7 from the first)
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const uint32_t p01 = __builtin_amdgcn_perm(sy[4 * g + 1], sy[4 * g], 0x0C0C0602u);
    const uint32_t p23 = __builtin_amdgcn_perm(sy[4 * g + 3], sy[4 * g + 2], 0x06020C0Cu);
    qw[g] = p01 | p23;
    // called bases = reference bases: code -> character
    sw[g] = __builtin_amdgcn_perm(0u, 0x47544341u, cd4[g] >> 2);  // "ACTG"
  }
  acc_out = acc;
}

// Base index of an item's un-shifted template window in the 2-bit copy the read walks forwards -- the forward copy, or
// for a reverse read the reverse-complement copy, where the fragment [A, A + n) starts at total - A - n: output
// position p of the read is base `index + p - 8c + 5` of that copy.  Idle lanes read a harmless in-bounds index.
template <bool PAIRED>
__device__ __forceinline__ uint32_t fast_src(const DevBatch& B, uint32_t m, const uint4 m0, const uint4 m1, uint32_t c, bool active) {
  const uint32_t flen = m1.x & 0x3FFFFFFFu;
  const bool rev = PAIRED ? (m == 1u) : ((m1.x >> 31) != 0);
  const uint32_t a = rev ? (uint32_t)B.chains_total - m0.x - flen : m0.x;  // the straight-line kernel runs on buffers < 2^31 bases
  return active ? a + 8u * c - 5u : 128u;
}

template <bool PAIRED, bool DIAG>
__device__ __forceinline__ bool fast_item(const DevProfile& P, const DevBatch& B, uint32_t img_at, const uint32_t* lut,
                                          const uint32_t* code4, uint32_t TI, uint32_t m, const uint4 m0, const uint4 m1,
                                          uint32_t slot, uint32_t c, bool active, uint4* tail_row, int d0, uint32_t n_in,
                                          uint32_t ew_in, uint32_t tj_in, uint32_t src0, uint2 wpre,
                                          __amdgpu_buffer_rsrc_t out_rsrc, uint32_t& fix, uint32_t& cw_out) {
  const uint32_t bins = (uint32_t)P.bins;
  const uint32_t dg = DIAG ? B.diag : 0u;  // SG_FDIAG timing ablations: compiled out of the production kernel
  const bool rev = PAIRED ? (m == 1u) : ((m1.x >> 31) != 0);
  // An idle lane may be looking at a row whose read is finished; its fragment offset, reciprocal and event word then
  // hold the parked last item (see below), so an idle lane must not follow them -- it reads a harmless in-bounds
  // window and has no event.  The caller has reduced the read's sequencing indels to what this item sees: d0 =
  // template index minus output index at the item's first position (events before it), n_in = events reaching into
  // it (0, 1, or 2 = too many for the two-window code), ew_in = that event with its OUTPUT position, tj_in = its
  // template position (the address of its draws).
  const uint32_t np = m1.y & 0xFFFFu, nev = active ? n_in : 0u, hdr = m1.y >> 22;
  const uint32_t inv = m1.z;
  const uint32_t i0 = 8u * c;
  // 13 source codes, 2 bits each, for output positions i0-5 .. i0+7: 26 bits of the 2-bit copy starting at base index
  // src (src0 = the un-shifted window, fast_src, already loaded into wpre by the caller one step ahead; events before
  // the item shift it by d0).  Reads touching a non-ACGT block never get here with valid codes: they are queued whole.
  const uint8_t* copy2 = rev ? B.chains2_rc : B.chains2_fwd;
  auto window = [&](uint32_t idx) -> uint32_t {
    uint2 v;
    __builtin_memcpy(&v, copy2 + (idx >> 2), 8);
    return (uint32_t)((((uint64_t)v.y << 32) | v.x) >> (2u * (idx & 3u)));
  };
  const uint32_t src = src0 + (uint32_t)d0;
  if (__ballot(d0 != 0) != 0ull) {
    if (d0 != 0) __builtin_memcpy(&wpre, copy2 + (src >> 2), 8);
  }
  const uint32_t bad = (m1.x >> 30) & 1u;
  uint32_t cw;
  if (dg & 4u) cw = slot * 2654435761u + c;  // ablation: no haplotype fetch
  else cw = (uint32_t)((((uint64_t)wpre.y << 32) | wpre.x) >> (2u * (src & 3u)));

  // reads with exactly one sequencing indel: past the event the window is shifted by +-len
  if (__ballot(nev == 1u) != 0ull) {
    int q_ins = 1, q_end = 0;  // an insertion's own positions inside the item's window
    if (nev == 1u) {
      const uint32_t ew = ew_in;
      const int ej = (int)(ew & 0xFFFFu), elen = (int)((ew >> 16) & 0x7FFFu);
      const bool del = (ew >> 31) != 0;
      const int delta = del ? elen : -elen;
      const uint32_t cw2 = window(src + (uint32_t)delta);
      const int first_shifted = del ? ej : ej + elen + 1;       // first output position reading the shifted window
      const int q0 = first_shifted - ((int)i0 - 5);               // its index in the item window
      const uint32_t keep = q0 <= 0 ? 0u : (q0 >= 13 ? 0xFFFFFFFFu : ((1u << (2 * q0)) - 1u));
      cw = (cw & keep) | (cw2 & ~keep);
      if (!del) {
        // inserted run = output positions ej+1 .. ej+elen: randomInteger(0, N-1), never the last base
        // (Profile.cpp:1564); flat draw f = p - ej of stream (slot, ej); codes are `bases` indexes
        // (window indexes q = p - (i0 - 5) in 3 .. 12 of those positions; the wave loops as often as its longest run needs)
        q_ins = max(3, ej + 1 - ((int)i0 - 5));
        q_end = min(12, ej + elen - ((int)i0 - 5));
      }
    }
#pragma unroll 1
    for (; __ballot(q_ins <= q_end) != 0ull; q_ins++) {
      if (q_ins <= q_end) {
        const int p = (int)i0 - 5 + q_ins, ej = (int)(ew_in & 0xFFFFu);
        const uint32_t prof = __umulhi(aux_draw(B, slot, tj_in, (uint32_t)(p - ej), m), 3u);
        const uint32_t nat = (P.inv_remap_packed >> (2u * prof)) & 3u;
        cw = (cw & ~(3u << (2 * q_ins))) | (nat << (2 * q_ins));
      }
    }
  }
  // ---- two Philox calls: the heads words of the item's eight positions (call = i/4, word = i%4) ----
  uint32_t x[8];
  const uint32_t c3b = dev_ctx(KIND_BASE, m, B.batch_id);
  if (dg & 8u) {  // ablation: no Philox
#pragma unroll
    for (int z = 0; z < 8; z++) x[z] = (slot * 2654435761u) ^ (c * 40503u + z * 0x9E3779B9u);
  } else {
    philox_base(slot + B.slot_offset, 2u * c, 0, c3b, B.k0, B.k1, x);
    philox_base(slot + B.slot_offset, 2u * c + 1u, 0, c3b, B.k0, B.k1, x + 4);
  }

  // ---- table offsets of the eight positions: bin * block bytes (+ the short-context region of a read's first two
  // bases).  Reads of the profile's own length take them from the item's look-up row; reads that a sequencing indel
  // changed (their bins are i * binCount / n') and items past the row compute them. ----
  const uint32_t lgW = P.lgW, blk_bytes = P.fast_stride * 4u;
  uint32_t so[8], qo0, qo1;
  const bool lut_ok = np == (uint32_t)P.L && c < TI;
  if (__ballot(active && !lut_ok) == 0ull) {
    const uint4* row = (const uint4*)(lut + (lut_ok ? c : 0u) * LUT_ROW);
    const uint4 r0 = row[0], r1 = row[1], r2 = row[2];
    so[0] = r0.x; so[1] = r0.y; so[2] = r0.z; so[3] = r0.w; so[4] = r1.x; so[5] = r1.y; so[6] = r1.z; so[7] = r1.w;
    qo0 = r2.x; qo1 = r2.y;
  } else {
    const uint32_t ib0 = __umul24(i0, bins);
#pragma unroll
    for (int h = 0; h < 8; h++) {
      // idle lanes may compute a bin past the table: LDS reads beyond the allocation return 0, results unused
      const uint32_t bin = __umulhi(ib0 + (uint32_t)h * bins, inv);
      so[h] = __umul24(active ? bin : 0u, blk_bytes) + img_at;
    }
    qo0 = so[0]; qo1 = so[1];
    if (c == 0u) { so[0] += 128u * 4u; so[1] += 64u * 4u; }
  }
  uint32_t qw[2], sw[2], acc;
  sample8(code4, lgW, cw, x, so, qo0, qo1, sw, qw, acc);
  const bool slow = active && (bad != 0u || nev >= 2u);  // queued for the generic item code by the caller
  // flagged bases are redone exactly by the group's fix-up pass (the flags of a queued item do not matter)
  fix = (active && !slow && !(dg & 16u)) ? (acc >> 17) & 0xFFu & ((1u << min(8u, np - i0)) - 1u) : 0u;
  cw_out = cw;
  const bool go = active && !slow;
  // A whole item is two 8-byte stores.  They are buffer stores through the read group's descriptor (base = the group's
  // first record, offsets 32-bit), issued by EVERY lane on every path: a lane with nothing to store gives an offset past
  // the descriptor's range and the hardware drops it.  The step loop so holds a fixed number of vector-memory
  // operations, and the wait for the next step's prefetched window is a counted s_waitcnt (vmcnt(3)) instead of one
  // that also drains these stores (with the stores under a branch the compiler had to assume the path without them:
  // vmcnt(1), a full store round trip exposed in every step).  The read's last, partial item (np % 8 bases) is parked
  // in the read's own LDS row: the per-read pass after the step loop merges it with the record separators, so the
  // byte-granular stores run once per read group rather than in every step.
  // A last item of exactly seven bases (np % 8 == 7: every read without an indel of the 151-base profile) is a whole
  // item too -- its eighth byte is the line break that follows the bases / the qualities -- so it leaves with the steps,
  // next to its neighbours, instead of coming back to half-written lines from the per-read pass.
  {
    const bool last7 = i0 + 7u == np;
    const bool whole = i0 + 8u <= np || last7;
    const bool st = go && whole && !(dg & 1u);
    const uint32_t so_ = st ? m0.z + hdr + i0 : 0xFFFFFFFFu;
    const uint32_t qo_ = st ? so_ + np + 3u : 0xFFFFFFFFu;
    const uint32_t s1 = last7 ? (sw[1] & 0x00FFFFFFu) | 0x0A000000u : sw[1];
    const uint32_t q1 = last7 ? (qw[1] & 0x00FFFFFFu) | 0x0A000000u : qw[1];
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{sw[0], s1}, out_rsrc, so_, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{qw[0], q1}, out_rsrc, qo_, 0, 0);
    // ... and the "+\n" between the two line breaks, by the same lane (a third store on every path, dropped elsewhere)
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)0x0A2Bu, out_rsrc, st && last7 ? so_ + 8u : 0xFFFFFFFFu, 0, 0);
    if (active && i0 + 8u > np && !(last7 && !slow)) {
      // parked over the fields no lane needs once the last item has been sampled (fragment offset, 2^32/n',
      // event): base characters are never 0xFF
      tail_row[0].x = slow ? 0xFFFFFFFFu : sw[0];
      tail_row[0].y = sw[1];
      tail_row[1].z = qw[0];
      tail_row[1].w = qw[1];
    }
  }
  return slow;
}

// n (< 16) bytes of the 128-bit value (lo, hi) to q, as TWO overlapping pieces of the largest power of two <= n: its
// first bytes and its last (one piece when n is that power).  (Lane = record here: every store instruction touches 64 different lines, so the number of
// pieces is what these stores cost; 8/4/2/1-byte pieces made three or four of a 15-byte name.)
__device__ __forceinline__ void store_var(uint8_t* q, uint64_t lo, uint64_t hi, uint32_t n) {
  if (n >= 8u) {
    __builtin_memcpy(q, &lo, 8);
    if (n > 8u) {
      const uint32_t k = 8u * (n - 8u);   // bit offset of the last eight bytes (8 .. 56)
      const uint64_t v = (lo >> k) | (hi << (64u - k));
      __builtin_memcpy(q + n - 8u, &v, 8);
    }
  } else if (n >= 4u) {
    const uint32_t a = (uint32_t)lo, b = (uint32_t)(lo >> (8u * (n - 4u)));
    __builtin_memcpy(q, &a, 4);
    if (n > 4u) __builtin_memcpy(q + n - 4u, &b, 4);
  } else if (n >= 2u) {
    const uint16_t a = (uint16_t)lo, b = (uint16_t)(lo >> (8u * (n - 2u)));
    __builtin_memcpy(q, &a, 2);
    if (n > 2u) __builtin_memcpy(q + n - 2u, &b, 2);
  } else if (n == 1u) {
    *q = (uint8_t)lo;
  }
}

#define SLOW_CAP 128  // per-wave queue of items deferred to the generic code
#define OVF_CAP 192   // per-wave list of single items: the plain reads' last items (<= 63) + appended items (<= 126)
#define FIX_CAP 128   // per-wave list of flagged bases waiting for the group's fix-up pass
#define CLEAN_CAP 256 // per-wave list of the one-indel reads' clean items (~10 such reads of 19 items in a group at XTen rates)

template <bool PAIRED, bool DIAG>
__global__ __launch_bounds__(EMIT_THREADS) void emit_fast_kernel(DevProfile P, DevBatch B, uint32_t TI, uint32_t inv_TI, uint32_t clean_cap) {
  extern __shared__ uint4 smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t m = blockIdx.y;
  const uint32_t tm = PAIRED ? m : 0u;
  // launched before the host knew the text size (sg_api.cpp run_pass): a text that does not fit the buffers is left to
  // a second launch into larger ones
  if (B.totals[m] + 64u > B.out_cap[m]) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr((unsigned long long*)&B.totals[3], 4ull);
    return;
  }
  const uint32_t bins = (uint32_t)P.bins;
  const uint32_t dg = DIAG ? B.diag : 0u;  // SG_FDIAG timing ablations: compiled out of the production kernel
  // ---- LDS: [table image: bins x fast_stride words][look-up rows: TI x 12][256 words: four 2-bit codes -> bytes code * 4]
  //           [per-wave read rows][slow-item queues][fix-up lists][read order][appended items] ----
  const uint32_t img_words = bins * P.fast_stride;
  uint32_t* img = (uint32_t*)smem;
  const uint32_t img_at = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)img;  // its LDS byte address
  uint32_t* lut = img + ((img_words + 3u) & ~3u);
  uint32_t* code4 = lut + TI * LUT_ROW;
  uint4* lds_meta_all = (uint4*)(code4 + 256);
  uint32_t* slow_all = (uint32_t*)(lds_meta_all + EMIT_WAVES * 64 * (META_ROW / 16));
  uint2* fix_all = (uint2*)(slow_all + EMIT_WAVES * SLOW_CAP);
  uint8_t* perm_all = (uint8_t*)(fix_all + EMIT_WAVES * FIX_CAP);
  uint8_t* permp_all = perm_all + EMIT_WAVES * 64;                  // the plain reads of a group, in lane order
  uint16_t* ovf_all = (uint16_t*)(permp_all + EMIT_WAVES * 64);     // single items of the general stream (read | item << 8)
  uint16_t* clean_all = ovf_all + EMIT_WAVES * OVF_CAP;             // clean items of one-indel reads (read | after << 6 | item << 7)
  {
    const uint4* src = (const uint4*)(P.fast_lds + (size_t)tm * P.fast_mate_words);  // 16-byte aligned on the host
    for (uint32_t i = tid; i < (img_words + 3u) / 4u; i += EMIT_THREADS) ((uint4*)img)[i] = src[i];
    // look-up rows for reads of the profile's own length: bin = i * binCount / L (Profile.cpp:1672)
    const uint32_t blk_bytes = P.fast_stride * 4u;
    for (uint32_t i = tid; i < TI * LUT_ROW; i += EMIT_THREADS) {
      const uint32_t c = i / LUT_ROW, f = i - c * LUT_ROW;
      const uint32_t h = f < 8u ? f : f - 8u;
      const uint32_t pos = 8u * c + h;
      const uint32_t bin = min(pos * bins / (uint32_t)P.L, bins - 1u);
      uint32_t v = bin * blk_bytes;
      if (f < 8u && c == 0u) v += f == 0u ? 128u * 4u : f == 1u ? 64u * 4u : 0u;
      lut[i] = f < 10u ? v + img_at : 0u;
    }
    for (uint32_t i = tid; i < 256u; i += EMIT_THREADS) {
      uint32_t v = 0;
      for (uint32_t z = 0; z < 4u; z++) v |= (((i >> (2u * z)) & 3u) * 4u) << (8u * z);
      code4[i] = v;
    }
  }
  __syncthreads();
  uint4* meta_rows = lds_meta_all + (size_t)wv * 64 * (META_ROW / 16);
  uint32_t* slow_list = slow_all + wv * SLOW_CAP;
  uint2* fix_list = fix_all + wv * FIX_CAP;
  uint8_t* perm = perm_all + wv * 64;
  uint8_t* permp = permp_all + wv * 64;
  uint16_t* ovf = ovf_all + wv * OVF_CAP;
  uint16_t* clean = clean_all + wv * clean_cap;
  // Items per plain read that run through the plain steps: all but the last one, whose store is partial and followed by the
  // record separators -- unless the last item holds seven bases (L % 8 == 7, the 151-base profile): with the line break as
  // its eighth byte it is a whole item like the others, plus the two bytes of the "+" line.  (No fewer instructions that
  // way, but the record's last bytes leave with their neighbours instead of fifteen steps later, when the line has left
  // L2: WRITE_SIZE 6.43 -> see profiles/r03_final.)
  const bool tail_whole = ((uint32_t)P.L & 7u) == 7u && ((uint32_t)P.L + 7u) / 8u == TI;
  const uint32_t TIp = tail_whole ? TI : TI - 1u;
  // 16-base items per plain read (the plain16 steps below).  A whole last item of seven bases (tail_whole) carries the line
  // break and the "+" line and only the 8-base steps write those: it stays out of the pairs (L = 303: 38 items, 18 pairs +
  // items 36 and 37 on their own; found by fuzz seeds 222 / 224, whose 303-base reads lost their line break to a sixteenth base)
  const uint32_t TI16 = (tail_whole ? TIp - 1u : TIp) >> 1;
  const uint32_t inv_TI16 = TI16 ? (1u << 20) / TI16 + 1u : 0u;   // ceil-reciprocal (i / TI16 exact while i * TI16 < 2^20)

  // Lane -> (read, item) map over the first TI = ceil(L / 8) items of a group's G = 63 reads: their items form one
  // stream, 64 per step: lane l of step s does stream item i = 64 s + l = item i % TI of the (i / TI)-th read in step
  // order; every lane busy whatever TI is (63 reads, so that TI steps hold the 63 TI map items plus up to TI appended
  // ones).  inv_TI = ceil-reciprocal of TI.
  const uint32_t G = 63u;
  const uint32_t ngroups = (B.n_slots + G - 1u) / G;
  GroupRuns runs(B.totals, 0u, m, ngroups, wv, EMIT_WAVES);
  for (; runs.more(); runs.next(lane)) {
    const uint32_t g = runs.g;
    const uint32_t t = g * G + lane;
    uint32_t items = 0;
    uint4 my0 = make_uint4(0, 0, 0, 0), my1 = make_uint4(0, 0, 0, 0);
    uint64_t ooff = 0;
    if (lane < G && t < B.n_slots) {
      const size_t idx = (size_t)m * B.n_slots + t;
      my0 = B.meta[idx * 3];
      my1 = B.meta[idx * 3 + 1];
      ooff = rec_offset(B, m, t);
      if (my1.x & 0x7FFFFFFFu) {
        items = ((my1.y & 0xFFFFu) + 7u) / 8u;  // separators: per-read pass below
        // The read's name, stored in front of where the step loop will put the bases: the batch's prefix, then the
        // read's own part from its row (indel_kernel made the text), two pieces of <= 16 bytes.
        if (B.prefix_len <= 16u && !(dg & 2u)) {
          uint8_t* rec = B.out[m] + ooff;
          const uint32_t nv = (my1.y >> 22) - B.prefix_len;
          if (__builtin_expect(nv > 16u, 0)) {
            write_name(rec, B.prefix, B.prefix_len, my0.z, my0.w, PAIRED ? 1u : 0u, m);
          } else {
            const uint64_t p_lo = ((uint64_t)B.prefix_w[1] << 32) | B.prefix_w[0], p_hi = ((uint64_t)B.prefix_w[3] << 32) | B.prefix_w[2];
            const uint4 tx = B.meta[idx * 3 + 2];
            const uint64_t b0 = ((uint64_t)tx.y << 32) | tx.x, b1 = ((uint64_t)tx.w << 32) | tx.z;
            const uint32_t plen = B.prefix_len, hb = plen + nv;   // the name's bytes: prefix, then the read's own text
            if (plen < 16u && hb >= 16u && hb <= 24u) {
              // the usual case ("@sim#20#" + "592246#30000/1\n"): the name as ONE string (both parts are zero beyond their
              // ends), its first sixteen bytes and its last eight -- two stores where prefix + two text pieces were three
              // (lane = record: every store instruction touches 64 lines, and on the 75-base profiles the names and tails
              // are a quarter of the kernel's time)
              const uint32_t sh = 8u * plen;
              uint64_t w0, w1, w2;
              if (sh < 64u) {
                w0 = p_lo | (b0 << sh);
                w1 = p_hi | (b1 << sh) | (sh ? b0 >> (64u - sh) : 0ull);
                w2 = sh ? b1 >> (64u - sh) : 0ull;
              } else {
                const uint32_t t2 = sh - 64u;
                w0 = p_lo;
                w1 = p_hi | (b0 << t2);
                w2 = (b1 << t2) | (t2 ? b0 >> (64u - t2) : 0ull);
              }
              struct { uint64_t a, b; } first = {w0, w1};
              __builtin_memcpy(rec, &first, 16);
              const uint32_t n2 = hb - 16u;
              if (n2) {
                const uint32_t k = 8u * n2;
                const uint64_t v = k == 64u ? w2 : (w1 >> k) | (w2 << (64u - k));
                __builtin_memcpy(rec + hb - 8u, &v, 8);
              }
            } else {
              if (plen == 16u) { __builtin_memcpy(rec, &p_lo, 8); __builtin_memcpy(rec + 8, &p_hi, 8); }
              else store_var(rec, p_lo, p_hi, plen);
              uint8_t* q = rec + plen;
              if (nv == 16u) __builtin_memcpy(q, &tx, 16);
              else store_var(q, b0, b1, nv);
            }
          }
        }
      }
    }
    // The group's text is one contiguous range starting at its first record: a buffer descriptor on that address, the
    // records at 32-bit offsets from it (row word 2).  Offsets past 2^31 are the "no store" value of idle lanes.
    const uint64_t gbase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ooff >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ooff);
    uint8_t* const gout = B.out[m] + gbase;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(gout, 0, 0x80000000u, 0x00020000);
    my0.z = (uint32_t)(ooff - gbase);
    {
      // what the plain steps need of the read, in the two row words nothing else uses: the base index of output position -5
      // in the 2-bit copy the read walks (bit 31: that copy is the reverse complement), the offset of its first base
      const uint32_t flen_l = my1.x & 0x3FFFFFFFu;
      const bool rev_l = PAIRED ? (m == 1u) : ((my1.x >> 31) != 0u);
      my0.y = ((rev_l ? (uint32_t)B.chains_total - my0.x - flen_l : my0.x) - 5u) | (rev_l ? 0x80000000u : 0u);
      my0.w = my0.z + (my1.y >> 22);
    }
    meta_rows[lane * 2] = my0;
    meta_rows[lane * 2 + 1] = my1;
    wave_lds_sync();
    uint32_t nslow = 0, nfix = 0;  // wave-uniform
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
    // The fix-up pass, lane = flagged base: the bases the heads could not decide (a possible substitution, a head equal
    // to a threshold's) are sampled again exactly -- full identity-first row and alias column from L2, the tails call
    // only when a head sits on a threshold -- and patched into the text already stored (or into the parked last item).
    // Run once per read group (and whenever the list fills): one pass serves ~40 flagged bases, where a loop inside the
    // step would run for one or two lanes in five steps out of six.
    auto flush_fix = [&]() {
      wave_lds_sync();
      // the items' own stores must have landed before single bytes of them are rewritten
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      const uint32_t lgW = P.lgW, u_mask = (1u << (16u - lgW)) - 1u;
      const uint4* gsub = P.fast_sub + (size_t)tm * bins * FAST_CTX;
      const uint32_t c3b = dev_ctx(KIND_BASE, m, B.batch_id);
      for (uint32_t b0 = 0; b0 < nfix; b0 += 64u) {
        // lane = flagged item: read | flags of its eight bases << 6 | item << 17, packed codes; its bases one after the
        // other (nine items in ten have one flagged base, so the loop body nearly always runs once per 64 items)
        const uint2 fe = fix_list[b0 + lane < nfix ? b0 + lane : 0u];
        uint32_t todo = b0 + lane < nfix ? (fe.x >> 6) & 0xFFu : 0u;
        const uint32_t r = fe.x & 63u, c = fe.x >> 17, cw = fe.y;
        const uint32_t slot = g * G + r + B.slot_offset;
        const uint4 r0 = meta_rows[r * 2], r1 = meta_rows[r * 2 + 1];
        const uint32_t np = r1.y & 0xFFFFu, hl = r1.y >> 22;
        // stored like a whole item (a last item of seven bases included), or the read's parked last item?
        const bool whole = 8u * c + 8u <= np + (uint32_t)((np & 7u) == 7u);
        // ceil(2^32 / n') again (indel_kernel's row word): a read's parked last item has overwritten it in the row
        const uint32_t inv = 0xFFFFFFFFu / max(np, 1u) + 1u;
        while (__ballot(todo != 0u) != 0ull) {
          const bool on = todo != 0u;
          const uint32_t h = on ? (uint32_t)__builtin_ctz(todo) : 0u;
          todo &= todo - 1u;
          const uint32_t i = 8u * c + h;
          const uint32_t bin = __umulhi(__umul24(i, bins), inv);
          uint32_t xw[4];
          philox_base(slot, 2u * c + (h >> 2), 0, c3b, B.k0, B.k1, xw);
          const uint32_t l = h & 3u;
          const uint32_t wh = l == 0u ? xw[0] : l == 1u ? xw[1] : l == 2u ? xw[2] : xw[3];
          const uint32_t region = c == 0u ? (h == 0u ? 128u : h == 1u ? 64u : 0u) : 0u;
          const uint32_t cdn = (cw >> (2u * h + 10u)) & 3u;
          const uint4 row = gsub[(size_t)bin * FAST_CTX + region + ((cw >> (2u * h + 6u)) & 63u)];
          // identity-first row on the 16-bit head: certain unless the head equals a threshold's
          const uint32_t sh = wh >> 16, h0 = row.x >> 16, h1 = row.y >> 16, h2 = row.z >> 16;
          const uint32_t jj = (uint32_t)(sh > h0) + (uint32_t)(sh > h1) + (uint32_t)(sh > h2);
          const bool amb_s = on && jj != (uint32_t)(sh >= h0) + (uint32_t)(sh >= h1) + (uint32_t)(sh >= h2);
          uint32_t kn = 0;
          uint2 e2 = make_uint2(0, 0);
          const uint32_t col = (wh & 0xFFFFu) >> (16u - lgW), uh = wh & u_mask;
          auto column = [&](uint32_t j_) {
            const uint32_t jm = max(j_, row.w & 3u);
            kn = (row.w >> (2u * jm + 2u)) & 3u;
            e2 = P.fast_alias[((((size_t)cdn * 4u + kn) * bins + bin) << lgW) + col];
          };
          if (!amb_s) column(jj);
          const bool need_tail = amb_s || (on && uh == (e2.x >> 16) && (e2.y & 0xFFu) != (e2.y >> 8));
          uint32_t wt = 0;
          if (__ballot(need_tail) != 0ull) {
            if (need_tail) wt = tail_word(slot, c, h, c3b, B.k0, B.k1);
          }
          if (__ballot(amb_s) != 0ull) {
            if (amb_s) {
              const uint32_t xs = (wh & 0xFFFF0000u) | (wt >> 16);
              column((uint32_t)(xs > row.x) + (uint32_t)(xs > row.y) + (uint32_t)(xs > row.z));
            }
          }
          // side of the column: the head decides unless it sits on the threshold's (then the 16-bit tail does)
          const uint32_t th = e2.x >> 16;
          const bool lo_side = uh != th ? uh < th : ((uh << 16) | (wt & 0xFFFFu)) < e2.x;
          const uint32_t sym = (lo_side ? (e2.y & 0xFFu) : (e2.y >> 8)) + (uint32_t)P.min_qual;
          const uint32_t ch = (0x47544341u >> (8u * kn)) & 0xFFu;  // "ACTG"[kn]: natural code -> character
          if (on) {
            if (whole) {
              uint8_t* rec = gout + r0.z + hl + i;
              rec[0] = (uint8_t)ch;
              rec[np + 3u] = (uint8_t)sym;
            } else if (r0.x != 0xFFFFFFFFu) {
              // the read's parked last item: characters in words 0, 1, qualities in words 6, 7 of its row
              uint8_t* row8 = (uint8_t*)(meta_rows + r * 2);
              row8[h] = (uint8_t)ch;
              row8[24u + h] = (uint8_t)sym;
            }
          }
        }
      }
      nfix = 0;
      wave_lds_sync();
    };
    // The group's items go through two loops.
    //  * PLAIN steps: items 0 .. TI-2 of the plain reads -- no sequencing indel, the profile's own length, no non-ACGT
    //    block (84 % of the reads at XTen rates).  Every such item is eight bases stored whole, its bin offsets come from
    //    the look-up row, nothing about it is conditional: the step is the 8-base block, two Philox calls and ~50
    //    instructions around them (the general step: ~250).
    //  * GENERAL steps: everything else as one item stream -- all items of the other reads (reads with one indel before
    //    those with several, so that the two-window code and the event-list walk run in few steps), then single items: the
    //    plain reads' last items (partial or followed by the separators: the per-read pass and the parked-item logic
    //    belong to the general code) and the one or two items an insertion appended to a read.
    //  * CLEAN steps: a read with ONE sequencing indel is, item by item, a plain read nearly everywhere -- the items wholly
    //    before the event read the template unshifted, those wholly after it read it shifted by the event's length; what
    //    differs from a plain item is the bin arithmetic (n' != L: no look-up row) and that shift.  Only the one to
    //    three items the event reaches into, and the last, partial one, need the general code.  (Measured before this
    //    split, SQ_INSTS_VALU with either loop compiled out: the general steps took 34 % of the kernel's instructions for
    //    16 % of the reads -- every item of a one-indel read paid the two-window code and its insertion loop.)
    const uint32_t nev_l = (my1.y >> 16) & 0x3Fu;
    const uint32_t np_l = my1.y & 0xFFFFu;
    const bool act_l = items > 0u;
    const bool plain_l = act_l && nev_l == 0u && np_l == (uint32_t)P.L && items == TI && !((my1.x >> 30) & 1u) && TIp != 0u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // one-indel reads: item ranges.  fc = first output position that is not the unshifted template's, fs = first one that
    // reads the shifted template (an insertion's own bases lie between the two); an item's codes span positions
    // 8c - 5 .. 8c + 7
    uint32_t kb = 0, ca0 = 0, n_whole = 0, ka = 0, n_gen = 0, vscan = 0;
    bool split_l = false;
    // (only where a group has enough one-indel reads for about two clean steps: with fewer the split cannot pay, see below,
    // and the other profiles' groups -- a third to a twentieth of XTen's one-indel reads -- skip its bookkeeping)
    const bool cand0_l = act_l && nev_l == 1u && !((my1.x >> 30) & 1u);
    if (clean_cap != 0u && (uint32_t)__popcll(__ballot(cand0_l)) * (TI - 2u) >= 96u) {
      const uint32_t ew = my1.w;
      const uint32_t ej = ew & 0xFFFFu, elen = (ew >> 16) & 0x7FFFu;
      const bool del = (ew >> 31) != 0u;
      const uint32_t fc = del ? ej : ej + 1u, fs = del ? ej : ej + elen + 1u;
      n_whole = np_l >> 3;
      kb = min(fc >> 3, n_whole);                 // items 0 .. kb-1: wholly before
      ca0 = (fs + 12u) >> 3;                      // items ca0 .. n_whole-1: wholly after (8c - 5 >= fs)
      ka = n_whole > ca0 ? n_whole - ca0 : 0u;
      n_gen = ka ? (ca0 - kb) + (items - n_whole) : items - kb;   // what is left for the general steps
      const bool cand_l = cand0_l && items <= 255u && kb + ka != 0u;
      // as many of them, in lane order, as the two lists hold (behind the other single items, counted below)
      uint32_t v = cand_l ? ((kb + ka) << 16) | n_gen : 0u;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(v, d, 64);
        if ((int)lane >= d) v += up;
      }
      vscan = v;   // (clean items << 16 | general items) of the candidates up to and including this lane
      const uint32_t extra_l = items > TI ? items - TI : 0u;
      const uint32_t others = (tail_whole ? 0u : (uint32_t)__popcll(__ballot(plain_l))) +
                              (uint32_t)__popcll(__ballot(extra_l == 1u || extra_l == 2u)) + (uint32_t)__popcll(__ballot(extra_l == 2u));
      split_l = cand_l && (v >> 16) <= clean_cap && others + (v & 0xFFFFu) <= OVF_CAP;
      // (a prefix in lane order: the sums only grow)
      // Does the split pay in this group?  Steps are whole: a few clean items make a clean step of their own, and what
      // they leave behind may still need as many general steps as before (the 125-base profile: one read in ten has one
      // indel, and every plain read sends its partial last item through the general steps anyway -- with the split
      // unconditional its launch ran 3.6 % MORE instructions).  Step counts either way from the sums at hand, priced at
      // the loops' measured instructions per step (general 390-590 by profile: 480, clean 260, composing the lists ~250).
      const unsigned long long ms = __ballot(split_l);
      if (ms) {
        const uint32_t tot_s = (uint32_t)__shfl((int)v, 63 - __builtin_clzll(ms), 64);
        const uint32_t n_cl = tot_s >> 16, n_gn = tot_s & 0xFFFFu;
        const uint32_t whole = (uint32_t)__popcll(__ballot(act_l && !plain_l && !split_l));   // reads that stay whole in the general steps
        const uint32_t s0 = (whole + (uint32_t)__popcll(ms)) * TI + others, s1 = whole * TI + others + n_gn;
        const uint32_t rem = n_cl & 63u, mv = (rem != 0u && rem <= ((0u - s1) & 63u)) ? rem : 0u;   // (the partial clean step that moves, below)
        const uint32_t before = 48u * ((s0 + 63u) >> 6);
        const uint32_t after = 26u * ((n_cl - mv + 63u) >> 6) + 48u * ((s1 + mv + 63u) >> 6) + 25u;
        if (after >= before) split_l = false;
      }
      kb = split_l ? kb : 0u;
      ka = split_l ? ka : 0u;
      n_gen = split_l ? n_gen : 0u;
    }
    const unsigned long long m_split = __ballot(split_l);
    const unsigned long long m_plain = __ballot(plain_l);
    const unsigned long long m_multi = __ballot(act_l && nev_l >= 2u);
    const unsigned long long m_one = __ballot(act_l && nev_l == 1u && !split_l);
    const unsigned long long m_rest = __ballot(act_l && !plain_l && nev_l == 0u);
    const uint32_t n_plain = (uint32_t)__popcll(m_plain);
    const uint32_t n_psingle = tail_whole ? 0u : n_plain;   // plain reads' last items left to the general steps
    const uint32_t n_rest = (uint32_t)__popcll(m_rest), n_one = (uint32_t)__popcll(m_one);
    const uint32_t n_fast = n_rest + n_one + (uint32_t)__popcll(m_multi);  // reads whose items all walk the general steps
    // TI = ceil(L / 8) exactly; a read that an insertion grew by one or two items appends them to the end of
    // the stream (they fill lanes of the last step that would idle anyway); only reads longer than that take
    // steps of their own below.
    const uint32_t extra = (items > TI && !split_l) ? items - TI : 0u;
    const bool small = extra == 1u || extra == 2u;
    const unsigned long long b1 = __ballot(small), b2 = __ballot(extra == 2u);
    const uint32_t n_single = n_psingle + (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2);
    // the split reads' totals (they are the first candidates in lane order: the scan above holds their sums and offsets)
    const uint32_t tot = m_split ? (uint32_t)__shfl((int)vscan, 63 - __builtin_clzll(m_split), 64) : 0u;
    uint32_t n_clean = tot >> 16;
    const uint32_t n_gen_all = tot & 0xFFFFu;
    // The clean list's last, partial step: if the general stream's last step has that many idle lanes, the items go there
    // as single items (the general code does them as well) and the step is saved.  They go FIRST among the single items:
    // a read's last item is parked in its LDS row over fields its other items need (fast_item), so nothing of a read may
    // follow its last item in the stream.
    uint32_t n_moved = 0;
    {
      const uint32_t rem = n_clean & 63u, idle = (0u - (n_fast * TI + n_single + n_gen_all)) & 63u;
      if (rem != 0u && rem <= idle && n_single + n_gen_all + rem <= OVF_CAP) n_moved = rem;
    }
    if (act_l && !split_l) {
      if (plain_l) {
        const uint32_t pos = (uint32_t)__popcll(m_plain & lt);
        permp[pos] = (uint8_t)lane;
        if (!tail_whole) ovf[n_moved + pos] = (uint16_t)(lane | (TIp << 8));   // its last item: a single item of the general stream
      } else {
        uint32_t pos;
        if ((m_rest >> lane) & 1ull) pos = (uint32_t)__popcll(m_rest & lt);
        else if ((m_one >> lane) & 1ull) pos = n_rest + (uint32_t)__popcll(m_one & lt);
        else pos = n_rest + n_one + (uint32_t)__popcll(m_multi & lt);
        perm[pos] = (uint8_t)lane;
      }
    }
    if (n_fast == 0u && lane == 0u) perm[0] = 0;   // (what idle lanes of a step look at)
    if (small) {
      const uint32_t o = n_moved + n_psingle + (uint32_t)__popcll(b1 & lt) + (uint32_t)__popcll(b2 & lt);
      ovf[o] = (uint16_t)(lane | (TI << 8));
      if (extra == 2u) ovf[o + 1u] = (uint16_t)(lane | ((TI + 1u) << 8));
    }
    if (split_l) {
      // the split reads' items into the two lists, each lane its own ranges (one loop over both clean ranges: the wave runs
      // as many iterations as its longest lane needs, and every split read has about TI - 2 clean items however they fall
      // before and after its indel)
      const uint32_t oc = (vscan >> 16) - (kb + ka);
      uint32_t og = n_moved + n_single + (vscan & 0xFFFFu) - n_gen;
      const uint32_t kt = kb + ka, behind = ((ca0 - kb) << 7) | 64u;
      for (uint32_t k = 0; k < kt; k++) clean[oc + k] = (uint16_t)((lane | (k << 7)) + (k >= kb ? behind : 0u));
      const uint32_t mid_end = ka ? ca0 : items;
      for (uint32_t c = kb; c < mid_end; c++) ovf[og++] = (uint16_t)(lane | (c << 8));
      if (ka) for (uint32_t c = n_whole; c < items; c++) ovf[og++] = (uint16_t)(lane | (c << 8));
    }
    const uint32_t n_ovf = n_moved + n_single + n_gen_all;
    unsigned long long more = (dg & 256u) ? 0ull : __ballot(extra > 2u);
    wave_lds_sync();
    if (n_moved) {
      n_clean -= n_moved;
      if (lane < n_moved) {
        const uint32_t e = clean[n_clean + lane];
        ovf[lane] = (uint16_t)((e & 63u) | ((e >> 7) << 8));
      }
      wave_lds_sync();
    }
    const uint32_t n_items = n_fast * TI, n_stream = n_items + n_ovf;
    const uint32_t nmain = (dg & 256u) ? 0u : (n_stream + 63u) / 64u;   // (ablations: no general / no plain steps)
    // The plain reads' items walk two loops (round 4).  PLAIN16 steps: lane = SIXTEEN bases, items 2c and 2c + 1 of one read
    // -- one map, one row fetch, one 8-byte window of the 2-bit copy (21 codes = 42 bits), two 16-byte stores for what were
    // two of each; only WHOLE steps run that way.  What is left -- the 16-base items of the last, partial step, as their two
    // halves, and the reads' odd last 8-base item (cnt8 = TIp mod 2) -- goes through the 8-base plain steps as one stream,
    // so that neither loop ends in a step with idle lanes of its own.
    // Reads of fewer than 18 whole items keep the 8-base steps alone: with four to seven pairs per read the whole 16-base steps
    // cover 70 % of a group's items and the rest pays two loops' prologues -- the 125-, 75- and 74-base profiles ran 6-7 %
    // SLOWER in pairs (2.75 / 3.36 / 3.41 ms against 2.59 / 3.14 / 3.18, profiles/r04_mid_all_profiles vs r03_final).
    const bool use16 = !(dg & 512u) && TIp >= 18u;                 // (SG_FDIAG 512: no 16-base steps)
    const uint32_t n_p16 = use16 ? n_plain * TI16 : 0u;
    const uint32_t np16steps = (dg & 128u) ? 0u : n_p16 >> 6, rem16 = n_p16 - (np16steps << 6);
    const uint32_t cnt8 = use16 ? TIp - 2u * TI16 : TIp, c8_base = use16 ? 2u * TI16 : 0u;
    const uint32_t n_pitems = 2u * rem16 + n_plain * cnt8, npsteps = (dg & 128u) ? 0u : (n_pitems + 63u) / 64u;
    uint32_t cb = TI;
    // One item per lane.  A step's lane -> (read, item) map, its read rows and its haplotype window are fetched ONE STEP
    // AHEAD (the chain LDS -> LDS -> L2 is ~1000 cycles; issued before the previous step's sampling it is covered by it).
    struct Stage { uint32_t r, c; bool active; uint4 m0, m1; uint32_t src; uint2 w; };
    auto fetch_item = [&](uint32_t r, uint32_t c, bool ok, uint32_t c_idle) -> Stage {
      Stage st;
      st.r = r;
      st.m0 = meta_rows[r * 2];
      st.m1 = meta_rows[r * 2 + 1];
      const uint32_t nitems = ((st.m1.y & 0xFFFFu) + 7u) / 8u;
      st.active = ok && (st.m1.x & 0x7FFFFFFFu) != 0u && c < nitems;
      st.c = st.active ? c : c_idle;
      st.src = fast_src<PAIRED>(B, m, st.m0, st.m1, st.c, st.active);
      __builtin_memcpy(&st.w, (((PAIRED ? m == 1u : (st.m1.x >> 31) != 0u)) ? B.chains2_rc : B.chains2_fwd) + (st.src >> 2), 8);
      return st;
    };
    auto fetch_step = [&](uint32_t step) -> Stage {  // the item stream, 64 items per step
      const uint32_t i = step * 64u + lane;
      const bool ok = i < n_stream, in_map = i < n_items;
      const uint32_t ri = __umul24(i, inv_TI) >> 20;  // i / TI (exact: i * TI < 2^20)
      const uint32_t e = ovf[(ok && !in_map) ? i - n_items : 0u];
      const uint32_t r = in_map ? perm[ri] : (ok ? (e & 0xFFu) : perm[0]);
      const uint32_t c = in_map ? i - __umul24(ri, TI) : (ok ? e >> 8 : 1u);
      return fetch_item(r, c, ok, c);
    };
    // windows with a non-ACGT base are queued for the generic code
    auto run_item = [&](const Stage& st) {
      const uint32_t r = st.r, c = st.c;
      const uint4 m0 = st.m0, m1 = st.m1;
      const bool active = st.active;
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
      uint32_t fix, cw;
      const bool slow = fast_item<PAIRED, DIAG>(P, B, img_at, lut, code4, TI, m, m0, m1, g * G + r, c, active, meta_rows + r * 2, d0, n_in, ew_in,
                                          tj_in, st.src, st.w, out_rsrc, fix, cw);
      const unsigned long long sm = __ballot(slow);
      if (sm) {
        if (slow) slow_list[nslow + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull))] = r | (c << 8);
        nslow += (uint32_t)__popcll(sm);
      }
      // flagged items -> the group's fix-up list: read | flags of the eight bases << 6 | item << 17 (items < 2^15), packed
      // codes.  One entry per item whatever the number of its flagged bases: no loop in the step.
      const unsigned long long fm = __ballot(fix != 0u);
      if (fm) {
        if (nfix + 64u > FIX_CAP) flush_fix();
        if (fix != 0u) fix_list[nfix + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = make_uint2(r | (fix << 6) | (c << 17), cw);
        nfix += (uint32_t)__popcll(fm);
      }
    };
    // The odd items leave in the MIDDLE of their reads' pairs (round 4): a read's last bytes written nine steps behind its
    // pairs met the line gone from L2 (WRITE_SIZE 5.80 -> 6.99 M KiB per launch, profiles/r04_start); with the singles' step
    // between the two halves of the 16-base steps no instalment of a read is more than ~four steps from the other.
    const uint32_t h16 = np16steps >> 1, s8_first = (use16 && cnt8 && npsteps) ? 1u : 0u;   // (8-base steps alone: one loop, phase 1)
#pragma unroll 1
    for (uint32_t ph = 0; ph < 2u; ph++) {
    const uint32_t a16 = ph ? h16 : 0u, b16 = ph ? np16steps : h16;
    const uint32_t a8 = ph ? s8_first : 0u, b8 = ph ? npsteps : s8_first;
    // ---- plain16 steps: whole steps of 16-base items ----
    if (a16 < b16) {
      struct P16 { uint32_t r, c, src, out; uint2 w; };
      uint32_t TI16_v = TI16, inv_TI16_v = inv_TI16, qual_at = (uint32_t)P.L + 3u;
      asm volatile("" : "+v"(TI16_v), "+v"(inv_TI16_v), "+v"(qual_at));   // (as scalars they were spilled, see the 8-base loop)
      auto fetch16 = [&](uint32_t step) -> P16 {
        P16 st;
        const uint32_t i = step * 64u + lane;                  // < n_p16: whole steps only, every lane has an item
        const uint32_t ri = __umul24(i, inv_TI16_v) >> 20;    // i / TI16
        st.c = i - __umul24(ri, TI16_v);
        st.r = permp[ri];
        const uint32_t* row = (const uint32_t*)(meta_rows + st.r * 2);
        const uint32_t A = row[1];
        st.src = (A & 0x7FFFFFFFu) + 16u * st.c;
        st.out = row[3] + 16u * st.c;
        const uint8_t* copy2 = (PAIRED ? m == 1u : (A >> 31) != 0u) ? B.chains2_rc : B.chains2_fwd;
        __builtin_memcpy(&st.w, copy2 + (st.src >> 2), 8);
        return st;
      };
      const uint32_t c3b = dev_ctx(KIND_BASE, m, B.batch_id);
      const uint32_t lgW = P.lgW;
      auto run16 = [&](const P16& st) {
        const uint32_t c8 = 2u * st.c;   // the lane's two 8-base items: c8, c8 + 1
        // 21 source codes (positions 16c - 5 .. 16c + 15): 42 bits from bit 2 (src mod 4) of the eight bytes on
        const uint64_t w64 = ((((uint64_t)st.w.y << 32) | st.w.x) >> (2u * (st.src & 3u)));
        uint32_t cwa = (uint32_t)w64, cwb = (uint32_t)(w64 >> 16);
        const uint32_t slot = g * G + st.r + B.slot_offset;
        if (dg & 4u) { cwa = (g * G + st.r) * 2654435761u + c8; cwb = cwa * 40503u + 1u; }  // ablation: no haplotype fetch
        uint32_t sw[4], qw[4], fixv[2];
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          uint32_t x[8];
          const uint32_t c = c8 + (uint32_t)hf, cw = hf ? cwb : cwa;
          if (dg & 8u) {  // ablation: no Philox
#pragma unroll
            for (int z = 0; z < 8; z++) x[z] = (slot * 2654435761u) ^ (c * 40503u + z * 0x9E3779B9u);
          } else {
            philox_base(slot, 2u * c, 0, c3b, B.k0, B.k1, x);
            philox_base(slot, 2u * c + 1u, 0, c3b, B.k0, B.k1, x + 4);
          }
          const uint4* lrow = (const uint4*)(lut + c * LUT_ROW);
          const uint4 r0 = lrow[0], r1 = lrow[1], r2 = lrow[2];
          const uint32_t so[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
          uint32_t s2[2], q2[2], acc;
          sample8(code4, lgW, cw, x, so, r2.x, r2.y, s2, q2, acc);
          sw[2 * hf] = s2[0]; sw[2 * hf + 1] = s2[1];
          qw[2 * hf] = q2[0]; qw[2 * hf + 1] = q2[1];
          fixv[hf] = (dg & 16u) ? 0u : (acc >> 17) & 0xFFu;
        }
        const uint32_t so_ = (dg & 1u) ? 0xFFFFFFFFu : st.out;
        const uint32_t qo_ = (dg & 1u) ? 0xFFFFFFFFu : st.out + qual_at;
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{sw[0], sw[1], sw[2], sw[3]}, out_rsrc, so_, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{qw[0], qw[1], qw[2], qw[3]}, out_rsrc, qo_, 0, 0);
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          const uint32_t fix = fixv[hf];
          const unsigned long long fm = __ballot(fix != 0u);
          if (fm) {
            if (nfix + 64u > FIX_CAP) flush_fix();
            if (fix != 0u)
              fix_list[nfix + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = make_uint2(st.r | (fix << 6) | ((c8 + (uint32_t)hf) << 17), hf ? cwb : cwa);
            nfix += (uint32_t)__popcll(fm);
          }
        }
      };
      P16 cur = fetch16(a16);
      // (two dropped stores: every iteration then has the same vector-memory operations behind its prefetch, see below)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, out_rsrc, 0xFFFFFFFFu, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, out_rsrc, 0xFFFFFFF0u, 0, 0);
      for (uint32_t step = a16; step < b16; step++) {
        const P16 nxt = fetch16(min(step + 1u, b16 - 1u));
        run16(cur);
        cur = nxt;
      }
    }
    // ---- plain steps (8-base items): the reads' own odd last items, then the halves of the 16-base items the whole steps
    // above left over ----
    if (a8 < b8) {
      struct PStage { uint32_t r, c, src, out; bool ok; uint2 w; };
      // (wave-uniform values of every step held in vector registers: as scalars they were spilled and came back
      // through a v_readlane_b32 in every step)
      uint32_t qual_at = (uint32_t)P.L + 3u, n_single = n_plain * cnt8, i16_base = np16steps << 6;
      asm volatile("" : "+v"(qual_at), "+v"(n_single), "+v"(i16_base));
      const uint32_t inv_cnt8 = cnt8 ? (1u << 20) / cnt8 + 1u : 0u;
      auto fetch_plain = [&](uint32_t step) -> PStage {
        PStage st;
        const uint32_t i_raw = step * 64u + lane;
        st.ok = i_raw < n_pitems;
        const uint32_t i = min(i_raw, n_pitems - 1u);     // idle lanes redo the stream's last item, their stores are dropped
        const bool half = i >= n_single;                  // a read's own 8-base item / a half of a left-over 16-base item
        const uint32_t j = i - n_single;
        const uint32_t q = half ? i16_base + (j >> 1) : i;
        const uint32_t d = half ? TI16 : cnt8;
        const uint32_t ri = __umul24(q, half ? inv_TI16 : inv_cnt8) >> 20;  // q / d
        const uint32_t rem = q - __umul24(ri, d);
        st.c = half ? 2u * rem + (j & 1u) : c8_base + rem;
        st.r = permp[ri];
        const uint32_t* row = (const uint32_t*)(meta_rows + st.r * 2);
        const uint32_t A = row[1];
        st.src = (A & 0x7FFFFFFFu) + 8u * st.c;
        st.out = row[3] + 8u * st.c;
        const uint8_t* copy2 = (PAIRED ? m == 1u : (A >> 31) != 0u) ? B.chains2_rc : B.chains2_fwd;
        __builtin_memcpy(&st.w, copy2 + (st.src >> 2), 8);
        return st;
      };
      const uint32_t c3b = dev_ctx(KIND_BASE, m, B.batch_id);
      const uint32_t lgW = P.lgW;
      auto run_plain = [&](const PStage& st) {
        const uint32_t c = st.c;
        uint32_t cw;
        if (dg & 4u) cw = (g * G + st.r) * 2654435761u + c;  // ablation: no haplotype fetch
        else cw = (uint32_t)((((uint64_t)st.w.y << 32) | st.w.x) >> (2u * (st.src & 3u)));
        uint32_t x[8];
        const uint32_t slot = g * G + st.r + B.slot_offset;
        if (dg & 8u) {  // ablation: no Philox
#pragma unroll
          for (int z = 0; z < 8; z++) x[z] = (slot * 2654435761u) ^ (c * 40503u + z * 0x9E3779B9u);
        } else {
          philox_base(slot, 2u * c, 0, c3b, B.k0, B.k1, x);
          philox_base(slot, 2u * c + 1u, 0, c3b, B.k0, B.k1, x + 4);
        }
        const uint4* lrow = (const uint4*)(lut + c * LUT_ROW);
        const uint4 r0 = lrow[0], r1 = lrow[1], r2 = lrow[2];
        const uint32_t so[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        uint32_t sw[2], qw[2], acc;
        sample8(code4, lgW, cw, x, so, r2.x, r2.y, sw, qw, acc);
        const bool st_ok = st.ok && !(dg & 1u);
        const uint32_t so_ = st_ok ? st.out : 0xFFFFFFFFu;
        const uint32_t qo_ = st_ok ? st.out + qual_at : 0xFFFFFFFFu;
        uint32_t fix_mask = 0xFFu;
        if (tail_whole) {  // (wave-uniform) the read's last item: its eighth byte is the line break, then "+\n"
          const bool last = c + 1u == TI;
          const uint32_t sel = last ? 0x04020100u : 0x03020100u;   // v_perm: byte 3 = the first operand's byte 0 / the second's own
          sw[1] = __builtin_amdgcn_perm(0x0Au, sw[1], sel);
          qw[1] = __builtin_amdgcn_perm(0x0Au, qw[1], sel);
          __builtin_amdgcn_raw_buffer_store_b16((unsigned short)0x0A2Bu, out_rsrc, st_ok && last ? so_ + 8u : 0xFFFFFFFFu, 0, 0);
          fix_mask = last ? 0x7Fu : 0xFFu;
        }
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{sw[0], sw[1]}, out_rsrc, so_, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{qw[0], qw[1]}, out_rsrc, qo_, 0, 0);
        const uint32_t fix = (st.ok && !(dg & 16u)) ? (acc >> 17) & fix_mask : 0u;
        const unsigned long long fm = __ballot(fix != 0u);
        if (fm) {
          if (nfix + 64u > FIX_CAP) flush_fix();
          if (fix != 0u) fix_list[nfix + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = make_uint2(st.r | (fix << 6) | (c << 17), cw);
          nfix += (uint32_t)__popcll(fm);
        }
      };
      PStage cur = fetch_plain(a8);
      // (two dropped stores: every iteration then has the same vector-memory operations behind its prefetch, see below)
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, out_rsrc, 0xFFFFFFFFu, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, out_rsrc, 0xFFFFFFF0u, 0, 0);
      if (tail_whole) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)0u, out_rsrc, 0xFFFFFFE0u, 0, 0);
      for (uint32_t step = a8; step < b8; step++) {
        const PStage nxt = fetch_plain(min(step + 1u, b8 - 1u));
        run_plain(cur);
        cur = nxt;
      }
    }
    }   // (the two phases)
    // ---- clean steps: the plain step with the read's own bins, length and (behind the indel) shifted template ----
    if (n_clean) {
      const uint32_t ncsteps = (dg & 128u) ? 0u : (n_clean + 63u) / 64u;
      struct CStage { uint32_t r, c, src, out, np, inv; bool ok; uint2 w; };
      auto fetch_clean = [&](uint32_t step) -> CStage {
        CStage st;
        const uint32_t i_raw = step * 64u + lane;
        st.ok = i_raw < n_clean;
        const uint32_t e = clean[min(i_raw, n_clean - 1u)];   // idle lanes redo the list's last item, their stores are dropped
        st.r = e & 63u;
        st.c = e >> 7;
        const uint4 q0 = meta_rows[st.r * 2], q1 = meta_rows[st.r * 2 + 1];
        const uint32_t elen = (q1.w >> 16) & 0x7FFFu;
        const uint32_t delta = (e & 64u) ? ((q1.w >> 31) ? elen : 0u - elen) : 0u;
        st.src = (q0.y & 0x7FFFFFFFu) + 8u * st.c + delta;
        st.out = q0.w + 8u * st.c;
        st.np = q1.y & 0xFFFFu;
        st.inv = q1.z;
        const uint8_t* copy2 = (PAIRED ? m == 1u : (q0.y >> 31) != 0u) ? B.chains2_rc : B.chains2_fwd;
        __builtin_memcpy(&st.w, copy2 + (st.src >> 2), 8);
        return st;
      };
      const uint32_t c3b = dev_ctx(KIND_BASE, m, B.batch_id);
      const uint32_t lgW = P.lgW, blk_bytes = P.fast_stride * 4u;
      auto run_clean = [&](const CStage& st) {
        const uint32_t c = st.c;
        uint32_t cw;
        if (dg & 4u) cw = (g * G + st.r) * 2654435761u + c;  // ablation: no haplotype fetch
        else cw = (uint32_t)((((uint64_t)st.w.y << 32) | st.w.x) >> (2u * (st.src & 3u)));
        uint32_t x[8];
        const uint32_t slot = g * G + st.r + B.slot_offset;
        if (dg & 8u) {  // ablation: no Philox
#pragma unroll
          for (int z = 0; z < 8; z++) x[z] = (slot * 2654435761u) ^ (c * 40503u + z * 0x9E3779B9u);
        } else {
          philox_base(slot, 2u * c, 0, c3b, B.k0, B.k1, x);
          philox_base(slot, 2u * c + 1u, 0, c3b, B.k0, B.k1, x + 4);
        }
        // bin = i * binCount / n' (Profile.cpp:1672) by the row's reciprocal
        uint32_t so[8];
        const uint32_t ib0 = __umul24(8u * c, bins);
#pragma unroll
        for (int h = 0; h < 8; h++) so[h] = __umul24(__umulhi(ib0 + (uint32_t)h * bins, st.inv), blk_bytes) + img_at;
        const uint32_t qo0 = so[0], qo1 = so[1];
        if (c == 0u) { so[0] += 128u * 4u; so[1] += 64u * 4u; }
        uint32_t sw[2], qw[2], acc;
        sample8(code4, lgW, cw, x, so, qo0, qo1, sw, qw, acc);
        const bool st_ok = st.ok && !(dg & 1u);
        const uint32_t so_ = st_ok ? st.out : 0xFFFFFFFFu;
        const uint32_t qo_ = st_ok ? st.out + st.np + 3u : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{sw[0], sw[1]}, out_rsrc, so_, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{qw[0], qw[1]}, out_rsrc, qo_, 0, 0);
        const uint32_t fix = (st.ok && !(dg & 16u)) ? (acc >> 17) & 0xFFu : 0u;
        const unsigned long long fm = __ballot(fix != 0u);
        if (fm) {
          if (nfix + 64u > FIX_CAP) flush_fix();
          if (fix != 0u) fix_list[nfix + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))] = make_uint2(st.r | (fix << 6) | (c << 17), cw);
          nfix += (uint32_t)__popcll(fm);
        }
      };
      if (ncsteps) {
        CStage cur = fetch_clean(0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, out_rsrc, 0xFFFFFFFFu, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, out_rsrc, 0xFFFFFFF0u, 0, 0);
        for (uint32_t step = 0; step < ncsteps; step++) {
          const CStage nxt = fetch_clean(min(step + 1u, ncsteps - 1u));
          run_clean(cur);
          cur = nxt;
        }
      }
    }
    // ---- general steps ----
    if (nmain) {
      Stage cur = fetch_step(0);
      // three (dropped) stores: the loop is then entered with the same vector-memory operations behind the first
      // prefetch as every later iteration has behind its own (three item stores + the next prefetch), and the wait for a
      // prefetched window stays a counted one instead of falling back to the entry path's vmcnt(1)
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, out_rsrc, 0xFFFFFFFFu, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, out_rsrc, 0xFFFFFFF0u, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b16((unsigned short)0u, out_rsrc, 0xFFFFFFE0u, 0, 0);
      for (uint32_t step = 0; step < nmain; step++) {
        if (nslow > SLOW_CAP - 64u) flush_slow();
        // (the last step fetches itself again: an unconditional fetch keeps the loaded registers free of copies until
        // the end of the iteration, so the wait for them sits after this step's sampling)
        const Stage nxt = fetch_step(min(step + 1u, nmain - 1u));
        run_item(cur);
        cur = nxt;
      }
    }
    while (more) {  // items past the map (reads grown by insertions, reads of more than 64 items): never a first item
      if (nslow > SLOW_CAP - 64u) flush_slow();
      const uint32_t r = (uint32_t)__builtin_ctzll(more);
      const uint32_t c = cb + lane;
      const uint32_t nitems = ((meta_rows[r * 2 + 1].y & 0xFFFFu) + 7u) / 8u;
      cb += 64u;
      if (cb >= nitems) { more &= more - 1ull; cb = TI; }
      const Stage st = fetch_item(r, c, true, 1u);
      run_item(st);
    }
    if (nslow) flush_slow();
    if (nfix) flush_fix();
    // ---- per-read pass, lane = read: last partial item, record separators ----
    // These byte-granular stores touch lines the steps above have just written from this wave, so
    // they merge in L2.  Reads whose last item went to the generic code (0xFFFFFFFF row) get only the
    // separators here; emit_slow_kernel writes the same separator bytes again (benign).
    wave_lds_sync();
    if (lane < G && t < B.n_slots && !(dg & 2u)) {
      const uint4 r0 = meta_rows[lane * 2], r1 = meta_rows[lane * 2 + 1];
      if (r1.x & 0x7FFFFFFFu) {
        const uint32_t np = r1.y & 0xFFFFu, hl = r1.y >> 22;
        uint8_t* rec = gout + r0.z;
        const uint4 tr = make_uint4(r0.x, r0.y, r1.z, r1.w);  // the parked last item (fast_item)
        if ((np & 7u) == 7u && tr.x != 0xFFFFFFFFu) {
          // the last item went out with the steps, line breaks and the "+" line included
        } else {
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
    }
    wave_lds_sync();
  }
}

// The queued items of emit_fast_kernel, one lane each, through the generic item code; the tables are
// staged in LDS like in emit_kernel when they fit (persistent workgroup per CU, grid-stride over the queue).
template <int KT, bool SUB_LDS>
__global__ __launch_bounds__(EMIT_THREADS) void emit_slow_kernel(DevProfile P, DevBatch B, uint32_t sub_rows) {
  extern __shared__ uint4 smem[];
  const uint32_t m = blockIdx.y, tid = threadIdx.x, lane = tid & 63u;
  const uint32_t tm = B.paired ? m : 0u;
  uint32_t n = B.slowq_count[m];
  if (n > B.slowq_cap) n = B.slowq_cap;  // overflow: the host reruns the batch through emit_kernel
  if ((blockIdx.x * EMIT_THREADS) >= n) return;  // nothing for this workgroup: skip the staging too
  uint4* lds_sub = smem;
  const uint4* gsub = P.sub + (size_t)tm * P.sub_mate_rows;
  if (SUB_LDS)
    for (uint32_t i = tid; i < sub_rows; i += EMIT_THREADS) lds_sub[i] = gsub[i];
  __syncthreads();
  const uint2* q = B.slowq + (size_t)m * B.slowq_cap;
  for (uint32_t b0 = (blockIdx.x * EMIT_THREADS + tid) & ~63u; b0 < n; b0 += gridDim.x * EMIT_THREADS) {
    const uint32_t i = b0 + lane;
    const bool act = i < n;
    const uint2 e = q[act ? i : b0];
    const size_t idx = (size_t)m * B.n_slots + e.x;
    uint4 m0 = B.meta[idx * 3];
    const uint4 m1 = B.meta[idx * 3 + 1];
    const uint64_t ooff = rec_offset(B, m, e.x);
    m0.z = (uint32_t)ooff;
    m0.w = (uint32_t)(ooff >> 32);
    emit_item<KT, SUB_LDS>(P, B, lds_sub, gsub, m, m0, m1, e.x, e.y, act);
  }
}

// ------------------------------------------------------------------------------------------------
// haplotype encoding: ASCII -> base code, in place (A0 C1 T2 G3, 'N' = 4, 'X' = 6, anything else = 5)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t encode4(uint32_t w) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t b = (w >> (8 * i)) & 0xFFu;
    const bool acgt = (b == 'A') | (b == 'C') | (b == 'G') | (b == 'T');
    const uint32_t v = acgt ? ((b >> 1) & 3u) : (b == 'N' ? 4u : (b == 'X' ? 6u : 5u));
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

// GC factor and weight of a window (Profile::getGCFactor, Profile.cpp:1507-1517; Segment.cpp:576,586,615): one lane
// per window, the normal variate interpolated from the quantile table in three rounded fp64 operations (no fused
// multiply-add: the host / oracle evaluation of the same table must give the same bits)
__global__ __launch_bounds__(256) void gc_weight_kernel(const int32_t* __restrict__ gc, const sg_gc_window* __restrict__ wins,
                                                        const uint32_t* __restrict__ seg_ord, const uint32_t* __restrict__ win_ord,
                                                        uint64_t n, const double* __restrict__ means, double std,
                                                        const double* __restrict__ Q, uint32_t lg_cells, uint32_t frag,
                                                        int32_t full_tile_form, uint32_t c3, uint32_t k0, uint32_t k1,
                                                        double* __restrict__ out) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  const int32_t g = gc[w];
  double f = 0.0;
  if (g >= 0 && g <= 100) {
    const double mean = means[g];
    const uint32_t tail_bits = 32u - lg_cells;
    const double scale = 1.0 / (double)(1ull << (tail_bits + 1u));
    for (uint32_t a = 0;; a++) {
      uint32_t x[4];
      philox4x32_10(win_ord[w], a, seg_ord[w], c3, k0, k1, x);
      const uint32_t k = x[0] >> tail_bits, fr = x[0] & ((1u << tail_bits) - 1u);
      const double t = __dmul_rn((double)(2u * fr + 1u), scale);
      const double d = __dsub_rn(Q[k + 1], Q[k]);
      const double z = __dadd_rn(Q[k], __dmul_rn(d, t));
      f = __dadd_rn(mean, __dmul_rn(std, z));
      if (f >= 0.0) break;
    }
  }
  const uint32_t len = wins[w].len;
  out[w] = (full_tile_form && len == frag) ? __ddiv_rn(f, (double)frag)
                                           : __ddiv_rn(__dmul_rn(f, (double)len), (double)((uint64_t)frag * frag));
}

// ------------------------------------------------------------------------------------------------
// sampling plan on the device (sg_windows_build / sg_plan_windows / sg_plan_range)
// ------------------------------------------------------------------------------------------------
// generator of window w: the last one whose prefix (first window) is <= w
__device__ __forceinline__ uint32_t gen_of(const uint64_t* __restrict__ prefix, uint32_t n_gens, uint64_t w) {
  uint32_t lo = 0, hi = n_gens - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (prefix[mid] <= w) lo = mid; else hi = mid - 1;
  }
  return lo;
}
// lane = window: geometry of the tile (Segment.cpp:566-590), its segment and its ordinal inside the segment
__global__ __launch_bounds__(256) void tile_kernel(const sg_window_gen* __restrict__ gens, const uint64_t* __restrict__ prefix,
                                                   uint32_t n_gens, uint64_t n, uint32_t frag, const uint64_t* __restrict__ seg_first,
                                                   sg_gc_window* __restrict__ out, uint32_t* __restrict__ seg_ord, uint32_t* __restrict__ win_ord) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  const uint32_t g = gen_of(prefix, n_gens, w);
  const sg_window_gen G = gens[g];
  const uint64_t t = w - prefix[g], off = t * frag;
  sg_gc_window o;
  o.start = G.hap_base + off;
  o.chain = G.chain;
  o.len = (uint32_t)(G.hap_len - off < frag ? G.hap_len - off : frag);
  out[w] = o;
  seg_ord[w] = G.seg;
  win_ord[w] = (uint32_t)(w - seg_first[G.seg]);
}
// workgroup = segment: weight sum in window order (the reference's summation order, Segment.cpp:627-630; fp64 addition
// does not reassociate).  The whole workgroup stages tiles of the weights in LDS (coalesced), its first lane adds them
// one after the other: the chain of dependent adds is all that is serial (one lane per segment reading its weights
// from memory itself took 0.27 ms for 65 segments of 2000 windows).
#define SEG_SUM_TILE 4096
__global__ __launch_bounds__(256) void seg_sum_kernel(const double* __restrict__ wt, const uint64_t* __restrict__ seg_first, uint32_t n_segs,
                                                      double* __restrict__ out) {
  __shared__ double tile[SEG_SUM_TILE];
  const uint32_t k = blockIdx.x;
  const uint64_t w0 = seg_first[k], w1 = seg_first[k + 1];
  double acc = 0.0;
  for (uint64_t base = w0; base < w1; base += SEG_SUM_TILE) {
    const uint32_t n = (uint32_t)(w1 - base < SEG_SUM_TILE ? w1 - base : SEG_SUM_TILE);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) tile[i] = wt[base + i];
    __syncthreads();
    if (threadIdx.x == 0u)
      for (uint32_t i = 0; i < n; i++) acc = __dadd_rn(acc, tile[i]);
    __syncthreads();
  }
  if (threadIdx.x == 0u) out[k] = acc;
}
// lane = window of an active segment: fragRCs[i] = (long)(fragWeights[i] * readCount / totalWL), Segment.cpp:466-470
__global__ __launch_bounds__(256) void window_reads_kernel(const sg_window_gen* __restrict__ gens, const uint64_t* __restrict__ prefix,
                                                           uint32_t n_gens, uint64_t n, uint32_t frag, const double* __restrict__ wt,
                                                           const sg_active_seg* __restrict__ act, sg_window* __restrict__ rows,
                                                           unsigned long long* __restrict__ seg_sum) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = w < n;
  uint32_t seg = 0xFFFFFFFFu;
  long long rc = 0;
  if (valid) {
    const uint32_t g = gen_of(prefix, n_gens, w);
    const sg_window_gen G = gens[g];
    const uint64_t t = w - prefix[g], off = t * frag;
    const sg_active_seg A = act[G.seg];
    const double total = __dadd_rn(A.weight, 2.2204e-16);
    rc = (long long)__ddiv_rn(__dmul_rn(wt[G.first_window + t], (double)A.reads), total);
    sg_window o;
    o.hap_base = G.hap_base;
    o.chain = G.chain;
    o.spos = (uint32_t)off;
    o.len = (uint32_t)(G.hap_len - off < frag ? G.hap_len - off : frag);
    o.n_reads = (int32_t)rc;
    o.seg = G.seg;
    o.slot_base = 0;
    rows[w] = o;
    seg = G.seg;
  }
  // The segment's sum (an integer: any order): a wave's windows are nearly always of ONE segment -- one atomic for the
  // wave then, not 64 on the same address (64 k single-address atomics were 0.37 ms of this kernel's 0.38).
  const uint32_t seg0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg);  // (the first lane of a wave that has one is valid)
  if (__ballot(valid && seg != seg0) == 0ull) {
    unsigned long long sum = (unsigned long long)rc;  // 0 on the lanes past n
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_down(sum, d, 64);
    if ((threadIdx.x & 63u) == 0u && valid) atomicAdd(seg_sum + seg0, sum);
  } else if (valid) {
    atomicAdd(seg_sum + seg, (unsigned long long)rc);
  }
}
// lane = active segment: the remainder goes to the segment's first window (Segment.cpp:472-474)
__global__ __launch_bounds__(64) void seg_remainder_kernel(const sg_active_seg* __restrict__ act, const uint32_t* __restrict__ seg_first,
                                                           uint32_t n_act, const unsigned long long* __restrict__ seg_sum,
                                                           sg_window* __restrict__ rows) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_act) return;
  const long long sum = (long long)seg_sum[a];
  if (sum < act[a].reads) rows[seg_first[a]].n_reads += (int32_t)(act[a].reads - sum);
}
__global__ __launch_bounds__(256) void planned_kernel(const sg_window* __restrict__ rows, uint64_t n, int32_t paired, uint32_t* __restrict__ planned) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  const int32_t r = rows[w].n_reads;
  planned[w] = r <= 0 ? 0u : (paired ? ((uint32_t)r + 1u) / 2u : (uint32_t)r);
}
__global__ __launch_bounds__(256) void slot_base_kernel(sg_window* __restrict__ rows, uint64_t n, const uint64_t* __restrict__ off) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n) rows[w].slot_base = (uint32_t)off[w];
}
// planned fragments before each active segment's first window (and the total in [n_act])
__global__ __launch_bounds__(64) void seg_slots_kernel(const uint64_t* __restrict__ off, const uint32_t* __restrict__ seg_first, uint32_t n_act,
                                                       const uint64_t* __restrict__ total, uint64_t* __restrict__ out) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < n_act) out[a] = off[seg_first[a]];
  if (a == n_act) out[a] = *total;
}
// rows [w_lo, w_lo + n) of the batch table as a batch of their own: segment ordinals and slots relative to the run
__global__ __launch_bounds__(256) void slice_kernel(const sg_window* __restrict__ all, uint64_t w_lo, uint64_t n, uint32_t a0, uint32_t slot_lo,
                                                    sg_window* __restrict__ out) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n) return;
  sg_window o = all[w_lo + w];
  o.seg -= a0;
  o.slot_base -= slot_lo;
  out[w] = o;
}

// ------------------------------------------------------------------------------------------------
// launchers (called from sg_api.cpp through plain C++ declarations)
// ------------------------------------------------------------------------------------------------
// a pass's sizes and flags (totals[0..4]) to the context's pinned mailbox: a kernel of five lanes writing host memory
// reaches the host sooner than a copy command of 40 bytes through the DMA engine
__global__ void mail_kernel(const uint64_t* __restrict__ totals, uint64_t* __restrict__ mail) {
  if (threadIdx.x < 5u) mail[threadIdx.x] = totals[threadIdx.x];
}
void launch_mail(const uint64_t* totals, uint64_t* mail, hipStream_t s) {
  hipLaunchKernelGGL(mail_kernel, dim3(1), dim3(64), 0, s, totals, mail);
}
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
  hipLaunchKernelGGL(indel_kernel, dim3((B.n_slots + 255) / 256), dim3(256), 0, s, P, B);
}
bool emit_uses_fast_kernel(const DevProfile& P, const DevBatch& B);
void launch_header(const DevProfile& P, const DevBatch& B, hipStream_t s) {
  if (!B.n_slots) return;
  const bool fast = emit_uses_fast_kernel(P, B);
  if (fast && B.prefix_len <= 16u) return;  // the straight-line kernel writes the names of its reads itself
  dim3 grid((B.n_slots + 255) / 256, B.paired ? 2 : 1);
  hipLaunchKernelGGL(header_kernel, grid, dim3(256), 0, s, B);
}
uint32_t scan_blocks(uint32_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
uint32_t record_seg_shift(uint32_t n_slots) {  // smallest shift with at most 16 segments of 2^shift blocks of 256 reads
  const uint32_t nblk = (n_slots + 255u) >> 8;
  uint32_t k = 0;
  while (((nblk + (1u << k) - 1u) >> k) > 16u) k++;
  return k;
}
void launch_scan(const DevBatch& B, hipStream_t s) {  // block sums of indel_kernel -> block / segment bases, text sizes
  if (!B.n_slots) return;
  hipLaunchKernelGGL(block_base_kernel, dim3(16, B.paired ? 2 : 1), dim3(1024), 0, s, B);
}
// exclusive scan of n u32 values into u64 offsets (one row); total -> *total
void launch_scan_u32(const uint32_t* in, uint32_t n, uint64_t* bsum, uint64_t* out, uint64_t* total, hipStream_t s) {
  if (!n) return;
  const uint32_t nblk = scan_blocks(n);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblk, 1), dim3(SCAN_BLOCK), 0, s, in, n, bsum, nblk);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, s, bsum, nblk, total);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nblk, 1), dim3(SCAN_BLOCK), 0, s, in, n, bsum, nblk, out);
}
// LDS budget: the generic kernels stage the substitution rows when they fit (the alias columns are read through L2);
// the straight-line kernel needs its whole table image.
static const size_t kLdsBytes = 160 * 1024;
template <int KT, bool SL>
static void launch_emit_variant(const DevProfile& P, const DevBatch& B, dim3 grid, size_t lds, uint32_t sub_rows, uint32_t TI,
                                uint32_t RPI, hipStream_t s) {
  (void)hipFuncSetAttribute((const void*)emit_kernel<KT, SL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((emit_kernel<KT, SL>), grid, dim3(EMIT_THREADS), lds, s, P, B, sub_rows, TI, RPI);
}
struct EmitLds { uint32_t sub_rows, fast_TI; size_t lds, lds_fast; bool sub_lds, fast_fits; };
static EmitLds emit_lds(const DevProfile& P) {
  EmitLds e;
  uint32_t kmer_count = 0;
  for (int m = 1, p = 1; m <= P.kmer; m++) { p *= 4; kmer_count += p; }
  e.sub_rows = kmer_count * (uint32_t)P.bins;
  const size_t meta_b = (size_t)EMIT_WAVES * 64 * META_ROW;
  const size_t sub_b = (size_t)e.sub_rows * 16;
  e.sub_lds = meta_b + sub_b <= kLdsBytes;
  e.lds = meta_b + (e.sub_lds ? sub_b : 0);
  // straight-line kernel: table image + look-up rows + character table + read rows + queues + read order + appended items
  e.fast_TI = ((uint32_t)P.L + 7u) / 8u;
  if (e.fast_TI > 64u) e.fast_TI = 64u;  // longer reads finish in steps of their own
  const size_t img_b = (((size_t)P.bins * P.fast_stride + 3) & ~(size_t)3) * 4;
  e.lds_fast = img_b + (size_t)e.fast_TI * LUT_ROW * 4 + 256 * 4 + meta_b + (size_t)EMIT_WAVES * SLOW_CAP * 4 + (size_t)EMIT_WAVES * FIX_CAP * 8 +
               (size_t)EMIT_WAVES * 64 * 2 + (size_t)EMIT_WAVES * OVF_CAP * 2;
  e.fast_fits = P.kmer == 3 && P.fast_lds != nullptr && e.lds_fast <= kLdsBytes && P.bins < 256;
  return e;
}
// 0: generic kernel, 1: straight-line kernel
static int emit_fast_mode(const DevProfile& P) {
  if (getenv("SG_DIAG") != nullptr) return 0;
  return emit_lds(P).fast_fits ? 1 : 0;
}
int emit_variant(const DevProfile& P) { return emit_fast_mode(P); }
bool emit_uses_fast_kernel(const DevProfile& P, const DevBatch& B) {
  // the straight-line kernel addresses the 2-bit haplotype copies with 32-bit base indexes
  return emit_fast_mode(P) != 0 && B.chains_total < (1ull << 31);
}
void launch_emit(const DevProfile& P, const DevBatch& B, hipStream_t s, bool force_generic, hipEvent_t after_main) {
  if (!B.n_slots) {
    if (after_main) (void)hipEventRecord(after_main, s);
    return;
  }
  const EmitLds e = emit_lds(P);
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const uint32_t nm = B.paired ? 2 : 1;
  // generic kernel, fixed lane map: TI items of 8 bases per read, RPI reads per wave iteration
  uint32_t TI = ((uint32_t)P.L + 3u + 7u) / 8u;
  if (TI > 64u) TI = 64u;  // longer reads finish in the clean-up loop
  const uint32_t RPI = 64u / TI;
  const uint32_t G = RPI * (64u / RPI);
  uint32_t gx = (uint32_t)cus / nm;  // one 1024-thread workgroup per CU, persistent over read groups
  if (gx < 1) gx = 1;
  const uint32_t need = ((B.n_slots + G - 1u) / G + EMIT_WAVES - 1) / EMIT_WAVES;
  if (gx > need) gx = need;
  const dim3 grid(gx, nm);
  const int mode = (force_generic || !emit_uses_fast_kernel(P, B)) ? 0 : 1;
  if (mode != 0) {
    // straight-line kernel: item-stream map, 63 reads per group
    const uint32_t TIf = e.fast_TI;
    const uint32_t fneed = ((B.n_slots + 62u) / 63u + EMIT_WAVES - 1) / EMIT_WAVES;
    uint32_t fgx = (uint32_t)cus / nm;
    if (fgx < 1) fgx = 1;
    if (fgx > fneed) fgx = fneed;
    const dim3 fgrid(fgx, nm);
    const uint32_t inv_TI = (1u << 20) / TIf + 1u;
    // the list of the one-indel reads' clean items, where the table image leaves room for it (else those reads stay whole
    // in the general steps: same output)
    // (CLEAN_CAP entries per wave, or as many 64s as fit beside a large image: the 41-symbol profiles leave 6.9 KB)
    uint32_t clean_cap = (uint32_t)std::min<size_t>(CLEAN_CAP, ((kLdsBytes - e.lds_fast) / (EMIT_WAVES * 2)) & ~(size_t)63);
    if (getenv("SG_NO_CLEAN_STEPS") != nullptr) clean_cap = 0;
    if (const char* e = getenv("SG_CLEAN_CAP")) clean_cap = std::min(clean_cap, (uint32_t)strtoul(e, nullptr, 10) & ~63u);  // (tests: a list that overflows)
    const size_t lds_fast = e.lds_fast + (size_t)EMIT_WAVES * clean_cap * 2;
    auto launch_fast = [&](auto kern) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fast);
      hipLaunchKernelGGL(kern, fgrid, dim3(EMIT_THREADS), lds_fast, s, P, B, TIf, inv_TI, clean_cap);
    };
    if (B.diag) {  // timing ablations (SG_FDIAG)
      if (B.paired) launch_fast(emit_fast_kernel<true, true>);
      else launch_fast(emit_fast_kernel<false, true>);
    } else {
      if (B.paired) launch_fast(emit_fast_kernel<true, false>);
      else launch_fast(emit_fast_kernel<false, false>);
    }
    if (after_main) (void)hipEventRecord(after_main, s);
    // the queued items through the generic code (reference-order tables)
    const size_t slow_sub = e.sub_lds ? (size_t)e.sub_rows * 16 : 0;
    auto launch_slow = [&](auto kern) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slow_sub);
      hipLaunchKernelGGL(kern, grid, dim3(EMIT_THREADS), slow_sub, s, P, B, e.sub_rows);
    };
    if (slow_sub) launch_slow(emit_slow_kernel<3, true>);
    else launch_slow(emit_slow_kernel<3, false>);
  } else {
    if (P.kmer == 3 && e.sub_lds) launch_emit_variant<3, true>(P, B, grid, e.lds, e.sub_rows, TI, RPI, s);
    else if (e.sub_lds) launch_emit_variant<0, true>(P, B, grid, e.lds, e.sub_rows, TI, RPI, s);
    else launch_emit_variant<0, false>(P, B, grid, e.lds, e.sub_rows, TI, RPI, s);
    if (after_main) (void)hipEventRecord(after_main, s);
  }
}
void launch_encode(uint8_t* buf, size_t bytes, hipStream_t s) {  // bytes is a multiple of 16
  if (!bytes) return;
  const size_t n16 = bytes / 16;
  uint32_t grid = (uint32_t)std::min<size_t>((n16 + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(encode_kernel, dim3(grid), dim3(256), 0, s, (uint4*)buf, n16);
}
void launch_gc_weight(const int32_t* gc, const sg_gc_window* wins, const uint32_t* seg_ord, const uint32_t* win_ord, uint64_t n,
                      const double* means, double std, const double* Q, uint32_t lg_cells, uint32_t frag, int32_t full_tile_form,
                      uint32_t ctx24, uint64_t seed, double* out, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(gc_weight_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, gc, wins, seg_ord, win_ord, n, means, std, Q,
                     lg_cells, frag, full_tile_form, KIND_GC | (ctx24 << 8), (uint32_t)seed, (uint32_t)(seed >> 32), out);
}
static inline uint32_t blocks256(uint64_t n) { return (uint32_t)((n + 255) / 256); }
void launch_tile(const sg_window_gen* gens, const uint64_t* prefix, uint32_t n_gens, uint64_t n, uint32_t frag, const uint64_t* seg_first,
                 sg_gc_window* out, uint32_t* seg_ord, uint32_t* win_ord, hipStream_t s) {
  if (n) hipLaunchKernelGGL(tile_kernel, dim3(blocks256(n)), dim3(256), 0, s, gens, prefix, n_gens, n, frag, seg_first, out, seg_ord, win_ord);
}
void launch_seg_sum(const double* wt, const uint64_t* seg_first, uint32_t n_segs, double* out, hipStream_t s) {
  if (n_segs) hipLaunchKernelGGL(seg_sum_kernel, dim3(n_segs), dim3(256), 0, s, wt, seg_first, n_segs, out);
}
void launch_window_reads(const sg_window_gen* gens, const uint64_t* prefix, uint32_t n_gens, uint64_t n, uint32_t frag, const double* wt,
                         const sg_active_seg* act, const uint32_t* seg_first, uint32_t n_act, sg_window* rows, unsigned long long* seg_sum,
                         int32_t paired, uint32_t* planned, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(window_reads_kernel, dim3(blocks256(n)), dim3(256), 0, s, gens, prefix, n_gens, n, frag, wt, act, rows, seg_sum);
  hipLaunchKernelGGL(seg_remainder_kernel, dim3((n_act + 63) / 64), dim3(64), 0, s, act, seg_first, n_act, seg_sum, rows);
  hipLaunchKernelGGL(planned_kernel, dim3(blocks256(n)), dim3(256), 0, s, rows, n, paired, planned);
}
void launch_slot_base(sg_window* rows, uint64_t n, const uint64_t* off, const uint32_t* seg_first, uint32_t n_act, const uint64_t* total,
                      uint64_t* seg_slots, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(slot_base_kernel, dim3(blocks256(n)), dim3(256), 0, s, rows, n, off);
  hipLaunchKernelGGL(seg_slots_kernel, dim3((n_act + 64) / 64), dim3(64), 0, s, off, seg_first, n_act, total, seg_slots);
}
void launch_slice(const sg_window* all, uint64_t w_lo, uint64_t n, uint32_t a0, uint32_t slot_lo, sg_window* out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(slice_kernel, dim3(blocks256(n)), dim3(256), 0, s, all, w_lo, n, a0, slot_lo, out);
}
void launch_gc(const uint8_t* chains, const uint64_t* chain_off, const sg_gc_window* wins, uint64_t n, int32_t* out, hipStream_t s) {
  if (!n) return;
  uint32_t grid = (uint32_t)((n + 3) / 4);
  hipLaunchKernelGGL(gc_kernel, dim3(grid), dim3(256), 0, s, chains, chain_off, wins, n, out);
}

}  // namespace sg
