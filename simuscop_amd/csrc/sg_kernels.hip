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
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
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
// plan: one lane per window
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plan_kernel(DevProfile P, DevBatch B) {
  uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= B.n_windows) return;
  const sg_window win = B.windows[w];
  const int n = win.n_reads;
  const uint32_t planned = n <= 0 ? 0u : (B.paired ? ((uint32_t)n + 1u) / 2u : (uint32_t)n);
  const uint64_t clen = B.chain_len[win.chain];
  const uint32_t c3 = dev_ctx(KIND_PLAN, 0, B.batch_id);
  const uint32_t L = (uint32_t)P.L;
  uint32_t done = 0, fail = 0, attempt = 0;
  while (done < planned) {
    uint32_t x[4];
    philox4x32_10((uint32_t)w + B.win_offset, attempt++, 0, c3, B.k0, B.k1, x);
    // threadPool->randomInteger(spos, epos+1): (long)(start + (end-start)*(x/2^32)) in fp64
    double frac = __dmul_rn((double)x[0], 1.0 / 4294967296.0);
    double v = __dadd_rn((double)win.spos, __dmul_rn((double)win.len, frac));
    uint32_t pos = (uint32_t)(long long)v;
    uint32_t isz;
    if (B.paired) isz = P.isz_row ? (uint32_t)P.isz_min + row_search(P.isz_row, P.isz_lg, x[1]) : (uint32_t)P.fixed_isz;
    else isz = win.len;
    uint64_t avail = clen - (win.hap_base + pos);
    uint32_t flen = avail < (uint64_t)isz ? (uint32_t)avail : isz;
    if (flen < L) {
      if (++fail > 1000) break;
      continue;
    }
    uint32_t strand = B.paired ? 0u : (x[2] >> 31);  // randomInteger(0,2)
    PairRec r;
    r.win = (uint32_t)w; r.relpos = pos - win.spos; r.fl = flen | (strand << 31); r.k = done;
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
  if (!flen) { B.rlen[idx] = 0; B.reclen[idx] = 0; return; }
  const int L = P.L;
  int j = 0, dl = 0;
  uint32_t nev = 0;
  uint32_t* ev = B.events + idx * SG_MAX_EVENTS;
  const uint32_t c3 = dev_ctx(KIND_INDEL, m, B.batch_id);
  for (int c = 0; 2 * c < L; c++) {
    uint32_t x[4];
    philox4x32_10(t + B.slot_offset, (uint32_t)c, 0, c3, B.k0, B.k1, x);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int jj = 2 * c + h;
      if (jj >= L || jj < j) continue;
      const uint32_t xi = x[2 * h], xd = x[2 * h + 1];
      j = jj + 1;
      if (xi <= P.Tins) {
        uint32_t len = row_search(P.ins_row, P.ins_lg, aux_draw(B, t, (uint32_t)jj, 0, m));
        if (len > 0) {
          if (nev < SG_MAX_EVENTS) ev[nev] = ev_pack((uint32_t)jj, len, 0);
          nev++;
          dl += (int)len;
        }
      } else if (xd < P.Cdel) {
        uint32_t len = row_search(P.del_row, P.del_lg, aux_draw(B, t, (uint32_t)jj, 0, m));
        if (len > 0) {
          uint32_t k = min((uint32_t)(L - jj), len);
          if (nev < SG_MAX_EVENTS) ev[nev] = ev_pack((uint32_t)jj, k, 1);
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
  const uint32_t namepos = (win.spos + rec.relpos) % B.seg_size[win.seg];
  const uint32_t fragcount = B.win_namebase[rec.win] + rec.k + 1u;
  B.reclen[idx] = B.prefix_len + ndigits(namepos) + 1u + ndigits(fragcount) + (B.paired ? 2u : 0u) + 1u + 2u * np + 4u;
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
// emit: one lane per read; samples substitution + quality per base and writes the FASTQ record
// ------------------------------------------------------------------------------------------------
struct Packer {  // byte stream -> unaligned dword stores
  uint8_t* p;
  uint32_t w, nb;
  __device__ __forceinline__ void init(uint8_t* dst) { p = dst; w = 0; nb = 0; }
  __device__ __forceinline__ void push(uint32_t b) {
    w |= b << (nb * 8);
    if (++nb == 4) {
      __builtin_memcpy(p, &w, 4);
      p += 4; w = 0; nb = 0;
    }
  }
  __device__ __forceinline__ void flush() {
    for (uint32_t i = 0; i < nb; i++) p[i] = (uint8_t)(w >> (8 * i));
    p += nb; nb = 0; w = 0;
  }
  __device__ __forceinline__ void push_dec(uint32_t v) {
    uint32_t nd = ndigits(v);
    uint32_t div = 1;
    for (uint32_t i = 1; i < nd; i++) div *= 10;
    for (uint32_t i = 0; i < nd; i++) {
      uint32_t d = v / div;
      push('0' + d);
      v -= d * div;
      div /= 10;
    }
  }
};

__global__ __launch_bounds__(256) void emit_kernel(DevProfile P, DevBatch B) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t m = blockIdx.y;
  if (t >= B.n_slots) return;
  const PairRec rec = B.pairs[t];
  const uint32_t flen = rec.fl & 0x7FFFFFFFu;
  if (!flen) return;
  const size_t idx = (size_t)m * B.n_slots + t;
  const uint64_t off = B.recoff[idx];
  const uint32_t rl = B.rlen[idx];
  const uint32_t np = rl & 0xFFFFu;
  const uint32_t nev = rl >> 16;
  if (off + B.reclen[idx] > B.out_cap[m]) return;  // host re-checks totals before launching
  const sg_window win = B.windows[rec.win];
  const uint32_t pos = win.spos + rec.relpos;
  const uint8_t* frag = B.chains + B.chain_off[win.chain] + win.hap_base + pos;
  const bool rev = B.paired ? (m == 1) : ((rec.fl >> 31) != 0);
  const uint32_t tm = B.paired ? m : 0u;  // SE always samples from the mate-1 tables (Segment.cpp:770,777)
  const uint32_t L = (uint32_t)P.L;

  // ---- header: @popu#chr#pos%segsize#fragCount[/m]\n  (Segment.cpp:780,809,824) ----
  uint8_t* o = B.out[m] + off;
  Packer ps;
  ps.init(o);
  for (uint32_t i = 0; i < B.prefix_len; i++) ps.push(B.prefix[i]);
  const uint32_t namepos = pos % B.seg_size[win.seg];
  const uint32_t fragcount = B.win_namebase[rec.win] + rec.k + 1u;
  ps.push_dec(namepos);
  ps.push('#');
  ps.push_dec(fragcount);
  if (B.paired) { ps.push('/'); ps.push('1' + m); }
  ps.push('\n');
  const uint32_t hdr = B.prefix_len + ndigits(namepos) + 1u + ndigits(fragcount) + (B.paired ? 2u : 0u) + 1u;
  Packer pq;
  pq.init(o + hdr + np + 3u);

  // ---- event cursor ----
  const uint32_t* ev = B.events + idx * SG_MAX_EVENTS;
  uint32_t e = 0;
  uint32_t nextw = nev ? ev[0] : 0xFFFFFFFFu;  // j field 0xFFFF never matches (L < 65535)
  uint32_t ins_left = 0, ins_j = 0, ins_f = 1;

  // ---- source walk state ----
  uint32_t j = 0;                         // next reference position of the L-base read template
  uint32_t cw = 0, cidx = 0xFFFFFFFFu;    // cached 4 haplotype bytes
  const uint32_t K = (uint32_t)P.kmer;
  const uint32_t ctxmask = (1u << (2 * K)) - 1u;
  uint32_t ctxv = 0, vc = 0;
  uint32_t bin = 0, acc = 0;              // bin = i*bins/np, acc = i*bins - bin*np
  const uint32_t bins = (uint32_t)P.bins;
  const uint32_t c3b = dev_ctx(KIND_BASE, m, B.batch_id);
  const uint4* subt = P.sub + (size_t)tm * P.sub_mate_rows;

  for (uint32_t i0 = 0; i0 < np; i0 += 2) {
    uint32_t x[4];
    philox4x32_10(t + B.slot_offset, i0 >> 1, 0, c3b, B.k0, B.k1, x);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t i = i0 + h;
      if (i >= np) break;
      const uint32_t xs = x[2 * h], xq = x[2 * h + 1];
      // -- next source base: profile code 0..3, or 4 = not in `bases` --
      uint32_t code;
      if (ins_left) {
        code = __umulhi(aux_draw(B, t, ins_j, ins_f, m), 3u);  // randomInteger(0, N-1): never the last base
        ins_f++;
        ins_left--;
      } else {
        while ((nextw & 0xFFFFu) == j && (nextw >> 31)) {      // deletion(s) starting here
          j += (nextw >> 16) & 0x7FFFu;
          e++;
          nextw = e < nev ? ev[e] : 0xFFFFFFFFu;
        }
        const uint32_t f = rev ? flen - 1u - j : j;
        const uint32_t widx = f >> 2;
        if (widx != cidx) {
          __builtin_memcpy(&cw, frag + (size_t)widx * 4, 4);
          cidx = widx;
        }
        const uint32_t byte = (cw >> ((f & 3u) * 8u)) & 0xFFu;
        const bool valid = (byte == 'A') | (byte == 'C') | (byte == 'G') | (byte == 'T');
        uint32_t nat = (byte >> 1) & 3u;   // A0 C1 T2 G3
        if (rev) nat ^= 2u;                 // complement (Segment.cpp:81-103)
        code = valid ? ((P.remap_packed >> (2u * nat)) & 3u) : 4u;
        if ((nextw & 0xFFFFu) == j) {      // insertion after this base
          ins_left = (nextw >> 16) & 0x7FFFu;
          ins_j = j;
          ins_f = 1;
          e++;
          nextw = e < nev ? ev[e] : 0xFFFFFFFFu;
        }
        j++;
      }
      // -- k-mer context ending at i (Profile::initKmers order, Profile.cpp:70-124) --
      const bool valid = code < 4u;
      ctxv = ((ctxv << 2) | (code & 3u)) & ctxmask;
      vc = valid ? min(vc + 1u, K) : 0u;
      const uint32_t mlen = min(i + 1u, K);
      int k;
      if (vc >= mlen) {
        const uint32_t kidx = P.kmer_off[mlen] + (ctxv & ((1u << (2u * mlen)) - 1u));
        const uint4 row = subt[(size_t)kidx * bins + bin];
        uint32_t c = (xs > row.x) + (xs > row.y) + (xs > row.z);
        k = (int)max(c, row.w);
      } else {
        k = valid ? (int)code : -1;
      }
      uint32_t ch, q;
      if (k >= 0) {
        ch = (P.bases_packed >> (8u * (uint32_t)k)) & 0xFFu;
        const uint32_t bp = code * 4u + (uint32_t)k;
        q = (uint32_t)P.min_qual + row_search(P.qual + (size_t)(bp * bins + bin) * P.qual_stride, P.qual_lg, xq);
      } else {
        ch = 'N';
        q = (uint32_t)P.min_qual + __umulhi(xq, 20u);  // getRandBaseQuality, Profile.cpp:1582-1584
      }
      ps.push(ch);
      pq.push(q);
      acc += bins;
      while (acc >= np) { acc -= np; bin++; }
    }
  }
  ps.push('\n'); ps.push('+'); ps.push('\n');
  ps.flush();
  pq.push('\n');
  pq.flush();
  (void)L;
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
      else if (left < 4) x &= (1u << (8 * left)) - 1u;
      gc += count_eq_bytes(x, 'G') + count_eq_bytes(x, 'C');
      nn += count_eq_bytes(x, 'N');
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
  uint32_t grid = (uint32_t)((B.n_windows + 255) / 256);
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
uint32_t scan_blocks(uint32_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
void launch_scan(const DevBatch& B, uint64_t* bsum, hipStream_t s) {
  if (!B.n_slots) return;
  const uint32_t nm = B.paired ? 2 : 1;
  const uint32_t nblk = scan_blocks(B.n_slots);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblk, nm), dim3(SCAN_BLOCK), 0, s, B.reclen, B.n_slots, bsum, nblk);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(nm), dim3(1024), 0, s, bsum, nblk, B.totals);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nblk, nm), dim3(SCAN_BLOCK), 0, s, B.reclen, B.n_slots, bsum, nblk, B.recoff);
}
void launch_emit(const DevProfile& P, const DevBatch& B, hipStream_t s) {
  if (!B.n_slots) return;
  dim3 grid((B.n_slots + 255) / 256, B.paired ? 2 : 1);
  hipLaunchKernelGGL(emit_kernel, grid, dim3(256), 0, s, P, B);
}
void launch_gc(const uint8_t* chains, const uint64_t* chain_off, const sg_gc_window* wins, uint64_t n, int32_t* out, hipStream_t s) {
  if (!n) return;
  uint32_t grid = (uint32_t)((n + 3) / 4);
  hipLaunchKernelGGL(gc_kernel, dim3(grid), dim3(256), 0, s, chains, chain_off, wins, n, out);
}

}  // namespace sg
