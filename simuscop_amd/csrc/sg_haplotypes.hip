// sg_haplotypes.hip -- reference ingest and haplotype assembly on the device.
//
// The reference reads one <= 1 Mbp slice per segment through the .fai index, twice per run
// (lib/fastahack/Fasta.cpp:304-334 via lib/segment/Segment.cpp:137), upper-cases it (Segment.cpp:143),
// replicates it per haplotype copy and edits the std::strings in place (Segment.cpp:210-447).  Here the
// FASTA file is streamed to HBM as it is, the contigs are turned into base codes once (newlines
// dropped by index arithmetic), and a haplotype chain is a list of copies: pieces of the encoded
// reference or of literal (inserted) bases, followed by single-base patches (SNP / SNV alleles).
// All of it is HBM-bound byte movement: 1 B read + 1 B written per haplotype base.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "simuscop_amd.h"

namespace sg {

// base codes shared with sg_kernels.hip: A0 C1 T2 G3, 'N' = 4, 'X' = 6 (the k-mer trie's place holder, template_code in
// sg_kernels.hip), anything else = 5
// A0 C1 T2 G3, N 4, X 6 (the k-mer trie's place holder, Profile.cpp:94-101), anything else 5; letters in either case
// (Segment.cpp:143 upper-cases the reference slice, :456 the variant alleles).  Without a branch: bytes 0x40..0x5F after
// bit 5 is cleared index a 32 x 4-bit table held in two constants -- written with compares, the compiler made a chain of
// wave-level branches per byte and the ingest kernel ran at 0.25 TB/s.
__device__ __forceinline__ uint32_t encode_base(uint32_t b) {
  const uint32_t idx = (b & 0xDFu) - 0x40u;                 // 'A' 1, 'C' 3, 'G' 7, 'N' 14, 'T' 20, 'X' 24
  const uint64_t t0 = 0x5455555535551505ull, t1 = 0x5555555655525555ull;   // entries 0-15, 16-31
  const uint64_t t = (idx & 16u) ? t1 : t0;
  const uint32_t v = (uint32_t)(t >> ((idx & 15u) * 4u)) & 0xFu;
  return idx < 32u ? v : 5u;
}

// ---- header scan: offsets of '>' / '@' at a line start; ';' comment lines raise flag 1 ----------
__global__ __launch_bounds__(256) void ref_scan_kernel(const uint8_t* __restrict__ raw, uint64_t n, uint64_t* __restrict__ list,
                                                       uint32_t cap, uint32_t* __restrict__ count, uint32_t* __restrict__ flags) {
  for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (uint64_t)gridDim.x * blockDim.x * 16) {
    uint4 w = *(const uint4*)(raw + i);  // the raw buffer is padded to a multiple of 16
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    // sixteen bytes without any of the three characters (all but a handful of a genome's blocks): one test, no byte loop
    // (a branch per byte made this scan 0.94 G scalar instructions for 3.1 GB)
    auto has = [](uint32_t x, uint32_t pat) { const uint32_t y = x ^ pat; return (y - 0x01010101u) & ~y & 0x80808080u; };
    uint32_t any = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) any |= has(ws[q], 0x3E3E3E3Eu) | has(ws[q], 0x40404040u) | has(ws[q], 0x3B3B3B3Bu);
    if (!any) continue;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const uint32_t b = (ws[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
      if (b != '>' && b != '@' && b != ';') continue;
      const uint64_t p = i + (uint64_t)k;
      if (p >= n) continue;
      const bool line_start = p == 0 || raw[p - 1] == '\n';
      if (!line_start) continue;
      if (b == ';') { atomicOr(flags, 1u); continue; }
      const uint32_t slot = atomicAdd(count, 1u);
      if (slot < cap) list[slot] = p;
    }
  }
}

// ---- contig ingest: raw FASTA lines of fixed width -> base codes ---------------------------------
struct DevContig {
  uint64_t raw_off;    // first base in the raw buffer
  uint64_t code_off;   // first code in the encoded reference
  uint64_t length;     // bases
  uint32_t line_bases, line_width;
  uint64_t first_block;  // exclusive prefix of 16-base blocks over the contigs
};

// lane = 16 consecutive bases of one contig.  Line structure is verified on the way: a line break
// byte that is not '\n' / '\r', or a line break byte inside a line, sets flag 2 (the host then falls
// back to its general parser).  An image without line ends (line_width == line_bases: what the general
// parser uploads, one line per contig) may hold carriage returns: fastahack keeps them as bases
// (Fasta.cpp:150-199) and so does the default reading here; they encode as "other".
__global__ __launch_bounds__(256) void ref_ingest_kernel(const uint8_t* __restrict__ raw, uint8_t* __restrict__ codes,
                                                         const DevContig* __restrict__ contigs, uint32_t n_contigs,
                                                         uint64_t n_blocks, uint32_t* __restrict__ flags) {
  for (uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; blk < n_blocks; blk += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t lo = 0, hi = n_contigs - 1;  // last contig with first_block <= blk
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) >> 1;
      if (contigs[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const DevContig c = contigs[lo];
    const uint64_t i0 = (blk - c.first_block) * 16;
    // (a contig holds fewer than 2^32 bases -- the host refuses longer ones --, so line and column are 32-bit divisions:
    // the 64-bit one was most of this kernel's 12.7 ms on a 3.1 Gbp genome)
    const uint32_t line = (uint32_t)i0 / c.line_bases;
    uint32_t r = (uint32_t)i0 - line * c.line_bases;
    uint64_t src = c.raw_off + (uint64_t)line * c.line_width + r;
    const uint32_t nl = c.line_width - c.line_bases;
    bool bad = false;
    const uint32_t nb = (uint32_t)(c.length - i0 < 16 ? c.length - i0 : 16);
    if (nb == 16u && c.line_bases >= 16u) {
      // Sixteen bases that meet at most one line end (every lane but a contig's last): two unaligned 16-byte loads -- the
      // bytes up to the line end from the first, the bytes behind it from the second, taken nl bytes on -- instead of a
      // byte loop that the whole wave walked for the one lane in four that crosses a line end.
      const uint32_t first_len = c.line_bases - r;            // bases left in this line (>= 1)
      uint4 va, vb;
      __builtin_memcpy(&va, raw + src, 16);
      vb = va;
      if (first_len < 16u) __builtin_memcpy(&vb, raw + src + nl, 16);
      if (first_len <= 16u && i0 + first_len < c.length)      // the line end inside or right behind the block: its bytes
        for (uint32_t t = 0; t < nl; t++) { const uint32_t e = raw[src + first_len + t]; bad |= !((e == '\n') | (e == '\r')); }
      auto pick = [&](uint32_t a, uint32_t b2, uint32_t q) {   // word q: bytes below first_len from a, the rest from b2
        const uint32_t lo4 = 4u * q;
        const uint32_t m = first_len >= lo4 + 4u ? 0xFFFFFFFFu : (first_len <= lo4 ? 0u : (1u << (8u * (first_len - lo4))) - 1u);
        return (a & m) | (b2 & ~m);
      };
      auto enc4 = [&](uint32_t w) {
        uint32_t o = 0;
#pragma unroll
        for (int z = 0; z < 4; z++) {
          const uint32_t b = (w >> (8 * z)) & 0xFFu;
          bad |= (b == '\n') | ((b == '\r') & (nl != 0u));
          o |= encode_base(b) << (8 * z);
        }
        return o;
      };
      const uint4 o = make_uint4(enc4(pick(va.x, vb.x, 0)), enc4(pick(va.y, vb.y, 1)), enc4(pick(va.z, vb.z, 2)), enc4(pick(va.w, vb.w, 3)));
      if (bad) atomicOr(flags, 2u);
      *(uint4*)(codes + c.code_off + i0) = o;
      continue;
    }
    // (a contig's last block, or lines shorter than a block: byte by byte)
    uint64_t wlo = 0x0404040404040404ull, whi = 0x0404040404040404ull;
    for (uint32_t k = 0; k < nb; k++) {
      const uint32_t b = raw[src];
      bad |= (b == '\n') | ((b == '\r') & (nl != 0u));
      const uint64_t v8 = (uint64_t)encode_base(b), sh = (uint64_t)(k & 7u) * 8u;
      if (k < 8u) wlo = (wlo & ~(0xFFull << sh)) | (v8 << sh);
      else whi = (whi & ~(0xFFull << sh)) | (v8 << sh);
      src++;
      if (++r == c.line_bases) {
        if (i0 + k + 1 < c.length)  // not after the contig's last base (the file may end without a newline)
          for (uint32_t t = 0; t < nl; t++) { const uint32_t e = raw[src + t]; bad |= !((e == '\n') | (e == '\r')); }
        src += nl;
        r = 0;
      }
    }
    if (bad) atomicOr(flags, 2u);
    *(uint4*)(codes + c.code_off + i0) = make_uint4((uint32_t)wlo, (uint32_t)(wlo >> 32), (uint32_t)whi, (uint32_t)(whi >> 32));  // code_off is 16-aligned
  }
}

// ---- haplotype assembly ---------------------------------------------------------------------------
struct DevPiece { uint64_t dst; uint64_t src; uint32_t len; uint32_t pad; };  // absolute byte offsets

// workgroup per piece (the host splits long pieces): aligned 16-byte stores, unaligned 16-byte loads
__global__ __launch_bounds__(256) void hap_copy_kernel(uint8_t* __restrict__ chains, const uint8_t* __restrict__ ref_codes,
                                                       const uint8_t* __restrict__ literals, const DevPiece* __restrict__ pieces,
                                                       uint64_t n_pieces) {
  for (uint64_t pi = blockIdx.x; pi < n_pieces; pi += gridDim.x) {
    const DevPiece p = pieces[pi];
    const uint8_t* src = ((p.pad & 1u) ? literals : ref_codes) + p.src;
    uint8_t* dst = chains + p.dst;
    const uint64_t d0 = p.dst, d1 = p.dst + p.len;
    const uint64_t a0 = (d0 + 15) & ~(uint64_t)15, a1 = d1 & ~(uint64_t)15;  // aligned interior [a0, a1)
    if (a0 >= a1) {
      for (uint32_t i = threadIdx.x; i < p.len; i += blockDim.x) dst[i] = src[i];
      continue;
    }
    const uint32_t head = (uint32_t)(a0 - d0), tail = (uint32_t)(d1 - a1);
    if (threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
    if (threadIdx.x < tail) dst[p.len - tail + threadIdx.x] = src[p.len - tail + threadIdx.x];
    const uint64_t nblk = (a1 - a0) / 16;
    for (uint64_t b = threadIdx.x; b < nblk; b += blockDim.x) {
      uint4 w;
      __builtin_memcpy(&w, src + head + b * 16, 16);
      *(uint4*)(dst + head + b * 16) = w;
    }
  }
}

struct DevPatch { uint64_t dst; uint32_t base; uint32_t pad; };
__global__ __launch_bounds__(256) void hap_patch_kernel(uint8_t* __restrict__ chains, const DevPatch* __restrict__ patches, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    chains[patches[i].dst] = (uint8_t)encode_base(patches[i].base);
}

__global__ __launch_bounds__(256) void encode_bytes_kernel(uint8_t* __restrict__ buf, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    buf[i] = (uint8_t)encode_base(buf[i]);
}

// ---- 2-bit copies for the straight-line emit kernel ---------------------------------------------------
// lane = 16 consecutive bases of the chains buffer (T bytes, a multiple of 1024, guard bytes included):
//   fwd2  base a at bits 2 (a % 16) of word a / 16, code & 3;
//   rc2   the reverse complement of the WHOLE buffer, rc2 base j = complement of base T - 1 - j: a reverse read of the
//         fragment [A, A + n) is the forward read at T - A - n of rc2 (complement in code space = xor 2);
//   bad   one bit per 64 bases: some base of the block is not A/C/G/T (reads touching such a block take the generic code).
__global__ __launch_bounds__(256) void pack2_kernel(const uint8_t* __restrict__ chains, uint64_t n16, uint32_t* __restrict__ fwd2,
                                                    uint32_t* __restrict__ rc2, uint16_t* __restrict__ bad) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // n16 is a multiple of 64: whole waves
  if (g >= n16) return;
  const uint4 w = *(const uint4*)(chains + g * 16);
  const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
  uint32_t word = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t v = ws[k] & 0x03030303u;
    word |= ((v | (v >> 6) | (v >> 12) | (v >> 18)) & 0xFFu) << (8 * k);
  }
  const bool any_bad = ((w.x | w.y | w.z | w.w) & 0xFCFCFCFCu) != 0u;
  fwd2[g] = word;
  uint32_t r = __builtin_bitreverse32(word);                       // field order reversed, the two bits of a field swapped
  r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);         // swap them back
  rc2[n16 - 1 - g] = r ^ 0xAAAAAAAAu;                              // A0 <-> T2, C1 <-> G3
  const unsigned long long b = __ballot(any_bad);                  // one bit per 16 bases, 1024 bases per wave
  if ((threadIdx.x & 63u) == 0u) {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) m |= (uint32_t)(((b >> (4 * k)) & 0xFull) != 0ull) << k;
    bad[g >> 6] = (uint16_t)m;
  }
}

// ---- launchers -------------------------------------------------------------------------------------
static uint32_t grid_for(uint64_t items, uint32_t per_block, uint32_t max_blocks) {
  uint64_t g = (items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  return (uint32_t)(g > max_blocks ? max_blocks : g);
}
void launch_ref_scan(const uint8_t* raw, uint64_t n, uint64_t* list, uint32_t cap, uint32_t* count, uint32_t* flags, hipStream_t s) {
  hipLaunchKernelGGL(ref_scan_kernel, dim3(grid_for((n + 15) / 16, 256, 256 * 32)), dim3(256), 0, s, raw, n, list, cap, count, flags);
}
void launch_ref_ingest(const uint8_t* raw, uint8_t* codes, const void* contigs, uint32_t n_contigs, uint64_t n_blocks,
                       uint32_t* flags, hipStream_t s) {
  if (!n_blocks) return;
  hipLaunchKernelGGL(ref_ingest_kernel, dim3(grid_for(n_blocks, 256, 256 * 64)), dim3(256), 0, s, raw, codes,
                     (const DevContig*)contigs, n_contigs, n_blocks, flags);
}
void launch_hap_copy(uint8_t* chains, const uint8_t* ref_codes, const uint8_t* literals, const void* pieces, uint64_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(hap_copy_kernel, dim3(grid_for(n, 1, 256 * 64)), dim3(256), 0, s, chains, ref_codes, literals,
                     (const DevPiece*)pieces, n);
}
void launch_hap_patch(uint8_t* chains, const void* patches, uint64_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(hap_patch_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, chains, (const DevPatch*)patches, n);
}
void launch_pack2(const uint8_t* chains, uint64_t bytes, uint32_t* fwd2, uint32_t* rc2, uint16_t* bad, hipStream_t s) {  // bytes % 1024 == 0
  const uint64_t n16 = bytes / 16;
  if (!n16) return;
  hipLaunchKernelGGL(pack2_kernel, dim3((uint32_t)((n16 + 255) / 256)), dim3(256), 0, s, chains, n16, fwd2, rc2, bad);
}
void launch_encode_bytes(uint8_t* buf, uint64_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(encode_bytes_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, buf, n);
}

}  // namespace sg
