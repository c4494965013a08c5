// sg_deflate.hip -- device side of the block-gzip (BGZF) FASTQ sink (see sg_deflate.h).
//
// One workgroup of 512 lanes turns one 32 KB chunk of FASTQ text into one gzip member: lane = 64
// consecutive input bytes.  The lane -> byte map is aligned to the END of the chunk (a short last
// chunk leaves its leading lanes empty), so that the CRC-32 combine tree uses the same nine "advance by
// 64 * 2^k bytes" operators for every chunk length: leading zero bytes do not change a CRC register
// that starts at zero.
//   gz_hist_kernel    byte histogram of a 1/16 sample of the text (the host builds the Huffman code)
//   gz_size_kernel    sum of code lengths per chunk -> member size (offsets by the u32 -> u64 scan)
//   gz_encode_kernel  prefix (gzip header + block header), literal codes packed LSB-first through LDS
//                     (ds_or on 32-bit words), end-of-block, CRC-32, ISIZE; then one contiguous copy out
// All three are HBM-class passes over the text (4.2 GB per C2 batch); encode adds ~10 integer ops and
// two LDS operations per byte.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sg_deflate.h"

namespace sg {

struct DevDeflate {
  const uint8_t* text;
  uint64_t bytes;
  uint32_t n_chunks;
  const uint32_t* code;       // [257] reversed code | length << 16
  const uint32_t* prefix;     // member prefix words (BSIZE = 0)
  uint32_t prefix_words, prefix_bits;
  const uint32_t* crc_tab;    // [4][256]
  const uint32_t* crc_shift;  // [kGzLevels][32]
  uint32_t crc_init_full, crc_init_last;
  uint32_t* next;             // [2] chunk counter of the encode kernel at [1] (zeroed by the host)
  uint32_t* msize;            // [n_chunks] member bytes
  const uint64_t* moff;       // [n_chunks] member offsets
  uint8_t* out;
};

__global__ __launch_bounds__(256) void gz_hist_kernel(const uint8_t* __restrict__ text, uint64_t bytes,
                                                      unsigned long long* __restrict__ hist) {
  __shared__ uint32_t h[4][256];
  for (uint32_t i = threadIdx.x; i < 1024; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  const uint32_t wv = threadIdx.x >> 6;
  for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 256; i + 16 <= bytes; i += (uint64_t)gridDim.x * 256 * 256) {
    const uint4 w = *(const uint4*)(text + i);  // i is a multiple of 256
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 16; k++) atomicAdd(&h[wv][(ws[k >> 2] >> ((k & 3) * 8)) & 0xFFu], 1u);
  }
  __syncthreads();
  const uint32_t s = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
  if (s) atomicAdd(&hist[threadIdx.x], (unsigned long long)s);
}

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

__global__ __launch_bounds__(kGzThreads) void gz_size_kernel(DevDeflate D) {
  __shared__ uint32_t code[257];
  __shared__ uint32_t wave_tot[kGzThreads / 64];
  for (uint32_t i = threadIdx.x; i < 257; i += kGzThreads) code[i] = D.code[i];
  __syncthreads();
  for (uint32_t c = blockIdx.x; c < D.n_chunks; c += gridDim.x) {
    const uint64_t left = D.bytes - (uint64_t)c * kGzChunk;
    const uint32_t n = left < kGzChunk ? (uint32_t)left : kGzChunk;
    uint32_t w[16];
    const uint32_t first = gz_load_lane(D, c, threadIdx.x, n, w);
    uint32_t bits = 0;
#pragma unroll
    for (uint32_t k = 0; k < kGzLaneBytes; k++)
      if (k >= first) bits += code[(w[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu] >> 16;
    uint32_t total;
    gz_block_scan(bits, wave_tot, &total);
    if (threadIdx.x == 0) D.msize[c] = (D.prefix_bits + total + (code[256] >> 16) + 7u) / 8u + 8u;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kGzThreads) void gz_encode_kernel(DevDeflate D, uint32_t stage_words) {
  extern __shared__ uint32_t gz_smem[];
  uint32_t* stage = gz_smem;                       // [stage_words]
  uint32_t* code = stage + stage_words;            // [257] (+3 pad)
  uint32_t* crc_tab = code + 260;                  // [4][256]
  uint32_t* crc_shift = crc_tab + 1024;            // [kGzLevels][32]
  uint32_t* crcs = crc_shift + kGzLevels * 32;     // [kGzThreads]
  uint32_t* wave_tot = crcs + kGzThreads;          // [8]
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < 257; i += kGzThreads) code[i] = D.code[i];
  for (uint32_t i = lane; i < 1024; i += kGzThreads) crc_tab[i] = D.crc_tab[i];
  for (uint32_t i = lane; i < kGzLevels * 32; i += kGzThreads) crc_shift[i] = D.crc_shift[i];
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
    // ---- bit offsets ----
    uint32_t bits = 0;
#pragma unroll
    for (uint32_t k = 0; k < kGzLaneBytes; k++)
      if (k >= first) bits += code[(w[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu] >> 16;
    uint32_t total;
    const uint32_t excl = gz_block_scan(bits, wave_tot, &total);  // its barrier also orders the zeroing above
    // ---- prefix, BSIZE ----
    if (lane < D.prefix_words) atomicOr(&stage[lane], D.prefix[lane]);
    if (lane == 0) atomicOr(&stage[4], (msize - 1u) & 0xFFFFu);  // bytes 16-17 of the member
    // ---- literal codes ----
    {
      uint32_t pos = D.prefix_bits + excl;
      uint32_t wi = pos >> 5;
      uint32_t nacc = pos & 31u;
      uint64_t acc = 0;
#pragma unroll
      for (uint32_t k = 0; k < kGzLaneBytes; k++) {
        if (k >= first) {
          const uint32_t e = code[(w[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu];
          acc |= (uint64_t)(e & 0xFFFFu) << nacc;
          nacc += e >> 16;
          if (nacc >= 32u) {
            atomicOr(&stage[wi++], (uint32_t)acc);
            acc >>= 32;
            nacc -= 32u;
          }
        }
      }
      if (lane == kGzThreads - 1u) {  // end-of-block after the chunk's last byte
        const uint32_t e = code[256];
        acc |= (uint64_t)(e & 0xFFFFu) << nacc;
        nacc += e >> 16;
      }
      if (nacc) {
        atomicOr(&stage[wi], (uint32_t)acc);
        if (nacc > 32u) atomicOr(&stage[wi + 1u], (uint32_t)(acc >> 32));
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
          uint32_t r = 0;
#pragma unroll 8
          for (uint32_t j = 0; j < 32u; j++) r ^= (0u - ((x >> j) & 1u)) & crc_shift[k * 32u + j];
          crcs[lane] = r ^ y;
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
void launch_gz_hist(const uint8_t* text, uint64_t bytes, unsigned long long* hist, hipStream_t s) {
  uint64_t blocks = (bytes / 256 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(gz_hist_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, text, bytes, hist);
}
uint32_t gz_stage_words(uint32_t prefix_bits) {  // worst case: every literal 15 bits
  return ((prefix_bits + 15u * kGzChunk + 15u + 7u) / 8u + 8u + 3u) / 4u + 4u;
}
void launch_gz_size(const void* d, uint32_t n_chunks, hipStream_t s) {
  if (!n_chunks) return;
  const uint32_t grid = n_chunks < 256u * 16u ? n_chunks : 256u * 16u;
  hipLaunchKernelGGL(gz_size_kernel, dim3(grid), dim3(kGzThreads), 0, s, *(const DevDeflate*)d);
}
void launch_gz_encode(const void* d, uint32_t n_chunks, uint32_t prefix_bits, hipStream_t s) {
  if (!n_chunks) return;
  const uint32_t sw = gz_stage_words(prefix_bits);
  const size_t lds = ((size_t)sw + 260 + 1024 + kGzLevels * 32 + kGzThreads + 8) * 4;
  (void)hipFuncSetAttribute((const void*)gz_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t grid = n_chunks < 512u ? n_chunks : 512u;  // two workgroups per CU fit in LDS
  hipLaunchKernelGGL(gz_encode_kernel, dim3(grid), dim3(kGzThreads), lds, s, *(const DevDeflate*)d, sw);
}

}  // namespace sg
