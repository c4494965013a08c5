// tools/pmc_calibrate.hip -- calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on the emit kernel's own access
// patterns (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern").  Not part of the product.
//
//   wr8_records   the body stores of emit_fast_kernel: records of 333 bytes back to back; per record the lanes
//                 of a wave write 19 x 8 bytes of "bases" at +27 and 19 x 8 bytes of "qualities" at +181
//                 (unaligned 8-byte stores, 152-byte runs); known bytes = records * 304
//   wr_full       the same records written completely (header, separators, tails as byte stores AFTER the body,
//                 like the per-read pass); known bytes = records * 333
//   rd16_windows  the haplotype window loads: lane loads 16 unaligned bytes at stride 8 (each byte fetched by
//                 two lanes), three runs of 19 lanes per wave at distant places; known unique bytes = reads * 160
// Run under:  rocprofv3 --pmc WRITE_SIZE ... and again with --pmc FETCH_SIZE (separate passes).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr uint32_t REC = 333, HDR = 27, NB = 151;

__global__ __launch_bounds__(1024) void wr8_records(uint8_t* out, uint32_t n_rec, int full) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t waves = gridDim.x * 16u;
  const uint32_t sub = lane / 19u, c = lane % 19u;
  for (uint32_t g = blockIdx.x * 16u + wv; g * 63u < n_rec; g += waves) {   // 63 records per wave group
    for (uint32_t step = 0; step < 21u; step++) {
      const uint32_t r = g * 63u + step * 3u + sub;
      if (sub < 3u && r < n_rec) {
        uint8_t* rec = out + (size_t)r * REC;
        if (8u * c + 8u <= NB) {
          const uint64_t v = 0x4141414141414141ull + c;
          __builtin_memcpy(rec + HDR + 8u * c, &v, 8);
          __builtin_memcpy(rec + HDR + NB + 3u + 8u * c, &v, 8);
        }
      }
    }
    if (full) {  // per-read pass: header, tail bases + separators, tail qualities + newline
      const uint32_t r = g * 63u + lane;
      if (lane < 63u && r < n_rec) {
        uint8_t* rec = out + (size_t)r * REC;
        for (uint32_t i = 0; i < HDR; i++) rec[i] = '@';
        for (uint32_t i = 144; i < NB + 3u; i++) rec[HDR + i] = 'x';
        for (uint32_t i = 144; i < NB + 1u; i++) rec[HDR + NB + 3u + i] = 'y';
      }
    }
  }
}

__global__ __launch_bounds__(1024) void rd16_windows(const uint8_t* in, uint32_t* sink, uint32_t n_reads, size_t span) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t waves = gridDim.x * 16u;
  const uint32_t sub = lane / 19u, c = lane % 19u;
  uint32_t acc = 0;
  for (uint32_t g = blockIdx.x * 16u + wv; g * 63u < n_reads; g += waves) {
    for (uint32_t step = 0; step < 21u; step++) {
      const uint32_t r = g * 63u + step * 3u + sub;
      if (sub < 3u && r < n_reads) {
        // read r starts 10 bases after read r-1 (30x coverage of 151-base reads): its 19 windows overlap by 8
        const uint8_t* p = in + ((size_t)r * 10u) % span + 8u * c + 3u;
        uint4 v;
        __builtin_memcpy(&v, p, 16);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  if (acc == 0x12345u) sink[0] = acc;
}

int main() {
  const uint32_t n = 12867792;  // reads of one C2 pass (both mates)
  uint8_t* out; hipMalloc(&out, (size_t)n * REC + 1024);
  uint8_t* in; const size_t span = (size_t)128 << 20; hipMalloc(&in, span + 4096);
  uint32_t* sink; hipMalloc(&sink, 64);
  hipMemset(out, 0, (size_t)n * REC + 1024);
  hipMemset(in, 1, span + 4096);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(wr8_records, dim3(256), dim3(1024), 0, 0, out, n, 0);
    hipLaunchKernelGGL(wr8_records, dim3(256), dim3(1024), 0, 0, out, n, 1);
    hipLaunchKernelGGL(rd16_windows, dim3(256), dim3(1024), 0, 0, in, sink, n, span);
  }
  hipDeviceSynchronize();
  printf("known bytes: wr8 body %.3f GB, full records %.3f GB, rd16 unique %.3f GB (requested %.3f GB)\n", n * 304e-9, n * 333e-9,
         n * 10e-9, n * 19 * 16e-9);
  return 0;
}
