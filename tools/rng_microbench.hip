// tools/rng_microbench.hip -- price counter-based RNG variants on gfx950 (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

// the production form (sg_kernels.hip): v_mad_u64_u32 products, three-input xor in one v_bitop3_b32
__device__ __forceinline__ void philox_prod(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t o[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
    c1 = lo1;
    c2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
    c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

template <int ROUNDS, bool MAD64>
__device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t o[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    uint32_t hi0, lo0, hi1, lo1;
    if (MAD64) {
      uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
      hi0 = p0 >> 32; lo0 = (uint32_t)p0; hi1 = p1 >> 32; lo1 = (uint32_t)p1;
    } else {
      hi0 = __umulhi(M0, c0); lo0 = M0 * c0; hi1 = __umulhi(M1, c2); lo1 = M1 * c2;
    }
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
template <int ROUNDS>
__device__ __forceinline__ void threefry4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t o[4]) {
  const int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
  uint32_t ks[5] = {k0, k1, 0, 0, 0x1BD11BDA ^ k0 ^ k1};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1], x2 = c2 + ks[2], x3 = c3 + ks[3];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    if (r % 2 == 0) { x0 += x1; x1 = rotl(x1, R[r % 8][0]) ^ x0; x2 += x3; x3 = rotl(x3, R[r % 8][1]) ^ x2; }
    else { x0 += x3; x3 = rotl(x3, R[r % 8][0]) ^ x0; x2 += x1; x1 = rotl(x1, R[r % 8][1]) ^ x2; }
    if (r % 4 == 3) { int s = r / 4 + 1; x0 += ks[s % 5]; x1 += ks[(s + 1) % 5]; x2 += ks[(s + 2) % 5]; x3 += ks[(s + 3) % 5] + s; }
  }
  o[0] = x0; o[1] = x1; o[2] = x2; o[3] = x3;
}

template <int V>
__global__ __launch_bounds__(256) void bench(uint32_t* out, int iters, uint32_t k0, uint32_t k1) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    uint32_t o[4];
    if (V == 0) philox<10, false>(t, i, 0, 6, k0, k1, o);
    if (V == 1) philox<10, true>(t, i, 0, 6, k0, k1, o);
    if (V == 2) philox<7, true>(t, i, 0, 6, k0, k1, o);
    if (V == 3) threefry4x32<20>(t, i, 0, 6, k0, k1, o);
    if (V == 4) threefry4x32<12>(t, i, 0, 6, k0, k1, o);
    if (V == 5) philox_prod(t, i, 0, 6, k0, k1, o);
    acc ^= o[0] + o[1] + o[2] + o[3];
  }
  out[t] = acc;
}

int main() {
  const int blocks = 256 * 8, iters = 2000;
  uint32_t* d; hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[] = {"philox4x32-10 mul_lo+mul_hi", "philox4x32-10 mad_u64_u32", "philox4x32-7 mad_u64", "threefry4x32-20", "threefry4x32-12", "philox4x32-10 mad_u64 + bitop3"};
  for (int v = 0; v < 6; v++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(a);
      if (v == 0) hipLaunchKernelGGL(bench<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u, 2u);
      if (v == 1) hipLaunchKernelGGL(bench<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u, 2u);
      if (v == 2) hipLaunchKernelGGL(bench<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u, 2u);
      if (v == 3) hipLaunchKernelGGL(bench<3>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u, 2u);
      if (v == 4) hipLaunchKernelGGL(bench<4>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u, 2u);
      if (v == 5) hipLaunchKernelGGL(bench<5>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u, 2u);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      double calls = (double)blocks * 256 * iters;
      if (rep) printf("%-30s %8.3f ms  %7.2f Gcalls/s  (%.1f G u32/s)\n", names[v], ms, calls / ms / 1e6, 4 * calls / ms / 1e6);
    }
  }
  return 0;
}
