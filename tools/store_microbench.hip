// tools/store_microbench.hip -- cost of byte-misaligned coalesced stores/loads on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// each wave writes contiguous runs: lane l writes W bytes at base + W*l + mis
template <int W>
__global__ __launch_bounds__(256) void wr(uint8_t* out, size_t per_wave, int iters, int mis) {
  const uint32_t lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  uint8_t* p = out + wave * per_wave + mis + (size_t)lane * W;
  for (int i = 0; i < iters; i++) {
    if (W == 4) { uint32_t v = i * 2654435761u + lane; __builtin_memcpy(p, &v, 4); }
    if (W == 16) { uint4 v = make_uint4(i, lane, i ^ lane, 7); __builtin_memcpy(p, &v, 16); }
    if (W == 2) { uint16_t v = (uint16_t)(i + lane); __builtin_memcpy(p, &v, 2); }
    if (W == 1) { *p = (uint8_t)(i + lane); }
    p += 64 * W;
  }
}
template <int W>
__global__ __launch_bounds__(256) void rd(const uint8_t* in, uint32_t* sink, size_t per_wave, int iters, int mis, int stride) {
  const uint32_t lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint8_t* p = in + wave * per_wave + mis + (size_t)lane * stride;
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    if (W == 16) { uint4 v; __builtin_memcpy(&v, p, 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (W == 4) { uint32_t v; __builtin_memcpy(&v, p, 4); acc += v; }
    p += 64 * stride;
  }
  if (acc == 0x12345) sink[0] = acc;
}

int main() {
  const int blocks = 256 * 8, iters = 512;
  const size_t waves = (size_t)blocks * 4;
  const size_t per_wave = (size_t)iters * 64 * 16 + 256;
  uint8_t* d; hipMalloc(&d, waves * per_wave);
  uint32_t* sink; hipMalloc(&sink, 64);
  hipMemset(d, 1, waves * per_wave);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto time = [&](auto launch, const char* name, double bytes) {
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms %9.1f GB/s\n", name, best, bytes / best / 1e6);
  };
  char nm[128];
  for (int mis = 0; mis < 4; mis++) {
    snprintf(nm, sizeof nm, "store dword  contiguous, misalign %d", mis);
    time([&]() { hipLaunchKernelGGL(wr<4>, dim3(blocks), dim3(256), 0, 0, d, per_wave, iters, mis); }, nm, (double)waves * iters * 256);
  }
  for (int mis : {0, 1, 4, 5}) {
    snprintf(nm, sizeof nm, "store dwordx4 contiguous, misalign %d", mis);
    time([&]() { hipLaunchKernelGGL(wr<16>, dim3(blocks), dim3(256), 0, 0, d, per_wave, iters, mis); }, nm, (double)waves * iters * 1024);
  }
  for (int mis : {0, 1}) {
    snprintf(nm, sizeof nm, "store ushort contiguous, misalign %d", mis);
    time([&]() { hipLaunchKernelGGL(wr<2>, dim3(blocks), dim3(256), 0, 0, d, per_wave, iters, mis); }, nm, (double)waves * iters * 128);
  }
  time([&]() { hipLaunchKernelGGL(wr<1>, dim3(blocks), dim3(256), 0, 0, d, per_wave, iters, 0); }, "store byte contiguous", (double)waves * iters * 64);
  for (int mis : {0, 3}) {
    snprintf(nm, sizeof nm, "load dwordx4 stride 4 (overlap), misalign %d", mis);
    time([&]() { hipLaunchKernelGGL(rd<16>, dim3(blocks), dim3(256), 0, 0, d, sink, per_wave, iters, mis, 4); }, nm, (double)waves * iters * 256);
    snprintf(nm, sizeof nm, "load dwordx4 stride 16, misalign %d", mis);
    time([&]() { hipLaunchKernelGGL(rd<16>, dim3(blocks), dim3(256), 0, 0, d, sink, per_wave, iters, mis, 16); }, nm, (double)waves * iters * 1024);
  }
  return 0;
}
