// tools/valu_microbench.hip -- issue cost of the integer instructions the emit / indel kernels are made of,
// measured at the occupancy those kernels run at (not part of the product).
// Findings are recorded in DESIGN.md section 5 ("cost model").
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_microbench tools/valu_microbench.hip && tools/valu_microbench
//
// Every test is an unrolled stream of ONE instruction kind on independent registers (8 chains per lane),
// launched as one workgroup per CU with 1, 4 or 8 waves per SIMD.  Reported: shader cycles per
// wave-instruction per SIMD = launch time x clock / (instructions issued on one SIMD), with the clock
// taken from s_memtime / s_memrealtime inside the same launch.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHAINS 8
#define UNROLL 16  // instructions per chain per loop trip

enum Op { OP_ADD, OP_BITOP3, OP_MAD_U64_U32, OP_MUL_LO, OP_MUL_HI, OP_MUL_U24, OP_MAD_U24, OP_BFE, OP_CNDMASK, OP_PERM,
          OP_LSHL_OR, OP_ADD_LSHL, OP_CMP_ADDC, OP_ALIGNBIT, OP_LDS_B32, OP_LDS_B64, OP_LDS_B128, OP_PHILOX,
          OP_AND, OP_XOR, OP_LSHLREV, OP_LSHRREV, OP_LSHR_CONST, OP_SUB, OP_MAX, OP_ADD3, OP_AND_OR, OP_OR3, OP_CMP_E32, OP_CMP_E64,
          OP_CNDMASK_E64, OP_CNDMASK_SET, OP_MOV, OP_FMA, OP_ADD_SDWA, OP_XAD, OP_LSHL_ADD, OP_SAD, OP_MED3, OP_MIX_ADD_BFE, OP_MIX_3ADD_BFE, OP_ADD_2CHAINS, OP_AND_LSHR_SUB,
          OP_PK_SUB_U16, OP_PK_ADD_U16, OP_PK_MIN_U16, OP_PK_MAX_I16, OP_PK_ASHR_I16, OP_PK_LSHR_B16, OP_PK_LSHL_B16, OP_PK_MUL_LO_U16, OP_PK_MAD_U16,
          OP_SUB_SDWA2, OP_AND_SDWA, OP_SUB_U16, OP_ADD_DPP_ROW_SHR, OP_MOV_DPP_ROW_SHR, OP_ADD_DPP_QUAD, OP_DS_BPERMUTE, OP_DS_SWIZZLE,
          OP_LDS_U16_D16, OP_LDS_U16_D16_PAIR, OP_LDS_U8, OP_LDS_READ2_B32, OP_DOT4_U8, OP_MIX_PK_PERM_BITOP, OP_COUNT };
static const char* kNames[OP_COUNT] = {"v_add_u32", "v_bitop3_b32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24",
                                       "v_mad_u32_u24", "v_bfe_u32", "v_cndmask_b32 (vcc never written)", "v_perm_b32", "v_lshl_or_b32", "v_add_lshl_u32",
                                       "v_cmp_gt_u32 + v_addc", "v_alignbit_b32", "ds_read_b32 (random bank)", "ds_read_b64 (random bank)",
                                       "ds_read_b128 (random bank)", "philox4x32-10 call (20 mad_u64 + 20 bitop3)",
                                       "v_and_b32", "v_xor_b32", "v_lshlrev_b32 (vgpr amount)", "v_lshrrev_b32 (vgpr amount)", "v_lshrrev_b32 (const amount)",
                                       "v_sub_u32", "v_max_u32", "v_add3_u32", "v_and_or_b32", "v_or3_b32", "v_cmp_gt_u32_e32 (vcc)",
                                       "v_cmp_gt_u32_e64 (sgpr pair)", "v_cndmask_b32_e64 (sgpr pair mask)", "v_cndmask_b32 (vcc written once before the loop)",
                                       "v_mov_b32", "v_fma_f32", "v_add_u32_sdwa (WORD_1 operand)", "v_xad_u32", "v_lshl_add_u32", "v_sad_u32", "v_med3_u32",
                                       "mix: v_add_u32, v_bfe_u32 alternating", "mix: 3 x v_add_u32, 1 x v_bfe_u32", "v_add_u32, 2 dependent chains per wave",
                                       "mix: v_and_b32, v_lshrrev_b32, v_sub_u32 (full-rate only)",
                                       "v_pk_sub_u16", "v_pk_add_u16", "v_pk_min_u16", "v_pk_max_i16", "v_pk_ashrrev_i16 (vgpr amount)", "v_pk_lshrrev_b16 (vgpr amount)",
                                       "v_pk_lshlrev_b16 (vgpr amount)", "v_pk_mul_lo_u16", "v_pk_mad_u16",
                                       "v_sub_u32_sdwa (WORD_0 - WORD_0 of two registers)", "v_and_b32_sdwa (WORD_1 operand)", "v_sub_u16",
                                       "v_add_u32_dpp row_shr:1", "v_mov_b32_dpp row_shr:1", "v_add_u32_dpp quad_perm:[1,0,3,2]", "ds_bpermute_b32", "ds_swizzle_b32 (swap 1)",
                                       "ds_read_u16_d16 (random bank)", "ds_read_u16_d16 + ds_read_u16_d16_hi into one register (random banks; per pair)",
                                       "ds_read_u8 (random bank)", "ds_read2_b32 (two random banks)", "v_dot4_u32_u8",
                                       "mix: v_pk_sub_u16, v_perm_b32, v_bitop3_b32 alternating"};

template <int OP>
__global__ __launch_bounds__(1024) void bench(uint32_t* out, uint64_t* clk, int trips, uint32_t seed) {
  __shared__ uint32_t lds[8192];
  for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 2654435761u + seed;
  __syncthreads();
  uint32_t v[CHAINS], w[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) { v[c] = threadIdx.x * 2654435761u + c * 40503u + seed; w[c] = v[c] ^ 0x9E3779B9u; }
  if (OP == OP_CNDMASK_SET) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(v[0]), "v"(w[1]) : "vcc");
  if (OP == OP_CNDMASK_E64) asm volatile("v_cmp_gt_u32 s[22:23], %0, %1" : : "v"(v[0]), "v"(w[1]) : "s22", "s23");
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < trips; t++) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        if (OP == OP_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_MAD_U64_U32) {
          uint64_t p;
          asm volatile("v_mad_u64_u32 %0, s[0:1], %1, %2, 0" : "=v"(p) : "v"(v[c]), "v"(w[c]) : "s0", "s1");
          v[c] = (uint32_t)p; w[c] = (uint32_t)(p >> 32);
        }
        if (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(v[c]));
        if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_ADD_LSHL) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_CMP_ADDC) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(v[c]) : "v"(w[c]) : "vcc");
        if (OP == OP_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_LSHLREV) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_LSHRREV) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_LSHR_CONST) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(v[c]));
        if (OP == OP_SUB) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_MAX) asm volatile("v_max_u32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_OR3) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_CMP_E32) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(v[c]), "v"(w[c]) : "vcc");
        if (OP == OP_CMP_E64) asm volatile("v_cmp_gt_u32 s[20:21], %0, %1" : : "v"(v[c]), "v"(w[c]) : "s20", "s21");
        if (OP == OP_CNDMASK_E64) asm volatile("v_cndmask_b32 %0, %0, %1, s[22:23]" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_CNDMASK_SET) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v[c]) : "v"(w[c]));
        if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_ADD_SDWA) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_XAD) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_SAD) asm volatile("v_sad_u32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_MED3) asm volatile("v_med3_u32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_MIX_ADD_BFE) { if (c & 1) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(v[c])); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c])); }
        if (OP == OP_MIX_3ADD_BFE) { if ((c & 3) == 3) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(v[c])); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c])); }
        if (OP == OP_ADD_2CHAINS) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c & 1]) : "v"(w[c]));
        if (OP == OP_AND_LSHR_SUB) {
          if (c % 3 == 0) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
          else if (c % 3 == 1) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(v[c]));
          else asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        }

        if (OP == OP_PK_SUB_U16) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_PK_MIN_U16) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_PK_MAX_I16) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_PK_ASHR_I16) asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_PK_LSHR_B16) asm volatile("v_pk_lshrrev_b16 %0, %1, %0" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_PK_LSHL_B16) asm volatile("v_pk_lshlrev_b16 %0, %1, %0" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_PK_MUL_LO_U16) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_PK_MAD_U16) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_SUB_SDWA2) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_AND_SDWA) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_SUB_U16) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_ADD_DPP_ROW_SHR) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_MOV_DPP_ROW_SHR) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_ADD_DPP_QUAD) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_DS_BPERMUTE) { uint32_t r; asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r) : "v"(w[c]), "v"(v[c])); w[c] ^= r; v[c] += 0x9E3779B9u; }
        if (OP == OP_DS_SWIZZLE) { uint32_t r; asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,1)" : "=v"(r) : "v"(v[c])); w[c] ^= r; v[c] += 0x9E3779B9u; }
        if (OP == OP_LDS_U16_D16) {
          uint32_t a = (v[c] & 16383u) << 1, r = w[c];
          asm volatile("ds_read_u16_d16 %0, %1" : "+v"(r) : "v"(a));
          w[c] ^= r;
          v[c] += 0x9E3779B9u;
        }
        if (OP == OP_LDS_U16_D16_PAIR) {
          uint32_t a = (v[c] & 16383u) << 1, b = (w[c] & 16383u) << 1, r = 0;
          asm volatile("ds_read_u16_d16 %0, %1\n ds_read_u16_d16_hi %0, %2" : "+v"(r) : "v"(a), "v"(b));
          w[c] ^= r;
          v[c] += 0x9E3779B9u;
        }
        if (OP == OP_LDS_U8) {
          uint32_t a = (v[c] & 32767u), r;
          asm volatile("ds_read_u8 %0, %1" : "=v"(r) : "v"(a));
          w[c] ^= r;
          v[c] += 0x9E3779B9u;
        }
        if (OP == OP_LDS_READ2_B32) {
          uint32_t a = (v[c] & 4095u) << 2;
          uint64_t r;
          asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:37" : "=v"(r) : "v"(a));
          w[c] ^= (uint32_t)r ^ (uint32_t)(r >> 32);
          v[c] += 0x9E3779B9u;
        }
        if (OP == OP_DOT4_U8) asm volatile("v_dot4_u32_u8 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == OP_MIX_PK_PERM_BITOP) {
          if (c % 3 == 0) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
          else if (c % 3 == 1) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(w[c]));
          else asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(v[c]) : "v"(w[c]));
        }
        if (OP == OP_LDS_B32) {
          uint32_t a = (v[c] & 8191u) << 2, r;
          asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(a));
          w[c] ^= r;  // consumed one trip later
          v[c] += 0x9E3779B9u;
        }
        if (OP == OP_LDS_B64) {
          uint32_t a = (v[c] & 4095u) << 3;
          uint64_t r;
          asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(a));
          w[c] ^= (uint32_t)r ^ (uint32_t)(r >> 32);
          v[c] += 0x9E3779B9u;
        }
        if (OP == OP_LDS_B128) {
          uint32_t a = (v[c] & 2047u) << 4;
          uint4 r;
          asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(a));
          w[c] ^= r.x ^ r.y ^ r.z ^ r.w;
          v[c] += 0x9E3779B9u;
        }
      }
      if (OP == OP_PHILOX && u < 2) {  // two calls per trip on the first four chains as counter
        uint32_t c0 = v[0], c1 = v[1], c2 = v[2], c3 = v[3], k0 = seed, k1 = seed ^ 0x5555u;
#pragma unroll
        for (int r = 0; r < 10; r++) {
          const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
          c0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
          c1 = (uint32_t)p1;
          c2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
          c3 = (uint32_t)p0;
          k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        v[0] = c0; v[1] = c1; v[2] = c2; v[3] = c3;
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc ^= v[c] ^ w[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP>
static void run(uint32_t* d_out, uint64_t* d_clk, int cus) {
  const int trips = OP == OP_PHILOX ? 2000 : (((OP >= OP_LDS_B32 && OP <= OP_PHILOX) || (OP >= OP_DS_BPERMUTE && OP <= OP_LDS_READ2_B32)) ? 300 : 1000);
  for (int wps : {1, 4, 8}) {  // waves per SIMD; 8 needs two 1024-thread workgroups per CU
    const int threads = wps == 1 ? 256 : 1024, blocks = cus * (wps == 8 ? 2 : 1);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(a);
      hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_clk, trips, 12345u + rep);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    std::vector<uint64_t> clk(2 * blocks);
    hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, mhz = 0;
    for (int i = 0; i < blocks; i++) { cyc += (double)clk[2 * i]; mhz += (double)clk[2 * i] / (double)clk[2 * i + 1] * 100.0; }
    cyc /= blocks; mhz /= blocks;
    // Instructions issued on one SIMD.  Time = the whole launch minus an empty launch of the same shape (the waves of a
    // SIMD do not finish together -- the oldest wave wins the arbitration -- so one wave's own stamps undercount).
    float base = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(a);
      hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, d_clk, 0, 1u);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (ms < base) base = ms;
    }
    (void)cyc;
    const double per_wave = OP == OP_PHILOX ? (double)trips * 2 : (double)trips * UNROLL * CHAINS * (OP == OP_CMP_ADDC ? 2 : 1) /* a d16 pair counts once */;
    const double insts = per_wave * wps;
    const double ns = (double)(best - base) * 1e6 / insts;
    printf("%-50s %d waves/SIMD: %7.3f ns = %6.2f cycles per wave-%s per SIMD  (in-kernel clock %.0f MHz, launch %.3f ms)\n", kNames[OP], wps,
           ns, ns * mhz * 1e-3, OP == OP_PHILOX ? "call" : "instruction", mhz, best);
    hipEventDestroy(a); hipEventDestroy(b);
  }
}

int main(int argc, char** argv) {
  const bool only_new = argc > 1 && argv[1][0] == 'n';  // `valu_microbench new`: the rows added in round 4 only
  int dev = 0, cus = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  uint32_t* d_out;
  uint64_t* d_clk;
  hipMalloc(&d_out, (size_t)cus * 2 * 1024 * 4);
  hipMalloc(&d_clk, (size_t)cus * 2 * 16);
  printf("CUs %d; one workgroup per CU (two at 8 waves/SIMD); %d independent chains per lane\n", cus, CHAINS);
  if (!only_new) {
  run<OP_ADD>(d_out, d_clk, cus);
  run<OP_BITOP3>(d_out, d_clk, cus);
  run<OP_MAD_U64_U32>(d_out, d_clk, cus);
  run<OP_MUL_LO>(d_out, d_clk, cus);
  run<OP_MUL_HI>(d_out, d_clk, cus);
  run<OP_MUL_U24>(d_out, d_clk, cus);
  run<OP_MAD_U24>(d_out, d_clk, cus);
  run<OP_BFE>(d_out, d_clk, cus);
  run<OP_CNDMASK>(d_out, d_clk, cus);
  run<OP_PERM>(d_out, d_clk, cus);
  run<OP_LSHL_OR>(d_out, d_clk, cus);
  run<OP_ADD_LSHL>(d_out, d_clk, cus);
  run<OP_CMP_ADDC>(d_out, d_clk, cus);
  run<OP_ALIGNBIT>(d_out, d_clk, cus);
  run<OP_LDS_B32>(d_out, d_clk, cus);
  run<OP_LDS_B64>(d_out, d_clk, cus);
  run<OP_LDS_B128>(d_out, d_clk, cus);
  run<OP_PHILOX>(d_out, d_clk, cus);
  run<OP_AND>(d_out, d_clk, cus);
  run<OP_XOR>(d_out, d_clk, cus);
  run<OP_LSHLREV>(d_out, d_clk, cus);
  run<OP_LSHRREV>(d_out, d_clk, cus);
  run<OP_LSHR_CONST>(d_out, d_clk, cus);
  run<OP_SUB>(d_out, d_clk, cus);
  run<OP_MAX>(d_out, d_clk, cus);
  run<OP_ADD3>(d_out, d_clk, cus);
  run<OP_AND_OR>(d_out, d_clk, cus);
  run<OP_OR3>(d_out, d_clk, cus);
  run<OP_CMP_E32>(d_out, d_clk, cus);
  run<OP_CMP_E64>(d_out, d_clk, cus);
  run<OP_CNDMASK_E64>(d_out, d_clk, cus);
  run<OP_CNDMASK_SET>(d_out, d_clk, cus);
  run<OP_MOV>(d_out, d_clk, cus);
  run<OP_FMA>(d_out, d_clk, cus);
  run<OP_ADD_SDWA>(d_out, d_clk, cus);
  run<OP_XAD>(d_out, d_clk, cus);
  run<OP_LSHL_ADD>(d_out, d_clk, cus);
  run<OP_SAD>(d_out, d_clk, cus);
  run<OP_MED3>(d_out, d_clk, cus);
  run<OP_MIX_ADD_BFE>(d_out, d_clk, cus);
  run<OP_MIX_3ADD_BFE>(d_out, d_clk, cus);
  run<OP_ADD_2CHAINS>(d_out, d_clk, cus);
  run<OP_AND_LSHR_SUB>(d_out, d_clk, cus);
  }
  run<OP_PK_SUB_U16>(d_out, d_clk, cus);
  run<OP_PK_ADD_U16>(d_out, d_clk, cus);
  run<OP_PK_MIN_U16>(d_out, d_clk, cus);
  run<OP_PK_MAX_I16>(d_out, d_clk, cus);
  run<OP_PK_ASHR_I16>(d_out, d_clk, cus);
  run<OP_PK_LSHR_B16>(d_out, d_clk, cus);
  run<OP_PK_LSHL_B16>(d_out, d_clk, cus);
  run<OP_PK_MUL_LO_U16>(d_out, d_clk, cus);
  run<OP_PK_MAD_U16>(d_out, d_clk, cus);
  run<OP_SUB_SDWA2>(d_out, d_clk, cus);
  run<OP_AND_SDWA>(d_out, d_clk, cus);
  run<OP_SUB_U16>(d_out, d_clk, cus);
  run<OP_ADD_DPP_ROW_SHR>(d_out, d_clk, cus);
  run<OP_MOV_DPP_ROW_SHR>(d_out, d_clk, cus);
  run<OP_ADD_DPP_QUAD>(d_out, d_clk, cus);
  run<OP_DS_BPERMUTE>(d_out, d_clk, cus);
  run<OP_DS_SWIZZLE>(d_out, d_clk, cus);
  run<OP_LDS_U16_D16>(d_out, d_clk, cus);
  run<OP_LDS_U16_D16_PAIR>(d_out, d_clk, cus);
  run<OP_LDS_U8>(d_out, d_clk, cus);
  run<OP_LDS_READ2_B32>(d_out, d_clk, cus);
  run<OP_DOT4_U8>(d_out, d_clk, cus);
  run<OP_MIX_PK_PERM_BITOP>(d_out, d_clk, cus);
  return 0;
}
