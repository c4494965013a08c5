// sg_device.h -- device-side data layout shared by the kernels and the host API (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/simuscop_amd.h"

namespace sg {

// DevBatch::totals: 16 words of results, then (byte 128 on) the read-group counters of the two emit kernels:
// [kernel 0 fast / 1 generic][mate][XCD partition], one 128-byte line each
// then [2][16] u64 segment bases of the record offsets (block_base_kernel) and its two arrival counters
constexpr size_t kTotalsSegBase = 128 + 2 * 2 * 8 * 128;
constexpr size_t kTotalsBytes = kTotalsSegBase + 2 * 16 * 8 + 64;

// RNG stream kinds (DESIGN.md "RNG addressing"); c3 = kind | (ctx24 << 8)
// Rounds of the per-base draws (KIND_BASE: two calls per eight sampled bases, 97 % of all calls).  Seven is the smallest
// round count of Philox4x32 its authors report as passing BigCrush (Salmon et al., SC'11; Random123's documented
// minimum); every other stream keeps their default of ten.  Specified in oracle/philox.h alike.
constexpr int kBaseRounds = 7;

enum : uint32_t { KIND_HAP = 1, KIND_GC = 2, KIND_PLAN = 3, KIND_INDEL = 4, KIND_AUX = 5, KIND_BASE = 6 };

// One planned fragment (16 B).  Written by plan_kernel, read by the indel and emit kernels.
struct PairRec {   // 32 bytes per planned fragment: everything indel_kernel needs of the fragment's window but its name base
  uint32_t win;     // window index in the batch
  uint32_t namepos; // fragment start inside its segment: (window.spos + draw) % segment size (Segment.cpp:780)
  uint32_t fl;      // bits 0..30 fragment length after chain-end clipping (0 = slot unused), bit 31 = SE reverse strand
  uint32_t k;       // ordinal of the fragment inside its window
  uint64_t foff;    // offset of the fragment's first base in the chains buffer
  uint64_t pad;
};

// Sequencing-indel event (Profile::getIndelSeq outcome): j | len<<16 | del<<31
__host__ __device__ inline uint32_t ev_pack(uint32_t j, uint32_t len, uint32_t del) { return j | (len << 16) | (del << 31); }

struct DevProfile {
  // Per-base sampling tables (sg_tables.h): each outcome owns exactly the draws the reference's
  // `r <= cdf[k]` scan gives it; substitution rows are ordered identity first, quality rows are alias columns.
  const uint4* sub;            // [mate][kmer_count][bins] rows {D0, D1, D2, j0 | o0<<2 | o1<<4 | o2<<6 | o3<<8}: j = max(j0, #{x > D_i}), k = o_j
  uint32_t sub_mate_rows;      // rows per mate table (0 when mate 2 shares mate 1's table)
  const uint2* alias;          // [16][bins][W] columns {thr, lo | hi << 8}: symbol = u < thr ? lo : hi
  uint32_t lgW;                // W = 2^lgW columns (4..128)
  // straight-line kernel (kmer 3), indexed by its own context ids and natural base codes
  const uint32_t* fast_lds;    // [mate][bins][fast_stride]: [0,192) keep_h - 1 by context slot, then [cdn][W] diagonal alias columns (head form)
  uint32_t fast_mate_words;    // words per mate image (0 when mate 2 shares mate 1's)
  uint32_t fast_stride;        // 192 + 4 W
  const uint4* fast_sub;       // [mate][bins][192] full rows, outcomes as natural codes
  const uint2* fast_alias;     // [cdn][kn][bins][W] full columns
  const uint32_t* ins_row; uint32_t ins_lg;
  const uint32_t* del_row; uint32_t del_lg;
  const uint32_t* isz_row; uint32_t isz_lg;  // isz_row == nullptr -> fixed insert size
  int32_t isz_min, fixed_isz;
  int32_t isz_lo, isz_hi;      // smallest / largest insert size that can be drawn
  // sequencing indels by skipping ahead (indel_kernel): evA / 2^64 = P(insertion at a position), evB / 2^64 = P(insertion or
  // deletion); gap_row[k] = P(no candidate in the next k positions) * 2^64, k = 1 .. L ([0] unused)
  uint64_t evA, evB;
  const uint64_t* gap_row;
  int32_t L, bins, kmer, min_qual;
  uint32_t remap_packed;       // natural index (A0 C1 T2 G3) -> profile base code, 2 bits each
  uint32_t bases_packed;       // profile base code -> ASCII, 8 bits each
  uint32_t kmer_off[8];        // kmer_off[m] = first table index of contexts with m real bases
  uint32_t inv_remap_packed;   // profile base code -> natural index (A0 C1 T2 G3), 2 bits each
};

struct DevBatch {
  const uint8_t* chains;        // all chains, each padded; chain c starts at chain_off[c]
  const uint64_t* chain_off;
  const uint64_t* chain_len;
  // 2-bit copies of the whole chains buffer for the straight-line emit kernel (pack2_kernel): forward, reverse
  // complement (base j = complement of base chains_total - 1 - j), and one "not all A/C/G/T" bit per 64 bases
  const uint8_t* chains2_fwd;
  const uint8_t* chains2_rc;
  const uint16_t* chains_bad;
  uint64_t chains_total;
  const sg_window* windows;
  uint64_t n_windows;
  const uint32_t* seg_size;
  const uint32_t* seg_first_window;
  uint32_t n_segs;
  uint32_t n_slots;             // planned fragments in the batch
  uint32_t batch_id;
  uint32_t win_offset, slot_offset;  // philox address of local window 0 / local slot 0 (sharded batches)
  int32_t paired;
  const uint8_t* prefix;        // "@popu#chr#"
  uint32_t prefix_len;
  uint32_t prefix_w[4];         // first 16 prefix bytes by value (longer prefixes fall back to the global copy)
  uint32_t k0, k1;              // philox key
  uint32_t diag;                // SG_DIAG timing ablations (0 in production; outputs are wrong otherwise)
  uint32_t strict_bases;        // 1: a literal X of the genome is an unknown base (sg_set_strict_bases); 0: the trie's place holder
  // work buffers
  PairRec* pairs;               // [n_slots]
  uint32_t* win_actual;         // [n_windows] fragments actually produced per window
  uint32_t* win_namebase;       // [n_windows] fragments produced by earlier windows of the same segment
  uint32_t* events;             // [2][n_slots][SG_MAX_EVENTS]
  uint32_t* recloc;             // [2][n_slots] exclusive prefix of reclen inside the read's block of 256 (indel_kernel)
  uint64_t* blkbase;            // [2][ceil(n_slots / 256)] offset of a block's first record inside its segment of 2^seg_shift blocks
                                // (sums by indel_kernel, scanned in place by block_base_kernel; the segments' own bases: totals + kTotalsSegBase)
  uint32_t seg_shift;
  uint4* meta;                  // [2][n_slots][3] per-read 48-byte rows for the emit kernels: m0, m1, name text (indel_kernel)
  // (from byte 128 on: the emit kernels' read-group counters, one 128-byte line each, see GroupRuns in sg_kernels.hip)
  uint64_t* totals;             // [0],[1] bytes per mate; [2] fragments produced; [3] flags (1 events, 2 slow queue full); [4] slow-queue counts;
  uint2* slowq;                 // [2][slowq_cap] (slot, item) left to emit_slow_kernel by the fast emit kernel
  uint32_t* slowq_count;        // [2] entries appended per mate (may exceed slowq_cap: overflow)
  uint32_t slowq_cap;
  uint8_t* out[2];
  uint64_t out_cap[2];
};

}  // namespace sg
