// sg_deflate.h -- host side of the block-gzip (BGZF) FASTQ sink, SURVEY section 8(f)-2.
//
// The reference writes plain FASTQ text (lib/seqwriter/SeqWriter.cpp:41-54).  With --gzip the text of a
// batch is cut into 32 KB chunks and every chunk becomes one independent gzip member in the BGZF
// layout (a gzip member with a 'BC' extra field holding its size; readable by zcat/gzip/zlib and
// seekable by bgzip-aware tools): one final dynamic-Huffman DEFLATE block of literals only (RFC 1951
// 3.2.7), all members of a batch sharing one code that is built here from a byte histogram.  The
// device kernels (sg_deflate.hip) only look codes up, pack bits and compute CRC-32s.
#pragma once
#include <cstdint>
#include <vector>

namespace sg {

constexpr uint32_t kGzChunk = 32768;     // input bytes per member
constexpr uint32_t kGzThreads = 512;     // lanes per member, 64 input bytes each
constexpr uint32_t kGzLaneBytes = kGzChunk / kGzThreads;
constexpr uint32_t kGzMemberHeader = 18; // gzip header with the BGZF extra field
constexpr uint32_t kGzLevels = 9;        // log2(kGzThreads): CRC combine tree

struct DeflatePlan {
  uint8_t lit_len[257];          // code lengths of the literals and of end-of-block (256), 1..15
  uint32_t lit_code[257];        // canonical codes, bit-reversed for LSB-first packing
  std::vector<uint32_t> prefix;  // the first prefix_bits of every member (header, BSIZE = 0, block header), LSB-first words
  uint32_t prefix_bits = 0;
  // CRC-32 machinery (reflected polynomial 0xEDB88320)
  uint32_t crc_table[4][256];             // slicing-by-4
  uint32_t crc_shift[kGzLevels][32];      // column j of "advance by 64 * 2^k zero bytes"
  uint32_t crc_init_full = 0;             // state reached from 0xFFFFFFFF over kGzChunk zero bytes
};

// counts[b]: (sampled) occurrences of byte b.  Every byte value gets a code, so the plan is valid for
// any text whatever the sample missed.
void deflate_build_plan(const uint64_t counts[256], DeflatePlan* plan);
// state reached from `state` over `n` zero bytes (crc32(data) = ~(advance(~0, n) ^ raw(data)))
uint32_t crc_advance(const DeflatePlan& plan, uint32_t state, uint64_t n);
// bytes of one member holding `data_bits` bits of literal codes
inline uint32_t member_bytes(const DeflatePlan& p, uint64_t data_bits) {
  return (uint32_t)((p.prefix_bits + data_bits + p.lit_len[256] + 7) / 8 + 8);
}

}  // namespace sg
