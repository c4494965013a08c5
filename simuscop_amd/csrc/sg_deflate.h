// sg_deflate.h -- host side of the block-gzip (BGZF) FASTQ sink, SURVEY section 8(f)-2.
//
// The reference writes plain FASTQ text (lib/seqwriter/SeqWriter.cpp:41-54).  With --gzip the text of a
// batch is cut into 32 KB chunks and every chunk becomes one independent gzip member in the BGZF
// layout (a gzip member with a 'BC' extra field holding its size; readable by zcat/gzip/zlib and
// seekable by bgzip-aware tools): one final dynamic-Huffman DEFLATE block (RFC 1951 3.2.7) of literals and
// LZ77 matches.  The matches come from two sources the device can find without a sequential pass over the
// member: the FIRST occurrence (at an even position) of the position's 8-byte gram in the member (reads of one GC window overlap
// on the template, so most of a read's bases have been written before), and the previous byte (runs, which
// is what FASTQ quality lines are made of).  All members of a batch share one pair of Huffman codes, built
// here from a token histogram of 512 members spread over the text.  The device kernels (sg_deflate.hip) find the matches,
// look codes up, pack bits and compute CRC-32s.
#pragma once
#include <cstdint>
#include <vector>

namespace sg {

constexpr uint32_t kGzChunk = 32768;     // input bytes per member
constexpr uint32_t kGzThreads = 512;     // lanes per member, 64 input bytes each
constexpr uint32_t kGzLaneBytes = kGzChunk / kGzThreads;
constexpr uint32_t kGzMemberHeader = 18; // gzip header with the BGZF extra field
constexpr uint32_t kGzLevels = 9;        // log2(kGzThreads): CRC combine tree

constexpr int kGzLitSyms = 286;          // literal / length alphabet: 256 literals, end-of-block, 29 length symbols
constexpr int kGzDistSyms = 30;          // distance alphabet
constexpr uint32_t kGzGram = 8;          // bytes hashed per position
constexpr uint32_t kGzHashBits = 13;     // first-occurrence table: 8192 entries in LDS
constexpr uint32_t kGzMinGramMatch = 12; // shortest copy taken from the table where literals are cheap (quality lines); base lines and names: 8
constexpr uint32_t kGzMinRun = 5;        // shortest match taken at distance 1
constexpr uint32_t kGzMaxMatch = 64;     // a lane's match never leaves its 64 bytes ...
constexpr uint32_t kGzMaxToken = 256;    // ... but matches of up to four neighbouring lanes that continue one copy leave as one token (RFC limit: 258)
constexpr uint32_t kGzLenTokens = 260;   // entries of the length-token table (index = match length)
constexpr uint32_t kGzLaneMatches = 6;   // matches per lane; what follows them in the lane is literals
constexpr uint32_t kGzSamples = 512;     // members sampled for the token histogram (16 MB of text), spread evenly

// length symbol (index into the 29 length codes, i.e. symbol - 257) of a match length, RFC 1951 3.2.5
int deflate_length_symbol(uint32_t len, uint32_t* extra_bits, uint32_t* extra_value);
// distance symbol of a distance 1..32768
int deflate_distance_symbol(uint32_t dist, uint32_t* extra_bits, uint32_t* extra_value);

struct DeflatePlan {
  uint8_t lit_len[kGzLitSyms];     // code lengths of the literals, of end-of-block (256) and of the length symbols, 1..15
  uint32_t lit_code[kGzLitSyms];   // canonical codes, bit-reversed for LSB-first packing
  uint8_t dist_len[kGzDistSyms];   // the distance code, likewise
  uint32_t dist_code[kGzDistSyms];
  // the length half of a match token of length L (3..256): length code and its extra bits, LSB-first in the low 24
  // bits, the bit count in the top 8
  uint32_t len_token[kGzLenTokens];
  std::vector<uint32_t> prefix;    // the first prefix_bits of every member (header, BSIZE = 0, block header), LSB-first words
  uint32_t prefix_bits = 0;
  // CRC-32 machinery (reflected polynomial 0xEDB88320)
  uint32_t crc_table[4][256];             // slicing-by-4
  uint32_t crc_shift[kGzLevels][8][16];   // "advance by 64 * 2^k zero bytes" applied to nibble i holding value v
  uint32_t crc_init_full = 0;             // state reached from 0xFFFFFFFF over kGzChunk zero bytes
};

// lit_counts[s] / dist_counts[s]: (sampled) occurrences of literal / length symbol s (256 = end-of-block: the number
// of sampled members) and of distance symbol s.  Every symbol gets a code, so the plan is valid for any token stream
// whatever the sample missed.
void deflate_build_plan(const uint64_t lit_counts[kGzLitSyms], const uint64_t dist_counts[kGzDistSyms], DeflatePlan* plan);
// state reached from `state` over `n` zero bytes (crc32(data) = ~(advance(~0, n) ^ raw(data)))
uint32_t crc_advance(const DeflatePlan& plan, uint32_t state, uint64_t n);
// bytes of one member holding `data_bits` bits of token codes
inline uint32_t member_bytes(const DeflatePlan& p, uint64_t data_bits) {
  return (uint32_t)((p.prefix_bits + data_bits + p.lit_len[256] + 7) / 8 + 8);
}

}  // namespace sg
