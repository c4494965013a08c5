// sg_deflate.cpp -- see sg_deflate.h: Huffman code construction, the DEFLATE dynamic-block header
// (RFC 1951 section 3.2.7), the BGZF member prefix and the CRC-32 tables for the block-gzip sink.
#include "sg_deflate.h"

#include <algorithm>
#include <cstring>
#include <queue>

namespace sg {

namespace {

// Code lengths (0 for unused symbols) of a complete prefix code with no length above `maxbits`.
void limited_lengths(const uint64_t* freq, int n, int maxbits, uint8_t* len) {
  std::memset(len, 0, (size_t)n);
  std::vector<int> used;
  for (int i = 0; i < n; i++)
    if (freq[i]) used.push_back(i);
  if (used.empty()) return;
  if (used.size() == 1) {  // a complete code needs two leaves: pair the symbol with a neighbour
    len[used[0]] = 1;
    len[used[0] == 0 ? 1 : used[0] - 1] = 1;
    return;
  }
  // Huffman tree: nodes 0..n-1 are leaves, parents appended
  struct Node { uint64_t w; int id; };
  auto cmp = [](const Node& a, const Node& b) { return a.w > b.w || (a.w == b.w && a.id > b.id); };
  std::priority_queue<Node, std::vector<Node>, decltype(cmp)> heap(cmp);
  std::vector<int> parent((size_t)n + used.size(), -1);
  for (int i : used) heap.push(Node{freq[i], i});
  int next = n;
  while (heap.size() > 1) {
    Node a = heap.top(); heap.pop();
    Node b = heap.top(); heap.pop();
    parent[a.id] = parent[b.id] = next;
    heap.push(Node{a.w + b.w, next});
    next++;
  }
  for (int i : used) {
    int d = 0;
    for (int v = i; parent[v] >= 0; v = parent[v]) d++;
    len[i] = (uint8_t)std::min(d, maxbits);
  }
  // Kraft sum in units of 2^-maxbits: lengthen the deepest codes that still can grow until the code
  // fits, then hand any slack back by shortening the most frequent codes of maximal length.
  const uint64_t full = 1ull << maxbits;
  auto kraft = [&]() { uint64_t k = 0; for (int i : used) k += full >> len[i]; return k; };
  uint64_t k = kraft();
  while (k > full) {
    int best = -1;
    for (int i : used)
      if (len[i] < maxbits && (best < 0 || len[i] > len[best] || (len[i] == len[best] && freq[i] < freq[best]))) best = i;
    k -= full >> (len[best] + 1);
    len[best]++;
  }
  while (k < full) {  // only after the loop above ran: symbols at maxbits exist
    int best = -1;
    for (int i : used)
      if (len[i] == maxbits && (best < 0 || freq[i] > freq[best])) best = i;
    if (best < 0) break;
    len[best]--;
    k += 1;
  }
}

uint32_t reverse_bits(uint32_t v, int n) {
  uint32_t r = 0;
  for (int i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
  return r;
}

// canonical codes (RFC 1951 3.2.2), returned bit-reversed
void canonical_codes(const uint8_t* len, int n, uint32_t* code) {
  uint32_t bl_count[16] = {0}, next_code[16] = {0};
  for (int i = 0; i < n; i++) bl_count[len[i]]++;
  bl_count[0] = 0;
  uint32_t c = 0;
  for (int b = 1; b < 16; b++) { c = (c + bl_count[b - 1]) << 1; next_code[b] = c; }
  for (int i = 0; i < n; i++) code[i] = len[i] ? reverse_bits(next_code[len[i]]++, len[i]) : 0;
}

struct BitWriter {
  std::vector<uint32_t> w;
  uint32_t bits = 0;
  void put(uint32_t v, int n) {  // n <= 24, LSB first
    for (int i = 0; i < n; i++) {
      if ((bits & 31u) == 0) w.push_back(0);
      w.back() |= ((v >> i) & 1u) << (bits & 31u);
      bits++;
    }
  }
};

}  // namespace

int deflate_length_symbol(uint32_t len, uint32_t* extra_bits, uint32_t* extra_value) {
  static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t ebits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  int i = 28;
  while (i > 0 && base[i] > len) i--;
  *extra_bits = ebits[i];
  *extra_value = len - base[i];
  return i;
}

int deflate_distance_symbol(uint32_t dist, uint32_t* extra_bits, uint32_t* extra_value) {
  const uint32_t d = dist - 1u;
  if (d < 4u) { *extra_bits = 0; *extra_value = 0; return (int)d; }
  uint32_t lg = 31;
  while (!(d >> lg)) lg--;
  *extra_bits = lg - 1u;
  *extra_value = d & ((1u << (lg - 1u)) - 1u);
  return (int)(2u * lg + ((d >> (lg - 1u)) & 1u));
}

uint32_t crc_advance(const DeflatePlan& plan, uint32_t state, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) state = (state >> 8) ^ plan.crc_table[0][state & 0xFFu];
  return state;
}

void deflate_build_plan(const uint64_t lit_counts[kGzLitSyms], const uint64_t dist_counts[kGzDistSyms], DeflatePlan* plan) {
  // ---- literal / length code: every byte value, end-of-block and every length symbol present ----
  uint64_t freq[kGzLitSyms];
  for (int i = 0; i < kGzLitSyms; i++) freq[i] = lit_counts[i] + 1;  // [256]: one end-of-block per sampled member
  limited_lengths(freq, kGzLitSyms, 15, plan->lit_len);
  canonical_codes(plan->lit_len, kGzLitSyms, plan->lit_code);
  for (uint32_t L = 0; L < kGzLenTokens; L++) {
    plan->len_token[L] = 0;
    if (L < 3 || L > 258) continue;
    uint32_t eb, ev;
    const int sym = 257 + deflate_length_symbol(L, &eb, &ev);
    const uint32_t nb = plan->lit_len[sym];
    plan->len_token[L] = (plan->lit_code[sym] | (ev << nb)) | ((nb + eb) << 24);
  }
  // ---- distance code: all thirty symbols present ----
  uint64_t dfreq[kGzDistSyms];
  for (int i = 0; i < kGzDistSyms; i++) dfreq[i] = dist_counts[i] + 1;
  limited_lengths(dfreq, kGzDistSyms, 15, plan->dist_len);
  canonical_codes(plan->dist_len, kGzDistSyms, plan->dist_code);

  // ---- code-length sequence: 286 literal/length lengths, then 30 distance lengths (one run-length stream) ----
  std::vector<uint8_t> seq(plan->lit_len, plan->lit_len + kGzLitSyms);
  seq.insert(seq.end(), plan->dist_len, plan->dist_len + kGzDistSyms);
  struct Tok { uint8_t sym, extra, extra_bits; };
  std::vector<Tok> toks;
  for (size_t i = 0; i < seq.size();) {
    size_t j = i;
    while (j < seq.size() && seq[j] == seq[i]) j++;
    size_t run = j - i;
    if (seq[i] == 0) {
      while (run >= 11) { size_t r = std::min<size_t>(run, 138); toks.push_back(Tok{18, (uint8_t)(r - 11), 7}); run -= r; }
      if (run >= 3) { toks.push_back(Tok{17, (uint8_t)(run - 3), 3}); run = 0; }
      while (run--) toks.push_back(Tok{0, 0, 0});
    } else {
      toks.push_back(Tok{seq[i], 0, 0});
      run--;
      while (run >= 3) { size_t r = std::min<size_t>(run, 6); toks.push_back(Tok{16, (uint8_t)(r - 3), 2}); run -= r; }
      while (run--) toks.push_back(Tok{seq[i], 0, 0});
    }
    i = j;
  }
  uint64_t cl_freq[19] = {0};
  for (const Tok& t : toks) cl_freq[t.sym]++;
  uint8_t cl_len[19];
  uint32_t cl_code[19];
  limited_lengths(cl_freq, 19, 7, cl_len);
  canonical_codes(cl_len, 19, cl_code);
  static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int hclen = 19;
  while (hclen > 4 && cl_len[order[hclen - 1]] == 0) hclen--;

  // ---- member prefix: gzip header with the BGZF extra field, then the block header ----
  BitWriter bw;
  static const uint8_t head[kGzMemberHeader] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0, 0};  // BSIZE filled per member
  for (uint8_t b : head) bw.put(b, 8);
  bw.put(1, 1);            // BFINAL
  bw.put(2, 2);            // BTYPE = dynamic Huffman
  bw.put(kGzLitSyms - 257, 5);  // HLIT: 286 literal/length codes
  bw.put(kGzDistSyms - 1, 5);   // HDIST: 30 distance codes
  bw.put((uint32_t)(hclen - 4), 4);
  for (int i = 0; i < hclen; i++) bw.put(cl_len[order[i]], 3);
  for (const Tok& t : toks) {
    bw.put(cl_code[t.sym], cl_len[t.sym]);
    if (t.extra_bits) bw.put(t.extra, t.extra_bits);
  }
  plan->prefix = bw.w;
  plan->prefix_bits = bw.bits;

  // ---- CRC-32 ----
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    plan->crc_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; i++)
    for (int t = 1; t < 4; t++)
      plan->crc_table[t][i] = (plan->crc_table[t - 1][i] >> 8) ^ plan->crc_table[0][plan->crc_table[t - 1][i] & 0xFFu];
  for (uint32_t k = 0; k < kGzLevels; k++) {
    uint32_t col[32];  // the operator is linear over GF(2): its value on a nibble is the xor of its columns
    for (int j = 0; j < 32; j++) col[j] = crc_advance(*plan, 1u << j, (uint64_t)kGzLaneBytes << k);
    for (int i = 0; i < 8; i++)
      for (uint32_t v = 0; v < 16; v++) {
        uint32_t r = 0;
        for (int b = 0; b < 4; b++)
          if ((v >> b) & 1u) r ^= col[4 * i + b];
        plan->crc_shift[k][i][v] = r;
      }
  }
  plan->crc_init_full = crc_advance(*plan, 0xFFFFFFFFu, kGzChunk);
}

}  // namespace sg
