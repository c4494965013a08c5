// sg_train.hip -- profile training, counting half: what Profile::processRead
// (lib/profile/Profile.cpp:228-510) adds to its count matrices, for lines of `samtools view` text.
//
//   train_parse_kernel   lane = line: the eleven mandatory fields, the filters of :262-288, the CIGAR walk of :294-388
//                        (insertion / deletion length counts; only a single nM goes on), the read's place on the
//                        resident reference codes -> one descriptor per line
//   train_count_kernel   wave = read, lane = base: k-mer context of the reference bases in read orientation
//                        (Profile::getKmerIndx order, :70-124, :220-226), bin = i * bins / n, then the three counters
//                        of :421-441 and :455-480 as 64-bit atomics; the insert size (:445-450) by lane 0
// Integer work throughout; results are exact counts.  Not restated: Profile::countGC (:512-703) -- sequential over the
// file -- and known variants (the VCF side of seqToProfile); see oracle/train_oracle.cpp.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sg_train.h"

namespace sg {

__device__ __forceinline__ bool is_digit(char c) { return c >= '0' && c <= '9'; }

// atoi / atol on a field [p, e): optional sign, digits (fields hold no white space)
__device__ long long field_int(const char* p, const char* e) {
  bool neg = false;
  if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
  long long v = 0;
  while (p < e && is_digit(*p)) { v = v * 10 + (*p - '0'); p++; }
  return neg ? -v : v;
}

// abbrOfChr (lib/mydefine/MyDefine.cpp:212-225): what follows the first "chrom", else the first "chr", else the name
__device__ void abbr_of_chr(const char*& p, const char* e) {
  for (int pass = 0; pass < 2; pass++) {
    const int n = pass == 0 ? 5 : 3;
    for (const char* q = p; q + n <= e; q++) {
      bool hit = q[0] == 'c' && q[1] == 'h' && q[2] == 'r';
      if (pass == 0) hit = hit && q[3] == 'o' && q[4] == 'm';
      if (hit) { p = q + n; return; }
    }
  }
}

__global__ __launch_bounds__(256) void train_parse_kernel(TrainJob J) {
  const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= J.n_lines) return;
  TrainRead R;
  R.len = 0; R.flags = 0; R.tlen = 0; R.seq_off = R.qual_off = R.ref_off = 0;
  const char* p = J.text + J.line_off[li];
  const char* le = J.text + J.line_off[li + 1] - 1;  // the line break
  // ---- the first eleven fields (:244-251) ----
  const char* fb[11];
  const char* fe[11];
  int nf = 0;
  const char* a = p;
  for (const char* q = p; q <= le && nf < 11; q++) {
    if (q == le || *q == '\t') {
      fb[nf] = a; fe[nf] = q; nf++;
      a = q + 1;
    }
  }
  auto done = [&]() { J.reads[li] = R; };
  if (nf < 11) { atomicOr(J.flags, 1u); done(); return; }
  const long long position = field_int(fb[3], fe[3]);
  const int mapq = (int)field_int(fb[4], fe[4]);
  const int tlen = (int)field_int(fb[8], fe[8]);
  if (position == 0) { done(); return; }                 // :262
  if (mapq < 15) { done(); return; }                     // :266
  const char* cb = fb[2];
  abbr_of_chr(cb, fe[2]);
  int contig = -1;
  for (uint32_t c = 0; c < J.n_contigs; c++) {           // :270-274
    const char* key = J.keys + (size_t)c * kTrainKeyBytes;
    uint32_t i = 0;
    while (cb + i < fe[2] && key[i] != 0 && key[i] == cb[i]) i++;
    if (cb + i == fe[2] && key[i] == 0) { contig = (int)c; break; }
  }
  if (contig < 0) { done(); return; }
  const uint32_t slen = (uint32_t)(fe[9] - fb[9]);
  if (slen == 1 && fb[9][0] == '*') { done(); return; }  // :276
  // ---- CIGAR (:294-388) ----
  const char* cg = fb[5];
  const int n_c = (int)(fe[5] - fb[5]);
  atomicAdd(J.scalars + kTrainCigarChars, (unsigned long long)n_c);   // `baseCount += n`, n = strlen(cigar) (:296)
  int sIndx = 0, k = 0;
  bool hard = false;
  for (int i = 0; i < n_c; i++) {
    const char c = cg[i];
    if (is_digit(c)) { k++; continue; }
    if (c == 'H') { atomicAdd(J.scalars + kTrainCigarChars, (unsigned long long)(-(long long)n_c)); hard = true; break; }   // :302-305
    if (c == 'S') sIndx = i + 1;
    else if (c == 'I' || c == 'D') {                     // :309-368 (no known variants: every event counts)
      const long long len = field_int(cg + sIndx, cg + i);
      unsigned long long* row = J.scalars + (c == 'I' ? kTrainInsLen : kTrainDelLen);
      if (len >= 0 && len < 256) atomicAdd(row + len, 1ull);
      atomicAdd(J.scalars + (c == 'I' ? kTrainInsEvents : kTrainDelEvents), 1ull);
      sIndx = i + 1;
    } else sIndx = i + 1;
  }
  if (hard) { done(); return; }
  if (n_c == 0 || k != n_c - 1 || cg[n_c - 1] != 'M') { done(); return; }   // :386-388
  const TrainContig C = J.contigs[contig];
  if ((uint64_t)(position - 1) + slen > C.length) { atomicAdd(J.scalars + kTrainOverhang, 1ull); done(); return; }
  R.len = slen;
  R.tlen = tlen;
  R.seq_off = (uint64_t)(fb[9] - J.text);
  R.qual_off = (uint64_t)(fb[10] - J.text);
  R.ref_off = C.code_off + (uint64_t)(position - 1);
  R.flags = 1u | (tlen < 0 ? 2u : 0u) | ((uint32_t)(fe[10] - fb[10]) == slen ? 4u : 0u);
  done();
}

__global__ __launch_bounds__(256) void train_count_kernel(TrainJob J) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t K = J.kmer, bins = J.bins, kc = J.kmer_count;
  for (uint64_t li = wave; li < J.n_lines; li += n_waves) {
    const TrainRead R = J.reads[li];
    if (!(R.flags & 1u)) continue;
    const bool rev = (R.flags & 2u) != 0u;               // tlen < 0: everything reverse-complemented, mate 2 (:394-403)
    const uint32_t n = R.len;
    unsigned long long* subs = rev ? J.subs2 : J.subs1;
    const char* seq = J.text + R.seq_off;
    const char* qual = J.text + R.qual_off;
    const uint8_t* ref = J.ref_codes + R.ref_off;
    for (uint32_t i = lane; i < n; i += 64u) {
      const uint32_t j = rev ? n - 1u - i : i;           // index of read position i in the line's strings / the reference window
      // read base -> index in `bases` (getIndexOfBase, MyDefine.cpp:228-236), complemented first on the reverse strand
      // (Segment::getComplementSeq keeps the case, so lower-case bases stay unknown)
      char c = seq[j];
      if (rev) c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'N';
      const int b = c == J.bases[0] ? 0 : c == J.bases[1] ? 1 : c == J.bases[2] ? 2 : c == J.bases[3] ? 3 : -1;
      // reference bases in read orientation: natural code (A0 C1 T2 G3; complement = code ^ 2) -> index in `bases`
      auto ref_idx = [&](uint32_t pos_in_read) -> int {
        const uint32_t w = rev ? n - 1u - pos_in_read : pos_in_read;
        uint32_t code = ref[w];
        if (code > 3u) return -1;
        if (rev) code ^= 2u;
        return (int)((J.remap >> (2u * code)) & 3u);
      };
      const uint32_t bin = (uint32_t)(((uint64_t)i * bins) / n);
      const int r0 = ref_idx(i);
      if (b >= 0) {                                      // :421-441
        const uint32_t m = i + 1u < K ? i + 1u : K;       // real bases of the context, the rest is 'X'
        int kidx = (int)J.kmer_off[m];
        int v = 0;
        for (uint32_t t = 0; t < m; t++) {                // oldest base in the highest digit
          const int x = ref_idx(i + 1u - m + t);
          if (x < 0) { kidx = -1; break; }
          v = v * 4 + x;
        }
        if (kidx >= 0) {
          kidx += v;
          atomicAdd(subs + ((size_t)kidx * bins + bin) * 4u + (uint32_t)b, 1ull);
          atomicAdd(J.kmers + (size_t)bin * kc + (uint32_t)kidx, 1ull);
        }
      }
      if ((R.flags & 4u) && r0 >= 0 && b >= 0) {         // :455-480
        const int q = (int)(signed char)qual[j];
        if (q >= 33 && q <= 126) atomicAdd(J.quality + (((size_t)(r0 * 4 + b)) * bins + bin) * 94u + (uint32_t)(q - 33), 1ull);
      }
    }
    if (lane == 0u) {
      if (R.tlen > 0) {                                  // :445-450
        if ((uint32_t)R.tlen < J.n_isize) atomicAdd(J.isize + R.tlen, 1ull);
        else atomicAdd(J.scalars + kTrainIsizeOverflow, 1ull);
      }
      atomicAdd(J.scalars + kTrainReads, 1ull);          // :482
    }
  }
}

void launch_train(const TrainJob& J, hipStream_t s) {
  if (!J.n_lines) return;
  hipLaunchKernelGGL(train_parse_kernel, dim3((uint32_t)((J.n_lines + 255) / 256)), dim3(256), 0, s, J);
  const uint64_t waves = J.n_lines;
  const uint32_t grid = (uint32_t)(waves * 64 / 256 + 1 < 256u * 32u ? waves * 64 / 256 + 1 : 256u * 32u);
  hipLaunchKernelGGL(train_count_kernel, dim3(grid), dim3(256), 0, s, J);
}

}  // namespace sg
