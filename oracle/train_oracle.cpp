// oracle/train_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the COUNTING half of SimuSCoP's profile training: what
// Profile::processRead (lib/profile/Profile.cpp:228-510) adds to its count matrices for one line of
// `samtools view` text -- the filters (:262-288), the CIGAR walk with its insertion / deletion length counts
// (:294-385), the per-(bin, k-mer, base) substitution counts of either mate (:405-441), the insert-size counts
// (:443-450) and the per-(bin, reference x read base, quality) counts (:452-480).  It is the checker of
// sg_train_count (include/simuscop_amd.h).
//
// PARITY UNPINNED.  The reference reads its input through popen("samtools view ...") (Profile.cpp:1448-1449) and
// neither samtools nor a BAM file exists in this image, so this restatement cannot be run against the reference
// binary; it follows the source line by line instead, and the tests hold it against the profile tables the reads
// were sampled from.  Left out, and why:
//   * Profile::countGC (:512-703), the per-window read-count state machine that also decides whether a read is
//     counted at all (:284-287): it is sequential over the file, and what it feeds (estimateGCParas, :713-834)
//     reads an uninitialised array (:735,739).  Every read that passes the other filters is counted here.
//   * known variants (the VCF of seqToProfile -v): altSequence = refSequence (Genome.cpp:466-475 without SNVs),
//     no known insertions / deletions (:311-322, :345-356 never "found").
//   * reads hanging over the end of their contig: the reference indexes refSeq past its end there (:458 with
//     n = strlen(readSeq)); such a line is skipped (counted in `skipped_overhang`).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "oracle.h"

namespace {

using std::string;

// lib/mydefine/MyDefine.cpp:212-225
string abbrOfChr(string chr) {
  size_t i = chr.find("chrom");
  if (i == string::npos) {
    i = chr.find("chr");
    if (i != string::npos) chr = chr.substr(i + 3, chr.size() - 3);
  } else {
    chr = chr.substr(i + 5, chr.size() - 5);
  }
  return chr;
}

// Segment::getComplementSeq, lib/segment/Segment.cpp:81-103
char complementOf(char c) {
  switch (c) {
    case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
    case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
    case 'N': return 'N'; default: return 'N';
  }
}

struct Trainer {
  string bases;
  int N = 4, kmer = 3, bins = 50, kmerCount = 0;
  std::map<string, int> kmerIndex;   // Profile::initKmers, Profile.cpp:70-124
  std::map<string, string> ref;      // upper-cased contigs (Genome.cpp:466,529)

  void initKmers() {
    kmerCount = 0;
    int p = 1;
    for (int m = 1; m <= kmer; m++) { p *= N; kmerCount += p; }
    std::vector<int> tmp(kmer, 0);
    int k = 0;
    for (int j = kmer - 1; j >= 0; j--) {
      for (int i = j; i < kmer; i++) tmp[i] = 0;
      while (tmp[j] < N) {
        string s(kmer, 'X');
        for (int i = j; i < kmer; i++) s[i] = bases[tmp[i]];
        kmerIndex[s] = k++;
        int n = 1;
        for (int i = kmer - 1; i > j; i--) {
          if (n == 0) break;
          tmp[i] += n;
          if (tmp[i] == N) { tmp[i] = 0; n = 1; } else { n = 0; }
        }
        tmp[j] += n;
      }
    }
  }
  int getKmerIndx(const string& s) const {  // Profile.cpp:220-226 (a path the trie does not hold: -1)
    auto it = kmerIndex.find(s);
    return it == kmerIndex.end() ? -1 : it->second;
  }
  int getIndexOfBase(char c) const {  // lib/mydefine/MyDefine.cpp:228-236
    for (int i = 0; i < N; i++)
      if (bases[i] == c) return i;
    return -1;
  }
  void loadFasta(const char* path) {
    std::ifstream ifs(path);
    string line;
    string* dst = nullptr;
    while (std::getline(ifs, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      if (!line.empty() && line[0] == '>') {
        string name = line.substr(1);
        name = name.substr(0, name.find_first_of(" \t"));
        dst = &ref[abbrOfChr(name)];
        dst->clear();
      } else if (dst) {
        for (char c : line) dst->push_back((char)toupper((unsigned char)c));
      }
    }
  }
};

}  // namespace

extern "C" int orc_train_count(const char* sam_text, uint64_t sam_bytes, const char* fasta_path, const char* bases, int kmer, int bins,
                               uint32_t n_isize, orc_train_counts* out) {
  Trainer T;
  T.bases = bases;
  T.N = (int)T.bases.size();
  T.kmer = kmer;
  T.bins = bins;
  T.initKmers();
  T.loadFasta(fasta_path);
  const int N = T.N, nq = 126 - 33 + 1;  // minBaseQuality 33, maxBaseQuality 126 (Profile::init, :173-174)
  const size_t subs_n = (size_t)T.kmerCount * bins * N;
  memset(out->subs1, 0, subs_n * 8);
  memset(out->subs2, 0, subs_n * 8);
  memset(out->kmers, 0, (size_t)bins * T.kmerCount * 8);
  memset(out->quality, 0, (size_t)N * N * bins * nq * 8);
  memset(out->isize, 0, (size_t)n_isize * 8);
  memset(out->ins_len, 0, sizeof out->ins_len);
  memset(out->del_len, 0, sizeof out->del_len);
  out->lines = out->reads_counted = out->cigar_chars = out->insert_events = out->delete_events = out->isize_overflow = out->skipped_overhang = 0;

  const char* p = sam_text;
  const char* end = sam_text + sam_bytes;
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* le = nl ? nl : end;
    string line(p, le);
    p = nl ? nl + 1 : end;
    if (line.empty()) continue;
    out->lines++;
    // ---- Profile::processRead, Profile.cpp:228-510 ----
    std::vector<string> el;
    {
      size_t a = 0;
      while (el.size() < 20) {
        size_t b = line.find('\t', a);
        el.push_back(line.substr(a, b == string::npos ? string::npos : b - a));
        if (b == string::npos) break;
        a = b + 1;
      }
    }
    if (el.size() < 11) return 1;                       // :246-251 (the reference exits)
    string chr = el[2];
    const long position = atol(el[3].c_str());
    const int mapQuality = atoi(el[4].c_str());
    string cigar = el[5];
    const int tlen = atoi(el[8].c_str());
    string readSeq = el[9], baseQuality = el[10];
    if (position == 0) continue;                        // :262
    if (mapQuality < 15) continue;                      // :266
    chr = abbrOfChr(chr);
    auto rit = T.ref.find(chr);
    if (rit == T.ref.end()) continue;                   // :270-274
    if (readSeq == "*") continue;                       // :276
    // (:281-287 countGC: not restated, see the header)
    const int n_c = (int)cigar.size();
    int sIndx = 0, k = 0;
    long refIndx = 0;
    out->cigar_chars += (uint64_t)n_c;                  // `baseCount += n` with n = strlen(cigar), :296
    bool hard = false;
    for (int i = 0; i < n_c; i++) {
      const char c = cigar[i];
      if (c >= '0' && c <= '9') { k++; continue; }
      if (c == 'H') { out->cigar_chars -= (uint64_t)n_c; hard = true; break; }   // :302-305
      if (c == 'S') sIndx = i + 1;
      else if (c == 'I') {                              // :309-337 (no known insertions: never "found")
        const int len = atoi(cigar.substr(sIndx, i - sIndx).c_str());
        if (len >= 0 && len < 256) out->ins_len[len]++;
        out->insert_events++;
        sIndx = i + 1;
      } else if (c == 'D') {                            // :338-368
        const int len = atoi(cigar.substr(sIndx, i - sIndx).c_str());
        if (len >= 0 && len < 256) out->del_len[len]++;
        out->delete_events++;
        refIndx += len;
        sIndx = i + 1;
      } else if (c == 'M') {
        refIndx += atoi(cigar.substr(sIndx, i - sIndx).c_str());
        sIndx = i + 1;
      } else {
        sIndx = i + 1;
      }
    }
    if (hard) continue;
    if (k != n_c - 1 || n_c == 0 || cigar[n_c - 1] != 'M') continue;   // :386-388: only a single nM reaches the counters
    const string& contig = rit->second;
    const size_t rl = readSeq.size();
    if ((size_t)(position - 1) + rl > contig.size()) { out->skipped_overhang++; continue; }
    string refSeq = contig.substr((size_t)(position - 1), rl);       // :390 (altSeq == refSeq here)
    int isRead1 = 1;
    if (tlen < 0) {                                     // :394-403
      auto rc = [](string& s) { std::string t(s.rbegin(), s.rend()); for (char& ch : t) ch = complementOf(ch); s = t; };
      rc(refSeq);
      rc(readSeq);
      baseQuality = string(baseQuality.rbegin(), baseQuality.rend());
      isRead1 = 0;
    }
    const int n = (int)refSeq.size();
    string seq(kmer - 1, 'X');                          // :409-420 (altSeq == refSeq: seq = the reference bases)
    seq += refSeq;
    uint64_t* subs = isRead1 ? out->subs1 : out->subs2;
    for (int i = 0; i < n; i++) {                       // :421-441
      const int baseIndx = T.getIndexOfBase(readSeq[i]);
      const int binIndx = i * bins / n;
      if (baseIndx != -1) {
        const int kmerIndx = T.getKmerIndx(seq.substr(i, kmer));
        if (kmerIndx == -1) continue;
        subs[((size_t)kmerIndx * bins + binIndx) * N + baseIndx]++;
        out->kmers[(size_t)binIndx * T.kmerCount + kmerIndx]++;
      }
    }
    if (tlen > 0) {                                     // :445-450 (the reference grows its row; here the row is n_isize long)
      if ((uint32_t)tlen < n_isize) out->isize[tlen]++;
      else out->isize_overflow++;
    }
    if (baseQuality.size() == readSeq.size()) {         // :455-480
      const int m = (int)readSeq.size();
      for (int i = 0; i < m; i++) {
        const int r = T.getIndexOfBase(refSeq[i]);
        const int binIndx = i * bins / m;
        const int b = T.getIndexOfBase(readSeq[i]);
        if (r == -1 || b == -1) continue;
        const int indx = r * N + b;
        const int j = (int)(signed char)baseQuality[i];
        if (j >= 33 && j <= 126) out->quality[((size_t)indx * bins + binIndx) * nq + (j - 33)]++;
      }
    }
    out->reads_counted++;                               // :482
  }
  return 0;
}
