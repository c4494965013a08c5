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
// binary; it follows the source line by line instead.  What a reference run CAN pin is pinned: the unmodified binary
// loads the profile files written from these counts (Profile::load, :934-1238) and simulates from them exactly as
// oracle(mt) does (tests/test_trained_profile_vs_reference.py), and the tests hold the counts against the profile
// tables the reads were sampled from.
//
// orc_train restates the whole of Profile::train (:1442-1484): processRead with Profile::countGC (:512-703) as the
// sequential state machine it is, the known variants of the VCF (lib/vcfparser/vcfparser.cpp:26-106,
// Genome::generateChrSequence Genome.cpp:452-531), exome targets (Genome::loadTargets / divideTargets, Genome.cpp:238-299,
// 684-739), estimateGCParas (:713-834), normParas(false) (:836-900) and saveResults (:1240-1365).  Where the reference's
// behaviour is undefined, this file and the product make the same, stated choice:
//   * reads hanging over the end of their contig: the reference indexes refSeq past its end (:458 with n =
//     strlen(readSeq)); the line is skipped after its CIGAR walk (`skipped_overhang`).  Reads that START behind the end
//     (std::string::substr throws in the reference) or lie on an empty contig never reach countGC.
//   * estimateGCParas thins its samples with `int *curCount = new int[bins]`, never initialised (:735, read at :739):
//     zero-initialised here.  `gcs[i]*bins` reaches `bins` itself for a window of G/C only (:720,738: one element past
//     the arrays): the arrays have bins + 1 cells.  `counts[i]/expectCount` divides by zero with fewer than 50 windows
//     (:723-726): the step is 1 then.  The median of no windows (:1473) is 0.
//     With no GC percent holding more than 20 windows the tails are extrapolated from gcMeans[-1] (:815-820): all means
//     stay 0 then.
//   * normParas(false) reads iSizeDist past its row when five times the most frequent insert size exceeds the largest
//     one seen (:884-889): columns past the row count as 0.
//   * SNVs outside their contig (written past the string, Genome.cpp:471-474) are skipped.
//   * an alternative allele that is no letter of ACGTN never equals a read base here (the product compares base codes).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
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


namespace {
// ---- the pieces of the reference that Profile::train touches, restated ----
struct Target { long spos, epos; };
struct KnownIndel { long pos; int len; };

struct TrainState : Trainer {
  std::vector<string> chromosomes;                   // index order (Genome::loadRefSeq, Genome.cpp:217-236)
  std::map<string, std::vector<Target>> inTargets;   // Genome::loadTargets + divideTargets
  struct SNV { long pos; char alt; bool homo; };
  std::map<string, std::vector<SNV>> snvs;
  std::map<string, std::vector<KnownIndel>> inserts, dels;
  string curChr = "nodefined", refSequence, altSequence;
  bool count_gc = true;
  // Profile's training state
  std::vector<double> subs1, subs2, kmersDist, quality, iSizeDist{std::vector<double>(10, 0.0)}, insFreqs{0.0}, delFreqs{0.0};
  double insertRate = 0, delRate = 0, baseCount = 0;
  std::vector<double> gcs, readCounts;
  uint64_t lines = 0, reads_counted = 0, skipped_overhang = 0, gc_rejected = 0, gc_windows = 0;
  uint64_t maxCount = 300000000;   // Profile::processRead's `maxCount` (:236): training stops at so many counted reads (twice
  bool stopped = false;            // as many with targets, :497-507)
  // countGC's statics (Profile.cpp:514-525)
  string preChr = "";
  long refLen = 0, rightPos = 0, leftPos = -1;
  unsigned int winSize = 1000;   // Segment::getFragmentSize()
  double GC = -1;
  int rc = 0, targetIndx = -1;

  void loadFastaOrdered(const char* path) {
    loadFasta(path);
    std::ifstream ifs(path);
    string line;
    while (std::getline(ifs, line))
      if (!line.empty() && line[0] == '>') {
        string name = line.substr(1);
        if (!name.empty() && name.back() == '\r') name.pop_back();
        name = abbrOfChr(name.substr(0, name.find_first_of(" \t")));
        if (std::find(chromosomes.begin(), chromosomes.end(), name) == chromosomes.end()) chromosomes.push_back(name);
      }
  }
  long getChromLen(const string& chr) const {        // Genome.cpp:384-397
    if (std::find(chromosomes.begin(), chromosomes.end(), chr) == chromosomes.end()) return 0;
    return (long)ref.at(chr).size();
  }
  // lib/vcfparser/vcfparser.cpp:26-106
  int parseVcf(const char* path) {
    FILE* fp = fopen(path, "r");
    if (!fp) return 1;
    char buf[20000];
    int wrong = 0;
    while (fgets(buf, sizeof buf, fp)) {
      if (buf[0] == '#') continue;
      std::vector<string> el;
      {
        string l(buf);
        size_t a = 0;
        while (el.size() < 10) {     // split() into at most ten fields, the rest of the line stays in the last one
          size_t b = l.find('\t', a);
          if (el.size() == 9 || b == string::npos) { el.push_back(l.substr(a)); break; }
          el.push_back(l.substr(a, b - a));
          a = b + 1;
        }
      }
      if (el.size() < 10) { if (++wrong > 10) { fclose(fp); return 2; } continue; }
      const string& info = el[7];
      size_t indx = info.find("DP=");
      if (indx != string::npos) {
        size_t indx1 = info.find(";", indx);
        int depth = atoi(info.substr(indx + 3, indx1 - indx - 3).c_str());
        if (depth < 10) continue;
      }
      if ((float)atof(el[5].c_str()) < 20) continue;
      string chr = abbrOfChr(el[0]);
      long pos = atol(el[1].c_str());
      string gt = el[9].substr(0, el[9].find(':'));
      // (`1/1` is filed as heterozygous and everything else as homozygous, vcfparser.cpp:81-86: kept -- and so is what
      // fgets leaves: a sample column that is the genotype alone ends in the line break, "1/1\n" is not "1/1")
      const bool homo = gt != "1/1";
      if (el[3].size() > 1) dels[chr].push_back(KnownIndel{pos + 1, (int)el[3].size() - 1});
      else if (el[4].size() > 1) inserts[chr].push_back(KnownIndel{pos, (int)el[4].size() - 1});
      else snvs[chr].push_back(SNV{pos, el[4].empty() ? '\0' : el[4][0], homo});
    }
    fclose(fp);
    return 0;
  }
  // Genome::loadTargets, Genome.cpp:238-299; divideTargets, :684-739
  int loadTargets(const char* path) {
    std::ifstream ifs(path);
    if (!ifs.is_open()) return 1;
    std::map<string, std::vector<Target>> targets;
    string line;
    while (std::getline(ifs, line)) {
      std::vector<string> f;
      size_t a = 0;
      for (;;) { size_t b = line.find('\t', a); f.push_back(line.substr(a, b == string::npos ? string::npos : b - a)); if (b == string::npos) break; a = b + 1; }
      if (f.size() < 3) return 2;
      string chr = abbrOfChr(f[0]);
      long chrLen = getChromLen(chr);
      if (chrLen <= 0) continue;
      Target t;
      t.spos = std::max((long)1, atol(f[1].c_str()) - 50 + 1);
      long tmp = atol(f[2].c_str()) <= 0 ? chrLen - (-atol(f[2].c_str())) % chrLen : atol(f[2].c_str());
      t.epos = std::min(chrLen, tmp + 50);
      targets[chr].push_back(t);
    }
    const unsigned int targetMaxSize = 1000;   // Segment::fragSize
    for (auto& kv : targets)
      for (const Target& target : kv.second) {
        long spos = target.spos;
        long tsize = target.epos - target.spos + 1;
        int k = (int)(tsize / targetMaxSize);
        for (int i = 0; i < k; i++) {
          Target nt;
          nt.spos = spos;
          nt.epos = i == k - 1 ? target.epos : spos + targetMaxSize - 1;
          spos = nt.epos + 1;
          inTargets[kv.first].push_back(nt);
        }
        if (spos <= target.epos) inTargets[kv.first].push_back(Target{spos, target.epos});
      }
    return 0;
  }
  // Genome::generateChrSequence, Genome.cpp:452-531
  void generateChrSequence(const string& chr) {
    curChr = chr;
    refSequence = ref.at(chr);
    altSequence = refSequence;
    auto it = snvs.find(chr);
    if (it != snvs.end())
      for (const SNV& v : it->second) {
        if (v.pos < 1 || (size_t)v.pos > altSequence.size()) continue;
        altSequence[v.pos - 1] = v.alt;
        if (v.homo) refSequence[v.pos - 1] = v.alt;
      }
    for (char& c : refSequence) c = (char)toupper((unsigned char)c);
    for (char& c : altSequence) c = (char)toupper((unsigned char)c);
  }
  string subRef(const string& chr, long start, long length) { if (curChr != chr) generateChrSequence(chr); return refSequence.substr((size_t)start, (size_t)length); }
  string subAlt(const string& chr, long start, long length) { if (curChr != chr) generateChrSequence(chr); return altSequence.substr((size_t)start, (size_t)length); }
  // lib/mydefine/MyDefine.cpp:306-331
  static double calculateGCContent(const string& s) {
    if (s.empty()) return 0;
    int gc = 0, nc = 0;
    for (char c : s) { if (c == 'G' || c == 'C') gc++; else if (c == 'N') nc++; }
    if (nc > 0) return -1;
    return 1.0 * gc / (double)(s.size() - nc);
  }
  void pushWindow(bool wxs) {
    if (!wxs) { gcs.push_back(GC); readCounts.push_back(rc); }
    else {
      int targetSize = (int)(rightPos - leftPos + 1);
      rc = (int)(winSize * (unsigned)rc / (unsigned)targetSize);
      gcs.push_back(GC); readCounts.push_back(rc);
    }
  }
  // Profile::countGC, Profile.cpp:512-703
  int countGC(const string& chr, long position) {
    const bool wxs = !inTargets.empty();
    position -= 1;
    if (chr == "X" || chr == "Y" || chr == "M") return 0;
    bool haveSeq = false;
    string refSeq;
    if (preChr == chr) {
      if (refLen == 0) return 0;
      if (position < leftPos) return 0;
      if (position >= leftPos && position <= rightPos) { rc++; return 1; }
      if (GC > 0 && rc > 0) pushWindow(wxs);
      gc_windows++;
      if (!wxs) {
        rightPos += winSize;
        while (rightPos < position) rightPos += winSize;
        rightPos = std::min(rightPos, refLen - 1);
        leftPos = rightPos - winSize + 1;
        refSeq = subRef(chr, leftPos, winSize); haveSeq = true;
        rc = 1;
      } else {
        std::vector<Target>& targets = inTargets[chr];
        targetIndx++;
        for (; targetIndx < (int)targets.size(); targetIndx++)
          if (targets[targetIndx].epos - 1 >= position) break;
        if (targetIndx < (int)targets.size()) {
          rightPos = targets[targetIndx].epos - 1;
          leftPos = targets[targetIndx].spos;
          refSeq = subRef(chr, leftPos, rightPos - leftPos + 1); haveSeq = true;
          rc = leftPos <= position ? 1 : 0;
        } else { rc = 0; leftPos = refLen; rightPos = refLen; }
      }
      GC = haveSeq ? calculateGCContent(refSeq) : -1;
      return rc;
    }
    if (preChr != "" && GC > 0 && rc > 0) pushWindow(wxs);
    preChr = chr;
    refLen = getChromLen(chr);
    gc_windows++;
    if (refLen == 0) { rc = 0; leftPos = -1; rightPos = -1; }
    else {
      if ((long)winSize > refLen) winSize = (unsigned)refLen;
      rightPos = -1;
      if (!wxs) {
        rightPos += winSize;
        while (rightPos < position) rightPos += winSize;
        rightPos = std::min(rightPos, refLen - 1);
        leftPos = rightPos - winSize + 1;
        refSeq = subRef(chr, leftPos, winSize); haveSeq = true;
        rc = 1;
      } else {
        std::vector<Target>& targets = inTargets[chr];
        for (targetIndx = 0; targetIndx < (int)targets.size(); targetIndx++)
          if (targets[targetIndx].epos - 1 >= position) break;
        if (targetIndx < (int)targets.size()) {
          rightPos = targets[targetIndx].epos - 1;
          leftPos = targets[targetIndx].spos;
          refSeq = subRef(chr, leftPos, rightPos - leftPos + 1); haveSeq = true;
          rc = leftPos <= position ? 1 : 0;
        } else { rc = 0; leftPos = refLen; rightPos = refLen; }
      }
    }
    GC = haveSeq ? calculateGCContent(refSeq) : -1;
    return rc;
  }

  void init() {
    N = (int)bases.size();
    initKmers();
    subs1.assign((size_t)kmerCount * bins * N, 0.0);
    subs2 = subs1;
    kmersDist.assign((size_t)bins * kmerCount, 0.0);
    quality.assign((size_t)N * N * bins * 94, 0.0);
  }

  // Profile::processRead, Profile.cpp:228-510; returns 1 when a line has fewer than eleven fields (the reference exits)
  int processRead(const string& line) {
    if (line.empty()) return 0;
    lines++;
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
    if (el.size() < 11) return 1;
    string chr = el[2];
    const long position = atol(el[3].c_str());
    const int mapQuality = atoi(el[4].c_str());
    string cigar = el[5];
    const int tlen = atoi(el[8].c_str());
    string readSeq = el[9], baseQuality = el[10];
    if (position == 0) return 0;
    if (mapQuality < 15) return 0;
    chr = abbrOfChr(chr);
    if (std::find(chromosomes.begin(), chromosomes.end(), chr) == chromosomes.end()) return 0;
    if (readSeq == "*") return 0;
    const string& contig = ref.at(chr);
    if (count_gc) {
      if (chr == "X" || chr == "Y" || chr == "M") { gc_rejected++; return 0; }
      if (position - 1 < 0 || (size_t)(position - 1) >= contig.size()) { skipped_overhang++; return 0; }   // (see the header)
      if (countGC(chr, position) == 0) { gc_rejected++; return 0; }   // :281-285
    }
    const std::vector<KnownIndel>* insertsOfChr = inserts.count(chr) ? &inserts[chr] : nullptr;
    const std::vector<KnownIndel>* delsOfChr = dels.count(chr) ? &dels[chr] : nullptr;
    const int n_c = (int)cigar.size();
    int sIndx = 0, k = 0;
    long refIndx = 0;
    baseCount += n_c;                                   // :294
    for (int i = 0; i < n_c; i++) {
      const char c = cigar[i];
      if (c >= '0' && c <= '9') { k++; continue; }
      if (c == 'H') { baseCount -= n_c; return 0; }     // :300-303
      if (c == 'S') sIndx = i + 1;
      else if (c == 'I') {                              // :307-336
        const int insertLen = atoi(cigar.substr(sIndx, i - sIndx).c_str());
        const long pos = position + refIndx - 1;
        bool found = false;
        if (insertsOfChr)
          for (const KnownIndel& e : *insertsOfChr) {
            if (e.pos > pos) break;
            if (e.pos == pos && insertLen == e.len) { found = true; break; }
          }
        if (!found) {
          if (insertLen >= 0) {
            if (insertLen > (int)insFreqs.size() - 1) insFreqs.resize((size_t)insertLen + 1, 0.0);
            insFreqs[insertLen] += 1;
          }
          insertRate++;
        }
        sIndx = i + 1;
      } else if (c == 'D') {                            // :337-367
        const int delLen = atoi(cigar.substr(sIndx, i - sIndx).c_str());
        const long pos = position + refIndx;
        bool found = false;
        if (delsOfChr)
          for (const KnownIndel& e : *delsOfChr) {
            if (e.pos > pos) break;
            if (e.pos == pos && delLen == e.len) { found = true; break; }
          }
        if (!found) {
          if (delLen >= 0) {
            if (delLen > (int)delFreqs.size() - 1) delFreqs.resize((size_t)delLen + 1, 0.0);
            delFreqs[delLen] += 1;
          }
          delRate++;
        }
        refIndx += delLen;
        sIndx = i + 1;
      } else if (c == 'M') {
        refIndx += atoi(cigar.substr(sIndx, i - sIndx).c_str());
        sIndx = i + 1;
      } else {
        sIndx = i + 1;
      }
    }
    if (k != n_c - 1 || n_c == 0 || cigar[n_c - 1] != 'M') return 0;   // :380-382
    const size_t rl = readSeq.size();
    if ((size_t)(position - 1) + rl > contig.size()) { skipped_overhang++; return 0; }
    string refSeq = subRef(chr, position - 1, (long)rl);              // :384-385
    string altSeq = subAlt(chr, position - 1, (long)rl);
    int isRead1 = 1;
    if (tlen < 0) {                                     // :387-397
      auto rcmp = [](string& s) { std::string t(s.rbegin(), s.rend()); for (char& ch : t) ch = complementOf(ch); s = t; };
      rcmp(refSeq);
      rcmp(altSeq);
      rcmp(readSeq);
      baseQuality = string(baseQuality.rbegin(), baseQuality.rend());
      isRead1 = 0;
    }
    const int n = (int)refSeq.size();
    string seq(kmer - 1, 'X');                          // :404-415
    for (int i = 0; i < n; i++) seq.push_back(altSeq[i] == readSeq[i] ? altSeq[i] : refSeq[i]);
    std::vector<double>& subs = isRead1 ? subs1 : subs2;
    for (int i = 0; i < n; i++) {                       // :416-442
      const int baseIndx = getIndexOfBase(readSeq[i]);
      const int binIndx = i * bins / n;
      if (baseIndx != -1) {
        const int kmerIndx = getKmerIndx(seq.substr(i, kmer));
        if (kmerIndx == -1) continue;
        subs[((size_t)kmerIndx * bins + binIndx) * N + baseIndx] += 1;
        kmersDist[(size_t)binIndx * kmerCount + kmerIndx] += 1;
      }
    }
    if (tlen > 0) {                                     // :446-451
      if (tlen > (int)iSizeDist.size() - 1) iSizeDist.resize((size_t)tlen + 1, 0.0);
      iSizeDist[tlen] += 1;
    }
    if (baseQuality.size() == readSeq.size()) {         // :457-481
      const int m = (int)readSeq.size();
      for (int i = 0; i < m; i++) {
        int r = getIndexOfBase(refSeq[i]);
        const int binIndx = i * bins / m;
        const int b = getIndexOfBase(readSeq[i]);
        if (r == -1 || b == -1) continue;
        if (altSeq[i] == readSeq[i]) r = getIndexOfBase(altSeq[i]);
        const int indx = r * N + b;
        const int j = (int)(signed char)baseQuality[i];
        if (j >= 33 && j <= 126) quality[((size_t)indx * bins + binIndx) * 94 + (j - 33)] += 1;
      }
    }
    reads_counted++;                                    // :483
    if (reads_counted >= (inTargets.empty() ? maxCount : 2 * maxCount)) return 2;   // :497-507: the caller stops reading
    return 0;
  }

  int feed(const char* sam_text, uint64_t sam_bytes) {
    const char* p = sam_text;
    const char* end = sam_text + sam_bytes;
    while (p < end) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      const char* le = nl ? nl : end;
      if (stopped) return 0;                            // (Profile::train left its loop, :1461-1464)
      const int ret = processRead(string(p, le));
      if (ret == 1) return 1;
      if (ret == 2) stopped = true;
      p = nl ? nl + 1 : end;
    }
    return 0;
  }

  void exportCounts(uint32_t n_isize, uint32_t n_indel_len, orc_train_counts* out) const {
    auto cp = [](uint64_t* dst, const std::vector<double>& src, size_t n) { if (dst) for (size_t i = 0; i < n; i++) dst[i] = i < src.size() ? (uint64_t)src[i] : 0; };
    cp(out->subs1, subs1, subs1.size()); cp(out->subs2, subs2, subs2.size());
    cp(out->kmers, kmersDist, kmersDist.size()); cp(out->quality, quality, quality.size());
    cp(out->isize, iSizeDist, n_isize); cp(out->ins_len, insFreqs, n_indel_len); cp(out->del_len, delFreqs, n_indel_len);
    out->isize_overflow = 0; out->indel_len_overflow = 0;
    for (size_t i = n_isize; i < iSizeDist.size(); i++) out->isize_overflow += (uint64_t)iSizeDist[i];
    for (size_t i = n_indel_len; i < insFreqs.size(); i++) out->indel_len_overflow += (uint64_t)insFreqs[i];
    for (size_t i = n_indel_len; i < delFreqs.size(); i++) out->indel_len_overflow += (uint64_t)delFreqs[i];
    out->lines = lines; out->reads_counted = reads_counted; out->cigar_chars = (uint64_t)baseCount;
    out->insert_events = (uint64_t)insertRate; out->delete_events = (uint64_t)delRate;
    out->skipped_overhang = skipped_overhang; out->gc_rejected = gc_rejected; out->gc_windows = gc_windows; out->capped = stopped ? 1 : 0;
  }

  // ---- the second half of Profile::train ----
  double gcMeans[101];
  double gcStd = 0, stdISize = 0;
  static constexpr double ZERO_FINAL = 2.2204e-16;

  static double median(std::vector<double> v) {          // lib/mydefine/MyDefine.h:72-104
    if (v.empty()) return 0;                              // (see the header)
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    return n % 2 ? v[n / 2] : (v[n / 2] + v[n / 2 - 1]) / 2;
  }
  static double clampZero(double x) { return fabs(x) < ZERO_FINAL ? 0 : x; }   // Matrix::operator*, Matrix.h:683-697

  // Profile::estimateGCParas, Profile.cpp:713-834
  void estimateGCParas(const string& gcFile) {
    const int gbins = 50;
    std::vector<int> counts(gbins + 1, 0);
    for (size_t i = 0; i < gcs.size(); i++) counts[(int)(gcs[i] * gbins)]++;
    const int expectCount = std::min(150000, (int)gcs.size()) / gbins;
    std::vector<int> steps(gbins + 1, 1);
    for (int i = 0; i <= gbins; i++) steps[i] = expectCount > 0 ? std::max(1, counts[i] / expectCount) : 1;
    std::ofstream ofs(gcFile.c_str());
    std::vector<int> indxs;
    std::vector<int> curCount(gbins + 1, 0);
    const double med_rc = median(readCounts);
    for (size_t i = 0; i < readCounts.size(); i++) {
      const int j = (int)(gcs[i] * gbins);
      if (curCount[j] % steps[j] == 0) {
        readCounts[i] = readCounts[i] / (med_rc + ZERO_FINAL);
        if (readCounts[i] < 3) {
          ofs << readCounts[i] << '\t' << gcs[i] << std::endl;
          indxs.push_back((int)i);
        }
      }
      curCount[j]++;
    }
    ofs.close();
    const double tau = 5, winSizeGC = 0.03;
    int minGC = -1, maxGC = -1;
    for (int k = 0; k <= 100; k++) {
      const double gc = k / 100.0;
      std::vector<double> g, r;
      for (int j : indxs)
        if (fabs(gc - gcs[j]) <= winSizeGC / 2) { g.push_back(gcs[j]); r.push_back(readCounts[j]); }
      if (g.size() > 20) {
        if (minGC == -1) minGC = k;
        maxGC = k;
        const size_t n = g.size();
        // beta = (B' W B)^-1 B' W y with the reference's Matrix arithmetic: every product is a running sum from 0 in index
        // order, clamped to 0 below ZERO_FINAL; W is diagonal, so the sums over it have one term that is not zero
        std::vector<double> w(n), btw0(n), btw1(n);
        for (size_t i = 0; i < n; i++) {
          w[i] = exp(-pow(g[i] - gc, 2) / (2 * tau));
          btw0[i] = clampZero(0 + 1 * w[i]);
          btw1[i] = clampZero(0 + g[i] * w[i]);
        }
        double M[2][2] = {{0, 0}, {0, 0}};
        for (size_t i = 0; i < n; i++) { M[0][0] += btw0[i] * 1; M[0][1] += btw0[i] * g[i]; M[1][0] += btw1[i] * 1; M[1][1] += btw1[i] * g[i]; }
        for (auto& row : M) for (double& x : row) x = clampZero(x);
        // Matrix::determinant (Matrix.h:223-268): elimination without pivoting unless the pivot is zero
        double c[2][2] = {{M[0][0], M[0][1]}, {M[1][0], M[1][1]}};
        int switchcount = 0;
        bool skip = false;
        if (c[0][0] == 0) {
          if (c[1][0] == 0) skip = true;
          else { switchcount++; std::swap(c[0][0], c[1][0]); std::swap(c[0][1], c[1][1]); }
        }
        if (!skip && c[1][0] != 0) {
          const double a = c[1][0] / c[0][0];
          c[1][0] -= a * c[0][0];
          c[1][1] -= a * c[0][1];
        }
        double det = 1;
        det *= c[0][0]; det *= c[1][1];
        if (switchcount % 2) det = -det;
        // Matrix::inverse (Matrix.h:184-220): cofactors over the determinant
        double inv[2][2];
        inv[0][0] = clampZero(M[1][1] / det);
        inv[1][0] = clampZero(-M[1][0] / det);
        inv[0][1] = clampZero(-M[0][1] / det);
        inv[1][1] = clampZero(M[0][0] / det);
        double beta[2] = {0, 0};
        for (int i = 0; i < 2; i++) {
          double acc = 0;
          for (size_t kk = 0; kk < n; kk++) {
            double p = 0;
            p += inv[i][0] * 1; p += inv[i][1] * g[kk];   // (inverse * B')[i][kk]
            p = clampZero(p);
            const double pw = clampZero(0 + p * w[kk]);    // (... * W)[i][kk]
            acc += pw * r[kk];
          }
          beta[i] = clampZero(acc);
        }
        double yp = 0;
        yp += 1 * beta[0]; yp += gc * beta[1];
        yp = clampZero(yp);
        gcMeans[k] = std::max(0.0, yp);
      } else {
        gcMeans[k] = 0;
      }
    }
    if (minGC >= 0) {   // (no GC percent with more than 20 windows: the reference reads gcMeans[-1] here; all means stay 0)
      for (int k = 0; k < minGC; k++) gcMeans[k] = gcMeans[minGC] * k / minGC;
      for (int k = maxGC + 1; k <= 100; k++) gcMeans[k] = gcMeans[maxGC] - gcMeans[maxGC] * (k - maxGC) / (100 - maxGC);
    }
    gcStd = 0;
    for (int j : indxs) {
      const int k = (int)(gcs[j] * 100);
      gcStd += pow(readCounts[j] - gcMeans[k], 2);
    }
    gcStd = sqrt(gcStd / indxs.size());
  }

  static void normalizeRows(std::vector<double>& m, size_t rows, size_t cols) {   // Matrix::normalize(0), Matrix.h:483-503
    for (size_t i = 0; i < rows; i++) {
      double s = 0;
      for (size_t j = 0; j < cols; j++) s += m[i * cols + j];
      for (size_t j = 0; j < cols; j++) m[i * cols + j] /= (ZERO_FINAL + s);
    }
  }
  // Profile::normParas(false), Profile.cpp:836-900
  void normParas() {
    normalizeRows(kmersDist, bins, kmerCount);
    std::vector<string> kmerOf(kmerCount);
    for (auto& kv : kmerIndex) kmerOf[kv.second] = kv.first;
    for (int i = 0; i < kmerCount; i++)
      for (std::vector<double>* sd : {&subs1, &subs2}) {
        std::vector<double> block(sd->begin() + (size_t)i * bins * N, sd->begin() + (size_t)(i + 1) * bins * N);
        normalizeRows(block, bins, N);
        const int indx = getIndexOfBase(kmerOf[i][kmer - 1]);
        for (int j = 0; j < bins; j++) {
          double s = 0;
          for (int k = 0; k < N; k++) s += block[(size_t)j * N + k];
          if (s < ZERO_FINAL) block[(size_t)j * N + indx] = 1;
        }
        std::copy(block.begin(), block.end(), sd->begin() + (size_t)i * bins * N);
      }
    for (int i = 0; i < N * N; i++) {
      std::vector<double> block(quality.begin() + (size_t)i * bins * 94, quality.begin() + (size_t)(i + 1) * bins * 94);
      normalizeRows(block, bins, 94);
      std::copy(block.begin(), block.end(), quality.begin() + (size_t)i * bins * 94);
    }
    int maxCount = 0, j = 0;
    for (size_t i = 0; i < iSizeDist.size(); i++)
      if (iSizeDist[i] > maxCount) { maxCount = (int)iSizeDist[i]; j = (int)i; }
    for (size_t i = (size_t)j * 5; i < iSizeDist.size(); i++) iSizeDist[i] = 0;
    normalizeRows(iSizeDist, 1, iSizeDist.size());
    auto at = [&](int i) { return (size_t)i < iSizeDist.size() ? iSizeDist[i] : 0.0; };
    double meanTlen = 0;
    for (int i = 0; i < j * 5; i++) meanTlen += at(i) * i;
    stdISize = 0;
    for (int i = 0; i < j * 5; i++) stdISize += at(i) * pow(i - meanTlen, 2);
    stdISize = sqrt(stdISize);
    normalizeRows(insFreqs, 1, insFreqs.size());
    normalizeRows(delFreqs, 1, delFreqs.size());
    insertRate /= baseCount;
    delRate /= baseCount;
  }
  // Profile::saveResults, Profile.cpp:1240-1365
  int saveResults(const string& outFile, const string& bamFile, const string& stamp, int readLength) {
    std::ofstream ofs(outFile.c_str());
    if (!ofs.is_open()) return 1;
    std::ostream& ost = ofs;
    ost << "#model created at " << stamp;
    ost << "#reads: " << bamFile << std::endl << std::endl;
    ost << "bases: " << bases << std::endl;
    ost << "readLength: " << readLength << std::endl;
    ost << "binCount: " << bins << std::endl;
    ost << "kmer: " << kmer << std::endl << std::endl;
    ost << "\n[Insert Rate]" << std::endl << insertRate << std::endl << "[Insert Frequency]" << std::endl;
    for (size_t i = 0; i + 1 < insFreqs.size(); i++) ost << insFreqs[i] << '\t';
    ost << insFreqs.back() << std::endl;
    ost << "\n[Deletion Rate]" << std::endl << delRate << std::endl << "[Deletion Frequency]" << std::endl;
    for (size_t i = 0; i + 1 < delFreqs.size(); i++) ost << delFreqs[i] << '\t';
    ost << delFreqs.back() << std::endl;
    ost << "\n[Substitution Probs]" << std::endl;
    std::vector<string> kmerOf(kmerCount);
    for (auto& kv : kmerIndex) kmerOf[kv.second] = kv.first;
    for (int i = 0; i < kmerCount; i++) {
      ost << "kmer: " << kmerOf[i] << std::endl;
      for (const std::vector<double>* sd : {&subs1, &subs2})
        for (int j = 0; j < bins; j++)
          for (int k = 0; k < N; k++) {
            ost << (*sd)[((size_t)i * bins + j) * N + k];
            if (k < N - 1) ost << '\t'; else ost << std::endl;
          }
    }
    ost << "\n[Base Quality Distribution]" << std::endl;
    for (int i = 0; i < N * N; i++) {
      ost << "basePairIndx: " << i << std::endl;
      for (int j = 0; j < bins; j++)
        for (int k = 0; k < 94; k++) {
          ost << quality[((size_t)i * bins + j) * 94 + k];
          if (k < 93) ost << '\t'; else ost << std::endl;
        }
    }
    ost << "\n[Insert Size Standard Deviation]" << std::endl << stdISize << std::endl;
    ost << "\n[Log Ratio Mean Value]" << std::endl;
    for (int i = 0; i < 101; i++) ost << i << '\t' << gcMeans[i] << std::endl;
    ost << "\n[Log Ratio Standard Deviation]" << std::endl << gcStd << std::endl;
    return 0;
  }
};

// Profile::setReadLength, Profile.cpp:126-170: the number of the first line whose CIGAR is a single nM (the reference asks
// samtools for `-F 0xD04 -q 20` lines there; the text is what there is)
int readLengthOf(const char* sam_text, uint64_t sam_bytes) {
  const char* p = sam_text;
  const char* end = sam_text + sam_bytes;
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* le = nl ? nl : end;
    const char* f = p;
    int tabs = 0;
    while (f < le && tabs < 5) { if (*f == '\t') tabs++; f++; }
    if (tabs == 5) {
      const char* e = f;
      while (e < le && *e != '\t') e++;
      const int n = (int)(e - f);
      int i = 0;
      for (; i < n - 1; i++) if (!(f[i] >= '0' && f[i] <= '9')) break;
      if (n >= 1 && i == n - 1 && f[i] == 'M') return atoi(string(f, e).c_str());
    }
    p = nl ? nl + 1 : end;
  }
  return 0;
}

uint64_t g_max_reads = 0;   // orc_train_set_max_reads: 0 = the reference's 300,000,000

int setupState(TrainState& T, const char* fasta_path, const char* vcf_path, const char* bed_path, const char* bases, int kmer, int bins, int count_gc) {
  T.bases = bases;
  T.kmer = kmer;
  T.bins = bins;
  T.count_gc = count_gc != 0;
  if (g_max_reads) T.maxCount = g_max_reads;
  T.init();
  T.loadFastaOrdered(fasta_path);
  if (vcf_path && *vcf_path && T.parseVcf(vcf_path)) return 2;
  if (bed_path && *bed_path && T.loadTargets(bed_path)) return 3;
  return 0;
}

}  // namespace

// Profile::processRead's `maxCount` for the calls that follow (tests: a small cap shows the cut; 0 = the reference's)
extern "C" void orc_train_set_max_reads(uint64_t n) { g_max_reads = n; }

extern "C" int orc_train_count(const char* sam_text, uint64_t sam_bytes, const char* fasta_path, const char* bases, int kmer, int bins,
                               uint32_t n_isize, uint32_t n_indel_len, orc_train_counts* out) {
  TrainState T;
  if (setupState(T, fasta_path, nullptr, nullptr, bases, kmer, bins, 0)) return 2;
  if (T.feed(sam_text, sam_bytes)) return 1;
  T.exportCounts(n_isize, n_indel_len, out);
  return 0;
}

// the counters of a whole Profile::train pass (countGC gating, known variants, targets) and the (GC, read count) pairs
extern "C" int orc_train(const char* sam_text, uint64_t sam_bytes, const char* fasta_path, const char* vcf_path, const char* bed_path,
                         const char* bases, int kmer, int bins, uint32_t n_isize, uint32_t n_indel_len, orc_train_counts* out,
                         double* gc, double* rc, uint64_t cap, uint64_t* n_gc) {
  TrainState T;
  if (int e = setupState(T, fasta_path, vcf_path, bed_path, bases, kmer, bins, 1)) return e;
  if (T.feed(sam_text, sam_bytes)) return 1;
  T.exportCounts(n_isize, n_indel_len, out);
  if (n_gc) *n_gc = T.gcs.size();
  for (size_t i = 0; i < T.gcs.size() && i < cap; i++) { if (gc) gc[i] = T.gcs[i]; if (rc) rc[i] = T.readCounts[i]; }
  return 0;
}

// Profile::train as a whole: the profile file (and `<out>.gc` when the GC model is fitted).  `stamp` replaces the
// asctime() line of saveResults (:1263-1265), `bam_label` the name in its "#reads:" line.
extern "C" int orc_train_profile(const char* sam_text, uint64_t sam_bytes, const char* fasta_path, const char* vcf_path, const char* bed_path,
                                 const char* bases, int kmer, int bins, const char* out_path, const char* bam_label, const char* stamp) {
  TrainState T;
  const int readLength = readLengthOf(sam_text, sam_bytes);
  if (bins > readLength) bins = readLength;            // Profile::init, :184-188
  if (int e = setupState(T, fasta_path, vcf_path, bed_path, bases, kmer, bins, 1)) return e;
  if (T.feed(sam_text, sam_bytes)) return 1;
  const double med_rc = TrainState::median(T.readCounts);   // :1471-1481
  if (med_rc < 5) { for (double& m : T.gcMeans) m = 1; T.gcStd = 1.0e-5; }
  else T.estimateGCParas(string(out_path) + ".gc");
  T.normParas();
  return T.saveResults(out_path, bam_label, stamp, readLength) ? 4 : 0;
}
