// oracle/oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Sequential CPU restatement of the reference's simuReads path.  Every function cites the
// reference file:line it follows (paths relative to the SimuSCoP source tree).  The code is
// written for fidelity, not speed: fp64 CDFs with linear scans, char-by-char sequence walks.
//
// Randomness goes through `Rng`, which has two modes:
//   mt     -- libstdc++ mt19937 x2 (worker "real"/"int" generators), default_random_engine +
//             normal_distribution (GC factors), glibc srand/rand (haplotype choices): the
//             reference's own generators, seeded from the frozen clock value.  Sequential draws.
//   philox -- counter-addressed Philox4x32-10 (philox.h); every draw has an address, so the
//             GPU can produce the same value from any lane.
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

#include "philox.h"

namespace orc {

using std::map;
using std::string;
using std::vector;

static const double ZERO_FINAL = 2.2204e-16;  // lib/mydefine/MyDefine.cpp:20

// (long) of a double the way the reference's machine code does it (cvttsd2si): a NaN or a value out of range gives
// LONG_MIN.  The reference divides by weighted lengths that can be zero (Genome.cpp:803-817: a chromosome, or a whole
// population, whose every window holds an N or lies outside the targets) and casts the NaN; what follows -- sums that wrap,
// negative read counts that sample nothing -- is what its binary does, and what a user of such inputs sees: no reads from
// that chromosome, the run goes on.  Restated as such instead of refused.
static inline long to_long_x86(double v) {
  if (!(v > -9223372036854775808.0 && v < 9223372036854775808.0)) return (long)0x8000000000000000ull;
  return (long)v;
}
static inline long wrap_add(long a, long b) { return (long)((unsigned long)a + (unsigned long)b); }
static inline long wrap_sub(long a, long b) { return (long)((unsigned long)a - (unsigned long)b); }

struct Fail : std::runtime_error {
  explicit Fail(const string& m) : std::runtime_error(m) {}
};

// ---------------------------------------------------------------------------------------------
// string helpers
// ---------------------------------------------------------------------------------------------
// lib/mydefine/MyDefine.cpp:197-209
static string trim(const string& str, const char* charlist = " \t\r\n") {
  string ret(str);
  size_t indx = ret.find_first_not_of(charlist);
  if (indx != string::npos) {
    ret.erase(0, indx);
    indx = ret.find_last_not_of(charlist);
    ret.erase(indx + 1);
  } else {
    ret.erase();
  }
  return ret;
}

// lib/split/split.cpp:3-16 (getline-based: trailing empty field is dropped)
static vector<string> split(const string& s, char delim) {
  vector<string> elems;
  std::stringstream ss(s);
  string item;
  while (std::getline(ss, item, delim)) elems.push_back(item);
  return elems;
}

// lib/split/split.cpp:18-29 (strtok-based: empty tokens skipped)
static vector<string> split_any(const string& s, const string& delims) {
  vector<string> elems;
  size_t i = 0;
  while (i < s.size()) {
    size_t b = s.find_first_not_of(delims, i);
    if (b == string::npos) break;
    size_t e = s.find_first_of(delims, b);
    if (e == string::npos) e = s.size();
    elems.push_back(s.substr(b, e - b));
    i = e;
  }
  return elems;
}

// lib/mydefine/MyDefine.cpp:212-225 == lib/snp/snp.cpp:131-145 == lib/fastahack/Fasta.cpp:59-68
static string abbrOfChr(string chr) {
  size_t i = chr.find("chrom");
  if (i == string::npos) {
    i = chr.find("chr");
    if (i != string::npos) chr = chr.substr(i + 3, chr.size() - 3);
  } else {
    chr = chr.substr(i + 5, chr.size() - 5);
  }
  return chr;
}

// lib/mydefine/MyDefine.cpp:239-251
static bool getNextLine(std::ifstream& ifs, string& line, int& lineNum) {
  line = "";
  while (std::getline(ifs, line)) {
    lineNum++;
    if (!line.empty() && line.at(0) != '#') break;
  }
  return !line.empty();
}

// ---------------------------------------------------------------------------------------------
// Config (lib/config/Config.cpp:14-175)
// ---------------------------------------------------------------------------------------------
struct Config {
  map<string, string> s;
  map<string, int> i;
  map<string, double> r;
  vector<string> popuNames;

  Config() {
    const char* names[] = {"bam", "profile", "ref", "variation", "snp", "vcf", "target",
                           "bases", "output", "abundance", "layout", "samtools"};
    for (const char* n : names) s[n] = "";
    s["layout"] = "SE";
    s["bases"] = "ACTG";
    i["kmer"] = 0; i["bins"] = 0; i["threads"] = 1; i["verbose"] = 1; i["readLength"] = 0;
    i["coverage"] = 0; i["ploidy"] = 2; i["insertSize"] = 350;
    r["indelRate"] = 0.00025;
  }
  bool paired() const { return s.at("layout") == "PE"; }

  void load(const string& file) {
    std::ifstream ifs(file.c_str());
    if (!ifs.is_open()) throw Fail("Error: can not open configuration file" + file);
    string line;
    int lineNum = 0;
    while (std::getline(ifs, line)) {
      lineNum++;
      line = trim(line);
      if (line.empty() || line[0] == '#') continue;
      size_t indx = line.find('=');
      if (indx == string::npos)
        throw Fail("ERROR: line " + std::to_string(lineNum) + " is incorrectly formatted in file " + file);
      string key = trim(line.substr(0, indx));
      string value = trim(line.substr(indx + 1));
      if (s.count(key)) s[key] = value;
      else if (i.count(key)) i[key] = atoi(value.c_str());
      else if (r.count(key)) r[key] = atof(value.c_str());
      else if (key == "name") {
        popuNames = split(value, ',');
        for (auto& p : popuNames) p = trim(p);
      } else {
        throw Fail("ERROR: unrecognized item \"" + key + "\" @line " + std::to_string(lineNum) + " in file " + file);
      }
    }
    check();
  }
  // Config.cpp:101-175
  void check() {
    if (s["profile"].empty()) throw Fail("Error: sequencing profile must be specified!");
    if (s["ref"].empty()) throw Fail("Error: reference file not specified!");
    if (popuNames.empty()) throw Fail("Error: population names not specified!");
    if (popuNames.size() > 1 && s["abundance"].empty()) throw Fail("Error: abundance file not specified!");
    if (s["output"].empty()) throw Fail("Error: output directory not specified!");
    if (s["layout"].empty()) s["layout"] = "SE";
    else if (s["layout"] != "SE" && s["layout"] != "PE") throw Fail("Error: sequence layout incorrectly specified!");
    if (i["threads"] < 1) throw Fail("Error: number of threads should be a positive integer!");
    if (i["coverage"] < 1) throw Fail("Error: sequence coverage should be a positive integer!");
    if (i["ploidy"] < 1) throw Fail("Error: genome ploidy should be a positive integer!");
    if (s["layout"] == "PE" && i["insertSize"] < i["readLength"]) throw Fail("Error: insert size should be not smaller than read length!");
    if (r["indelRate"] < 0 || r["indelRate"] > 0.001) throw Fail("Error: indel error rate should be a value between 0 to 0.001!");
  }
};

// ---------------------------------------------------------------------------------------------
// RNG
// ---------------------------------------------------------------------------------------------
struct Rng {
  bool philox = false;
  // mt mode state (lib/threadpool/ThreadPool.cpp:41-49: two generators, same seed)
  std::mt19937 realGen, intGen;
  unsigned clockSeed = 0;
  // philox mode state
  uint32_t k0 = 0, k1 = 0;

  void initMt(uint64_t sec, uint64_t nsec) {
    philox = false;
    // chrono::system_clock::now().time_since_epoch().count() is in ns; assigned to `unsigned`
    long long cnt = (long long)sec * 1000000000LL + (long long)nsec;
    clockSeed = (unsigned)cnt;
    realGen.seed(clockSeed);
    intGen.seed(clockSeed);
  }
  void initPhilox(uint64_t seed) {
    philox = true;
    k0 = (uint32_t)seed;
    k1 = (uint32_t)(seed >> 32);
  }
  uint32_t ph(uint32_t kind, uint32_t ctx24, uint32_t c0, uint32_t c1, uint32_t c2, int lane) const {
    Philox4 o = philox4x32_10(c0, c1, c2, kind | (ctx24 << 8), k0, k1);
    return o.v[lane];
  }
};

// lib/threadpool/ThreadPool.cpp:203-207 with minRandNumber = 0, maxRandNumber = 2^32-1
static inline double u32ToDouble(uint32_t x, double start, double end) {
  double number = (double)x;
  return start + (end - start) * ((number - 0.0) / (4294967295.0 - 0.0 + 1.0));
}
// lib/threadpool/ThreadPool.cpp:208-212 (double -> long truncation)
static inline long u32ToInteger(uint32_t x, long start, long end) {
  double number = (double)x;
  return (long)(start + (end - start) * ((number - 0.0) / (4294967295.0 - 0.0 + 1.0)));
}

// lib/mydefine/MyDefine.cpp:176-184 given the 32-bit draw
static inline int randIndxFrom(uint32_t x, const double* cdf, int ac) {
  double r = u32ToDouble(x, ZERO_FINAL, 1);
  for (int k = 0; k < ac; k++)
    if (r <= cdf[k]) return k;
  return ac - 1;
}

// Philox mode only.  randIndx's predicate `r <= c` (MyDefine.cpp:176-184) is monotone in the 32-bit draw x, so for
// every fp64 CDF value c it holds exactly for the draws x < countLe(c): countLe(c) = #{x : r(x) <= c}, by bisection
// on the reference's own fp64 expression.  A CDF row therefore partitions the 2^32 draws into integer masses
//   n[k] = countLe(cdf[k]) - countLe(cdf[k-1])   (k < ac-1),   n[ac-1] = 2^32 - countLe(cdf[ac-2])
// (the fall-through `return ac-1`), and ANY map from a uniform 32-bit draw to outcomes that gives outcome k exactly
// n[k] draws samples the reference's distribution exactly.  The philox mode uses two such maps:
// an identity-first order for substitutions and Walker/Vose alias columns for qualities.
static uint64_t countLe(double c) {
  uint64_t lo = 0, hi = 1ull << 32;  // predicate true for x < result
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (u32ToDouble((uint32_t)mid, ZERO_FINAL, 1) <= c) lo = mid + 1; else hi = mid;
  }
  return lo;
}
static void rowMasses(const double* cdf, int ac, vector<uint64_t>& n) {
  n.assign(ac, 0);
  uint64_t prev = 0;
  for (int k = 0; k + 1 < ac; k++) {
    uint64_t c = countLe(cdf[k]);
    if (c < prev) c = prev;  // a CDF is non-decreasing; guards against a malformed row
    n[k] = c - prev;
    prev = c;
  }
  n[ac - 1] = (1ull << 32) - prev;
}

// ---------------------------------------------------------------------------------------------
// Profile (lib/profile/Profile.cpp)
// ---------------------------------------------------------------------------------------------
struct Profile {
  string bases = "ACTG";
  int N = 4, kmer = 0, bins = 0, readLength = 0, kmerCount = 0;
  int minQ = 33, maxQ = 126, nQual = 94;
  double insertRate = 0, delRate = 0, stdISize = 0, gcStd = 0;
  uint64_t cntIns = 0, cntDel = 0;  // philox mode: the numbers of 32-bit draws passing the two tests (countIndelDraws), see buildIndelGaps
  void countIndelDraws() {
    // number of 32-bit draws x with x/2^32 <= insertRate, resp. x/2^32 < delRate/(1-insertRate): both
    // predicates are monotone in x, so a bisection over [0, 2^32] finds the exact counts
    const double dthr = delRate / (1 - insertRate);
    auto count = [&](bool strict, double thr) {
      uint64_t lo = 0, hi = 1ull << 32;  // predicate true for x < result
      while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const double v = (double)mid / 4294967296.0;
        if (strict ? v < thr : v <= thr) lo = mid + 1; else hi = mid;
      }
      return lo;
    };
    cntIns = count(false, insertRate);
    cntDel = count(true, dthr);
    buildIndelGaps();
  }
  // philox mode: sequencing indels by skipping ahead.  With A = cntIns * 2^32 and B = A + (2^32 - cntIns) * cntDel
  // (in units of 2^-64) a template position carries an indel candidate with probability B / 2^64 -- an insertion with
  // probability A / B, else a deletion -- independently of every other position (Profile.cpp:1560-1570 draws them one
  // position at a time).  The distance to the next candidate is therefore geometric, and it is drawn directly:
  //   gapS[k] = P(no candidate in the next k positions) * 2^64:  gapS[1] = 2^64 - B,  gapS[k+1] = floor(gapS[k] * gapS[1] / 2^64)
  //   one 64-bit uniform x:  g = #{k >= 1 : x < gapS[k]}  positions are skipped, the candidate sits on the next one.
  // (The floors make P(g >= k) differ from (1 - B/2^64)^k by less than k * 2^-64.)
  uint64_t evA = 0, evB = 0;
  vector<uint64_t> gapS;  // [0 .. readLength]; [0] unused
  void buildIndelGaps() {
    evA = cntIns << 32;
    evB = evA + ((1ull << 32) - cntIns) * cntDel;
    const int L = std::max(readLength, 1);
    gapS.assign((size_t)L + 1, 0xFFFFFFFFFFFFFFFFull);
    if (evB == 0) return;  // no candidates ever (x < 2^64 - 1 for every draw but one)
    gapS[1] = 0ull - evB;
    for (int k = 1; k < L; k++) gapS[(size_t)k + 1] = (uint64_t)(((unsigned __int128)gapS[(size_t)k] * gapS[1]) >> 64);
  }
  // ---- philox mode: integer sampling tables (built from the fp64 CDFs by buildIntegerTables) ----
  // Substitution row (kmerIndx, bin): outcomes in the order o = [cd, the other base indexes ascending], cd = index of the
  // context's last base (the reference base itself); cumulative masses c0 <= c1 <= c2 of o[0..2]:
  //   j = #{i : xs >= c_i},  k = o[j]          -- "no substitution" is the single test xs < c0.
  // Quality row (refIndx*N + k, bin): alias columns.  W = 2^lgW >= the largest number of symbols with mass in any row
  // (>= 4), C = 2^32 / W draws per column; column col holds symbol aLo for u < aThr and aHi for u >= aThr, where
  //   col = xq >> (32 - lgW),  u = xq & (C - 1).
  vector<uint64_t> subCum1, subCum2;  // [kmerCount * bins][3]
  vector<uint8_t> subOrd;             // [kmerCount][4]
  int lgW = 2;
  vector<uint32_t> aThr;              // [N*N * bins][W], in [0, C)
  vector<uint8_t> aLo, aHi;
  void buildIntegerTables() {
    vector<uint64_t> n;
    subOrd.assign((size_t)kmerCount * 4, 0);
    for (int i = 0; i < kmerCount; i++) {
      const int cd = baseIndex(kmers[i][kmer - 1]);
      uint8_t* o = &subOrd[(size_t)i * 4];
      o[0] = (uint8_t)cd;
      for (int k = 0, q = 1; k < N; k++) if (k != cd) o[q++] = (uint8_t)k;
    }
    for (int t = 0; t < 2; t++) {
      const vector<double>& src = t == 0 ? subs1 : subs2;
      vector<uint64_t>& dst = t == 0 ? subCum1 : subCum2;
      dst.assign((size_t)kmerCount * bins * 3, 0);
      if (t == 1 && !hasSub2) continue;
      for (size_t r = 0; r < (size_t)kmerCount * bins; r++) {
        rowMasses(src.data() + r * N, N, n);
        const uint8_t* o = &subOrd[(r / bins) * 4];
        uint64_t c = 0;
        for (int i = 0; i < 3; i++) { c += n[o[i]]; dst[r * 3 + i] = c; }
      }
    }
    // quality alias columns
    const size_t qrows = (size_t)N * N * bins;
    vector<vector<uint64_t>> masses(qrows);
    size_t most = 1;
    for (size_t r = 0; r < qrows; r++) {
      rowMasses(qual.data() + r * nQual, nQual, masses[r]);
      size_t m = 0;
      for (uint64_t v : masses[r]) m += v != 0;
      most = std::max(most, m);
    }
    lgW = 2;
    while ((1u << lgW) < most) lgW++;
    const uint32_t W = 1u << lgW;
    const uint64_t C = 1ull << (32 - lgW);
    aThr.assign(qrows * W, 0); aLo.assign(qrows * W, 0); aHi.assign(qrows * W, 0);
    for (size_t r = 0; r < qrows; r++) {
      // columns 0..m-1 start with the row's symbols in ascending order, the others empty
      vector<uint64_t> mass(W, 0);
      vector<int> sym(W, -1);
      uint32_t m = 0;
      for (int k = 0; k < nQual; k++) if (masses[r][k]) { mass[m] = masses[r][k]; sym[m] = k; m++; }
      vector<uint32_t> small, large;
      for (uint32_t c = 0; c < W; c++) (mass[c] < C ? small : large).push_back(c);
      vector<uint64_t> thr(W, C);
      vector<int> lo(sym), hi(sym);
      while (!small.empty() && !large.empty()) {
        const uint32_t sc = small.back(); small.pop_back();
        const uint32_t g = large.back(); large.pop_back();
        thr[sc] = mass[sc]; hi[sc] = sym[g];
        mass[g] -= C - mass[sc];
        (mass[g] < C ? small : large).push_back(g);
      }
      if (!small.empty()) throw Fail("alias construction: masses do not add up");
      for (uint32_t c = 0; c < W; c++) {
        // canonical form: thr in [0, C); a column of one symbol is thr = 0 with that symbol on both sides
        if (thr[c] == C) { thr[c] = 0; hi[c] = lo[c]; }
        if (thr[c] == 0) lo[c] = hi[c];
        if (lo[c] < 0 || hi[c] < 0) throw Fail("alias construction: empty column left without a symbol");
        aThr[r * W + c] = (uint32_t)thr[c]; aLo[r * W + c] = (uint8_t)lo[c]; aHi[r * W + c] = (uint8_t)hi[c];
      }
    }
  }
  int subSample(bool mate2, int kmerIndx, int binIndx, uint32_t xs) const {
    const size_t r = (size_t)kmerIndx * bins + binIndx;
    const uint64_t* c = (mate2 ? subCum2 : subCum1).data() + r * 3;
    const int j = (xs >= c[0]) + (xs >= c[1]) + (xs >= c[2]);
    return subOrd[(size_t)kmerIndx * 4 + j];
  }
  int qualSample(int bp, int binIndx, uint32_t xq) const {
    const size_t r = ((size_t)bp * bins + binIndx) << lgW;
    const uint32_t col = xq >> (32 - lgW), u = xq & ((1u << (32 - lgW)) - 1u);
    return u < aThr[r + col] ? aLo[r + col] : aHi[r + col];
  }
  vector<double> insFreqs, delFreqs, insCdf, delCdf;
  vector<string> kmers;
  map<string, int> kmerIndex;       // stands for the KmerIndex trie (Profile.h:18-23)
  vector<double> subs1, subs2;      // [kmerCount][bins][N]   (Dist, then turned into Cdf)
  bool hasSub2 = false;
  vector<double> qual;              // [N*N][bins][nQual]
  vector<int> iSizeAlphabet;
  vector<double> iSizeCdf;
  double gcMeans[101];
  bool paired = false;
  int insertSize = 350;
  // mt-mode GC generators (Profile.cpp:1409-1415)
  vector<std::default_random_engine> gcGen;
  vector<std::normal_distribution<double>> gcNorm;

  int baseIndex(char c) const {  // lib/mydefine/MyDefine.cpp:228-236
    for (int i = 0; i < N; i++)
      if (bases[i] == c) return i;
    return -1;
  }

  // Profile.cpp:70-124
  void initKmers() {
    kmerCount = 0;
    for (int i = kmer - 1; i >= 0; i--) kmerCount += (int)pow(N, kmer - i);
    kmers.assign(kmerCount, string(kmer, ' '));
    vector<int> tmp(kmer, 0);
    int k = 0;
    for (int j = kmer - 1; j >= 0; j--) {
      for (int i = j; i < kmer; i++) tmp[i] = 0;
      while (tmp[j] < N) {
        int i;
        for (i = 0; i < j; i++) kmers[k][i] = 'X';
        for (; i < kmer; i++) kmers[k][i] = bases[tmp[i]];
        kmerIndex[kmers[k]] = k;
        k++;
        int n = 1;
        for (i = kmer - 1; i > j; i--) {
          if (n == 0) break;
          tmp[i] += n;
          if (tmp[i] == N) { tmp[i] = 0; n = 1; } else { n = 0; }
        }
        tmp[j] += n;
      }
    }
  }
  // Profile.cpp:220-226 (unknown path -> -1)
  int getKmerIndx(const char* s, int len) const {
    auto it = kmerIndex.find(string(s, len));
    return it == kmerIndex.end() ? -1 : it->second;
  }

  // Profile.cpp:934-1238
  void load(const string& file) {
    std::ifstream ifs(file.c_str());
    if (!ifs.is_open()) throw Fail("can not open file " + file);
    string line;
    int lineNum = 0;
    string b = "";
    int binCount = -1, km = -1, rl = -1;
    string errMsg = "Error: malformed model file " + file + " @line ";
    while (getNextLine(ifs, line, lineNum)) {
      vector<string> f = split(line, ':');
      if (f.size() != 2) throw Fail(errMsg + std::to_string(lineNum));
      string key = trim(f[0]);
      if (key == "bases") { b = trim(f[1]); if (b.empty()) throw Fail(errMsg + std::to_string(lineNum)); }
      else if (key == "binCount") { binCount = atoi(trim(f[1]).c_str()); if (binCount <= 0) throw Fail(errMsg + std::to_string(lineNum)); }
      else if (key == "kmer") { km = atoi(trim(f[1]).c_str()); if (km <= 0) throw Fail(errMsg + std::to_string(lineNum)); }
      else if (key == "readLength") { rl = atoi(trim(f[1]).c_str()); if (rl <= 0) throw Fail(errMsg + std::to_string(lineNum)); }
      else throw Fail(errMsg + std::to_string(lineNum));
      if (!b.empty() && binCount > 0 && km > 0 && rl > 0) break;
    }
    if (b.empty() || binCount <= 0 || km <= 0 || rl <= 0) throw Fail("Error: malformed model file " + file);
    bases = b; N = (int)b.size(); kmer = km; readLength = rl;
    // Profile::init, Profile.cpp:172-218
    bins = binCount;
    if (bins > readLength) bins = readLength;
    insertRate = 0; delRate = 0; insFreqs.assign(1, 0); delFreqs.assign(1, 0); stdISize = 0;
    initKmers();
    subs1.assign((size_t)kmerCount * bins * N, 0.0);
    subs2.assign((size_t)kmerCount * bins * N, 0.0);
    qual.assign((size_t)N * N * bins * nQual, 0.0);
    for (int g = 0; g < 101; g++) gcMeans[g] = 0;

    int loaded = 0;
    vector<string> f;
    while (getNextLine(ifs, line, lineNum)) {
      if (line == "[Insert Rate]") {
        if (!getNextLine(ifs, line, lineNum)) throw Fail("Error: malformed profile file " + file);
        insertRate = atof(trim(line).c_str());
        loaded++;
      } else if (line == "[Insert Frequency]") {
        if (!getNextLine(ifs, line, lineNum)) throw Fail("Error: malformed profile file " + file);
        f = split(line, '\t');
        if (f.size() < 1) throw Fail(errMsg + std::to_string(lineNum));
        insFreqs.resize(f.size());
        for (size_t j = 0; j < f.size(); j++) insFreqs[j] = atof(trim(f[j]).c_str());
        loaded++;
      } else if (line == "[Deletion Rate]") {
        if (!getNextLine(ifs, line, lineNum)) throw Fail("Error: malformed profile file " + file);
        delRate = atof(trim(line).c_str());
        loaded++;
      } else if (line == "[Deletion Frequency]") {
        if (!getNextLine(ifs, line, lineNum)) throw Fail("Error: malformed profile file " + file);
        f = split(line, '\t');
        if (f.size() < 1) throw Fail(errMsg + std::to_string(lineNum));
        delFreqs.resize(f.size());
        for (size_t j = 0; j < f.size(); j++) delFreqs[j] = atof(trim(f[j]).c_str());
        loaded++;
      } else if (line == "[Substitution Probs]") {
        for (int i = 0; i < kmerCount; i++) {
          if (!getNextLine(ifs, line, lineNum)) throw Fail("Error: malformed profile file " + file);
          f = split(line, ':');
          if (f.size() != 2 || trim(f[0]) != "kmer") throw Fail(errMsg + std::to_string(lineNum));
          string ks = trim(f[1]);
          int kidx = getKmerIndx(ks.c_str(), (int)ks.size());
          if (kidx == -1) throw Fail("Error: unrecognized kmer @line " + std::to_string(lineNum));
          for (int j = 0; j < bins * 2; j++) {
            if (!getNextLine(ifs, line, lineNum)) throw Fail("Error: malformed profile file " + file);
            f = split(line, '\t');
            if ((int)f.size() != N) throw Fail(errMsg + std::to_string(lineNum));
            for (int k = 0; k < N; k++) {
              double prob = atof(trim(f[k]).c_str());
              if (j < bins) subs1[((size_t)kidx * bins + j) * N + k] = prob;
              else subs2[((size_t)kidx * bins + (j - bins)) * N + k] = prob;
            }
          }
        }
        loaded++;
      } else if (line == "[Base Quality Distribution]") {
        for (int i = 0; i < N * N; i++) {
          if (!getNextLine(ifs, line, lineNum)) throw Fail("Error: malformed profile file " + file);
          f = split(line, ':');
          if (f.size() != 2 || trim(f[0]) != "basePairIndx") throw Fail(errMsg + std::to_string(lineNum));
          int bp = atoi(trim(f[1]).c_str());
          if (bp < 0 || bp > N * N - 1) throw Fail("Error: unrecognized basePairIndx @line " + std::to_string(lineNum));
          for (int j = 0; j < bins; j++) {
            if (!getNextLine(ifs, line, lineNum)) throw Fail("Error: malformed profile file " + file);
            f = split(line, '\t');
            if ((int)f.size() != nQual) throw Fail(errMsg + std::to_string(lineNum));
            for (int k = 0; k < nQual; k++)
              qual[((size_t)bp * bins + j) * nQual + k] = atof(trim(f[k]).c_str());
          }
        }
        loaded++;
      } else if (line == "[Insert Size Standard Deviation]") {
        if (!getNextLine(ifs, line, lineNum)) throw Fail("Error: malformed model file " + file);
        stdISize = atof(trim(line).c_str());
        loaded++;
      } else if (line == "[Log Ratio Mean Value]") {
        for (int j = 0; j < 101; j++) {
          if (!getNextLine(ifs, line, lineNum)) throw Fail("Error: malformed model file " + file);
          f = split(line, '\t');
          if (f.size() != 2) throw Fail(errMsg + std::to_string(lineNum));
          int gc = atoi(f[0].c_str());
          if (gc < 0 || gc > 100) throw Fail(errMsg + std::to_string(lineNum));
          gcMeans[gc] = atof(f[1].c_str());
        }
        loaded++;
      } else if (line == "[Log Ratio Standard Deviation]") {
        if (!getNextLine(ifs, line, lineNum)) throw Fail("Error: malformed model file " + file);
        gcStd = atof(trim(line).c_str());
        loaded++;
      }
    }
    if (loaded < 9) throw Fail("Error: corrupted model file " + file + ", failed to load some parameters!");
  }

  // Matrix<double>::normalize(0), lib/matrix/Matrix.h:495-501 (+ sumCols :328-336), rows of width w
  static void normalizeRows(double* m, size_t rows, size_t w) {
    for (size_t i = 0; i < rows; i++) {
      double sum = 0;
      for (size_t j = 0; j < w; j++) sum += m[i * w + j];
      for (size_t j = 0; j < w; j++) m[i * w + j] /= (ZERO_FINAL + sum);
    }
  }
  // Matrix<double>::cumsum, lib/matrix/Matrix.h:506-522
  static void cumsumRows(double* m, size_t rows, size_t w) {
    for (size_t i = 0; i < rows; i++)
      for (size_t j = 1; j < w; j++) m[i * w + j] = m[i * w + j] + m[i * w + j - 1];
  }
  static double normpdf(double x, double mu, double sigma) {  // lib/mydefine/MyDefine.cpp:53-56
    double PI = 3.1415926;
    return exp(-pow(x - mu, 2) / (2 * pow(sigma, 2))) / (sqrt(2 * PI) * sigma);
  }

  // Profile::normParas(true), Profile.cpp:836-932
  void normParas() {
    for (int i = 0; i < kmerCount; i++) {
      int indx = baseIndex(kmers[i][kmer - 1]);
      for (vector<double>* sd : {&subs1, &subs2}) {
        double* m = sd->data() + (size_t)i * bins * N;
        normalizeRows(m, bins, N);
        for (int j = 0; j < bins; j++) {
          double sum = 0;
          for (int k = 0; k < N; k++) sum += m[j * N + k];
          if (sum < ZERO_FINAL) m[j * N + indx] = 1;
        }
      }
    }
    normalizeRows(qual.data(), (size_t)N * N * bins, nQual);
    if (paired && stdISize > 0) {
      int meanInsertSize = insertSize + 1;
      int intervalLen = 6 * stdISize;
      int minInsertSize = std::max(meanInsertSize - intervalLen / 2, readLength);
      int maxInsertSize = 2 * meanInsertSize - minInsertSize;
      int cols = maxInsertSize - minInsertSize + 1;
      if (cols <= 0) throw Fail("insert-size alphabet is empty");
      iSizeAlphabet.resize(cols);
      for (int i = 0; i < cols; i++) iSizeAlphabet[i] = minInsertSize++;
      iSizeCdf.resize(cols);
      for (int i = 0; i < cols; i++) iSizeCdf[i] = normpdf(iSizeAlphabet[i], meanInsertSize, stdISize);
      normalizeRows(iSizeCdf.data(), 1, cols);
    }
  }
  // Profile::initCDFs, Profile.cpp:1367-1434
  void initCDFs(const Rng& rng) {
    insCdf = insFreqs; cumsumRows(insCdf.data(), 1, insCdf.size());
    delCdf = delFreqs; cumsumRows(delCdf.data(), 1, delCdf.size());
    normalizeRows(qual.data(), (size_t)N * N * bins, nQual);   // second normalize, :1397
    cumsumRows(qual.data(), (size_t)N * N * bins, nQual);
    if (paired && stdISize > 0) cumsumRows(iSizeCdf.data(), 1, iSizeCdf.size());
    if (!rng.philox) {
      for (int l = 0; l < 101; l++) {
        gcGen.push_back(std::default_random_engine(rng.clockSeed));
        gcNorm.push_back(std::normal_distribution<double>(gcMeans[l], gcStd));
      }
    }
    cumsumRows(subs1.data(), (size_t)kmerCount * bins, N);
    hasSub2 = paired && stdISize > 0;
    if (hasSub2) cumsumRows(subs2.data(), (size_t)kmerCount * bins, N);
  }
  void train(const string& file, const Rng& rng) {  // Profile.cpp:1436-1440
    load(file);
    normParas();
    initCDFs(rng);
    countIndelDraws();
    if (rng.philox) { buildIntegerTables(); buildNormalQuantiles(); }
  }
  int maxInsertSize() const { return iSizeAlphabet.empty() ? insertSize : iSizeAlphabet.back(); }

  // Philox mode: the standard normal behind the GC factor comes from a fixed quantile table, so that a GPU can
  // reproduce it bit for bit without log / cos: NQ = 2^14 cells of equal probability, knots Q[k] = Phi^-1(k / NQ)
  // (Q[0] = Phi^-1(1 / (4 NQ)), Q[NQ - k] = -Q[k]), found by 64 bisection steps on 0.5 * erfc(-z / sqrt 2) in fp64.
  // A 32-bit draw x picks cell x >> 18 and interpolates linearly inside it at t = (2 (x & 0x3FFFF) + 1) / 2^19:
  //   z = Q[k] + (Q[k+1] - Q[k]) * t          (three fp64 operations, each rounded: no fused multiply-add)
  // The result is a normal variate up to the piecewise-linear quantile (total variation ~1e-5, tails end at 4.3 sigma).
  static const int kNormCells = 1 << 14;
  vector<double> gcQ;
  void buildNormalQuantiles() {
    const int N = kNormCells;
    gcQ.assign(N + 1, 0.0);
    auto quantile = [](double p) {
      double lo = -10.0, hi = 0.0;
      for (int it = 0; it < 64; it++) {
        const double mid = 0.5 * (lo + hi);
        if (0.5 * erfc(-mid * 0.7071067811865476) < p) lo = mid; else hi = mid;
      }
      return 0.5 * (lo + hi);
    };
    for (int k = 0; k < N / 2; k++) {
      const double q = quantile(k == 0 ? 0.25 / N : (double)k / N);
      gcQ[k] = q;
      gcQ[N - k] = -q;
    }
  }
  // Profile::getGCFactor, Profile.cpp:1507-1517.  philox: draws addressed by (popu, chr, seg ordinal, window
  // ordinal, attempt) through the quantile table above.
  double getGCFactor(int gc, Rng& rng, uint32_t ctx24, uint32_t segOrd, uint32_t winOrd) {
    if (gc < 0 || gc > 100) return 0;
    if (!rng.philox) {
      double v = gcNorm[gc](gcGen[gc]);
      while (v < 0) v = gcNorm[gc](gcGen[gc]);
      return v;
    }
    for (uint32_t a = 0;; a++) {
      Philox4 o = philox4x32_10(winOrd, a, segOrd, KIND_GC | (ctx24 << 8), rng.k0, rng.k1);
      const uint32_t k = o.v[0] >> 18, f = o.v[0] & 0x3FFFFu;
      const double t = (double)(2u * f + 1u) * (1.0 / 524288.0);
      const double d = gcQ[k + 1] - gcQ[k];
      const double z = gcQ[k] + d * t;
      const double v = gcMeans[gc] + gcStd * z;
      if (v >= 0) return v;
    }
  }
};

// Per-read RNG context for Profile::predict
struct ReadCtx {
  Rng* rng;
  uint32_t batch;  // batch id (philox)
  uint32_t slot;   // pair slot in batch (philox)
  int mate;        // 0: read 1 (or SE), 1: read 2
  uint32_t ctx24() const { return ((uint32_t)mate << 23) | (batch & 0xFFFFu); }
  uint32_t indel(int j, int which) {  // which: 0 insert test, 1 deletion test  (real stream; mt mode only)
    (void)j; (void)which;
    return rng->realGen();
  }
  // philox mode (Profile::buildIndelGaps): the e-th candidate of the read, looked for from template position `start`.
  // Call (KIND_INDEL, c0 = slot, c1 = e, c2 = 0): x = words (1,0) -> distance, y = words (3,2) -> kind.
  // Returns the candidate's position (n: none before the read's end) and its kind (1 insertion, 2 deletion).
  int indel_candidate(uint32_t e, int start, int n, const vector<uint64_t>& gapS, uint64_t evA, uint64_t evB, int* kind) {
    const uint64_t x = ((uint64_t)rng->ph(KIND_INDEL, ctx24(), slot, e, 0, 1) << 32) | rng->ph(KIND_INDEL, ctx24(), slot, e, 0, 0);
    const uint64_t y = ((uint64_t)rng->ph(KIND_INDEL, ctx24(), slot, e, 0, 3) << 32) | rng->ph(KIND_INDEL, ctx24(), slot, e, 0, 2);
    const int room = n - start;  // >= 1
    if (x < gapS[(size_t)room]) return n;
    int g = 0;
    while (g + 1 < room && x < gapS[(size_t)g + 1]) g++;
    *kind = (uint64_t)(((unsigned __int128)y * evB) >> 64) < evA ? 1 : 2;
    return start + g;
  }
  uint32_t aux(int j, int f, bool intStream) {  // f=0 indel length (real), f>=1 inserted base f-1 (int)
    if (!rng->philox) return intStream ? rng->intGen() : rng->realGen();
    return rng->ph(KIND_AUX, ctx24(), slot, (uint32_t)j, (uint32_t)(f >> 2), f & 3);
  }
  // philox mode: output position i owns word i%4 of two calls, (i/4, c2 = 0) "heads" and (i/4, c2 = 1) "tails":
  //   substitution draw xs = heads[31:16] << 16 | tails[31:16]
  //   quality draw      xq = heads[15:0]  << 16 | tails[15:0]
  // Both are uniform 32-bit values.  The split exists for the GPU: almost every base is decided by the heads alone
  // (no substitution / which side of an alias column), so the tails call is made only for the rare base that needs it.
  uint32_t cachedCall = 0xFFFFFFFFu;
  int cachedMate = -1;
  Philox4 heads{}, tails{};
  uint32_t base(int i, int which, bool intStream) {  // which: 0 substitution, 1 quality
    if (!rng->philox) return intStream ? rng->intGen() : rng->realGen();
    const uint32_t call = (uint32_t)(i >> 2), l = (uint32_t)i & 3u;
    if (call != cachedCall || mate != cachedMate) {
      cachedMate = mate;
      heads = philox4x32_r(kBaseRounds, slot, call, 0, KIND_BASE | (ctx24() << 8), rng->k0, rng->k1);
      tails = philox4x32_r(kBaseRounds, slot, call, 1, KIND_BASE | (ctx24() << 8), rng->k0, rng->k1);
      cachedCall = call;
    }
    const uint32_t wh = heads.v[l], wt = tails.v[l];
    return which == 0 ? (wh & 0xFFFF0000u) | (wt >> 16) : (wh << 16) | (wt & 0xFFFFu);
  }
};

// Profile::predict, Profile.cpp:1586-1701 (with getIndelSeq :1556-1574, getSubBaseIndx1/2
// :1527-1554, getBaseQuality :1576-1580, getRandBaseQuality :1582-1584).
// Returns n' and fills bases/quals (n' chars each).
static int predict(const Profile& P, const char* refSeq, int n, int isRead1, ReadCtx& rc,
                   string& outBases, string& outQuals) {
  const int N = P.N, kmer = P.kmer, binCount = P.bins;
  map<int, vector<int>> indelBaseIndxs;
  vector<int> indelLens;
  int indelLength = 0;
  // philox mode: the next candidate position and its kind (drawn ahead, see Profile::buildIndelGaps)
  uint32_t candOrd = 0;
  int candPos = n, candKind = 0;
  if (rc.rng->philox && n > 0) candPos = rc.indel_candidate(candOrd++, 0, n, P.gapS, P.evA, P.evB, &candKind);
  for (int j = 0; j < n;) {
    // getIndelSeq(indelBaseIndxs[j])
    vector<int>& baseIndxs = indelBaseIndxs[j];
    baseIndxs.clear();
    int k = 0;
    bool isIns, isDel = false;
    const bool atCand = rc.rng->philox && j == candPos;
    if (rc.rng->philox) {
      isIns = atCand && candKind == 1;
      isDel = atCand && candKind == 2;
    } else {
      double p = u32ToDouble(rc.indel(j, 0), 0, 1);
      isIns = p <= P.insertRate;
      if (!isIns) {
        p = u32ToDouble(rc.indel(j, 1), 0, 1);
        isDel = p < P.delRate / (1 - P.insertRate);
      }
    }
    if (isIns) {
      int len = randIndxFrom(rc.aux(j, 0, false), P.insCdf.data(), (int)P.insCdf.size());
      for (int i = 0; i < len; i++) baseIndxs.push_back((int)u32ToInteger(rc.aux(j, 1 + i, true), 0, N - 1));
      k = len;
    } else if (isDel) {
      k = randIndxFrom(rc.aux(j, 0, false), P.delCdf.data(), (int)P.delCdf.size());
    }
    if (baseIndxs.empty() && k > 0) {
      k = std::min(n - j, k);
      indelLength -= k;
      indelLens.push_back(k);
      for (int i = 1; i < k; i++) indelLens.push_back(0);
      j += k;
    } else {
      indelLength += k;
      j++;
      indelLens.push_back(k);
    }
    if (atCand) candPos = j < n ? rc.indel_candidate(candOrd++, j, n, P.gapS, P.evA, P.evB, &candKind) : n;
  }
  if (n + indelLength < 50) {
    indelLength = 0;
    indelBaseIndxs.clear();
    indelLens.assign(n, 0);
  }
  string sourceSeq;
  sourceSeq.reserve(n + indelLength);
  for (int j = 0; j < n;) {
    if (indelBaseIndxs[j].empty() && indelLens[j] > 0) {  // deletion
      j += indelLens[j];
      continue;
    } else if (indelLens[j] == 0) {
      sourceSeq.push_back(refSeq[j]);
      j++;
    } else {  // insertion
      sourceSeq.push_back(refSeq[j]);
      for (int b : indelBaseIndxs[j]) sourceSeq.push_back(P.bases[b]);
      j++;
    }
  }
  n += indelLength;
  string seq(kmer - 1, 'X');
  seq += sourceSeq;
  outBases.assign(n, '?');
  outQuals.assign(n, '?');
  const vector<double>& subs = (isRead1 || !P.hasSub2) ? P.subs1 : P.subs2;
  for (int j = 0; j < n; j++) {
    int refIndx = P.baseIndex(sourceSeq[j]);
    int binIndx = j * binCount / n;
    int k;
    int kmerIndx = P.getKmerIndx(&seq[j], kmer);
    if (kmerIndx == -1) {
      k = P.baseIndex(seq[j + kmer - 1]);
    } else {
      const uint32_t xs = rc.base(j, 0, false);
      k = rc.rng->philox ? P.subSample(!(isRead1 || !P.hasSub2), kmerIndx, binIndx, xs)
                         : randIndxFrom(xs, subs.data() + ((size_t)kmerIndx * binCount + binIndx) * N, N);
    }
    if (k == -1) {
      outBases[j] = 'N';
      outQuals[j] = (char)u32ToInteger(rc.base(j, 1, true), P.minQ, P.minQ + 20);
    } else {
      outBases[j] = P.bases[k];
      int bp = refIndx * N + k;
      const uint32_t xq = rc.base(j, 1, false);
      int qi = rc.rng->philox ? P.qualSample(bp, binIndx, xq)
                              : randIndxFrom(xq, P.qual.data() + ((size_t)bp * binCount + binIndx) * P.nQual, P.nQual);
      outQuals[j] = (char)(P.minQ + qi);
    }
  }
  return n;
}

// ---------------------------------------------------------------------------------------------
// FASTA (lib/fastahack/Fasta.cpp).  The reference fseeks per segment through a .fai; the oracle
// loads whole contigs (newlines stripped), which yields the same bytes for in-range requests.
// ---------------------------------------------------------------------------------------------
struct Fasta {
  vector<string> names;          // keyed names: first token, chr/chrom stripped (Fasta.cpp:58-69)
  map<string, string> seqs;      // raw (case preserved)
  void open(const string& file) {
    std::ifstream ifs(file.c_str());
    if (!ifs.is_open()) throw Fail("could not open " + file);
    string line, cur, ignored;
    string* dst = nullptr;
    while (std::getline(ifs, line)) {
      if (!line.empty() && line[0] == ';') continue;
      if (!line.empty() && (line[0] == '>' || line[0] == '@')) {
        string full = line.substr(1);
        vector<string> toks = split_any(full, " \t");
        string name = abbrOfChr(toks.empty() ? string("") : toks[0]);
        // A name met again: fastahack lists it once more (sequenceNames.push_back, Fasta.cpp:66,197) but its map keeps the
        // FIRST entry (std::map::insert does not overwrite, Fasta.cpp:67,198), and the index it writes and reads back is
        // sorted by offset (Fasta.cpp:84-97) -- the row of the first sequence, twice, at the first one's place.  The
        // chromosome list then holds the name twice, both resolving to the first sequence; the later one is never read.
        if (seqs.count(name)) {
          names.insert(std::find(names.begin(), names.end(), name) + 1, name);
          dst = &ignored;
        } else {
          names.push_back(name);
          dst = &seqs[name];
        }
        dst->clear();
      } else if (dst) {
        dst->append(line);
      }
    }
  }
  long length(const string& chr) const {
    auto it = seqs.find(chr);
    return it == seqs.end() ? 0 : (long)it->second.size();
  }
};

// ---------------------------------------------------------------------------------------------
// Variants
// ---------------------------------------------------------------------------------------------
enum VarType { HET, HOMO };
struct CNV { long spos, epos; float CN, mCN; };
struct SNV { long pos; char ref, alt; VarType type; };
struct Insert { long pos; string seq; VarType type; };
struct Deletion { long pos; int length; VarType type; };
struct SNP { long long pos; char nucleotide; };
struct Target { long spos, epos; };

static char complementOf(char c) {  // lib/snp/snp.cpp:84-97
  switch (c) {
    case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
    case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
    case 'N': return 'N'; default: return 'N';
  }
}

struct Genome;

// ---------------------------------------------------------------------------------------------
// Segment (lib/segment/Segment.cpp)
// ---------------------------------------------------------------------------------------------
struct Segment {
  static const unsigned int segMaxSize = 1000000;  // Segment.cpp:15
  static const unsigned int fragSize = 1000;       // Segment.cpp:16
  int segIndx; string chr; long start, end; int CN, mCN;
  bool hasSeqs = false;            // segSequences != NULL
  vector<vector<char>> haps;       // NUL-terminated; size()==0 <=> NULL entry
  vector<int> mIndx, seqReps;
  vector<int> targetIndxs;
  long readCount = 0;
  vector<long> fragStartPos, fragEndPos;
  vector<int> fragRCs;
  vector<double> fragWeights;
  vector<int> hapIndxs;

  unsigned int refSize() const { return (unsigned int)(end - start + 1); }
  size_t hapLen(int i) const { return haps[i].empty() ? 0 : haps[i].size() - 1; }
  bool hapNull(int i) const { return haps[i].empty(); }
  unsigned int seqSize() const {  // Segment.cpp:643-655
    if (!hasSeqs) return CN * refSize();
    unsigned int s = 0;
    for (size_t i = 0; i < haps.size(); i++) s += (unsigned int)hapLen((int)i);
    return s;
  }
  void clearSeqs() { haps.clear(); hasSeqs = false; }
};

// lib/mydefine/MyDefine.cpp:279-303 on a C string
static int calculateGCPercent(const char* s) {
  int gc = 0, nc = 0;
  if (s == nullptr || s[0] == '\0') return 0;
  int n = (int)strlen(s);
  for (int i = 0; i < n; i++) {
    if (s[i] == 'G' || s[i] == 'C') gc++;
    else if (s[i] == 'N') nc++;
  }
  if (nc > 0) return -1;
  return 100 * gc / (n - nc);
}

struct Genome {
  Config& cfg;
  Profile& prof;
  Rng& rng;
  Fasta fa;
  vector<string> chromosomes;
  map<string, map<string, vector<CNV>>> cnvs;
  map<string, map<string, vector<SNV>>> snvs;
  map<string, map<string, vector<Insert>>> inserts;
  map<string, map<string, vector<Deletion>>> dels;
  map<string, vector<SNP>> snps;
  map<string, vector<Target>> outTargets;
  vector<vector<float>> mixProps;
  map<string, map<string, vector<Segment>>> segments;
  string curPopu;
  uint64_t readsWritten = 0;
  uint32_t batchId = 0;

  Genome(Config& c, Profile& p, Rng& r) : cfg(c), prof(p), rng(r) {}

  long chromLen(const string& chr) const {  // Genome.cpp:383-396
    if (std::find(chromosomes.begin(), chromosomes.end(), chr) == chromosomes.end()) return 0;
    return fa.length(chr);
  }
  long genomeLength() const { long n = 0; for (auto& c : chromosomes) n += chromLen(c); return n; }
  long targetLength() const {  // Genome.cpp:405-419
    if (!outTargets.empty()) {
      long n = 0;
      for (auto& kv : outTargets) for (auto& t : kv.second) n += t.epos - t.spos + 1;
      return n;
    }
    return genomeLength();
  }
  int popuIdx(const string& p) const {
    return (int)(std::find(cfg.popuNames.begin(), cfg.popuNames.end(), p) - cfg.popuNames.begin());
  }
  int chrIdx(const string& c) const {  // index into the FASTA contig order (stable across target re-ordering)
    return (int)(std::find(fa.names.begin(), fa.names.end(), c) - fa.names.begin());
  }
  uint32_t hostCtx(const string& popu, const string& chr) const {
    return ((uint32_t)popuIdx(popu) << 16) | ((uint32_t)chrIdx(chr) & 0xFFFFu);
  }

  // Genome.cpp:41-206
  void loadAbers() {
    string file = cfg.s["variation"];
    if (file.empty()) return;
    std::ifstream ifs(file.c_str());
    if (!ifs.is_open()) throw Fail("can not open file " + file);
    string line;
    int lineNum = 0;
    auto bad = [&](const string& what) { return Fail("ERROR: " + what + " at line " + std::to_string(lineNum) + " in file " + file); };
    while (std::getline(ifs, line)) {
      lineNum++;
      if (line.empty()) continue;
      vector<string> f = split(line, '\t');
      string t = f[0];
      auto popuOf = [&](const string& name) {
        if (std::find(cfg.popuNames.begin(), cfg.popuNames.end(), name) == cfg.popuNames.end())
          throw bad("unrecognized population identifier");
        return name;
      };
      auto typeOf = [&](const string& code) {
        if (code != "homo" && code != "het") throw bad("unrecognized variant type");
        return code == "het" ? HET : HOMO;
      };
      if (t == "c") {
        if (f.size() != 7) throw bad("wrong number of fields");
        string popu = popuOf(f[1]);
        string chr = abbrOfChr(f[2]);
        long spos = atol(f[3].c_str()), epos = atol(f[4].c_str());
        float cn = atof(f[5].c_str()), mcn = atof(f[6].c_str());
        if (cn < mcn) throw bad("total copy number should be not lower than major copy number");
        if (cn - mcn > mcn) mcn = cn - mcn;
        cnvs[popu][chr].push_back(CNV{spos, epos, cn, mcn});
      } else if (t == "s") {
        if (f.size() != 7) throw bad("wrong number of fields");
        string popu = popuOf(f[1]);
        string chr = abbrOfChr(f[2]);
        long pos = atol(f[3].c_str());
        char ref = f[4].at(0), alt = f[5].at(0);
        if (ref == alt) throw bad("the mutated allele should be not same as the reference allele");
        snvs[popu][chr].push_back(SNV{pos, ref, alt, typeOf(f[6])});
      } else if (t == "i") {
        if (f.size() != 6) throw bad("wrong number of fields");
        string popu = popuOf(f[1]);
        string chr = abbrOfChr(f[2]);
        inserts[popu][chr].push_back(Insert{atol(f[3].c_str()), f[4], typeOf(f[5])});
      } else if (t == "d") {
        if (f.size() != 6) throw bad("wrong number of fields");
        string popu = popuOf(f[1]);
        string chr = abbrOfChr(f[2]);
        dels[popu][chr].push_back(Deletion{atol(f[3].c_str()), atoi(f[4].c_str()), typeOf(f[5])});
      } else {
        throw bad("unrecognized aberraton type");
      }
    }
  }

  // lib/snp/snp.cpp:147-203 + SNP::SNP :12-35
  void loadSNPs() {
    string file = cfg.s["snp"];
    if (file.empty()) return;
    FILE* fp = fopen(file.c_str(), "r");
    if (!fp) throw Fail("can not open SNP file " + file);
    char buf[1000];
    while (fgets(buf, 1000, fp)) {
      vector<char*> elems;
      // lib/split/split.cpp:36-48 (in-place split on '\t')
      char* s = buf;
      int j = 0;
      for (int i = 0; s[i] != '\0'; i++) {
        if (s[i] == '\t') { s[i] = '\0'; elems.push_back(&s[j]); j = i + 1; }
      }
      elems.push_back(&s[j]);
      if (elems.size() != 6) continue;  // reference only warns
      long long position = atoll(elems[2]);
      string observed = elems[3];
      char strand = *elems[4];
      char ref = *elems[5];
      vector<string> strs = split(observed, '/');
      if (strand == '-') ref = complementOf(ref);
      char nucleotide;
      if (strs[0].at(0) == ref) nucleotide = strs[1].at(0);
      else nucleotide = strs[0].at(0);
      if (strand == '-') nucleotide = complementOf(nucleotide);
      snps[abbrOfChr(elems[1])].push_back(SNP{position, nucleotide});
    }
    fclose(fp);
  }

  // Genome.cpp:217-236
  void loadRefSeq() {
    string tmp = cfg.s["ref"];
    if (tmp.empty()) throw Fail("genome sequence file not specified!");
    if (tmp.size() >= 3 && tmp.substr(tmp.size() - 3) == ".gz") {
      string cmd = "gzip -cd " + tmp + " > " + tmp.substr(0, tmp.size() - 3);
      if (system(cmd.c_str()) != 0) throw Fail("gzip failed on " + tmp);
      tmp = tmp.substr(0, tmp.size() - 3);
    }
    fa.open(tmp);
    chromosomes = fa.names;
    if (chromosomes.empty()) throw Fail("ERROR: reference sequence cannot be empty!");
  }

  // Genome.cpp:238-295
  void loadTargets() {
    string file = cfg.s["target"];
    if (file.empty()) return;
    std::ifstream ifs(file.c_str());
    if (!ifs.is_open()) throw Fail("can not open target file " + file);
    string line;
    int lineNum = 0;
    while (std::getline(ifs, line)) {
      lineNum++;
      vector<string> f = split(line, '\t');
      if (f.size() < 3) throw Fail("ERROR: line " + std::to_string(lineNum) + " should have at least 3 fields in file " + file);
      string chr = abbrOfChr(f[0]);
      long chrLen = chromLen(chr);
      if (chrLen <= 0) continue;
      Target t;
      t.spos = std::max((long)1, atol(f[1].c_str()) - 50 + 1);
      long tmp;
      if (atol(f[2].c_str()) <= 0) tmp = chrLen - (-atol(f[2].c_str())) % chrLen;
      else tmp = atol(f[2].c_str());
      t.epos = std::min(chrLen, tmp + 50);
      outTargets[chr].push_back(t);
    }
  }
  // Genome.cpp:684-739
  void divideTargets() {
    unsigned int targetMaxSize = Segment::fragSize;
    map<string, vector<Target>> nt;
    for (auto& kv : outTargets) {
      for (const Target& target : kv.second) {
        long spos = target.spos;
        long tsize = target.epos - target.spos + 1;
        int k = tsize / targetMaxSize;
        for (int i = 0; i < k; i++) {
          Target t;
          t.spos = spos;
          t.epos = (i == k - 1) ? target.epos : spos + targetMaxSize - 1;
          spos = t.epos + 1;
          nt[kv.first].push_back(t);
        }
        if (spos <= target.epos) nt[kv.first].push_back(Target{spos, target.epos});
      }
    }
    outTargets = nt;
  }
  // Genome.cpp:297-339
  void loadAbundance() {
    string file = cfg.s["abundance"];
    if (file.empty()) return;
    std::ifstream ifs(file.c_str());
    if (!ifs.is_open()) throw Fail("can not open abundance file " + file);
    string line;
    int lineNum = 0;
    while (std::getline(ifs, line)) {
      lineNum++;
      vector<string> f = split(line, '\t');
      if (f.size() != cfg.popuNames.size()) throw Fail("ERROR: line " + std::to_string(lineNum) + " has wrong number of fields in file " + file);
      vector<float> props;
      float sum = 0;
      for (auto& x : f) { float p = atof(x.c_str()); sum += p; props.push_back(p); }
      if (fabs(1 - sum) > 0.001) throw Fail("ERROR: the sum of abundances is not equal to one at line " + std::to_string(lineNum));
      mixProps.push_back(props);
    }
  }
  void loadData() {  // Genome.cpp:17-30
    loadAbers();
    loadSNPs();
    loadRefSeq();
    loadTargets();
    divideTargets();
    loadAbundance();
  }

  // Segment ctor + initTargets, Segment.cpp:42-79
  Segment makeSegment(int segIndx, const string& chr, long s, long e, int CN, int mCN) {
    Segment g;
    g.segIndx = segIndx; g.chr = chr; g.start = s; g.end = e; g.CN = CN; g.mCN = mCN;
    if (!outTargets.empty()) {
      vector<Target>& ts = outTargets[chr];
      for (size_t i = 0; i < ts.size(); i++) {
        long spos = ts[i].spos, epos = ts[i].epos;
        if ((spos >= s && spos <= e) || (epos >= s && epos <= e) || (spos < s && epos > e)) g.targetIndxs.push_back((int)i);
      }
    }
    return g;
  }
  // Genome.cpp:741-763
  void divideSegment(const string& popu, const string& chr, long segStartPos, long segEndPos, int CN, int mCN, int& segIndx) {
    unsigned int segMaxSize = Segment::segMaxSize;
    vector<Segment>& out = segments[popu][chr];
    long segSize = segEndPos - segStartPos + 1;
    int n = segSize / segMaxSize;
    unsigned int m = segSize - n * segMaxSize;
    for (int i = 0; i < n; i++) {
      if (i == n - 1 && m < segMaxSize / 2) {
        out.push_back(makeSegment(segIndx++, chr, segStartPos, segEndPos, CN, mCN));
        segStartPos = segEndPos + 1;
      } else {
        out.push_back(makeSegment(segIndx++, chr, segStartPos, segStartPos + segMaxSize - 1, CN, mCN));
        segStartPos += segMaxSize;
      }
    }
    if (segStartPos <= segEndPos) out.push_back(makeSegment(segIndx++, chr, segStartPos, segEndPos, CN, mCN));
  }
  // Genome.cpp:634-682
  void generateSegments() {
    int ploidy = cfg.i["ploidy"];
    int mCN = (int)ceil((float)ploidy / 2);
    if (!outTargets.empty()) {
      chromosomes.clear();
      for (auto& kv : outTargets)
        if (std::find(chromosomes.begin(), chromosomes.end(), kv.first) == chromosomes.end()) chromosomes.push_back(kv.first);
    }
    for (const string& popu : cfg.popuNames) {
      for (const string& chr : chromosomes) {
        int segIndx = 0;
        vector<CNV>& cs = cnvs[popu][chr];
        long segStartPos = 1;
        segments[popu][chr];
        for (size_t k = 0; k < cs.size(); k++) {
          if (segStartPos > chromLen(chr)) break;
          CNV cnv = cs[k];
          cnv.epos = std::min(cnv.epos, chromLen(chr));
          if (segStartPos < cnv.spos) divideSegment(popu, chr, segStartPos, cnv.spos - 1, ploidy, mCN, segIndx);
          divideSegment(popu, chr, cnv.spos, cnv.epos, (int)cnv.CN, (int)cnv.mCN, segIndx);
          segStartPos = cnv.epos + 1;
        }
        if (segStartPos <= chromLen(chr)) divideSegment(popu, chr, segStartPos, chromLen(chr), ploidy, mCN, segIndx);
      }
    }
  }

  // lib/mydefine/MyDefine.cpp:192-194 (rand()-based) used by Segment.cpp:153,170,190,199
  long hapRandomInteger(long start, long end, uint32_t ctx24, uint32_t segOrd, uint32_t& drawIdx) {
    if (!rng.philox) return (long)(start + (end - start) * (rand() / (RAND_MAX + 1.0)));
    uint32_t x = rng.ph(KIND_HAP, ctx24, segOrd, drawIdx++, 0, 0);
    return u32ToInteger(x, start, end);
  }

  // Segment::generateSegSequences, Segment.cpp:124-460
  void generateSegSequences(Segment& g, uint32_t segOrd) {
    int ploidy = cfg.i["ploidy"];
    vector<SNP>& snpsOfChr = snps[g.chr];
    vector<SNV>& snvsOfChr = snvs[curPopu][g.chr];
    vector<Insert>& insertsOfChr = inserts[curPopu][g.chr];
    vector<Deletion>& delsOfChr = dels[curPopu][g.chr];
    g.clearSeqs();
    if (g.CN == 0) return;
    const string& contig = fa.seqs[g.chr];
    if ((long)contig.size() < g.start) return;
    string refSeq = contig.substr(g.start - 1, g.refSize());
    for (auto& c : refSeq) c = (char)toupper((unsigned char)c);
    unsigned int refSize = (unsigned int)refSeq.size();
    uint32_t ctx = hostCtx(curPopu, g.chr);
    uint32_t drawIdx = 0;
    auto inM = [&](int j) { return std::find(g.mIndx.begin(), g.mIndx.end(), j) != g.mIndx.end(); };

    if (g.mIndx.empty()) {
      if (g.CN < ploidy) {
        for (int i = 0; i < g.CN; i++) {
          while (1) {
            int j = (int)hapRandomInteger(0, ploidy, ctx, segOrd, drawIdx);
            if (std::find(g.seqReps.begin(), g.seqReps.end(), j) == g.seqReps.end()) { g.seqReps.push_back(j); break; }
          }
        }
        for (int i = 0; i < g.mCN; i++) g.mIndx.push_back(g.seqReps[i]);
      } else {
        for (int i = 0; i < ploidy; i++) g.seqReps.push_back(1);
        int n = g.CN - ploidy;
        int k = (int)hapRandomInteger(0, ploidy, ctx, segOrd, drawIdx);
        int i;
        for (i = n; i >= 0; i--) {
          if (g.seqReps[k] + i == g.mCN) {
            g.seqReps[k] += i;
            g.mIndx.push_back(k);
            break;
          } else if (g.seqReps[k] + i == g.CN - g.mCN) {
            g.seqReps[k] += i;
            for (int j = 0; j < ploidy; j++) if (j != k) g.mIndx.push_back(j);
            break;
          }
        }
        if (i >= 0) {
          n -= i;
          // the reference spins forever here when ploidy == 1 (every draw equals k, Segment.cpp:188-197); a
          // checker that hangs is of no use, so this one input is refused (the GPU host refuses it too)
          if (n > 0 && ploidy == 1) throw Fail("ERROR: a copy-number gain cannot be placed on a haploid genome (ploidy = 1)");
          while (n > 0) {
            int j = (int)hapRandomInteger(0, ploidy, ctx, segOrd, drawIdx);
            if (j != k) { g.seqReps[j]++; n--; }
          }
        } else {
          while (n > 0) {
            int j = (int)hapRandomInteger(0, ploidy, ctx, segOrd, drawIdx);
            g.seqReps[j]++;
            n--;
          }
          for (i = 0; i < ploidy; i++) g.mIndx.push_back(i);
        }
      }
    }

    vector<string> segSeqs;
    if (g.CN < ploidy) {
      for (int i = 0; i < ploidy; i++) {
        if (std::find(g.seqReps.begin(), g.seqReps.end(), i) != g.seqReps.end()) segSeqs.push_back(refSeq);
        else segSeqs.push_back("");
      }
    } else {
      for (int i = 0; i < ploidy; i++) {
        string tmp;
        for (int j = 0; j < g.seqReps[i]; j++) tmp += refSeq;
        segSeqs.push_back(tmp);
      }
    }

    // SNPs, :234-265
    int k = 0;
    for (size_t i = 0; i < snpsOfChr.size(); i++) {
      long pos = (long)snpsOfChr[i].pos;
      if (pos >= g.start && pos <= g.end) {
        int sindx = (int)(pos - g.start);
        for (int j = 0; j < ploidy; j++) {
          bool major = inM(j);
          if ((k == 0) != major) continue;
          string& s = segSeqs[j];
          unsigned int segLen = (unsigned int)s.length();
          for (unsigned int t = 0; t < segLen / refSize; t++) s[sindx + t * refSize] = snpsOfChr[i].nucleotide;
        }
        k = (k + 1) % 2;
      }
    }
    // SNVs, :267-311
    k = 0;
    for (size_t i = 0; i < snvsOfChr.size(); i++) {
      const SNV& snv = snvsOfChr[i];
      if (snv.pos >= g.start && snv.pos <= g.end) {
        int sindx = (int)(snv.pos - g.start);
        for (int j = 0; j < ploidy; j++) {
          if (snv.type != HOMO) {
            bool major = inM(j);
            if ((k == 0) != major) continue;
          }
          string& s = segSeqs[j];
          unsigned int segLen = (unsigned int)s.length();
          for (unsigned int t = 0; t < segLen / refSize; t++) s[sindx + t * refSize] = snv.alt;
        }
        if (snv.type != HOMO) k = (k + 1) % 2;
      }
    }
    // Inserts, :313-370
    map<int, map<int, int>> insertsPerhaploidy;
    vector<int> insertLens(ploidy, 0);
    k = 0;
    for (size_t i = 0; i < insertsOfChr.size(); i++) {
      const Insert& ins = insertsOfChr[i];
      if (ins.pos >= g.start && ins.pos <= g.end) {
        int sindx = (int)(ins.pos + 1 - g.start);
        for (int j = 0; j < ploidy; j++) {
          if (ins.type != HOMO) {
            bool major = inM(j);
            if ((k == 0 && !major) || (k == 1 && major)) continue;
          }
          int offset = 0;
          map<int, int>& insertedSeq = insertsPerhaploidy[j];
          for (auto& kv : insertedSeq) if (kv.first <= sindx) offset += kv.second;
          string& s = segSeqs[j];
          int n = (int)(s.length() / (refSize + insertLens[j]));
          int len = (int)ins.seq.length();
          for (int t = 0; t < n; t++) s.insert(sindx + offset + t * (refSize + insertLens[j] + len), ins.seq);
          insertLens[j] += len;
          insertedSeq.insert(std::make_pair(sindx, len));
        }
        if (ins.type != HOMO) k = (k + 1) % 2;
      }
    }
    // Deletions, :372-444
    map<int, map<int, int>> delsPerhaploidy;
    vector<int> delLens(ploidy, 0);
    k = 0;
    for (size_t i = 0; i < delsOfChr.size(); i++) {
      const Deletion& del = delsOfChr[i];
      if (del.pos >= g.start && del.pos <= g.end) {
        int sindx = (int)(del.pos - g.start);
        int delLen = del.length;
        for (int j = 0; j < ploidy; j++) {
          if (del.type != HOMO) {
            bool major = inM(j);
            if ((k == 0 && !major) || (k == 1 && major)) continue;
          }
          int offset = 0;
          for (auto& kv : insertsPerhaploidy[j]) if (kv.first <= sindx) offset += kv.second;
          map<int, int>& delSeq = delsPerhaploidy[j];
          for (auto& kv : delSeq) if (kv.first <= sindx) offset -= kv.second;
          if (sindx + offset < 0) continue;
          string& s = segSeqs[j];
          int n = (int)(s.length() / (refSize + insertLens[j] - delLens[j]));
          for (int t = 0; t < n; t++) {
            size_t at = sindx + offset + t * (refSize + insertLens[j] - delLens[j] - delLen);
            if (at > s.size()) throw Fail("deletion outside haplotype (reference would throw std::out_of_range)");
            s.erase(at, delLen);
          }
          delLens[j] += delLen;
          delSeq.insert(std::make_pair(sindx, delLen));
        }
        if (del.type != HOMO) k = (k + 1) % 2;
      }
    }
    g.haps.assign(ploidy, vector<char>());
    for (int i = 0; i < ploidy; i++) {
      if (segSeqs[i].empty()) continue;
      vector<char>& h = g.haps[i];
      h.assign(segSeqs[i].begin(), segSeqs[i].end());
      for (auto& c : h) c = (char)toupper((unsigned char)c);
      h.push_back('\0');
    }
    g.hasSeqs = true;
  }

  // Segment::getWeightedLength, Segment.cpp:550-641
  double getWeightedLength(Segment& g, uint32_t segOrd) {
    double weightLen = 0;
    int ploidy = cfg.i["ploidy"];
    if (g.fragWeights.empty()) {
      int flag = 0;
      if (!g.hasSeqs) { flag = 1; generateSegSequences(g, segOrd); }
      if (!g.hasSeqs) {
        // CN == 0 (or contig shorter than the segment start): the reference dereferences NULL
        // here.  The oracle gives such a segment one zero-weight window so it draws no reads.
        g.fragStartPos.push_back(0); g.fragEndPos.push_back(0); g.fragWeights.push_back(0); g.hapIndxs.push_back(0);
        return 0;
      }
      unsigned int fragSize = Segment::fragSize;
      uint32_t ctx = hostCtx(curPopu, g.chr);
      uint32_t winOrd = 0;
      if (outTargets.empty()) {
        for (int i = 0; i < ploidy; i++) {
          if (g.hapNull(i)) continue;
          char* p = g.haps[i].data();
          size_t plen = g.hapLen(i);
          int k = (int)(plen / fragSize);
          for (int j = 0; j < k; j++) {
            long spos = (long)j * fragSize;
            long epos = (long)(j + 1) * fragSize - 1;
            char c = p[epos + 1];
            p[epos + 1] = '\0';
            int gc = calculateGCPercent(p + spos);
            p[epos + 1] = c;
            double weight = prof.getGCFactor(gc, rng, ctx, segOrd, winOrd++) / fragSize;
            g.fragStartPos.push_back(spos); g.fragEndPos.push_back(epos); g.fragWeights.push_back(weight); g.hapIndxs.push_back(i);
            weightLen += weight;
          }
          if ((size_t)k * fragSize < plen) {
            long spos = (long)k * fragSize;
            int gc = calculateGCPercent(p + spos);
            double weight = prof.getGCFactor(gc, rng, ctx, segOrd, winOrd++) * (plen - spos) / (fragSize * fragSize);
            g.fragStartPos.push_back(spos); g.fragEndPos.push_back((long)plen - 1); g.fragWeights.push_back(weight); g.hapIndxs.push_back(i);
            weightLen += weight;
          }
        }
      } else if (!g.targetIndxs.empty()) {
        vector<Target>& ts = outTargets[g.chr];
        for (int i = 0; i < ploidy; i++) {
          if (g.hapNull(i)) continue;
          char* p = g.haps[i].data();
          size_t plen = g.hapLen(i);
          int n = ((int)g.seqReps.size() < ploidy) ? 1 : g.seqReps[i];
          long refLen = (long)(plen / n);
          for (int k = 0; k < n; k++) {
            for (size_t j = 0; j < g.targetIndxs.size(); j++) {
              int m = g.targetIndxs[j];
              long spos = std::max(ts[m].spos, g.start) - g.start;
              long epos = std::min(ts[m].epos, g.start + refLen - 1) - g.start;
              long spos_k = (long)(spos + k * plen / n);
              long epos_k = (long)(epos + k * plen / n);
              if (epos_k + 1 < 0 || (size_t)(epos_k + 1) > plen || spos_k < 0 || (size_t)spos_k > plen)
                throw Fail("target window outside haplotype (reference reads out of bounds)");
              char c = p[epos_k + 1];
              p[epos_k + 1] = '\0';
              int gc = calculateGCPercent(p + spos_k);
              p[epos_k + 1] = c;
              double weight = prof.getGCFactor(gc, rng, ctx, segOrd, winOrd++) * (epos_k - spos_k + 1) / (fragSize * fragSize);
              g.fragStartPos.push_back(spos_k); g.fragEndPos.push_back(epos_k); g.fragWeights.push_back(weight); g.hapIndxs.push_back(i);
              weightLen += weight;
            }
          }
        }
      } else {
        g.fragStartPos.push_back(0); g.fragEndPos.push_back(0); g.fragWeights.push_back(0); g.hapIndxs.push_back(0);
      }
      if (flag == 1) g.clearSeqs();
    } else {
      for (double w : g.fragWeights) weightLen += w;
    }
    return weightLen;
  }

  // Segment::setReadCount, Segment.cpp:462-476
  void segSetReadCount(Segment& g, uint32_t segOrd, long readCount) {
    double totalWL = getWeightedLength(g, segOrd) + 2.2204e-16;
    long sum = 0;
    g.fragRCs.clear();
    for (size_t i = 0; i < g.fragWeights.size(); i++) {
      long rcv = to_long_x86(g.fragWeights[i] * readCount / totalWL);
      g.fragRCs.push_back((int)rcv);
      sum = wrap_add(sum, rcv);
    }
    if (sum < readCount) g.fragRCs[0] += (int)wrap_sub(readCount, sum);
    g.readCount = readCount;
  }

  // Genome::calculateACNs, Genome.cpp:765-781
  void calculateACNs(map<string, double>& ACNs) {
    for (auto& kv : segments) {
      long sum = 0;
      for (auto& kc : kv.second) for (auto& g : kc.second) sum += g.seqSize();
      ACNs[kv.first] = (double)sum / genomeLength();
    }
  }

  // Genome::setReadCounts, Genome.cpp:783-825
  void setReadCounts(const string& popu, long reads) {
    auto& segsOfPopu = segments[popu];
    map<string, double> chrWLens;
    double WL = 0;
    for (auto& chr : chromosomes) {
      vector<Segment>& cs = segsOfPopu[chr];
      double chrWL = 0;
      for (size_t j = 0; j < cs.size(); j++) chrWL += getWeightedLength(cs[j], (uint32_t)j);
      WL += chrWL;
      chrWLens[chr] = chrWL;
    }
    long curReads = 0, chrReads;
    for (size_t i = 0; i < chromosomes.size(); i++) {
      const string& chr = chromosomes[i];
      vector<Segment>& cs = segsOfPopu[chr];
      double chrWL = chrWLens[chr];
      if (i < chromosomes.size() - 1) chrReads = to_long_x86(reads * (chrWL / WL));
      else chrReads = wrap_sub(reads, curReads);
      long sum = 0;
      for (size_t j = 0; j < cs.size(); j++) {
        if (j < cs.size() - 1) {
          double share = getWeightedLength(cs[j], (uint32_t)j) / chrWL;   // 0/0 for a chromosome without weight: see to_long_x86
          long segReadCount = to_long_x86(share * chrReads);
          segSetReadCount(cs[j], (uint32_t)j, segReadCount);
          sum = wrap_add(sum, segReadCount);
        } else {
          segSetReadCount(cs[j], (uint32_t)j, wrap_sub(chrReads, sum));
        }
      }
      curReads = wrap_add(curReads, chrReads);
    }
  }

  // Genome::produceFragment, Genome.cpp:599-632
  bool produceFragment(vector<Segment>& cs, int startSegIndx, int segSeqIndx, int fragLen, string& out) {
    out.clear();
    if (fragLen <= 0) return false;
    size_t i;
    for (i = startSegIndx; i < cs.size(); i++) {
      if (!cs[i].hasSeqs || cs[i].hapNull(segSeqIndx)) continue;
      const char* segSeq = cs[i].haps[segSeqIndx].data();
      size_t sl = cs[i].hapLen(segSeqIndx);
      if (out.size() + sl >= (size_t)fragLen) {
        out.append(segSeq, fragLen - out.size());
        break;
      } else {
        out.append(segSeq, sl);
      }
    }
    return true;
  }
  // Segment::getFragSequence, Segment.cpp:1077-1103
  void getFragSequence(vector<Segment>& cs, Segment& g, int hap, long pos, int fragSize, string& s) {
    size_t hl = g.hapLen(hap);
    const char* h = g.haps[hap].data();
    if (hl - pos >= (size_t)fragSize) {
      s.assign(h + pos, fragSize);
    } else {
      int i = (int)(hl - pos);
      int k = fragSize - i;
      string p;
      s.assign(h + pos, i);
      if (produceFragment(cs, g.segIndx + 1, hap, k, p)) s += p;
    }
  }

  static void complementInPlace(char* s, int n) {  // Segment.cpp:81-103
    for (int i = 0; i < n; i++) {
      char c;
      switch (s[i]) {
        case 'A': c = 'T'; break; case 'T': c = 'A'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break;
        case 'a': c = 't'; break; case 't': c = 'a'; break; case 'c': c = 'g'; break; case 'g': c = 'c'; break;
        case 'N': c = 'N'; break; default: c = 'N';
      }
      s[i] = c;
    }
  }

  // Segment::yieldReads, Segment.cpp:673-871.  `winBase`/`slotBase` are the philox addresses of
  // this segment's first window / first pair slot inside the batch.
  void segYieldReads(vector<Segment>& cs, Segment& g, uint32_t batch, uint64_t winBase, uint64_t slotBase,
                     string& out1, string& out2) {
    bool paired = cfg.paired();
    int readLength = prof.readLength;
    unsigned int seqSize = g.seqSize();
    unsigned int segsize = seqSize / g.CN;
    int fragCount = 0;
    string fragSeq, b, q;
    char hdr[4096];
    uint64_t slot = slotBase;
    for (size_t i = 0; i < g.fragStartPos.size(); i++) {
      long spos = g.fragStartPos[i];
      long epos = g.fragEndPos[i];
      int hapIndx = g.hapIndxs[i];
      long fragSize = epos - spos + 1;
      int failCount = 0;
      int n = g.fragRCs[i];
      uint32_t widx = (uint32_t)(winBase + i);
      uint32_t attempt = 0;
      int planned = n <= 0 ? 0 : (paired ? (n + 1) / 2 : n);
      int done = 0;
      while (n > 0) {
        uint32_t a = attempt++;
        uint32_t xpos = rng.philox ? rng.ph(KIND_PLAN, batch & 0xFFFFu, widx, a, 0, 0) : rng.intGen();
        long pos = u32ToInteger(xpos, spos, epos + 1);
        if (!paired) {
          getFragSequence(cs, g, hapIndx, pos, (int)fragSize, fragSeq);
        } else {
          int insertSize;
          if (prof.iSizeAlphabet.empty()) insertSize = prof.insertSize;  // Profile.cpp:1486-1493
          else {
            uint32_t xi = rng.philox ? rng.ph(KIND_PLAN, batch & 0xFFFFu, widx, a, 0, 1) : rng.realGen();
            insertSize = prof.iSizeAlphabet[randIndxFrom(xi, prof.iSizeCdf.data(), (int)prof.iSizeCdf.size())];
          }
          getFragSequence(cs, g, hapIndx, pos, insertSize, fragSeq);
        }
        if ((int)fragSeq.size() < readLength) {
          failCount++;
          if (failCount > 1000) break;
          continue;
        }
        fragCount++;
        ReadCtx rc{&rng, batch, (uint32_t)(slot + done), 0};
        if (!paired) {
          uint32_t xs = rng.philox ? rng.ph(KIND_PLAN, batch & 0xFFFFu, widx, a, 0, 2) : rng.intGen();
          long k = u32ToInteger(xs, 0, 2);
          if (k == 0) {
            predict(prof, fragSeq.data(), readLength, 1, rc, b, q);
          } else {
            char* seq = &fragSeq[fragSeq.size() - readLength];
            complementInPlace(seq, readLength);
            std::reverse(seq, seq + readLength);
            predict(prof, seq, readLength, 1, rc, b, q);
          }
          snprintf(hdr, sizeof hdr, "@%s#%s#%ld#%d\n", curPopu.c_str(), g.chr.c_str(), pos % segsize, fragCount);
          out1 += hdr; out1 += b; out1 += "\n+\n"; out1 += q; out1 += '\n';
          readsWritten++;
          n--;
        } else {
          predict(prof, fragSeq.data(), readLength, 1, rc, b, q);
          snprintf(hdr, sizeof hdr, "@%s#%s#%ld#%d/1\n", curPopu.c_str(), g.chr.c_str(), pos % segsize, fragCount);
          out1 += hdr; out1 += b; out1 += "\n+\n"; out1 += q; out1 += '\n';
          char* seq = &fragSeq[fragSeq.size() - readLength];
          complementInPlace(seq, readLength);
          std::reverse(seq, seq + readLength);
          rc.mate = 1;
          predict(prof, seq, readLength, 0, rc, b, q);
          snprintf(hdr, sizeof hdr, "@%s#%s#%ld#%d/2\n", curPopu.c_str(), g.chr.c_str(), pos % segsize, fragCount);
          out2 += hdr; out2 += b; out2 += "\n+\n"; out2 += q; out2 += '\n';
          readsWritten += 2;
          n -= 2;
        }
        done++;
      }
      slot += planned;
    }
  }

  static int plannedOf(int n, bool paired) { return n <= 0 ? 0 : (paired ? (n + 1) / 2 : n); }

  // one (population, chromosome) batch: Genome.cpp:870-887 / :938-955
  void runBatch(const string& popu, const string& chr, FILE* f1, FILE* f2, int threads) {
    vector<Segment>& cs = segments[popu][chr];
    uint32_t batch = batchId++;
    if (batchId > 0xFFFF) throw Fail("more than 65535 (population, chromosome) batches");
    for (size_t k = 0; k < cs.size(); k++) generateSegSequences(cs[k], (uint32_t)k);
    // philox addresses: windows / pair slots of processed segments, in order
    vector<uint64_t> winBase(cs.size(), 0), slotBase(cs.size(), 0);
    vector<char> active(cs.size(), 0);
    uint64_t w = 0, s = 0;
    bool paired = cfg.paired();
    for (size_t k = 0; k < cs.size(); k++) {
      active[k] = cs[k].hasSeqs && cs[k].readCount != 0;  // Segment.cpp:675
      if (!active[k]) continue;
      winBase[k] = w; slotBase[k] = s;
      w += cs[k].fragStartPos.size();
      for (int n : cs[k].fragRCs) s += plannedOf(n, paired);
    }
    if (w > 0xFFFFFFFFull || s > 0xFFFFFFFFull) throw Fail("batch too large for 32-bit philox addresses");
    vector<string> o1(cs.size()), o2(cs.size());
    if (threads <= 1 || !rng.philox) {
      for (size_t k = 0; k < cs.size(); k++) {
        if (!active[k]) continue;
        segYieldReads(cs, cs[k], batch, winBase[k], slotBase[k], o1[k], o2[k]);
        if (f1) { fwrite(o1[k].data(), 1, o1[k].size(), f1); }
        if (f2) { fwrite(o2[k].data(), 1, o2[k].size(), f2); }
        o1[k].clear(); o1[k].shrink_to_fit(); o2[k].clear(); o2[k].shrink_to_fit();
      }
    } else {
      // philox draws are addressed, so segments may run in any order / in parallel; output is
      // still concatenated in segment order.  readsWritten is accumulated per worker.
      std::vector<std::thread> pool;
      std::vector<uint64_t> counts(threads, 0);
      std::atomic_size_t* next = new std::atomic_size_t(0);
      for (int t = 0; t < threads; t++) {
        pool.emplace_back([&, t]() {
          Genome local(cfg, prof, rng);  // shares read-only state; own counter
          local.curPopu = curPopu;
          for (;;) {
            size_t k = next->fetch_add(1);
            if (k >= cs.size()) break;
            if (!active[k]) continue;
            local.segYieldReads(cs, cs[k], batch, winBase[k], slotBase[k], o1[k], o2[k]);
          }
          counts[t] = local.readsWritten;
        });
      }
      for (auto& th : pool) th.join();
      delete next;
      for (int t = 0; t < threads; t++) readsWritten += counts[t];
      for (size_t k = 0; k < cs.size(); k++) {
        if (f1) fwrite(o1[k].data(), 1, o1[k].size(), f1);
        if (f2) fwrite(o2[k].data(), 1, o2[k].size(), f2);
      }
    }
    for (size_t k = 0; k < cs.size(); k++) cs[k].clearSeqs();
  }

  // Genome::yieldReads, Genome.cpp:827-960
  void yieldReads(const string& outDir, int threads) {
    vector<string>& popuNames = cfg.popuNames;
    long reads = targetLength() * cfg.i["coverage"] / prof.readLength;
    map<string, double> ACNs;
    calculateACNs(ACNs);
    if (!rng.philox) srand((unsigned)rng_time);  // Genome.cpp:852
    bool paired = cfg.paired();
    auto openOut = [&](const string& stem, FILE*& f1, FILE*& f2) {
      f1 = f2 = nullptr;
      if (paired) {
        f1 = fopen((outDir + "/" + stem + "_1.fq").c_str(), "w");
        f2 = fopen((outDir + "/" + stem + "_2.fq").c_str(), "w");
        if (!f1 || !f2) throw Fail("Error: can not open fastq file to save results: " + outDir + "/" + stem);
      } else {
        f1 = fopen((outDir + "/" + stem + ".fq").c_str(), "w");
        if (!f1) throw Fail("Error: can not open fastq file to save results: " + outDir + "/" + stem);
      }
    };
    if (mixProps.empty()) {
      curPopu = popuNames[0];
      FILE *f1, *f2;
      openOut(popuNames[0], f1, f2);
      setReadCounts(popuNames[0], reads);
      for (auto& chr : chromosomes) runBatch(popuNames[0], chr, f1, f2, threads);
      fclose(f1);
      if (f2) fclose(f2);
    } else {
      for (size_t m = 0; m < mixProps.size(); m++) {
        vector<float> props = mixProps[m];
        double w_acn = 0;
        char buf[1000];
        string fn;
        for (size_t i = 0; i < popuNames.size(); i++) {
          float prop = props[i];
          double acn = ACNs[popuNames[i]];
          w_acn += prop * acn;
          if (i == 0) snprintf(buf, sizeof buf, "%s_%.3f", popuNames[i].c_str(), prop);
          else snprintf(buf, sizeof buf, "+%s_%.3f", popuNames[i].c_str(), prop);
          fn += buf;
        }
        FILE *f1, *f2;
        openOut(fn, f1, f2);
        for (size_t i = 0; i < popuNames.size(); i++) {
          curPopu = popuNames[i];
          long popuReads = (long)(reads * props[i] * ACNs[curPopu] / w_acn);  // long*float -> float (Genome.cpp:935)
          setReadCounts(curPopu, popuReads);
          for (auto& chr : chromosomes) runBatch(curPopu, chr, f1, f2, threads);
        }
        fclose(f1);
        if (f2) fclose(f2);
      }
    }
  }
  uint64_t rng_time = 0;
};

}  // namespace orc

// ---------------------------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static uint64_t g_reads = 0;

extern "C" const char* orc_last_error(void) { return g_err.c_str(); }
extern "C" uint64_t orc_last_read_count(void) { return g_reads; }

extern "C" int orc_simulate(const char* config_path, int rng_mode, uint64_t seed_sec, uint64_t seed_nsec,
                            const char* output_dir_override, int threads) {
  using namespace orc;
  try {
    Config cfg;
    cfg.load(config_path);
    Rng rng;
    if (rng_mode == ORC_RNG_MT) {
      if (threads > 1) throw Fail("mt mode is sequential (threads must be 1)");
      rng.initMt(seed_sec, seed_nsec);
    } else {
      rng.initPhilox((seed_sec << 32) | (seed_nsec & 0xFFFFFFFFull));
    }
    Profile prof;
    prof.paired = cfg.paired();
    prof.insertSize = cfg.i["insertSize"];
    Genome genome(cfg, prof, rng);
    genome.rng_time = seed_sec;
    genome.loadData();                    // src/simuReads.cpp:53
    string outDir = output_dir_override && output_dir_override[0] ? output_dir_override : cfg.s["output"];
    mkdir(outDir.c_str(), 0755);          // src/simuReads.cpp:56-60
    prof.train(cfg.s["profile"], rng);    // src/simuReads.cpp:68
    genome.generateSegments();            // :71
    genome.yieldReads(outDir, threads);   // :73
    g_reads = genome.readsWritten;
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return 1;
  }
}

struct orc_profile {
  orc::Profile p;
  orc::Rng rng;
};

extern "C" orc_profile* orc_profile_load(const char* path, int paired, int insert_size) {
  try {
    std::unique_ptr<orc_profile> h(new orc_profile());
    h->rng.initPhilox(0);
    h->p.paired = paired != 0;
    h->p.insertSize = insert_size;
    h->p.train(path, h->rng);
    return h.release();
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
extern "C" void orc_profile_free(orc_profile* h) { delete h; }
extern "C" int orc_profile_info(const orc_profile* h, int what) {
  const orc::Profile& p = h->p;
  switch (what) {
    case 0: return p.N; case 1: return p.kmer; case 2: return p.bins; case 3: return p.readLength;
    case 4: return p.kmerCount; case 5: return p.nQual; case 6: return (int)p.insCdf.size();
    case 7: return (int)p.delCdf.size(); case 8: return (int)p.iSizeCdf.size(); case 9: return p.hasSub2 ? 1 : 0;
    case 10: return p.iSizeAlphabet.empty() ? p.insertSize : p.iSizeAlphabet[0];
    case 11: return p.lgW;
  }
  return -1;
}
extern "C" double orc_profile_rate(const orc_profile* h, int which) {
  switch (which) { case 0: return h->p.insertRate; case 1: return h->p.delRate; case 2: return h->p.stdISize; case 3: return h->p.gcStd; }
  return 0;
}
extern "C" const double* orc_profile_array(const orc_profile* h, int which) {
  const orc::Profile& p = h->p;
  switch (which) {
    case 0: return p.insCdf.data(); case 1: return p.delCdf.data(); case 2: return p.subs1.data();
    case 3: return p.hasSub2 ? p.subs2.data() : nullptr; case 4: return p.qual.data();
    case 5: return p.iSizeCdf.empty() ? nullptr : p.iSizeCdf.data(); case 6: return p.gcMeans;
  }
  return nullptr;
}
extern "C" void orc_profile_sub_row(const orc_profile* h, int mate2, int kmer_indx, int bin, uint64_t cum[3], uint8_t order[4]) {
  const orc::Profile& p = h->p;
  const size_t r = (size_t)kmer_indx * p.bins + bin;
  for (int i = 0; i < 3; i++) cum[i] = (mate2 ? p.subCum2 : p.subCum1)[r * 3 + i];
  for (int i = 0; i < 4; i++) order[i] = p.subOrd[(size_t)kmer_indx * 4 + i];
}
extern "C" void orc_profile_alias_row(const orc_profile* h, int base_pair, int bin, uint32_t* thr, uint8_t* lo, uint8_t* hi) {
  const orc::Profile& p = h->p;
  const size_t r = ((size_t)base_pair * p.bins + bin) << p.lgW;
  for (uint32_t c = 0; c < (1u << p.lgW); c++) { thr[c] = p.aThr[r + c]; lo[c] = p.aLo[r + c]; hi[c] = p.aHi[r + c]; }
}
extern "C" void orc_profile_kmer(const orc_profile* h, int i, char* out) {
  memcpy(out, h->p.kmers[i].data(), h->p.kmer);
}
// philox mode: the indel candidate law (Profile::buildIndelGaps): A, B in units of 2^-64 and the first n entries of
// the distance table gapS[1..]; returns the table's length (read length)
extern "C" int orc_profile_indel_gaps(const orc_profile* h, uint64_t ab[2], uint64_t* gaps, int n) {
  ab[0] = h->p.evA;
  ab[1] = h->p.evB;
  const int L = (int)h->p.gapS.size() - 1;
  for (int k = 1; k <= n && k <= L; k++) gaps[k - 1] = h->p.gapS[(size_t)k];
  return L;
}
extern "C" int orc_predict_philox(const orc_profile* h, const char* ref, int n, int is_read1, uint64_t seed,
                                  uint32_t batch_id, uint32_t pair_slot, char* out_bases, char* out_quals) {
  orc::Rng rng;
  rng.initPhilox(seed);
  orc::ReadCtx rc{&rng, batch_id, pair_slot, is_read1 ? 0 : 1};
  std::string b, q;
  int np = orc::predict(h->p, ref, n, is_read1, rc, b, q);
  memcpy(out_bases, b.data(), np);
  memcpy(out_quals, q.data(), np);
  return np;
}
extern "C" void orc_philox4x32_r(int rounds, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  orc::Philox4 o = orc::philox4x32_r(rounds, ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
  for (int i = 0; i < 4; i++) out[i] = o.v[i];
}
extern "C" int orc_base_rounds(void) { return orc::kBaseRounds; }
extern "C" void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  orc::Philox4 o = orc::philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
  for (int i = 0; i < 4; i++) out[i] = o.v[i];
}
