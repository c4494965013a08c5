// host/train.cpp -- see train.h.  The host keeps what the reference's seqToProfile does once per run -- options, the VCF,
// the BED file, the arithmetic on the finished count matrices (a few hundred thousand numbers) and the file format -- and
// streams the reads through the GPU: the reference to HBM (Fasta::open_on_device), the SAM text in pinned chunks of whole
// lines while a reader thread fills the next one (sg_train_feed).
#include "train.h"

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/simuscop_amd.h"
#include "common.h"
#include "fasta.h"
#include "reader_pool.h"

namespace simu {
namespace {

using Clock = std::chrono::steady_clock;
double since(Clock::time_point t) { return std::chrono::duration<double>(Clock::now() - t).count(); }

constexpr double kZeroFinal = 2.2204e-16;   // lib/mydefine/MyDefine.cpp:20
constexpr uint32_t kWindow = 1000;          // Segment::fragSize
constexpr uint32_t kIsizeCols = 1u << 20, kIndelCols = 1u << 16;
constexpr uint64_t kChunk = 64ull << 20;

struct Engine {
  sg_ctx* ctx = nullptr;
  ~Engine() { if (ctx) sg_destroy(ctx); }
  void check(int rc, const char* what) const {
    if (rc != SG_OK) throw Error(std::string("GPU engine error in ") + what + ": " + sg_last_error(ctx));
  }
};

// ---- lib/vcfparser/vcfparser.cpp:26-106 ----
struct KnownVariants {
  std::vector<uint32_t> snv_contig, ins_contig, del_contig;
  std::vector<int64_t> snv_pos, ins_pos, del_pos;
  std::string snv_alt;
  std::vector<uint8_t> snv_homo;
  std::vector<int32_t> ins_len, del_len;
  uint64_t n_snv = 0, n_ins = 0, n_del = 0;   // as the reference counts them (rows on contigs the FASTA lacks included)
};
void parse_vcf(const std::string& path, const std::map<std::string, uint32_t>& row_of, KnownVariants& kv, bool quiet) {
  if (path.empty()) { std::cerr << "Error: VCF file was not specified" << std::endl; return; }
  FILE* fp = fopen(path.c_str(), "r");
  if (!fp) throw Error("Error: cannot open VCF file " + path, -1);
  std::vector<char> buf(20000);
  long line_num = 0;
  int wrong = 0;
  while (fgets(buf.data(), (int)buf.size(), fp)) {
    line_num++;
    if (buf[0] == '#') continue;
    // ten fields, the rest of the line stays with the tenth
    char* f[10];
    int nf = 0;
    char* p = buf.data();
    f[nf++] = p;
    while (*p && nf < 10) {
      if (*p == '\t') { *p = '\0'; f[nf++] = p + 1; }
      p++;
    }
    if (nf < 10) {
      std::cerr << "Warning: malformed VCF file " << path << ", there should be at least 10 fields @line " << line_num << std::endl;
      if (++wrong > 10) { fclose(fp); throw Error("", 1); }
      continue;
    }
    const std::string info = f[7];
    const size_t dp = info.find("DP=");
    if (dp != std::string::npos) {
      const size_t semi = info.find(";", dp);
      if (atoi(info.substr(dp + 3, semi - dp - 3).c_str()) < 10) continue;   // depth_th
    }
    if ((float)atof(f[5]) < 20.0f) continue;                                  // quality_th
    const std::string chr = abbr_of_chr(f[0]);
    const long pos = atol(f[1]);
    std::string gt = f[9];
    gt = gt.substr(0, gt.find(':'));
    // (the parser files 1/1 as het and everything else as homo, :81-86: kept as it is -- including what fgets leaves behind:
    // a sample column that holds the genotype alone ends in the line break, and "1/1\n" is not "1/1")
    const bool homo = gt != "1/1";
    const auto it = row_of.find(chr);
    const size_t ref_len = strlen(f[3]), alt_len = strlen(f[4]);
    if (ref_len > 1) {
      kv.n_del++;
      if (it != row_of.end()) { kv.del_contig.push_back(it->second); kv.del_pos.push_back(pos + 1); kv.del_len.push_back((int32_t)ref_len - 1); }
    } else if (alt_len > 1) {
      kv.n_ins++;
      if (it != row_of.end()) { kv.ins_contig.push_back(it->second); kv.ins_pos.push_back(pos); kv.ins_len.push_back((int32_t)alt_len - 1); }
    } else {
      kv.n_snv++;
      if (it != row_of.end()) { kv.snv_contig.push_back(it->second); kv.snv_pos.push_back(pos); kv.snv_alt.push_back(f[4][0]); kv.snv_homo.push_back(homo ? 1 : 0); }
    }
  }
  fclose(fp);
  if (!quiet)
    std::cerr << "total " << kv.n_snv << " SNPs, " << kv.n_ins << " inserts and " << kv.n_del << " deletions were loaded from file " << path << std::endl;
}

// ---- Genome::loadTargets (Genome.cpp:238-299) + divideTargets (:684-739) into inTargets ----
struct Piece { long spos, epos; };
std::map<std::string, std::vector<Piece>> load_targets(const std::string& path, const Fasta& fa, bool quiet) {
  std::map<std::string, std::vector<Piece>> raw, out;
  if (path.empty()) return out;
  std::ifstream ifs(path.c_str());
  if (!ifs.is_open()) throw Error("can not open target file " + path, -1);
  std::string line;
  int line_num = 0, n = 0;
  while (std::getline(ifs, line)) {
    line_num++;
    const std::vector<std::string> f = split(line, '\t');
    if (f.size() < 3) throw Error("ERROR: line " + std::to_string(line_num) + " should have at least 3 fields in file " + path + "\n" + line);
    const std::string chr = abbr_of_chr(f[0]);
    const long len = fa.length(chr);
    if (len <= 0) continue;
    const long e = atol(f[2].c_str());
    Piece t;
    t.spos = std::max(1L, atol(f[1].c_str()) - 50 + 1);
    t.epos = std::min(len, (e <= 0 ? len - (-e) % len : e) + 50);
    raw[chr].push_back(t);
    n++;
  }
  if (!quiet) std::cerr << "\ntotal " << n << " targets were loaded from file " << path << std::endl;
  for (const auto& kv : raw)
    for (const Piece& t : kv.second) {
      long spos = t.spos;
      const int k = (int)((t.epos - t.spos + 1) / (long)kWindow);
      for (int i = 0; i < k; i++) {
        const Piece p{spos, i == k - 1 ? t.epos : spos + (long)kWindow - 1};
        spos = p.epos + 1;
        out[kv.first].push_back(p);
      }
      if (spos <= t.epos) out[kv.first].push_back(Piece{spos, t.epos});
    }
  return out;
}

// ---- where the lines come from: a file, standard input, or `samtools view` as the reference runs it ----
struct LineSource {
  FILE* fp = nullptr;
  bool piped = false;
  std::string what;
  ~LineSource() { close(); }
  void close() {
    if (fp && fp != stdin) { if (piped) pclose(fp); else fclose(fp); }
    fp = nullptr;
  }
  void open(const simu_train_options& o) {
    close();
    const std::string sam = o.sam ? o.sam : "";
    if (!sam.empty()) {
      what = sam;
      if (sam == "-") fp = stdin;
      else if (!(fp = fopen(sam.c_str(), "r"))) throw Error("cannot open SAM file " + sam, -1);
      return;
    }
    std::string samtools = o.samtools ? o.samtools : "";
    if (samtools.empty()) samtools = "samtools";
    what = o.bam ? o.bam : "";
    const std::string cmd = samtools + " view -F 0xD04 -q 20 " + what;   // Profile.cpp:1448
    piped = true;
    if (!(fp = popen(cmd.c_str(), "r"))) throw Error("cannot open BAM file " + what, -1);
  }
};

// CIGAR of a line when it is a single nM (Profile::setReadLength, Profile.cpp:155-163), else 0
int single_match_length(const char* p, const char* le) {
  int tabs = 0;
  while (p < le && tabs < 5) { if (*p == '\t') tabs++; p++; }
  if (tabs < 5) return 0;
  const char* e = p;
  while (e < le && *e != '\t') e++;
  const int n = (int)(e - p);
  int i = 0;
  for (; i < n - 1; i++) if (!(p[i] >= '0' && p[i] <= '9')) break;
  if (n >= 1 && i == n - 1 && p[i] == 'M') return atoi(std::string(p, e).c_str());
  return 0;
}

// Pinned chunks of whole lines, read ahead while the device works on the chunk before.  A regular file is read by a pool of
// threads (pread of 4 MB slices: one reader took text at 10 GB/s, a fifth of what the link and the kernels take); a pipe
// or standard input by the one thread there can be.
struct ChunkReader {
  sg_ctx* ctx;
  FILE* fp;
  int fd = -1;                // >= 0: a regular file, read with pread from `pos`
  uint64_t pos = 0, size = 0;
  std::unique_ptr<ReaderPool> pool;
  char* buf[2] = {nullptr, nullptr};
  uint64_t len[2] = {0, 0};
  std::string carry;          // the unfinished line behind a chunk's last line break
  bool eof = false;
  ChunkReader(sg_ctx* c, FILE* f, int threads) : ctx(c), fp(f) {
    struct stat sb;
    if (f != stdin && fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode)) {
      fd = fileno(f);
      size = (uint64_t)sb.st_size;
      pool.reset(new ReaderPool(std::min(16, std::max(1, threads))));
    }
    for (int i = 0; i < 2; i++) {
      void* p = nullptr;
      if (sg_host_alloc(ctx, kChunk + 16, &p) != SG_OK) throw Error(std::string("GPU engine error in sg_host_alloc: ") + sg_last_error(ctx));
      buf[i] = (char*)p;
    }
  }
  ~ChunkReader() { for (char* b : buf) if (b) sg_host_free(ctx, b); }
  // fills buf[i] with whole lines; false when nothing is left
  bool fill(int i) {
    if (eof && carry.empty()) { len[i] = 0; return false; }
    uint64_t have = carry.size();
    if (have > kChunk) throw Error("a line of the SAM text is longer than 64 MB");
    memcpy(buf[i], carry.data(), have);
    carry.clear();
    if (fd >= 0) {
      const uint64_t n = std::min<uint64_t>(kChunk - have, size - pos);
      if (n) parallel_pread(*pool, fd, (uint8_t*)buf[i] + have, pos, n);
      have += n;
      pos += n;
      if (pos >= size) eof = true;
    } else {
      while (!eof && have < kChunk) {
        const size_t got = fread(buf[i] + have, 1, kChunk - have, fp);
        if (got == 0) { eof = true; break; }
        have += got;
      }
    }
    if (!eof) {   // cut behind the last line break
      uint64_t cut = have;
      while (cut > 0 && buf[i][cut - 1] != '\n') cut--;
      if (cut == 0) throw Error("a line of the SAM text is longer than 64 MB");
      carry.assign(buf[i] + cut, have - cut);
      have = cut;
    }
    len[i] = have;
    return have > 0;
  }
};

// ---- the arithmetic of Profile::train behind the counting (Profile.cpp:1471-1483) ----
struct Model {
  std::string bases;
  int kmer = 3, bins = 50, kmer_count = 0, read_length = 0;
  std::vector<double> subs1, subs2, kmers, quality, isize, ins, del;   // the count matrices, then their normalised forms
  double insert_rate = 0, del_rate = 0, base_count = 0, std_isize = 0;
  std::vector<double> gcs, rcs;
  double gc_means[101];
  double gc_std = 0;
  bool gc_fitted = false;
};

double clamp0(double x) { return std::fabs(x) < kZeroFinal ? 0.0 : x; }   // Matrix::operator*, lib/matrix/Matrix.h:683-697

double median_of(std::vector<double> v) {   // lib/mydefine/MyDefine.h:72-104; no windows at all: 0
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  const size_t n = v.size();
  return (n & 1) ? v[n / 2] : (v[n / 2] + v[n / 2 - 1]) / 2;
}

// Profile::estimateGCParas (Profile.cpp:713-834): thin the windows to ~150,000, normalise the read counts by their median,
// fit a locally weighted line at every GC percent.  Four places where the reference reads memory it does not own are
// given a meaning here: the thinning counters start at zero, a window of G/C only has a cell of its
// own, fewer than 50 windows thin nothing, and without any fitted percent no tail is extrapolated.
void fit_gc_model(Model& M, const std::string& gc_file) {
  const int B = 50;
  std::vector<int> per_bin(B + 1, 0), step(B + 1, 1), seen(B + 1, 0);
  for (double g : M.gcs) per_bin[(int)(g * B)]++;
  const int expect = std::min(150000, (int)M.gcs.size()) / B;
  if (expect > 0)
    for (int b = 0; b <= B; b++) step[b] = std::max(1, per_bin[b] / expect);
  std::ofstream ofs(gc_file.c_str());
  std::vector<int> kept;
  const double med = median_of(M.rcs);
  for (size_t i = 0; i < M.rcs.size(); i++) {
    const int b = (int)(M.gcs[i] * B);
    if (seen[b] % step[b] == 0) {
      M.rcs[i] = M.rcs[i] / (med + kZeroFinal);
      if (M.rcs[i] < 3) {
        ofs << M.rcs[i] << '\t' << M.gcs[i] << std::endl;
        kept.push_back((int)i);
      }
    }
    seen[b]++;
  }
  ofs.close();
  const double tau = 5, half = 0.03 / 2;
  int lo = -1, hi = -1;
  std::vector<double> g, r, w;
  for (int k = 0; k <= 100; k++) {
    const double at = k / 100.0;
    g.clear(); r.clear();
    for (int j : kept)
      if (std::fabs(at - M.gcs[j]) <= half) { g.push_back(M.gcs[j]); r.push_back(M.rcs[j]); }
    if (g.size() <= 20) { M.gc_means[k] = 0; continue; }
    if (lo < 0) lo = k;
    hi = k;
    // beta = (B'WB)^-1 B'W y, B = [1 g], W = diag(w): the reference forms every product as a dense Matrix product -- running
    // sums from zero in index order, each result clamped to 0 below 2.2204e-16 -- and the sums over the diagonal W have one
    // term that is not zero.  Same operations, same order, without the n x n matrix.
    const size_t n = g.size();
    w.resize(n);
    double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
    for (size_t i = 0; i < n; i++) {
      w[i] = std::exp(-std::pow(g[i] - at, 2) / (2 * tau));
      const double u0 = clamp0(0.0 + 1.0 * w[i]), u1 = clamp0(0.0 + g[i] * w[i]);   // (B'W)[.][i]
      a00 += u0 * 1.0; a01 += u0 * g[i]; a10 += u1 * 1.0; a11 += u1 * g[i];
    }
    a00 = clamp0(a00); a01 = clamp0(a01); a10 = clamp0(a10); a11 = clamp0(a11);
    // determinant by elimination (Matrix.h:223-268), then the cofactor inverse (:184-220)
    double c00 = a00, c01 = a01, c10 = a10, c11 = a11;
    bool swapped = false, stuck = false;
    if (c00 == 0) {
      if (c10 == 0) stuck = true;
      else { std::swap(c00, c10); std::swap(c01, c11); swapped = true; }
    }
    if (!stuck && c10 != 0) {
      const double q = c10 / c00;
      c10 -= q * c00;
      c11 -= q * c01;
    }
    double det = 1.0;
    det *= c00;
    det *= c11;
    if (swapped) det = -det;
    const double i00 = clamp0(a11 / det), i01 = clamp0(-a01 / det), i10 = clamp0(-a10 / det), i11 = clamp0(a00 / det);
    double b0 = 0, b1 = 0;
    for (size_t i = 0; i < n; i++) {
      double p0 = 0, p1 = 0;
      p0 += i00 * 1.0; p0 += i01 * g[i];
      p1 += i10 * 1.0; p1 += i11 * g[i];
      b0 += clamp0(0.0 + clamp0(p0) * w[i]) * r[i];
      b1 += clamp0(0.0 + clamp0(p1) * w[i]) * r[i];
    }
    b0 = clamp0(b0); b1 = clamp0(b1);
    double y = 0;
    y += 1.0 * b0;
    y += at * b1;
    M.gc_means[k] = std::max(0.0, clamp0(y));
  }
  if (lo >= 0) {   // (no percent with more than 20 windows -- an exome with few targets: every mean stays 0)
    for (int k = 0; k < lo; k++) M.gc_means[k] = M.gc_means[lo] * k / lo;
    for (int k = hi + 1; k <= 100; k++) M.gc_means[k] = M.gc_means[hi] - M.gc_means[hi] * (k - hi) / (100 - hi);
  }
  double ss = 0;
  for (int j : kept) ss += std::pow(M.rcs[j] - M.gc_means[(int)(M.gcs[j] * 100)], 2);
  M.gc_std = std::sqrt(ss / kept.size());
}

// Matrix::normalize(0) (Matrix.h:483-503) on `rows` rows of `cols` numbers
void normalise_rows(double* m, size_t rows, size_t cols) {
  for (size_t i = 0; i < rows; i++) {
    double s = 0;
    for (size_t j = 0; j < cols; j++) s += m[i * cols + j];
    for (size_t j = 0; j < cols; j++) m[i * cols + j] /= (kZeroFinal + s);
  }
}

// the k-mer of context index i (Profile::initKmers, Profile.cpp:70-124): contexts of one real base first, X-padded
std::string kmer_text(const Model& M, int i) {
  int m = 1, first = 0, block = 4;
  while (i >= first + block) { first += block; block *= 4; m++; }
  int v = i - first;
  std::string s((size_t)M.kmer, 'X');
  for (int t = M.kmer - 1; t >= M.kmer - m; t--) { s[(size_t)t] = M.bases[(size_t)(v & 3)]; v >>= 2; }
  return s;
}

// Profile::normParas(false), Profile.cpp:836-900
void normalise(Model& M) {
  const int N = 4, bins = M.bins;
  normalise_rows(M.kmers.data(), (size_t)bins, (size_t)M.kmer_count);
  for (int i = 0; i < M.kmer_count; i++) {
    const std::string km = kmer_text(M, i);
    const int last = (int)M.bases.find(km[(size_t)M.kmer - 1]);
    for (std::vector<double>* sd : {&M.subs1, &M.subs2}) {
      double* blk = sd->data() + (size_t)i * bins * N;
      normalise_rows(blk, (size_t)bins, N);
      for (int j = 0; j < bins; j++) {   // an empty row becomes "no substitution" (:847-861)
        double s = 0;
        for (int k = 0; k < N; k++) s += blk[j * N + k];
        if (s < kZeroFinal) blk[j * N + last] = 1;
      }
    }
  }
  for (int i = 0; i < N * N; i++) normalise_rows(M.quality.data() + (size_t)i * bins * 94, (size_t)bins, 94);
  // insert sizes: the mode, everything from five times the mode on dropped, standard deviation of what is left (:869-891)
  int best = 0, mode = 0;
  for (size_t i = 0; i < M.isize.size(); i++)
    if (M.isize[i] > best) { best = (int)M.isize[i]; mode = (int)i; }
  for (size_t i = (size_t)mode * 5; i < M.isize.size(); i++) M.isize[i] = 0;
  normalise_rows(M.isize.data(), 1, M.isize.size());
  auto col = [&](int i) { return (size_t)i < M.isize.size() ? M.isize[(size_t)i] : 0.0; };   // (past the row: 0)
  double mean = 0;
  for (int i = 0; i < mode * 5; i++) mean += col(i) * i;
  double var = 0;
  for (int i = 0; i < mode * 5; i++) var += col(i) * std::pow(i - mean, 2);
  M.std_isize = std::sqrt(var);
  normalise_rows(M.ins.data(), 1, M.ins.size());
  normalise_rows(M.del.data(), 1, M.del.size());
  M.insert_rate /= M.base_count;
  M.del_rate /= M.base_count;
}

// Profile::saveResults, Profile.cpp:1240-1365
void write_profile(const Model& M, std::ostream& os, const std::string& reads_label, const std::string& stamp) {
  const int N = 4;
  os << "#model created at " << stamp;
  os << "#reads: " << reads_label << std::endl << std::endl;
  os << "bases: " << M.bases << std::endl;
  os << "readLength: " << M.read_length << std::endl;
  os << "binCount: " << M.bins << std::endl;
  os << "kmer: " << M.kmer << std::endl << std::endl;
  auto row = [&](const std::vector<double>& v) {
    for (size_t i = 0; i + 1 < v.size(); i++) os << v[i] << '\t';
    os << v.back() << std::endl;
  };
  os << "\n[Insert Rate]" << std::endl << M.insert_rate << std::endl << "[Insert Frequency]" << std::endl;
  row(M.ins);
  os << "\n[Deletion Rate]" << std::endl << M.del_rate << std::endl << "[Deletion Frequency]" << std::endl;
  row(M.del);
  os << "\n[Substitution Probs]" << std::endl;
  for (int i = 0; i < M.kmer_count; i++) {
    os << "kmer: " << kmer_text(M, i) << std::endl;
    for (const std::vector<double>* sd : {&M.subs1, &M.subs2}) {
      const double* p = sd->data() + (size_t)i * M.bins * N;
      for (int j = 0; j < M.bins; j++)
        for (int k = 0; k < N; k++) os << p[j * N + k] << (k < N - 1 ? '\t' : '\n');
    }
  }
  os << "\n[Base Quality Distribution]" << std::endl;
  for (int i = 0; i < N * N; i++) {
    os << "basePairIndx: " << i << std::endl;
    const double* p = M.quality.data() + (size_t)i * M.bins * 94;
    for (int j = 0; j < M.bins; j++)
      for (int k = 0; k < 94; k++) os << p[j * 94 + k] << (k < 93 ? '\t' : '\n');
  }
  os << "\n[Insert Size Standard Deviation]" << std::endl << M.std_isize << std::endl;
  os << "\n[Log Ratio Mean Value]" << std::endl;
  for (int i = 0; i < 101; i++) os << i << '\t' << M.gc_means[i] << std::endl;
  os << "\n[Log Ratio Standard Deviation]" << std::endl << M.gc_std << std::endl;
}

void run(const simu_train_options& o, simu_train_stats& st) {
  const auto t0 = Clock::now();
  const bool quiet = o.quiet != 0;
  const std::string ref = o.ref ? o.ref : "", vcf = o.vcf ? o.vcf : "", target = o.target ? o.target : "", output = o.output ? o.output : "";
  Engine eng;
  if (sg_create(&eng.ctx, o.device, 0) != SG_OK) throw Error(std::string("GPU engine error: ") + sg_last_error(nullptr));
  // ---- Genome::loadTrainData: VCF, reference, targets (Genome.cpp:32-39; the VCF needs the contig rows, so the reference
  // goes first here) ----
  auto t1 = Clock::now();
  Fasta fa;
  fa.open_on_device(ref, eng.ctx, std::max(1, o.threads));
  if (fa.names.empty()) throw Error("ERROR: reference sequence cannot be empty!");
  if (!quiet) std::cerr << "\nReference sequence was loaded from file " << ref << std::endl;
  st.t_reference = since(t1);
  std::vector<std::string> key_of_row(fa.contigs.size());
  for (const auto& kv : fa.contig_of) key_of_row[kv.second] = kv.first;
  KnownVariants kv;
  parse_vcf(vcf, fa.contig_of, kv, quiet);
  const std::map<std::string, std::vector<Piece>> targets = load_targets(target, fa, quiet);
  std::vector<uint64_t> tfirst(key_of_row.size() + 1, 0);
  std::vector<int64_t> tspos, tepos;
  for (size_t r = 0; r < key_of_row.size(); r++) {
    tfirst[r] = tspos.size();
    const auto it = targets.find(key_of_row[r]);
    if (it != targets.end())
      for (const Piece& p : it->second) { tspos.push_back(p.spos); tepos.push_back(p.epos); }
  }
  tfirst[key_of_row.size()] = tspos.size();
  // ---- Profile::init (Profile.cpp:172-218): the read length is the first single-nM CIGAR of the text ----
  LineSource src;
  src.open(o);
  std::unique_ptr<ChunkReader> rd(new ChunkReader(eng.ctx, src.fp, o.threads));
  int cur = 0;
  bool have = rd->fill(cur);
  int read_length = 0;
  {
    const char* p = rd->buf[cur];
    const char* end = p + rd->len[cur];
    while (p < end && !read_length) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      const char* le = nl ? nl : end;
      read_length = single_match_length(p, le);
      p = nl ? nl + 1 : end;
    }
  }
  if (read_length <= 0) throw Error("Error: no read with a single match in its CIGAR among the first lines of " + src.what);
  Model M;
  M.bases = "ACTG";   // (Config.cpp: the default of "bases"; seqToProfile has no option for it)
  M.kmer = o.kmer; M.bins = std::min(o.bins, read_length); M.read_length = read_length;
  for (int m = 1, p4 = 4; m <= M.kmer; m++, p4 *= 4) M.kmer_count += p4;
  // ---- Profile::train, the counting (Profile.cpp:1442-1469) ----
  t1 = Clock::now();
  std::vector<const char*> keys;
  for (const std::string& k : key_of_row) keys.push_back(k.c_str());
  sg_train_setup S;
  memset(&S, 0, sizeof S);
  S.contig_keys = keys.data(); S.n_contigs = (uint32_t)keys.size();
  S.bases = M.bases.c_str(); S.kmer = M.kmer; S.bins = M.bins;
  S.n_isize = kIsizeCols; S.n_indel_len = kIndelCols; S.count_gc = 1; S.window = kWindow; S.max_reads = o.max_reads;
  if (!tspos.empty()) { S.target_first = tfirst.data(); S.target_spos = tspos.data(); S.target_epos = tepos.data(); }
  S.n_snv = kv.snv_pos.size(); S.snv_contig = kv.snv_contig.data(); S.snv_pos = kv.snv_pos.data(); S.snv_alt = kv.snv_alt.data(); S.snv_homo = kv.snv_homo.data();
  S.n_ins = kv.ins_pos.size(); S.ins_contig = kv.ins_contig.data(); S.ins_pos = kv.ins_pos.data(); S.ins_len = kv.ins_len.data();
  S.n_del = kv.del_pos.size(); S.del_contig = kv.del_contig.data(); S.del_pos = kv.del_pos.data(); S.del_len = kv.del_len.data();
  eng.check(sg_train_begin(eng.ctx, &S), "sg_train_begin");
  // (a chunk's verdict -- a malformed line, the cap -- is known when the next chunk is fed or at sg_train_finish)
  auto engine_said = [&](int rc, const char* where) {
    if (rc == SG_OK) return;
    const std::string m = sg_last_error(eng.ctx);
    if (rc == SG_ERR_INVALID && m.find("malformed read") != std::string::npos) throw Error(m, 1);
    throw Error(std::string("GPU engine error in ") + where + ": " + m);
  };
  while (have) {
    // the next chunk is read while the device works on this one
    bool more = false;
    std::string reader_error;
    std::thread reader([&]() { try { more = rd->fill(cur ^ 1); } catch (const std::exception& e) { reader_error = e.what(); } });
    const int rc = sg_train_feed(eng.ctx, rd->buf[cur], rd->len[cur]);
    reader.join();
    engine_said(rc, "sg_train_feed");
    if (!reader_error.empty()) throw Error(reader_error);
    st.sam_bytes += rd->len[cur];
    cur ^= 1;
    have = more && !sg_train_capped(eng.ctx);   // (at the cap the reference leaves its loop: the rest of the input is not read)
  }
  const size_t subs_n = (size_t)M.kmer_count * M.bins * 4, kmers_n = (size_t)M.bins * M.kmer_count, qual_n = (size_t)16 * M.bins * 94;
  std::vector<uint64_t> c_subs1(subs_n), c_subs2(subs_n), c_kmers(kmers_n), c_qual(qual_n), c_isize(kIsizeCols), c_ins(kIndelCols), c_del(kIndelCols);
  sg_train_counts C;
  memset(&C, 0, sizeof C);
  C.subs1 = c_subs1.data(); C.subs2 = c_subs2.data(); C.kmers = c_kmers.data(); C.quality = c_qual.data();
  C.isize = c_isize.data(); C.ins_len = c_ins.data(); C.del_len = c_del.data();
  uint64_t n_gc = 0;
  {
    std::vector<double> gc(1), rc(1);
    int rcode = sg_train_finish(eng.ctx, &C, gc.data(), rc.data(), 0, &n_gc);
    (void)rcode;   // (gc_cap 0: the pairs are counted, the session stays open when there are any)
    if (rcode == SG_ERR_OVERFLOW) {
      M.gcs.resize(n_gc); M.rcs.resize(n_gc);
      engine_said(sg_train_finish(eng.ctx, &C, M.gcs.data(), M.rcs.data(), n_gc, &n_gc), "sg_train_finish");
    } else {
      engine_said(rcode, "sg_train_finish");
    }
  }
  st.t_reads = since(t1);
  rd.reset();
  src.close();
  if (C.isize_overflow || C.indel_len_overflow)
    throw Error("Error: insert sizes beyond " + std::to_string(kIsizeCols) + " or CIGAR insertions / deletions beyond " + std::to_string(kIndelCols) +
                " bases in the reads (" + std::to_string(C.isize_overflow) + " / " + std::to_string(C.indel_len_overflow) + " of them): not representable");
  auto as_double = [](const std::vector<uint64_t>& v, size_t n) { std::vector<double> d(n); for (size_t i = 0; i < n; i++) d[i] = (double)v[i]; return d; };
  auto grown = [](const std::vector<uint64_t>& v, size_t least) {   // the row as the reference grew it: up to the largest length met
    size_t n = least;
    for (size_t i = v.size(); i-- > least;) if (v[i]) { n = i + 1; break; }
    return n;
  };
  M.subs1 = as_double(c_subs1, subs_n); M.subs2 = as_double(c_subs2, subs_n); M.kmers = as_double(c_kmers, kmers_n); M.quality = as_double(c_qual, qual_n);
  M.isize = as_double(c_isize, grown(c_isize, 10)); M.ins = as_double(c_ins, grown(c_ins, 1)); M.del = as_double(c_del, grown(c_del, 1));
  M.insert_rate = (double)C.insert_events; M.del_rate = (double)C.delete_events; M.base_count = (double)C.cigar_chars;
  // ---- the GC model (Profile.cpp:1471-1481) ----
  const double med = median_of(M.rcs);
  if (med < 5) {
    if (!quiet) std::cerr << "\nWarning: no enough reads to evaluate GC-content effects!" << std::endl;
    for (double& m : M.gc_means) m = 1;   // Profile::initGCParas, :705-711
    M.gc_std = 1.0e-5;
  } else {
    fit_gc_model(M, output + ".gc");
    M.gc_fitted = true;
    if (!quiet) std::cerr << "\nread counts std: " << M.gc_std << std::endl;
  }
  normalise(M);
  if (!quiet) std::cerr << "insert rate: " << M.insert_rate << ", deletion rate: " << M.del_rate << std::endl;
  std::string stamp;
  if (o.stamp) stamp = o.stamp;
  else { time_t now; time(&now); stamp = asctime(gmtime(&now)); }
  if (!output.empty()) {
    std::ofstream ofs(output.c_str());
    if (!ofs.is_open()) throw Error("Error: cannot open file to save model training results:\n" + output, -1);
    write_profile(M, ofs, src.what, stamp);
  } else {
    write_profile(M, std::cout, src.what, stamp);
  }
  st.lines = C.lines; st.reads_counted = C.reads_counted; st.gc_rejected = C.gc_rejected; st.gc_windows = C.gc_windows;
  st.gc_pairs = n_gc; st.skipped_overhang = C.skipped_overhang; st.read_length = read_length; st.bins = M.bins;
  st.capped = (int32_t)C.capped;
  st.gc_fitted = M.gc_fitted ? 1 : 0; st.insert_rate = M.insert_rate; st.del_rate = M.del_rate; st.std_isize = M.std_isize; st.gc_std = M.gc_std;
  st.t_total = since(t0);
}

}  // namespace
}  // namespace simu

extern "C" void simu_train_default_options(simu_train_options* o) {
  std::memset(o, 0, sizeof *o);
  o->kmer = 3;
  o->bins = 50;
  o->threads = 8;
}

extern "C" int simu_train(const simu_train_options* opt, simu_train_stats* stats, char* err, size_t err_len) {
  auto set_err = [&](const std::string& m) {
    if (err && err_len) { strncpy(err, m.c_str(), err_len - 1); err[err_len - 1] = '\0'; }
  };
  if (!opt) { set_err("no options"); return 1; }
  simu_train_stats st;
  std::memset(&st, 0, sizeof st);
  try {
    simu::run(*opt, st);
    if (stats) *stats = st;
    return 0;
  } catch (const simu::Error& e) {
    set_err(e.what());
    return e.exit_code ? e.exit_code : 1;
  } catch (const std::exception& e) {
    set_err(e.what());
    return 1;
  }
}
