// host/profile.cpp -- see profile.h.
#include "profile.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace simu {

static const double ZERO_FINAL = 2.2204e-16;  // lib/mydefine/MyDefine.cpp:20

namespace {
struct LineReader {  // getNextLine, lib/mydefine/MyDefine.cpp:239-251: skips empty and '#' lines
  std::ifstream ifs;
  int line_num = 0;
  std::string file;
  bool next(std::string& line) {
    line.clear();
    while (std::getline(ifs, line)) {
      line_num++;
      if (!line.empty() && line[0] != '#') break;
    }
    return !line.empty();
  }
  [[noreturn]] void malformed(const std::string& line) const {
    throw Error("Error: malformed model file " + file + " @line " + std::to_string(line_num) + "\n" + line);
  }
  void need(std::string& line) {
    if (!next(line)) throw Error("Error: malformed profile file " + file);
  }
};

// row-normalise (Matrix::normalize(0), lib/matrix/Matrix.h:495-501) then cumulate (:506-522)
void normalize_rows(double* m, size_t rows, size_t w) {
  for (size_t r = 0; r < rows; r++) {
    double* row = m + r * w;
    double sum = 0;
    for (size_t j = 0; j < w; j++) sum += row[j];
    for (size_t j = 0; j < w; j++) row[j] /= (ZERO_FINAL + sum);
  }
}
void cumsum_rows(double* m, size_t rows, size_t w) {
  for (size_t r = 0; r < rows; r++) {
    double* row = m + r * w;
    for (size_t j = 1; j < w; j++) row[j] = row[j] + row[j - 1];
  }
}
}  // namespace

int Profile::base_index(char c) const {
  for (int i = 0; i < n_bases; i++)
    if (bases[i] == c) return i;
  return -1;
}

// Table order of Profile::initKmers: contexts with 1 real base first ("XXb"), then 2, ... each block
// in base-N counting order with the last base least significant.
int Profile::kmer_index(const std::string& s) const {
  if ((int)s.size() != kmer) return -1;
  int lead = 0;
  while (lead < kmer && s[lead] == 'X') lead++;
  int m = kmer - lead;
  if (m == 0) return -1;
  int off = 0, p = 1;
  for (int t = 1; t < m; t++) { p *= n_bases; off += p; }
  int v = 0;
  for (int i = lead; i < kmer; i++) {
    int b = base_index(s[i]);
    if (b < 0) return -1;
    v = v * n_bases + b;
  }
  return off + v;
}

void Profile::train(const std::string& file, bool paired_, int insert_size_) {
  paired = paired_;
  insert_size = insert_size_;
  LineReader rd;
  rd.file = file;
  rd.ifs.open(file.c_str());
  if (!rd.ifs.is_open()) throw Error("can not open file " + file, -1);
  std::string line;
  std::string b;
  int bin_count = -1, km = -1, rl = -1;
  // header: bases / binCount / kmer / readLength in any order (Profile.cpp:953-998)
  while (rd.next(line)) {
    std::vector<std::string> f = split(line, ':');
    if (f.size() != 2) rd.malformed(line);
    std::string key = trim(f[0]), val = trim(f[1]);
    if (key == "bases") { b = val; if (b.empty()) rd.malformed(line); }
    else if (key == "binCount") { bin_count = atoi(val.c_str()); if (bin_count <= 0) rd.malformed(line); }
    else if (key == "kmer") { km = atoi(val.c_str()); if (km <= 0) rd.malformed(line); }
    else if (key == "readLength") { rl = atoi(val.c_str()); if (rl <= 0) rd.malformed(line); }
    else rd.malformed(line);
    if (!b.empty() && bin_count > 0 && km > 0 && rl > 0) break;
  }
  if (b.empty() || bin_count <= 0 || km <= 0 || rl <= 0) throw Error("Error: malformed model file " + file);
  bases = b; n_bases = (int)b.size(); kmer = km; read_length = rl;
  bins = bin_count > read_length ? read_length : bin_count;  // Profile::init, Profile.cpp:184-188
  kmer_count = 0;
  for (int m = 1, p = 1; m <= kmer; m++) { p *= n_bases; kmer_count += p; }
  const int N = n_bases;
  subs1.assign((size_t)kmer_count * bins * N, 0.0);
  subs2.assign((size_t)kmer_count * bins * N, 0.0);
  qual.assign((size_t)N * N * bins * n_qual, 0.0);
  std::vector<double> ins_freq(1, 0.0), del_freq(1, 0.0);
  for (double& g : gc_means) g = 0;
  insert_rate = del_rate = std_isize = gc_std = 0;

  int loaded = 0;
  std::vector<std::string> f;
  auto read_vec = [&](std::vector<double>& dst) {
    rd.need(line);
    f = split(line, '\t');
    if (f.empty()) rd.malformed(line);
    dst.resize(f.size());
    for (size_t j = 0; j < f.size(); j++) dst[j] = atof(trim(f[j]).c_str());
  };
  while (rd.next(line)) {
    if (line == "[Insert Rate]") { rd.need(line); insert_rate = atof(trim(line).c_str()); loaded++; }
    else if (line == "[Insert Frequency]") { read_vec(ins_freq); loaded++; }
    else if (line == "[Deletion Rate]") { rd.need(line); del_rate = atof(trim(line).c_str()); loaded++; }
    else if (line == "[Deletion Frequency]") { read_vec(del_freq); loaded++; }
    else if (line == "[Substitution Probs]") {
      for (int i = 0; i < kmer_count; i++) {
        rd.need(line);
        f = split(line, ':');
        if (f.size() != 2 || trim(f[0]) != "kmer") rd.malformed(line);
        int kidx = kmer_index(trim(f[1]));
        if (kidx < 0) throw Error("Error: unrecognized kmer @line " + std::to_string(rd.line_num) + " in profile file " + file + "\n" + line);
        for (int j = 0; j < 2 * bins; j++) {
          rd.need(line);
          f = split(line, '\t');
          if ((int)f.size() != N) rd.malformed(line);
          double* dst = (j < bins ? subs1.data() : subs2.data()) + ((size_t)kidx * bins + (j % bins)) * N;
          for (int k = 0; k < N; k++) dst[k] = atof(trim(f[k]).c_str());
        }
      }
      loaded++;
    } else if (line == "[Base Quality Distribution]") {
      for (int i = 0; i < N * N; i++) {
        rd.need(line);
        f = split(line, ':');
        if (f.size() != 2 || trim(f[0]) != "basePairIndx") rd.malformed(line);
        int bp = atoi(trim(f[1]).c_str());
        if (bp < 0 || bp > N * N - 1)
          throw Error("Error: unrecognized basePairIndx @line " + std::to_string(rd.line_num) + " in profile file " + file + "\n" + line);
        for (int j = 0; j < bins; j++) {
          rd.need(line);
          f = split(line, '\t');
          if ((int)f.size() != n_qual) rd.malformed(line);
          double* dst = qual.data() + ((size_t)bp * bins + j) * n_qual;
          for (int k = 0; k < n_qual; k++) dst[k] = atof(trim(f[k]).c_str());
        }
      }
      loaded++;
    } else if (line == "[Insert Size Standard Deviation]") { rd.need(line); std_isize = atof(trim(line).c_str()); loaded++; }
    else if (line == "[Log Ratio Mean Value]") {
      for (int j = 0; j < 101; j++) {
        rd.need(line);
        f = split(line, '\t');
        if (f.size() != 2) rd.malformed(line);
        int gc = atoi(f[0].c_str());
        if (gc < 0 || gc > 100) rd.malformed(line);
        gc_means[gc] = atof(f[1].c_str());
      }
      loaded++;
    } else if (line == "[Log Ratio Standard Deviation]") { rd.need(line); gc_std = atof(trim(line).c_str()); loaded++; }
  }
  if (loaded < 9) throw Error("Error: corrupted model file " + file + ", failed to load some parameters!");

  // ---- normParas(true), Profile.cpp:836-932 ----
  for (int i = 0; i < kmer_count; i++) {
    // context id -> its last base: every block's size is a multiple of N and the last base is the
    // least significant digit, so it is simply id mod N
    const int last = i % N;
    for (std::vector<double>* tab : {&subs1, &subs2}) {
      double* m = tab->data() + (size_t)i * bins * N;
      normalize_rows(m, bins, N);
      for (int j = 0; j < bins; j++) {
        double sum = 0;
        for (int k = 0; k < N; k++) sum += m[j * N + k];
        if (sum < ZERO_FINAL) m[j * N + last] = 1;  // unseen context: copy the base (Profile.cpp:848-853)
      }
    }
  }
  normalize_rows(qual.data(), (size_t)N * N * bins, n_qual);
  isize_cdf.clear();
  if (paired && std_isize > 0) {  // discretised, truncated normal around insertSize+1 (Profile.cpp:912-930)
    int mean = insert_size + 1;
    int interval = 6 * std_isize;
    int lo = std::max(mean - interval / 2, read_length);
    int hi = 2 * mean - lo;
    if (hi < lo) throw Error("Error: empty insert size range");
    isize_min = lo;
    isize_cdf.resize(hi - lo + 1);
    const double PI = 3.1415926;  // lib/mydefine/MyDefine.cpp:54
    for (int i = 0; i <= hi - lo; i++)
      isize_cdf[i] = exp(-pow((double)(lo + i) - mean, 2) / (2 * pow(std_isize, 2))) / (sqrt(2 * PI) * std_isize);
    normalize_rows(isize_cdf.data(), 1, isize_cdf.size());
  }
  // ---- initCDFs, Profile.cpp:1367-1434 ----
  ins_cdf = ins_freq; cumsum_rows(ins_cdf.data(), 1, ins_cdf.size());
  del_cdf = del_freq; cumsum_rows(del_cdf.data(), 1, del_cdf.size());
  normalize_rows(qual.data(), (size_t)N * N * bins, n_qual);  // normalised a second time (:1397)
  cumsum_rows(qual.data(), (size_t)N * N * bins, n_qual);
  if (!isize_cdf.empty()) cumsum_rows(isize_cdf.data(), 1, isize_cdf.size());
  cumsum_rows(subs1.data(), (size_t)kmer_count * bins, N);
  has_sub2 = paired && std_isize > 0;
  if (has_sub2) cumsum_rows(subs2.data(), (size_t)kmer_count * bins, N);
}

sg_profile_cdf Profile::view() const {
  sg_profile_cdf v;
  std::memset(&v, 0, sizeof v);
  v.n_bases = n_bases;
  for (int i = 0; i < n_bases && i < 7; i++) v.bases[i] = bases[i];
  v.kmer = kmer; v.bins = bins; v.read_length = read_length; v.n_qual = n_qual; v.min_qual = min_qual;
  v.insert_rate = insert_rate; v.del_rate = del_rate;
  v.ins_cdf = ins_cdf.data(); v.n_ins = (int)ins_cdf.size();
  v.del_cdf = del_cdf.data(); v.n_del = (int)del_cdf.size();
  v.subs_cdf1 = subs1.data();
  v.subs_cdf2 = has_sub2 ? subs2.data() : nullptr;
  v.qual_cdf = qual.data();
  v.isize_cdf = isize_cdf.empty() ? nullptr : isize_cdf.data();
  v.n_isize = (int)isize_cdf.size();
  v.isize_min = isize_min;
  v.insert_size = insert_size;
  return v;
}

void Profile::build_gc_quantiles() {
  const int N = 1 << 14;
  gc_quantiles.assign(N + 1, 0.0);
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
    gc_quantiles[k] = q;
    gc_quantiles[N - k] = -q;
  }
}

}  // namespace simu
