// host/variants.cpp -- input parsers: variation file (c/s/i/d rows), SNP table, BED targets,
// abundance matrix.  Formats and error texts follow the reference
// (lib/genome/Genome.cpp:41-339, lib/snp/snp.cpp:12-35,147-203).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "genome.h"

namespace simu {

static char complement(char c) {  // lib/snp/snp.cpp:84-97
  switch (c) {
    case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
    case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
    default: return 'N';
  }
}

void Genome::load_variations() {
  const std::string file = cfg.str["variation"];
  if (file.empty()) return;
  std::ifstream ifs(file.c_str());
  if (!ifs.is_open()) throw Error("can not open file " + file, -1);
  std::string line;
  int line_num = 0;
  auto bad = [&](const std::string& what) {
    return Error("ERROR: " + what + " at line " + std::to_string(line_num) + " in file " + file + "\n" + line);
  };
  auto known_popu = [&](const std::string& p) {
    if (std::find(cfg.popu_names.begin(), cfg.popu_names.end(), p) == cfg.popu_names.end())
      throw bad("unrecognized population identifier");
  };
  auto zygosity = [&](const std::string& code, const char* what) {
    if (code != "homo" && code != "het") throw bad(std::string("unrecognized ") + what + " type");
    return code == "het" ? HET : HOMO;
  };
  while (std::getline(ifs, line)) {
    line_num++;
    if (line.empty()) continue;
    std::vector<std::string> f = split(line, '\t');
    const std::string kind = f.empty() ? std::string() : f[0];
    const size_t want = (kind == "c" || kind == "s") ? 7 : 6;
    if (kind != "c" && kind != "s" && kind != "i" && kind != "d") throw bad("unrecognized aberraton type");
    if (f.size() != want)
      throw Error("ERROR: line " + std::to_string(line_num) + " has wrong number of fields in file " + file + "\n" + line);
    known_popu(f[1]);
    const std::string chr = abbr_of_chr(f[2]);
    if (kind == "c") {
      float cn = atof(f[5].c_str()), mcn = atof(f[6].c_str());
      if (cn < mcn) throw bad("total copy number should be not lower than major copy number");
      if (cn - mcn > mcn) mcn = cn - mcn;
      cnvs[f[1]][chr].push_back(CNV{atol(f[3].c_str()), atol(f[4].c_str()), cn, mcn});
    } else if (kind == "s") {
      char ref = f[4].at(0), alt = f[5].at(0);
      if (ref == alt) throw bad("the mutated allele should be not same as the reference allele");
      snvs[f[1]][chr].push_back(SNV{atol(f[3].c_str()), ref, alt, zygosity(f[6], "SNV")});
    } else if (kind == "i") {
      inserts[f[1]][chr].push_back(Insertion{atol(f[3].c_str()), f[4], zygosity(f[5], "insert")});
    } else {
      dels[f[1]][chr].push_back(Deletion{atol(f[3].c_str()), atoi(f[4].c_str()), zygosity(f[5], "deletion")});
    }
  }
}

void Genome::load_snps() {
  const std::string file = cfg.str["snp"];
  if (file.empty()) return;
  FILE* fp = fopen(file.c_str(), "r");
  if (!fp) throw Error("can not open SNP file " + file, -1);
  char buf[1000];  // same line buffer size as the reference (snp.cpp:157)
  while (fgets(buf, sizeof buf, fp)) {
    // six tab-separated columns: id, chromosome, position, observed "X/Y", strand, reference base
    char* col[8];
    int n = 0;
    col[n++] = buf;
    for (char* p = buf; *p && n < 8; p++)
      if (*p == '\t') { *p = '\0'; col[n++] = p + 1; }
    if (n != 6) continue;  // the reference only warns about malformed rows
    char ref = *col[5];
    const char strand = *col[4];
    std::vector<std::string> alleles = split(col[3], '/');
    if (alleles.size() < 2 || alleles[0].empty() || alleles[1].empty()) continue;
    if (strand == '-') ref = complement(ref);
    char nuc = (alleles[0][0] == ref) ? alleles[1][0] : alleles[0][0];  // the non-reference allele
    if (strand == '-') nuc = complement(nuc);
    snps[abbr_of_chr(col[1])].push_back(SNP{atoll(col[2]), nuc});
  }
  fclose(fp);
}

void Genome::load_targets() {
  const std::string file = cfg.str["target"];
  if (file.empty()) return;
  std::ifstream ifs(file.c_str());
  if (!ifs.is_open()) throw Error("can not open target file " + file, -1);
  std::string line;
  int line_num = 0;
  while (std::getline(ifs, line)) {
    line_num++;
    std::vector<std::string> f = split(line, '\t');
    if (f.size() < 3)
      throw Error("ERROR: line " + std::to_string(line_num) + " should have at least 3 fields in file " + file + "\n" + line);
    const std::string chr = abbr_of_chr(f[0]);
    const long chr_len = chrom_len(chr);
    if (chr_len <= 0) continue;
    // BED start/end widened by 50 bp each side (Genome.cpp:270-279)
    Target t;
    t.spos = std::max(1L, atol(f[1].c_str()) - 50 + 1);
    const long e = atol(f[2].c_str());
    const long tmp = e <= 0 ? chr_len - (-e) % chr_len : e;
    t.epos = std::min(chr_len, tmp + 50);
    targets[chr].push_back(t);
  }
}

// Genome::divideTargets (Genome.cpp:684-739): pieces of kFragSize, the last piece absorbs the rest
void Genome::divide_targets() {
  std::map<std::string, std::vector<Target>> out;
  for (auto& kv : targets) {
    for (const Target& t : kv.second) {
      long spos = t.spos;
      const long size = t.epos - t.spos + 1;
      const int k = (int)(size / kFragSize);
      for (int i = 0; i < k; i++) {
        Target p{spos, i == k - 1 ? t.epos : spos + (long)kFragSize - 1};
        spos = p.epos + 1;
        out[kv.first].push_back(p);
      }
      if (spos <= t.epos) out[kv.first].push_back(Target{spos, t.epos});
    }
  }
  targets.swap(out);
}

void Genome::load_abundance() {
  const std::string file = cfg.str["abundance"];
  if (file.empty()) return;
  std::ifstream ifs(file.c_str());
  if (!ifs.is_open()) throw Error("can not open abundance file " + file, -1);
  std::string line;
  int line_num = 0;
  while (std::getline(ifs, line)) {
    line_num++;
    std::vector<std::string> f = split(line, '\t');
    if (f.size() != cfg.popu_names.size())
      throw Error("ERROR: line " + std::to_string(line_num) + " has wrong number of fields in file " + file + "\n" + line);
    std::vector<float> props;
    float sum = 0;
    for (const std::string& x : f) {
      float p = atof(x.c_str());
      sum += p;
      props.push_back(p);
    }
    if (fabs(1 - sum) > 0.001)
      throw Error("ERROR: the sum of abundances is not equal to one at line " + std::to_string(line_num) + " in file " + file + "\n" + line);
    mix_props.push_back(props);
  }
}

std::vector<int> Genome::assign_contigs(const std::vector<uint64_t>& lengths, int world) {
  std::vector<size_t> order(lengths.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return lengths[a] > lengths[b]; });
  std::vector<uint64_t> load((size_t)std::max(1, world), 0);
  std::vector<int> out(lengths.size(), 0);
  for (size_t i : order) {
    const size_t r = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
    out[i] = (int)r;
    load[r] += lengths[i];
  }
  return out;
}

void Genome::load_data() {
  load_variations();
  load_snps();
  const auto t0 = std::chrono::steady_clock::now();
  const int threads = (int)std::max<long long>(1, cfg.num["threads"]);
  bool sharded_ingest = false;
  if (device_haps && shard_contigs && shard_world > 1 && fa.load_index(cfg.str["ref"], threads)) {
    // ranks own whole contigs: ingest only those (the index comes from the .fai or a host header scan)
    std::vector<uint64_t> lens;
    for (const std::string& k : fa.names) lens.push_back(fa.contigs[fa.contig_of.at(k)].length);
    set_owners(fa.names, lens);
    std::vector<char> owned(fa.contigs.size(), 0);
    for (size_t i = 0; i < fa.names.size(); i++)
      if (owns(fa.names[i])) owned[fa.contig_of.at(fa.names[i])] = 1;
    fa.open_owned_on_device(cfg.str["ref"], engine, threads, owned);
    sharded_ingest = true;
  }
  if (!sharded_ingest) {
    if (device_haps) fa.open_on_device(cfg.str["ref"], engine, threads);
    else fa.open(cfg.str["ref"]);
    if (shard_contigs && shard_world > 1) {  // ingest replicated (host haplotypes, or not a plain FASTA): planning and sampling still sharded
      std::vector<uint64_t> lens;
      for (const std::string& k : fa.names) lens.push_back((uint64_t)fa.length(k));
      set_owners(fa.names, lens);
    }
  }
  t_reference = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  chromosomes = fa.names;
  load_targets();
  divide_targets();
  load_abundance();
}

}  // namespace simu
