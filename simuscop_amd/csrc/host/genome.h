// host/genome.h -- host-side orchestration state: inputs, segments, haplotype chains, windows.
//
// Mirrors the responsibilities of the reference's Genome + Segment classes
// (lib/genome/Genome.cpp, lib/segment/Segment.cpp) with a layout built for the GPU hand-off:
//   * one contiguous "chain" per haplotype index and chromosome (all segments concatenated), so a
//     fragment that spills over a segment end (Segment.cpp:1085-1101, Genome.cpp:599-632) is a plain
//     substring and the chains can be uploaded with one copy each;
//   * windows of a (population, chromosome) in flat SoA vectors instead of four std::vectors per
//     Segment object (Segment.h:45-49);
//   * haplotypes are built ONCE per (population, chromosome) and kept, instead of twice per run
//     (Genome.cpp:793 via getWeightedLength, then :876-878).
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../../include/simuscop_amd.h"
#include "config.h"
#include "fasta.h"
#include "profile.h"

namespace simu {

enum VarType { HET = 0, HOMO = 1 };
struct CNV { long spos, epos; float cn, mcn; };
struct SNV { long pos; char ref, alt; VarType type; };
struct Insertion { long pos; std::string seq; VarType type; };
struct Deletion { long pos; int length; VarType type; };
struct SNP { long long pos; char nucleotide; };
struct Target { long spos, epos; };

struct Segment {
  int index = 0;
  long start = 0, end = 0;  // 1-based inclusive
  int cn = 2, mcn = 1;
  std::vector<int> seq_reps, m_indx;  // chosen once (Segment.cpp:149-208)
  std::vector<int> targets;           // indexes into the chromosome's divided targets (Segment.cpp:67-79)
  bool has_seq = false;               // segSequences != NULL
  std::vector<uint64_t> hap_base, hap_len;  // per haplotype index: position / length inside chain h (len 0 = NULL)
  uint32_t w0 = 0, w1 = 0;            // window range in ChromPlan
  long read_count = 0;
  unsigned ref_size() const { return (unsigned)(end - start + 1); }
  unsigned seq_size() const {         // Segment::getSeqSize, Segment.cpp:643-655
    if (!has_seq) return (unsigned)cn * ref_size();
    unsigned s = 0;
    for (uint64_t l : hap_len) s += (unsigned)l;
    return s;
  }
};

struct ChromPlan {  // one (population, chromosome)
  std::vector<Segment> segs;
  std::vector<std::string> chains;  // [ploidy]  (host haplotypes only)
  // device haplotypes: the chains as copy lists for sg_build_haplotypes
  std::vector<uint64_t> chain_len;
  std::vector<sg_hap_piece> pieces;
  std::vector<sg_hap_patch> patches;
  std::string literals;
  bool chains_built = false, windows_built = false, weighed = false;
  std::vector<uint32_t> w_spos, w_len, w_hap;  // Segment::fragStartPos / (End-Start+1) / hapIndxs
  std::vector<double> w_weight;                // Segment::fragWeights
  // whole-genome runs (no targets): the windows exist on the device only (sg_windows_build / sg_plan_windows); the
  // host keeps one weight sum and one first-window index per segment
  bool dev_windows = false;
  std::vector<double> seg_w;
  std::vector<uint64_t> seg_win0;
  uint32_t store_id = 0;
};

struct Genome {
  static constexpr unsigned kSegMaxSize = 1000000;  // Segment.cpp:15
  static constexpr unsigned kFragSize = 1000;       // Segment.cpp:16

  Config& cfg;
  Fasta fa;
  std::vector<std::string> chromosomes;
  std::map<std::string, std::map<std::string, std::vector<CNV>>> cnvs;
  std::map<std::string, std::map<std::string, std::vector<SNV>>> snvs;
  std::map<std::string, std::map<std::string, std::vector<Insertion>>> inserts;
  std::map<std::string, std::map<std::string, std::vector<Deletion>>> dels;
  std::map<std::string, std::vector<SNP>> snps;
  std::map<std::string, std::vector<Target>> targets;  // after divide_targets
  std::vector<std::vector<float>> mix_props;
  std::map<std::string, std::map<std::string, ChromPlan>> plans;  // [popu][chr]

  // true: the reference lives on the device (Fasta::open_on_device) and build_chains() writes copy
  // lists instead of strings
  bool device_haps = false;
  ::sg_ctx* engine = nullptr;
  double t_reference = 0;  // seconds spent in Fasta::open / open_on_device
  // Multi-GPU, ranks own whole chromosomes: owner_of[name] = rank of that contig (empty: all here).  Only
  // the owned contigs are ingested, cut into haplotype chains, scanned and sampled by this process.  Keyed by name:
  // `chromosomes` is re-made in target-map order for runs with a BED file (generate_segments), the assignment is not.
  int shard_rank = 0, shard_world = 1;
  bool shard_contigs = false;
  std::map<std::string, int> owner_of;
  bool owns(const std::string& chr) const {
    if (owner_of.empty()) return true;
    const auto it = owner_of.find(chr);
    return it != owner_of.end() && it->second == shard_rank;
  }
  void set_owners(const std::vector<std::string>& names, const std::vector<uint64_t>& lengths) {
    const std::vector<int> r = assign_contigs(lengths, shard_world);
    owner_of.clear();
    for (size_t i = 0; i < names.size(); i++) owner_of[names[i]] = r[i];
  }
  // Longest-first greedy assignment of contigs to `world` ranks by length (ties: file order, lowest rank): the read
  // count of a chromosome follows its GC-weighted length, which follows its length.
  static std::vector<int> assign_contigs(const std::vector<uint64_t>& lengths, int world);

  explicit Genome(Config& c) : cfg(c) {}

  void load_data();          // Genome::loadData, Genome.cpp:17-30
  void generate_segments();  // Genome::generateSegments, Genome.cpp:634-682
  long chrom_len(const std::string& chr) const;
  long genome_length() const;
  long target_length() const;
  int popu_index(const std::string& p) const;
  int chr_index(const std::string& c) const;
  uint32_t host_ctx(const std::string& popu, const std::string& chr) const {
    return ((uint32_t)popu_index(popu) << 16) | ((uint32_t)chr_index(chr) & 0xFFFFu);
  }

  // Segment::generateSegSequences for every segment of the chromosome, appended to the chains.
  void build_chains(const std::string& popu, const std::string& chr, uint64_t seed);
  // Window geometry of Segment::getWeightedLength (weights are filled by the caller once GC% is known).
  void build_windows(const std::string& popu, const std::string& chr);

 private:
  void load_variations();
  void load_snps();
  void load_targets();
  void divide_targets();
  void load_abundance();
  void divide_segment(std::vector<Segment>& out, const std::string& chr, long s, long e, int cn, int mcn, int& idx);
  void choose_haplotypes(Segment& g, uint64_t seed, uint32_t ctx24, uint32_t seg_ord);
  void segment_haplotypes(const std::string& popu, const std::string& chr, Segment& g, std::vector<std::string>& out);
  void segment_pieces(const std::string& popu, const std::string& chr, Segment& g, ChromPlan& plan);
};

}  // namespace simu
