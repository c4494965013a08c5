// host/genome.cpp -- segments, haplotype chains and sampling windows (see genome.h).
#include "genome.h"

#include <algorithm>
#include <cmath>

namespace simu {

namespace {
// Non-inserting lookups: haplotypes of different chromosomes are built concurrently, so the shared
// maps must never be touched through operator[].
template <class V>
const std::vector<V>& lookup(const std::map<std::string, std::vector<V>>& m, const std::string& k) {
  static const std::vector<V> empty;
  auto it = m.find(k);
  return it == m.end() ? empty : it->second;
}
template <class V>
const std::vector<V>& lookup2(const std::map<std::string, std::map<std::string, std::vector<V>>>& m,
                              const std::string& a, const std::string& b) {
  static const std::vector<V> empty;
  auto it = m.find(a);
  return it == m.end() ? empty : lookup(it->second, b);
}
}  // namespace

long Genome::chrom_len(const std::string& chr) const {  // Genome::getChromLen, Genome.cpp:383-396
  if (std::find(chromosomes.begin(), chromosomes.end(), chr) == chromosomes.end()) return 0;
  return fa.length(chr);
}
long Genome::genome_length() const {
  long n = 0;
  for (const std::string& c : chromosomes) n += chrom_len(c);
  return n;
}
long Genome::target_length() const {  // Genome::getTargetLength, Genome.cpp:405-419
  if (targets.empty()) return genome_length();
  long n = 0;
  for (auto& kv : targets)
    for (const Target& t : kv.second) n += t.epos - t.spos + 1;
  return n;
}
int Genome::popu_index(const std::string& p) const {
  return (int)(std::find(cfg.popu_names.begin(), cfg.popu_names.end(), p) - cfg.popu_names.begin());
}
int Genome::chr_index(const std::string& c) const {
  return (int)(std::find(fa.names.begin(), fa.names.end(), c) - fa.names.begin());
}

// Genome::divideSegment, Genome.cpp:741-763: <=1 Mbp pieces; a tail shorter than half a piece is
// merged into the last full piece.
void Genome::divide_segment(std::vector<Segment>& out, const std::string& chr, long s, long e, int cn, int mcn, int& idx) {
  auto emit = [&](long a, long b) {
    Segment g;
    g.index = idx++;
    g.start = a; g.end = b; g.cn = cn; g.mcn = mcn;
    auto it = targets.find(chr);
    if (!targets.empty() && it != targets.end()) {  // Segment::initTargets, Segment.cpp:67-79
      const std::vector<Target>& ts = it->second;
      for (size_t i = 0; i < ts.size(); i++) {
        const long sp = ts[i].spos, ep = ts[i].epos;
        if ((sp >= a && sp <= b) || (ep >= a && ep <= b) || (sp < a && ep > b)) g.targets.push_back((int)i);
      }
    }
    out.push_back(std::move(g));
  };
  const long size = e - s + 1;
  const int n = (int)(size / kSegMaxSize);
  const unsigned m = (unsigned)(size - (long)n * kSegMaxSize);
  for (int i = 0; i < n; i++) {
    if (i == n - 1 && m < kSegMaxSize / 2) {
      emit(s, e);
      s = e + 1;
    } else {
      emit(s, s + kSegMaxSize - 1);
      s += kSegMaxSize;
    }
  }
  if (s <= e) emit(s, e);
}

void Genome::generate_segments() {
  const int ploidy = cfg.ploidy();
  const int mcn = (int)ceil((float)ploidy / 2);
  if (!targets.empty()) {  // only chromosomes that carry targets, in map (name) order -- Genome.cpp:640-654
    chromosomes.clear();
    for (auto& kv : targets) chromosomes.push_back(kv.first);
  }
  for (const std::string& popu : cfg.popu_names) {
    for (const std::string& chr : chromosomes) {
      ChromPlan& plan = plans[popu][chr];
      int idx = 0;
      long next = 1;
      const long clen = chrom_len(chr);
      for (CNV cnv : cnvs[popu][chr]) {  // file order, no sorting, overlaps kept as the reference does
        if (next > clen) break;
        cnv.epos = std::min(cnv.epos, clen);
        if (next < cnv.spos) divide_segment(plan.segs, chr, next, cnv.spos - 1, ploidy, mcn, idx);
        divide_segment(plan.segs, chr, cnv.spos, cnv.epos, (int)cnv.cn, (int)cnv.mcn, idx);
        next = cnv.epos + 1;
      }
      if (next <= clen) divide_segment(plan.segs, chr, next, clen, ploidy, mcn, idx);
    }
  }
}

// Which haplotype copies exist and which form the "major" set (Segment.cpp:149-208).  Draws are
// addressed Philox values: (long)(0 + ploidy * x/2^32), stream KIND_HAP, counter = draw ordinal.
void Genome::choose_haplotypes(Segment& g, uint64_t seed, uint32_t ctx24, uint32_t seg_ord) {
  if (!g.m_indx.empty()) return;
  const int ploidy = cfg.ploidy();
  uint32_t draw = 0;
  auto pick = [&]() {
    Philox4 o = philox4x32_10(seg_ord, draw++, 0, KIND_HAP | (ctx24 << 8), (uint32_t)seed, (uint32_t)(seed >> 32));
    return (int)(long)(0 + (double)ploidy * ((double)o.v[0] / 4294967296.0));
  };
  auto has = [](const std::vector<int>& v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };
  if (g.cn < ploidy) {
    while ((int)g.seq_reps.size() < g.cn) {
      int j = pick();
      if (!has(g.seq_reps, j)) g.seq_reps.push_back(j);
    }
    for (int i = 0; i < g.mcn; i++) g.m_indx.push_back(g.seq_reps[i]);
    return;
  }
  g.seq_reps.assign(ploidy, 1);
  int extra = g.cn - ploidy;
  const int k = pick();
  int i;
  for (i = extra; i >= 0; i--) {
    if (g.seq_reps[k] + i == g.mcn) {
      g.seq_reps[k] += i;
      g.m_indx.push_back(k);
      break;
    }
    if (g.seq_reps[k] + i == g.cn - g.mcn) {
      g.seq_reps[k] += i;
      for (int j = 0; j < ploidy; j++)
        if (j != k) g.m_indx.push_back(j);
      break;
    }
  }
  if (i >= 0) {
    extra -= i;
    // ploidy 1: every draw equals k and the reference spins forever (Segment.cpp:188-197)
    if (extra > 0 && ploidy == 1) throw Error("ERROR: a copy-number gain cannot be placed on a haploid genome (ploidy = 1)");
    while (extra > 0) {
      int j = pick();
      if (j != k) { g.seq_reps[j]++; extra--; }
    }
  } else {
    while (extra > 0) { g.seq_reps[pick()]++; extra--; }
    for (int j = 0; j < ploidy; j++) g.m_indx.push_back(j);
  }
}

// Segment::generateSegSequences (Segment.cpp:210-458): copy-number replication, then SNPs, SNVs,
// insertions and deletions in file order.  Heterozygous events alternate between the major set and
// its complement (running parity per event class).
void Genome::segment_haplotypes(const std::string& popu, const std::string& chr, Segment& g, std::vector<std::string>& haps) {
  const int ploidy = cfg.ploidy();
  const std::string& contig = fa.seqs.at(chr);
  const std::string ref = contig.substr(g.start - 1, g.ref_size());
  const unsigned ref_size = (unsigned)ref.size();
  haps.assign(ploidy, std::string());
  for (int h = 0; h < ploidy; h++) {
    int reps;
    if (g.cn < ploidy) reps = std::find(g.seq_reps.begin(), g.seq_reps.end(), h) != g.seq_reps.end() ? 1 : 0;
    else reps = g.seq_reps[h];
    haps[h].reserve((size_t)reps * ref_size + 64);
    for (int r = 0; r < reps; r++) haps[h] += ref;
  }
  auto major = [&](int h) { return std::find(g.m_indx.begin(), g.m_indx.end(), h) != g.m_indx.end(); };
  // heterozygous event number `parity` (0,1,0,...) goes to the major set when 0, to the others when 1
  auto carries = [&](int h, bool homo, int parity) { return homo || (parity == 0) == major(h); };
  auto substitute = [&](int h, int sindx, char c) {
    std::string& s = haps[h];
    const unsigned copies = (unsigned)s.size() / ref_size;
    for (unsigned t = 0; t < copies; t++) s[sindx + t * ref_size] = c;
  };
  int parity = 0;
  for (const SNP& snp : lookup(snps, chr)) {
    if (snp.pos < g.start || snp.pos > g.end) continue;
    for (int h = 0; h < ploidy; h++)
      if (carries(h, false, parity)) substitute(h, (int)(snp.pos - g.start), snp.nucleotide);
    parity ^= 1;
  }
  parity = 0;
  for (const SNV& v : lookup2(snvs, popu, chr)) {
    if (v.pos < g.start || v.pos > g.end) continue;
    const bool homo = v.type == HOMO;
    for (int h = 0; h < ploidy; h++)
      if (carries(h, homo, parity)) substitute(h, (int)(v.pos - g.start), v.alt);
    if (!homo) parity ^= 1;
  }
  // Length-changing edits.  The reference keeps, per haplotype, maps keyed by the segment-relative
  // position of the bases already inserted / deleted and shifts later edits by the entries at or before
  // them (Segment.cpp:328-334,387-398); std::map::insert keeps the FIRST entry of a repeated key.
  std::vector<std::map<int, int>> ins_at(ploidy), del_at(ploidy);
  std::vector<int> ins_total(ploidy, 0), del_total(ploidy, 0);
  auto shift = [](const std::map<int, int>& m, int sindx) {
    int s = 0;
    for (auto& kv : m) {
      if (kv.first > sindx) break;
      s += kv.second;
    }
    return s;
  };
  parity = 0;
  for (const Insertion& ins : lookup2(inserts, popu, chr)) {
    if (ins.pos < g.start || ins.pos > g.end) continue;
    const bool homo = ins.type == HOMO;
    const int sindx = (int)(ins.pos + 1 - g.start);  // inserted before reference position pos+1
    const int len = (int)ins.seq.size();
    for (int h = 0; h < ploidy; h++) {
      if (!carries(h, homo, parity)) continue;
      std::string& s = haps[h];
      const int offset = shift(ins_at[h], sindx);
      const int unit = (int)ref_size + ins_total[h];
      const int copies = (int)(s.size() / (size_t)unit);
      for (int t = 0; t < copies; t++) s.insert((size_t)(sindx + offset + t * (unit + len)), ins.seq);
      ins_total[h] += len;
      ins_at[h].insert(std::make_pair(sindx, len));
    }
    if (!homo) parity ^= 1;
  }
  parity = 0;
  for (const Deletion& d : lookup2(dels, popu, chr)) {
    if (d.pos < g.start || d.pos > g.end) continue;
    const bool homo = d.type == HOMO;
    const int sindx = (int)(d.pos - g.start);
    for (int h = 0; h < ploidy; h++) {
      if (!carries(h, homo, parity)) continue;
      const int offset = shift(ins_at[h], sindx) - shift(del_at[h], sindx);
      if (sindx + offset < 0) continue;
      std::string& s = haps[h];
      const int unit = (int)ref_size + ins_total[h] - del_total[h];
      const int copies = (int)(s.size() / (size_t)unit);
      for (int t = 0; t < copies; t++) {
        const size_t at = (size_t)(sindx + offset + t * (unit - d.length));
        if (at > s.size()) throw Error("ERROR: deletion at " + std::to_string(d.pos) + " falls outside its haplotype");
        s.erase(at, (size_t)d.length);
      }
      del_total[h] += d.length;
      del_at[h].insert(std::make_pair(sindx, d.length));
    }
    if (!homo) parity ^= 1;
  }
  for (std::string& s : haps)
    for (char& c : s)
      if (c >= 'a' && c <= 'z') c -= 32;  // variant alleles may be lower-case in the input (Segment.cpp:456)
}

// The same edits as segment_haplotypes(), kept as a piece table instead of bytes: every operation the
// reference applies to its std::strings (Segment.cpp:210-447) is an index operation -- assign at i,
// insert at i, erase [i, i+n) -- so the string's final content is a list of ranges of the replicated
// reference slice S0 (copy t of the slice at [t*ref_size, (t+1)*ref_size)) and of inserted literals.
// The substitutions all happen before the first length change, so they are positions of S0 and are
// carried through the table at the end.
void Genome::segment_pieces(const std::string& popu, const std::string& chr, Segment& g, ChromPlan& plan) {
  const int ploidy = cfg.ploidy();
  const int32_t dev_contig = fa.device_row(chr);
  if (dev_contig < 0) throw Error("ERROR: chromosome " + chr + " is not resident on this device");
  const uint32_t contig = (uint32_t)dev_contig;
  // contig.substr(start-1, ref_size) clips at the contig end
  const unsigned ref_size = (unsigned)std::min<long>((long)g.ref_size(), fa.length(chr) - (g.start - 1));
  struct Piece { uint32_t kind; uint64_t src; uint64_t len; };
  auto major = [&](int h) { return std::find(g.m_indx.begin(), g.m_indx.end(), h) != g.m_indx.end(); };
  auto carries = [&](int h, bool homo, int parity) { return homo || (parity == 0) == major(h); };

  std::vector<std::vector<Piece>> table((size_t)ploidy);
  std::vector<uint64_t> size((size_t)ploidy, 0);
  std::vector<std::map<uint64_t, char>> subs((size_t)ploidy);  // S0 index -> allele (a later one replaces an earlier one)
  std::vector<unsigned> copies0((size_t)ploidy, 0);
  for (int h = 0; h < ploidy; h++) {
    int reps;
    if (g.cn < ploidy) reps = std::find(g.seq_reps.begin(), g.seq_reps.end(), h) != g.seq_reps.end() ? 1 : 0;
    else reps = g.seq_reps[h];
    copies0[h] = (unsigned)reps;
    size[h] = (uint64_t)reps * ref_size;
    if (size[h]) table[h].push_back(Piece{0, 0, size[h]});
  }
  auto substitute = [&](int h, int sindx, char c) {
    for (unsigned t = 0; t < copies0[h]; t++) subs[h][(uint64_t)sindx + (uint64_t)t * ref_size] = c;
  };
  int parity = 0;
  for (const SNP& snp : lookup(snps, chr)) {
    if (snp.pos < g.start || snp.pos > g.end) continue;
    for (int h = 0; h < ploidy; h++)
      if (carries(h, false, parity)) substitute(h, (int)(snp.pos - g.start), snp.nucleotide);
    parity ^= 1;
  }
  parity = 0;
  for (const SNV& v : lookup2(snvs, popu, chr)) {
    if (v.pos < g.start || v.pos > g.end) continue;
    const bool homo = v.type == HOMO;
    for (int h = 0; h < ploidy; h++)
      if (carries(h, homo, parity)) substitute(h, (int)(v.pos - g.start), v.alt);
    if (!homo) parity ^= 1;
  }
  // piece-table forms of std::string::insert / erase
  auto split_at = [](std::vector<Piece>& tb, uint64_t at) -> size_t {  // index of the piece starting at `at`
    uint64_t pos = 0;
    for (size_t i = 0; i < tb.size(); i++) {
      if (pos == at) return i;
      if (at < pos + tb[i].len) {
        const uint64_t left = at - pos;
        Piece right{tb[i].kind, tb[i].src + left, tb[i].len - left};
        tb[i].len = left;
        tb.insert(tb.begin() + (long)i + 1, right);
        return i + 1;
      }
      pos += tb[i].len;
    }
    return tb.size();
  };
  auto insert_lit = [&](int h, uint64_t at, uint64_t lit_off, uint64_t len) {
    if (at > size[h]) throw Error("ERROR: insertion falls outside its haplotype on chromosome " + chr);
    if (!len) return;
    const size_t i = split_at(table[h], at);
    table[h].insert(table[h].begin() + (long)i, Piece{1, lit_off, len});
    size[h] += len;
  };
  auto erase = [&](int h, uint64_t at, uint64_t len) {
    len = std::min(len, size[h] - at);  // std::string::erase clamps
    if (!len) return;
    const size_t a = split_at(table[h], at), b = split_at(table[h], at + len);
    table[h].erase(table[h].begin() + (long)a, table[h].begin() + (long)b);
    size[h] -= len;
  };

  std::vector<std::map<int, int>> ins_at(ploidy), del_at(ploidy);
  std::vector<int> ins_total(ploidy, 0), del_total(ploidy, 0);
  auto shift = [](const std::map<int, int>& m, int sindx) {
    int s = 0;
    for (auto& kv : m) {
      if (kv.first > sindx) break;
      s += kv.second;
    }
    return s;
  };
  parity = 0;
  for (const Insertion& ins : lookup2(inserts, popu, chr)) {
    if (ins.pos < g.start || ins.pos > g.end) continue;
    const bool homo = ins.type == HOMO;
    const int sindx = (int)(ins.pos + 1 - g.start);
    const int len = (int)ins.seq.size();
    uint64_t lit_off = ~0ull;
    for (int h = 0; h < ploidy; h++) {
      if (!carries(h, homo, parity)) continue;
      if (lit_off == ~0ull) { lit_off = plan.literals.size(); plan.literals += ins.seq; }
      const int offset = shift(ins_at[h], sindx);
      const int unit = (int)ref_size + ins_total[h];
      const int copies = (int)(size[h] / (uint64_t)unit);
      for (int t = 0; t < copies; t++) insert_lit(h, (uint64_t)((long)sindx + offset + (long)t * (unit + len)), lit_off, (uint64_t)len);
      ins_total[h] += len;
      ins_at[h].insert(std::make_pair(sindx, len));
    }
    if (!homo) parity ^= 1;
  }
  parity = 0;
  for (const Deletion& d : lookup2(dels, popu, chr)) {
    if (d.pos < g.start || d.pos > g.end) continue;
    const bool homo = d.type == HOMO;
    const int sindx = (int)(d.pos - g.start);
    for (int h = 0; h < ploidy; h++) {
      if (!carries(h, homo, parity)) continue;
      const int offset = shift(ins_at[h], sindx) - shift(del_at[h], sindx);
      if (sindx + offset < 0) continue;
      const int unit = (int)ref_size + ins_total[h] - del_total[h];
      const int copies = (int)(size[h] / (uint64_t)unit);
      for (int t = 0; t < copies; t++) {
        const uint64_t at = (uint64_t)((long)sindx + offset + (long)t * (unit - d.length));
        if (at > size[h]) throw Error("ERROR: deletion at " + std::to_string(d.pos) + " falls outside its haplotype");
        erase(h, at, (uint64_t)d.length);
      }
      del_total[h] += d.length;
      del_at[h].insert(std::make_pair(sindx, d.length));
    }
    if (!homo) parity ^= 1;
  }

  // flatten: destination offsets, S0 ranges cut at copy boundaries -> contig coordinates
  for (int h = 0; h < ploidy; h++) {
    g.hap_base[h] = plan.chain_len[h];
    g.hap_len[h] = size[h];
    uint64_t dst = plan.chain_len[h];
    std::vector<std::pair<uint64_t, uint64_t>> s0_at;  // (S0 start of a kind-0 piece, its dst), ascending in both
    std::vector<uint64_t> s0_len;
    for (const Piece& p : table[h]) {
      if (p.kind == 1) {
        for (uint64_t o = 0; o < p.len; o += 0x40000000ull)
          plan.pieces.push_back(sg_hap_piece{dst + o, p.src + o, (uint32_t)std::min<uint64_t>(0x40000000ull, p.len - o), (uint32_t)h, 0, 1});
      } else {
        s0_at.emplace_back(p.src, dst);
        s0_len.push_back(p.len);
        uint64_t o = 0;
        while (o < p.len) {
          const uint64_t s0 = p.src + o, in_copy = s0 % ref_size;
          const uint64_t n = std::min<uint64_t>(p.len - o, ref_size - in_copy);
          plan.pieces.push_back(sg_hap_piece{dst + o, (uint64_t)(g.start - 1) + in_copy, (uint32_t)n, (uint32_t)h, contig, 0});
          o += n;
        }
      }
      dst += p.len;
    }
    plan.chain_len[h] = dst;
    for (const auto& kv : subs[h]) {  // a substituted base that was deleted afterwards is in no piece
      auto it = std::upper_bound(s0_at.begin(), s0_at.end(), std::pair<uint64_t, uint64_t>(kv.first, ~(uint64_t)0));
      if (it == s0_at.begin()) continue;
      --it;
      const size_t i = (size_t)(it - s0_at.begin());
      if (kv.first >= it->first + s0_len[i]) continue;
      plan.patches.push_back(sg_hap_patch{it->second + (kv.first - it->first), (uint32_t)h, (uint32_t)(uint8_t)kv.second});
    }
  }
}

void Genome::build_chains(const std::string& popu, const std::string& chr, uint64_t seed) {
  ChromPlan& plan = plans.at(popu).at(chr);
  if (plan.chains_built) return;
  const int ploidy = cfg.ploidy();
  plan.chains.assign(device_haps ? 0 : ploidy, std::string());
  plan.chain_len.assign(ploidy, 0);
  plan.pieces.clear(); plan.patches.clear(); plan.literals.clear();
  const uint32_t ctx = host_ctx(popu, chr);
  const long clen = fa.length(chr);
  std::vector<std::string> haps;
  for (size_t k = 0; k < plan.segs.size(); k++) {
    Segment& g = plan.segs[k];
    g.hap_base.assign(ploidy, 0);
    g.hap_len.assign(ploidy, 0);
    g.has_seq = false;
    if (g.cn == 0 || clen < g.start) continue;  // Segment.cpp:131-141
    choose_haplotypes(g, seed, ctx, (uint32_t)k);
    if (device_haps) {
      segment_pieces(popu, chr, g, plan);
    } else {
      segment_haplotypes(popu, chr, g, haps);
      for (int h = 0; h < ploidy; h++) {
        g.hap_base[h] = plan.chains[h].size();
        g.hap_len[h] = haps[h].size();
        plan.chains[h] += haps[h];
        plan.chain_len[h] = plan.chains[h].size();
      }
    }
    g.has_seq = true;
  }
  // A fragment that runs off its segment goes on in chrSegs[index + 1 ...] (Segment.cpp:1085-1101), `index` being the
  // number divideSegment gave the segment (Genome.cpp:741-763) -- its place in the list, except when the FASTA names a
  // contig twice: the list then holds the contig's segments twice, numbered from 0 both times, and the LAST segment of the
  // second round continues in the segment behind its own number instead of ending the chain (a single-segment contig: in
  // itself).  The chains get those segments once more behind their end, as a tail no window lies in.
  if (!plan.segs.empty() && (size_t)plan.segs.back().index + 1 < plan.segs.size()) {
    for (size_t k = (size_t)plan.segs.back().index + 1; k < plan.segs.size(); k++) {
      if (!plan.segs[k].has_seq) continue;
      Segment tail = plan.segs[k];   // (a copy: the segment keeps its own place in the chains)
      if (device_haps) {
        segment_pieces(popu, chr, tail, plan);
      } else {
        segment_haplotypes(popu, chr, tail, haps);
        for (int h = 0; h < ploidy; h++) {
          plan.chains[h] += haps[h];
          plan.chain_len[h] = plan.chains[h].size();
        }
      }
    }
  }
  plan.chains_built = true;
}

void Genome::build_windows(const std::string& popu, const std::string& chr) {
  ChromPlan& plan = plans.at(popu).at(chr);
  if (plan.windows_built) return;
  const int ploidy = cfg.ploidy();
  auto push = [&](uint32_t spos, uint32_t len, uint32_t hap) {
    plan.w_spos.push_back(spos);
    plan.w_len.push_back(len);
    plan.w_hap.push_back(hap);
  };
  const std::vector<Target>* ts = nullptr;
  if (!targets.empty()) ts = &lookup(targets, chr);
  for (Segment& g : plan.segs) {
    g.w0 = (uint32_t)plan.w_spos.size();
    if (!g.has_seq) {
      // CN == 0: the reference would dereference NULL (Segment.cpp:564); give the segment a single
      // zero-weight placeholder so that it draws no reads.
      push(0, 1, 0);
    } else if (targets.empty()) {
      for (int h = 0; h < ploidy; h++) {  // 1 kbp tiles + a shorter tail (Segment.cpp:563-592)
        const uint64_t len = g.hap_len[h];
        if (!len) continue;
        const uint64_t full = len / kFragSize;
        for (uint64_t j = 0; j < full; j++) push((uint32_t)(j * kFragSize), kFragSize, (uint32_t)h);
        if (full * kFragSize < len) push((uint32_t)(full * kFragSize), (uint32_t)(len - full * kFragSize), (uint32_t)h);
      }
    } else if (!g.targets.empty()) {
      // BED targets clipped to the segment, repeated for every copy of the haplotype.  Offsets are
      // computed in reference coordinates and applied to the edited string as the reference does
      // (Segment.cpp:595-624).
      for (int h = 0; h < ploidy; h++) {
        const uint64_t len = g.hap_len[h];
        if (!len) continue;
        const int n = ((int)g.seq_reps.size() < ploidy) ? 1 : g.seq_reps[h];
        const long ref_len = (long)(len / (uint64_t)n);
        for (int k = 0; k < n; k++) {
          for (int m : g.targets) {
            const long spos = std::max((*ts)[m].spos, g.start) - g.start;
            const long epos = std::min((*ts)[m].epos, g.start + ref_len - 1) - g.start;
            const long spos_k = (long)(spos + (uint64_t)k * len / (uint64_t)n);
            const long epos_k = (long)(epos + (uint64_t)k * len / (uint64_t)n);
            if (spos_k < 0 || epos_k < spos_k || (uint64_t)epos_k >= len)
              throw Error("ERROR: target window falls outside its haplotype on chromosome " + chr);
            push((uint32_t)spos_k, (uint32_t)(epos_k - spos_k + 1), (uint32_t)h);
          }
        }
      }
    } else {
      push(0, 1, 0);  // segment without targets: one zero-weight placeholder (Segment.cpp:625-630)
    }
    g.w1 = (uint32_t)plan.w_spos.size();
  }
  plan.w_weight.assign(plan.w_spos.size(), 0.0);
  plan.windows_built = true;
}

}  // namespace simu
