// host/simulate.cpp -- the `simuReads <config>` driver (src/simuReads.cpp:24-87 +
// Genome::yieldReads, lib/genome/Genome.cpp:827-960) on top of the C ABI.
//
// Order of work per population:
//   pass 1  for every chromosome: build haplotype chains once (host), upload, GC% per window on the
//           GPU (sg_gc_percent), GC-bias weights and weighted lengths on the host in fp64 -- the read
//           apportioning truncates fp64 products, so it stays where the reference computes it;
//   counts  chromosome / segment / window read counts (Genome::setReadCounts, Segment::setReadCount);
//   pass 2  one GPU batch per chromosome: sg_plan -> sg_sample -> sg_fetch -> FASTQ files.
#include "simulate.h"

#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <functional>
#include <memory>

#include "../../../include/simuscop_amd.h"
#include "genome.h"

namespace simu {

// (long) of a double as the reference's machine code does it (cvttsd2si: NaN or out of range -> LONG_MIN), and sums that
// wrap like its 64-bit registers: see set_read_counts
static inline long to_long_x86(double v) {
  if (!(v > -9223372036854775808.0 && v < 9223372036854775808.0)) return (long)0x8000000000000000ull;
  return (long)v;
}
static inline long wrap_add(long a, long b) { return (long)((unsigned long)a + (unsigned long)b); }
static inline long wrap_sub(long a, long b) { return (long)((unsigned long)a - (unsigned long)b); }
namespace {

using Clock = std::chrono::steady_clock;
inline double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

struct Engine {  // RAII around sg_ctx, turns status codes into simu::Error
  sg_ctx* ctx = nullptr;
  ~Engine() { if (ctx) sg_destroy(ctx); }
  void check(int rc, const char* what) {
    if (rc != SG_OK) throw Error(std::string("GPU engine error in ") + what + ": " + sg_last_error(ctx));
  }
};

struct Sink {  // FASTQ output: <output>/<stem>_1.fq + _2.fq, or <stem>.fq (Genome.cpp:857-866)
  FILE* f1 = nullptr;
  FILE* f2 = nullptr;
  bool bgzf = false, eof_block = true;
  void open(const std::string& dir, const std::string& stem, bool paired, const std::string& suffix0, bool gz = false) {
    close();
    bgzf = gz;
    const std::string suffix = (gz ? ".gz" : "") + suffix0;
    if (paired) {
      std::string a = dir + "/" + stem + "_1.fq" + suffix, b = dir + "/" + stem + "_2.fq" + suffix;
      f1 = fopen(a.c_str(), "wb");
      if (!f1) throw Error("Error: can not open fastq file to save results:\n" + a, -1);
      f2 = fopen(b.c_str(), "wb");
      if (!f2) throw Error("Error: can not open fastq file to save results:\n" + b, -1);
    } else {
      std::string a = dir + "/" + stem + ".fq" + suffix;
      f1 = fopen(a.c_str(), "wb");
      if (!f1) throw Error("Error: can not open fastq file to save results:\n" + a, -1);
    }
  }
  void close() {
    if (bgzf && eof_block) {  // BGZF end-of-file marker
      uint8_t eof[28];
      sg_bgzf_eof(eof);
      if (f1) fwrite(eof, 1, 28, f1);
      if (f2) fwrite(eof, 1, 28, f2);
    }
    if (f1) fclose(f1);
    if (f2) fclose(f2);
    f1 = f2 = nullptr;
  }
  ~Sink() { close(); }
};

struct Driver {
  Config cfg;
  Genome genome{cfg};
  Profile prof;
  Engine eng;
  simu_options opt;
  simu_stats st{};
  uint64_t seed = 0;
  uint32_t batch_id = 0;
  std::string resident;  // "popu\tchr" whose chains are on the device
  std::vector<char> host1, host2;
  std::vector<sg_window> wins;
  std::vector<uint32_t> seg_size, seg_first;
  std::vector<sg_gc_window> gcw;
  std::vector<int32_t> gcv;
  Sink sink;  // a member, so that an error path joins the drain in flight (below) before the files are closed

  ~Driver() {
    if (pending.active && pending.th.joinable()) pending.th.join();
    if (pending.active && pending.handle && eng.ctx) sg_release_outputs(eng.ctx, pending.handle);
    for (void* b : pinned)
      if (b && eng.ctx) sg_host_free(eng.ctx, b);
  }
  void log(const std::string& s) { if (!opt.quiet) std::cerr << s; }

  void upload(const std::string& popu, const std::string& chr) {
    const std::string key = popu + "\t" + chr;
    if (resident == key) return;
    ChromPlan& plan = genome.plans[popu][chr];
    auto t_up = Clock::now();
    struct Acc { double& d; Clock::time_point t; ~Acc() { d += since(t); } } acc{st.t_hap_device, t_up};
    if (genome.device_haps) {
      eng.check(sg_build_haplotypes(eng.ctx, (int32_t)plan.chain_len.size(), plan.chain_len.data(), plan.pieces.data(),
                                    plan.pieces.size(), plan.literals.data(), plan.literals.size(), plan.patches.data(),
                                    plan.patches.size()),
                "sg_build_haplotypes");
    } else {
      std::vector<const char*> ptr;
      std::vector<uint64_t> len;
      for (const std::string& c : plan.chains) { ptr.push_back(c.data()); len.push_back(c.size()); }
      eng.check(sg_upload_haplotypes(eng.ctx, (int32_t)ptr.size(), ptr.data(), len.data()), "sg_upload_haplotypes");
    }
    resident = key;
  }

  // pass 1 for one chromosome: chains, windows, GC%, weights (Segment::getWeightedLength)
  void weigh(const std::string& popu, const std::string& chr) {
    ChromPlan& plan = genome.plans[popu][chr];
    if (plan.weighed) return;
    auto t0 = Clock::now();
    genome.build_chains(popu, chr, seed);   // no-ops when prebuild() already did them
    if (!dev_plan()) genome.build_windows(popu, chr);
    st.t_haplotypes += since(t0);
    t0 = Clock::now();
    upload(popu, chr);
    if (dev_plan()) {
      // Windows, GC%, GC factors, weights and the per-segment weight sums on the device: the host never sees a window
      if (prof.gc_quantiles.empty()) prof.build_gc_quantiles();
      std::vector<sg_window_gen> gens;
      plan.seg_win0.assign(plan.segs.size(), 0);
      uint64_t nwin = 0;
      for (size_t k = 0; k < plan.segs.size(); k++) {
        Segment& g = plan.segs[k];
        plan.seg_win0[k] = nwin;
        g.w0 = (uint32_t)nwin;
        if (g.has_seq)
          for (size_t h = 0; h < g.hap_len.size(); h++) {  // 1 kbp tiles + a shorter tail per haplotype string (Segment.cpp:563-592)
            if (!g.hap_len[h]) continue;
            gens.push_back(sg_window_gen{g.hap_base[h], g.hap_len[h], (uint32_t)h, (uint32_t)k, 0});
            nwin += (g.hap_len[h] + Genome::kFragSize - 1) / Genome::kFragSize;
          }
        g.w1 = (uint32_t)nwin;
      }
      sg_gc_model model;
      model.means = prof.gc_means;
      model.std = prof.gc_std;
      model.quantiles = prof.gc_quantiles.data();
      model.lg_cells = 14;
      model.frag_size = Genome::kFragSize;
      model.full_tile_form = 1;
      model.ctx24 = genome.host_ctx(popu, chr);
      plan.store_id = model.ctx24;
      plan.seg_w.assign(plan.segs.size(), 0.0);
      uint64_t n_out = 0;
      eng.check(sg_windows_build(eng.ctx, plan.store_id, gens.data(), gens.size(), (uint32_t)plan.segs.size(), &model, plan.seg_w.data(), &n_out),
                "sg_windows_build");
      plan.dev_windows = true;
      plan.weighed = true;
      st.t_plan += since(t0);
      return;
    }
    // GC%, GC factor and weight of every window on the device (sg_window_weights); the draws of the factor are
    // addressed by (segment ordinal, window ordinal inside the segment)
    gcw.clear();
    std::vector<uint32_t> widx, seg_ord, win_ord;
    for (size_t k = 0; k < plan.segs.size(); k++) {
      const Segment& g = plan.segs[k];
      if (!g.has_seq || (!genome.targets.empty() && g.targets.empty())) continue;  // placeholder windows carry no GC draw
      for (uint32_t w = g.w0; w < g.w1; w++) {
        gcw.push_back(sg_gc_window{g.hap_base[plan.w_hap[w]] + plan.w_spos[w], plan.w_hap[w], plan.w_len[w]});
        widx.push_back(w);
        seg_ord.push_back((uint32_t)k);
        win_ord.push_back(w - g.w0);
      }
    }
    if (prof.gc_quantiles.empty()) prof.build_gc_quantiles();
    sg_gc_model model;
    model.means = prof.gc_means;
    model.std = prof.gc_std;
    model.quantiles = prof.gc_quantiles.data();
    model.lg_cells = 14;
    model.frag_size = Genome::kFragSize;
    // full 1 kbp tiles: factor/fragSize; tails and targets: factor*len/(fragSize*fragSize)
    // (Segment.cpp:576,586,615 -- the two forms round differently, keep both)
    model.full_tile_form = genome.targets.empty() ? 1 : 0;
    model.ctx24 = genome.host_ctx(popu, chr);
    std::vector<double> wts(gcw.size());
    eng.check(sg_window_weights(eng.ctx, gcw.data(), seg_ord.data(), win_ord.data(), gcw.size(), &model, wts.data(), nullptr),
              "sg_window_weights");
    for (size_t q = 0; q < widx.size(); q++) plan.w_weight[widx[q]] = wts[q];
    plan.weighed = true;
    st.t_plan += since(t0);
  }

  // windows made on the device (whole-genome runs; SIMU_HOST_PLAN=1 keeps the host planner for comparison runs)
  bool dev_plan() const {
    static const bool host_plan = getenv("SIMU_HOST_PLAN") != nullptr;
    return genome.targets.empty() && !host_plan;
  }
  static double seg_weight(const ChromPlan& plan, const Segment& g) {
    if (plan.dev_windows) return plan.seg_w[(size_t)(&g - plan.segs.data())];
    double s = 0;
    for (uint32_t w = g.w0; w < g.w1; w++) s += plan.w_weight[w];
    return s;
  }

  // Haplotype construction of a population's chromosomes is independent work: the reference does it
  // serially on the main thread, twice (Genome.cpp:793, :876-878); here once, on `threads` workers.
  void prebuild(const std::string& popu) {
    std::vector<std::string> todo;
    for (const std::string& chr : genome.chromosomes)
      if (genome.owns(chr) && !genome.plans[popu][chr].chains_built) todo.push_back(chr);
    const size_t nthreads = std::min<size_t>((size_t)std::max<long long>(1, cfg.num["threads"]), todo.size());
    if (nthreads <= 1) return;  // weigh() builds on demand
    auto t0 = Clock::now();
    std::atomic<size_t> next(0);
    std::mutex err_mu;
    std::string err;
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nthreads; t++) {
      pool.emplace_back([&]() {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= todo.size()) return;
          try {
            genome.build_chains(popu, todo[i], seed);
            if (!dev_plan()) genome.build_windows(popu, todo[i]);
          } catch (const std::exception& e) {
            std::lock_guard<std::mutex> lk(err_mu);
            if (err.empty()) err = e.what();
          }
        }
      });
    }
    for (std::thread& th : pool) th.join();
    st.t_haplotypes += since(t0);
    if (!err.empty()) throw Error(err);
  }

  // Genome::setReadCounts, Genome.cpp:783-825
  std::map<std::string, std::vector<double>> chr_wl_of;  // per population: GC-weighted length of every chromosome
  void set_read_counts(const std::string& popu, long reads) {
    prebuild(popu);
    std::vector<double>& chr_wl = chr_wl_of[popu];
    if (chr_wl.empty()) {
      for (const std::string& chr : genome.chromosomes) {
        double c = 0;
        if (genome.owns(chr)) {
          weigh(popu, chr);
          ChromPlan& plan = genome.plans[popu][chr];
          for (const Segment& g : plan.segs) c += seg_weight(plan, g);
        }
        chr_wl.push_back(c);
      }
      if (!genome.owner_of.empty() || (opt.shard_contigs && opt.exchange)) {
        // the other ranks' chromosomes: one small exchange per population (RCCL / gloo all-reduce in the torchrun front
        // end, the parent's pipes under `simuReads --gpus N`); each entry has one owner, so the sum is exact.  (A single
        // rank that was handed an exchange runs it too -- the sum of one contribution -- so that the transport can be
        // rehearsed on one GPU: tests/test_gpu_nccl_world1.py.)
        if (!opt.exchange) throw Error("ERROR: chromosome sharding needs an exchange callback (simu_options.exchange)");
        if (opt.exchange(opt.exchange_user, chr_wl.data(), (int32_t)chr_wl.size()) != 0)
          throw Error("ERROR: the weighted-length exchange between the ranks failed");
      }
    }
    double WL = 0;
    for (double c : chr_wl) WL += c;
    auto t0 = Clock::now();
    long cur = 0;
    for (size_t i = 0; i < genome.chromosomes.size(); i++) {
      ChromPlan& plan = genome.plans[popu][genome.chromosomes[i]];
      const double cw = chr_wl[i];
      // A chromosome -- or a whole population -- without weighted length (every window holds an N, or lies outside the
      // targets) makes these divisions 0/0 in the reference too (Genome.cpp:803-817).  Its binary casts the NaN (cvttsd2si:
      // LONG_MIN), the sums wrap, the read counts come out negative and sample nothing: no reads from that chromosome,
      // the run goes on.  The same arithmetic here (the oracle restates it alike); counts <= 0 are skipped below.
      const long chr_reads = i + 1 < genome.chromosomes.size() ? to_long_x86(reads * (cw / WL)) : wrap_sub(reads, cur);
      long sum = 0;
      for (size_t j = 0; genome.owns(genome.chromosomes[i]) && j < plan.segs.size(); j++) {
        Segment& g = plan.segs[j];
        if (j + 1 < plan.segs.size()) {
          const double share = seg_weight(plan, g) / cw;
          g.read_count = to_long_x86(share * chr_reads);
          sum = wrap_add(sum, g.read_count);
        } else {
          g.read_count = wrap_sub(chr_reads, sum);
        }
      }
      cur = wrap_add(cur, chr_reads);
    }
    st.t_plan += since(t0);
  }

  // one (population, chromosome): Genome.cpp:870-887
  // The sampling plan of one (population, chromosome): windows of the segments the reference would process
  // (Segment.cpp:675) with their fragRCs (Segment::setReadCount, Segment.cpp:462-476), numbered inside the
  // whole batch.  Any contiguous run of its segments can then be handed to the engine on its own
  // (plan_range): the run of a multi-GPU shard, or a memory-bounded piece of it -- the draws are addressed
  // by the batch-wide numbers, so the pieces' texts concatenate to the text of the whole.
  struct Active { size_t seg; uint32_t w_first; uint64_t slots; };
  struct BatchPlan {
    std::string popu, chr, prefix;
    uint32_t bid = 0;
    bool paired = false;
    std::vector<Active> act;
    uint64_t slots = 0;
    size_t a0 = 0, a1 = 0;  // this process's run of active segments (multi-GPU shard)
    bool dev = false;       // the window table lives on the device (sg_plan_windows)
  } cur;

  // false: nothing to sample in the batch (for this process)
  bool build_batch(const std::string& popu, const std::string& chr) {
    ChromPlan& plan = genome.plans[popu][chr];
    cur = BatchPlan();
    cur.popu = popu; cur.chr = chr;
    cur.bid = batch_id++;  // every rank numbers every batch alike: the id addresses the draws
    if (cur.bid > 0xFFFF) throw Error("ERROR: more than 65535 (population, chromosome) batches");
    if (!genome.owns(chr)) return false;
    st.batches++;
    const bool paired = cur.paired = cfg.paired();
    cur.prefix = "@" + popu + "#" + chr + "#";
    auto t0 = Clock::now();
    std::vector<Active>& act = cur.act;
    wins.clear(); seg_size.clear(); seg_first.clear();
    uint64_t slot = 0;
    if (plan.dev_windows) {
      // per-window read counts, the remainder rule, planned pairs and their prefix sums on the device (sg_plan_windows);
      // the host keeps what it needs to cut the batch into shards and pieces: the planned fragments per segment
      upload(popu, chr);
      std::vector<sg_window_gen> gens;
      std::vector<sg_active_seg> active;
      for (size_t k = 0; k < plan.segs.size(); k++) {
        const Segment& g = plan.segs[k];
        if (!g.has_seq || g.read_count <= 0) continue;   // (negative: what a weightless chromosome's NaN shares turn into)
        uint64_t w = plan.seg_win0[k];
        bool any = false;
        for (size_t h = 0; h < g.hap_len.size(); h++) {
          if (!g.hap_len[h]) continue;
          gens.push_back(sg_window_gen{g.hap_base[h], g.hap_len[h], (uint32_t)h, (uint32_t)active.size(), w});
          w += (g.hap_len[h] + Genome::kFragSize - 1) / Genome::kFragSize;
          any = true;
        }
        if (!any) throw Error("ERROR: sampling window on an absent haplotype (chromosome " + chr + ")");
        active.push_back(sg_active_seg{(int64_t)g.read_count, plan.seg_w[k], g.seq_size() / (unsigned)g.cn, 0});
        seg_size.push_back(g.seq_size() / (unsigned)g.cn);
        act.push_back(Active{k, 0, 0});
      }
      std::vector<uint64_t> slots(active.size(), 0);
      uint64_t nwin = 0;
      if (!active.empty())
        eng.check(sg_plan_windows(eng.ctx, plan.store_id, gens.data(), gens.size(), active.data(), (uint32_t)active.size(), Genome::kFragSize,
                                  cur.bid, paired ? 1 : 0, cur.prefix.c_str(), slots.data(), &nwin),
                  "sg_plan_windows");
      for (size_t i = 0; i < act.size(); i++) { act[i].slots = slots[i]; slot += slots[i]; }
      cur.dev = true;
      cur.slots = slot;
      st.windows += nwin;
      st.segments += act.size();
      st.t_plan += since(t0);
      if (act.empty() || nwin == 0) return false;
      return shard_range(slot);
    }
    for (size_t k = 0; k < plan.segs.size(); k++) {
      const Segment& g = plan.segs[k];
      if (!g.has_seq || g.read_count <= 0) continue;   // (negative: what a weightless chromosome's NaN shares turn into)
      const double total = seg_weight(plan, g) + 2.2204e-16;
      const uint32_t first = (uint32_t)wins.size();
      const uint64_t slot0 = slot;
      long sum = 0;
      for (uint32_t w = g.w0; w < g.w1; w++) {
        const long rc = (long)(plan.w_weight[w] * g.read_count / total);
        sum += rc;
        const uint32_t h = plan.w_hap[w];
        if (g.hap_len[h] == 0) throw Error("ERROR: sampling window on an absent haplotype (chromosome " + chr + ")");
        wins.push_back(sg_window{g.hap_base[h], h, plan.w_spos[w], plan.w_len[w], (int32_t)rc, (uint32_t)act.size(), 0});
      }
      if (sum < g.read_count) wins[first].n_reads += (int32_t)(g.read_count - sum);
      for (uint32_t i = first; i < wins.size(); i++) {
        wins[i].slot_base = (uint32_t)slot;
        const int n = wins[i].n_reads;
        slot += n <= 0 ? 0 : (paired ? ((uint64_t)n + 1) / 2 : (uint64_t)n);
      }
      seg_size.push_back(g.seq_size() / (unsigned)g.cn);
      seg_first.push_back(first);
      act.push_back(Active{k, first, slot - slot0});
    }
    seg_first.push_back((uint32_t)wins.size());
    if (slot > 0xFFFFFFF0ull) throw Error("ERROR: more than 2^32 fragments on chromosome " + chr);
    cur.slots = slot;
    st.windows += wins.size();
    st.segments += act.size();
    st.t_plan += since(t0);
    if (wins.empty()) return false;
    return shard_range(slot);
  }

  // shard by runs of segments (multi-GPU): contiguous, balanced by planned fragments
  bool shard_range(uint64_t slot) {
    const std::vector<Active>& act = cur.act;
    cur.a0 = 0; cur.a1 = act.size();
    if (opt.shard_world > 1 && !opt.shard_contigs) {
      const uint64_t per = (slot + opt.shard_world - 1) / opt.shard_world;
      uint64_t acc = 0;
      cur.a0 = cur.a1 = act.size();
      bool started = false;
      for (size_t i = 0; i < act.size(); i++) {
        const int owner = per ? (int)std::min<uint64_t>(acc / per, (uint64_t)opt.shard_world - 1) : 0;
        if (owner == opt.shard_rank) { if (!started) { cur.a0 = i; started = true; } cur.a1 = i + 1; }
        acc += act[i].slots;
      }
      if (!started) return false;
    }
    return true;
  }

  // chains on the device, sg_plan for the active segments [a0, a1) of the current batch
  void plan_range(size_t a0, size_t a1) {
    if (cur.dev) {
      auto t0d = Clock::now();
      upload(cur.popu, cur.chr);
      auto t_pl = Clock::now();
      eng.check(sg_plan_range(eng.ctx, (uint32_t)a0, (uint32_t)a1), "sg_plan_range");
      st.t_plan_api += since(t_pl);
      st.t_sample += since(t0d);
      if (getenv("SIMU_TRACE_PIECES") != nullptr)
        fprintf(stderr, "[piece] %s %s segments %zu..%zu: upload %.4f s, sg_plan_range %.4f s\n", cur.popu.c_str(), cur.chr.c_str(), a0, a1,
                since(t0d) - since(t_pl), since(t_pl));
      return;
    }
    const std::vector<Active>& act = cur.act;
    const uint32_t w_lo = act[a0].w_first;
    const uint32_t w_hi = a1 < act.size() ? act[a1].w_first : (uint32_t)wins.size();
    const uint32_t slot_lo = wins[w_lo].slot_base;
    std::vector<sg_window> shard(wins.begin() + w_lo, wins.begin() + w_hi);
    std::vector<uint32_t> sh_first, sh_size;
    for (size_t i = a0; i < a1; i++) { sh_first.push_back(act[i].w_first - w_lo); sh_size.push_back(seg_size[i]); }
    sh_first.push_back(w_hi - w_lo);
    for (sg_window& w : shard) { w.seg -= (uint32_t)a0; w.slot_base -= slot_lo; }

    auto t0 = Clock::now();
    upload(cur.popu, cur.chr);
    sg_batch b;
    std::memset(&b, 0, sizeof b);
    b.batch_id = cur.bid;
    b.paired = cur.paired ? 1 : 0;
    b.name_prefix = cur.prefix.c_str();
    b.windows = shard.data();
    b.n_windows = shard.size();
    b.seg_size = sh_size.data();
    b.seg_first_window = sh_first.data();
    b.n_segs = (uint32_t)sh_size.size();
    b.first_window = w_lo;
    b.first_slot = slot_lo;
    auto t_pl = Clock::now();
    eng.check(sg_plan(eng.ctx, &b), "sg_plan");
    st.t_plan_api += since(t_pl);
    st.t_sample += since(t0);
  }

  // the whole shard as one engine batch (step-by-step sessions: bench.py keeps it resident)
  bool prepare_batch(const std::string& popu, const std::string& chr) {
    if (!build_batch(popu, chr)) return false;
    plan_range(cur.a0, cur.a1);
    return true;
  }

  // Pieces of at most kPieceSlots planned fragments (whole segments; a segment is <= 1 Mbp per copy): the
  // work buffers of a pass are ~0.5 KB per fragment and its text ~0.7 KB, so a 250 Mbp chromosome at 30x
  // is a handful of pieces of ~10 GB, and a 1000x run still fits the card.  Each piece drains while the
  // next one is sampled.
  static constexpr uint64_t kPieceSlots = 12u << 20;
  void run_batch(const std::string& popu, const std::string& chr, Sink& sink) {
    if (!build_batch(popu, chr)) return;
    uint64_t piece_slots = kPieceSlots;
    if (const char* e = getenv("SIMU_PIECE_SLOTS")) piece_slots = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    for (size_t c0 = cur.a0; c0 < cur.a1;) {
      size_t c1 = c0;
      uint64_t acc = 0;
      while (c1 < cur.a1 && (c1 == c0 || acc + cur.act[c1].slots <= piece_slots)) acc += cur.act[c1++].slots;
      run_piece(c0, c1, sink);
      c0 = c1;
    }
  }

  void run_piece(size_t c0, size_t c1, Sink& sink) {
    plan_range(c0, c1);
    const bool paired = cur.paired;
    auto t0 = Clock::now();
    uint64_t n1 = 0, n2 = 0, nf = 0;
    const int reps = opt.repeat_sample > 1 ? opt.repeat_sample : 1;
    static const bool trace = getenv("SIMU_TRACE_PIECES") != nullptr;   // where a run's sampling time goes, piece by piece
    for (int r = 0; r < reps; r++) {
      auto ts = Clock::now();
      eng.check(sg_sample(eng.ctx), "sg_sample");
      const double d_sample = since(ts);
      eng.check(sg_result(eng.ctx, &n1, &n2, &nf), "sg_result");
      if (trace) fprintf(stderr, "[piece] %s %s segments %zu..%zu: sg_sample %.4f s, with sg_result %.4f s, %llu fragments\n", cur.popu.c_str(),
                         cur.chr.c_str(), c0, c1, d_sample, since(ts), (unsigned long long)nf);
      float ms[SG_K_COUNT];
      sg_kernel_times(eng.ctx, ms);
      for (int i = 0; i < SG_K_COUNT; i++) st.kernel_ms[i] += ms[i];
      uint64_t queued = 0;
      int requeued = 0;
      sg_emit_info(eng.ctx, &queued, &requeued);
      st.queued_items += queued;
      st.requeued_batches += (uint64_t)requeued;
    }
    st.t_sample += since(t0);
    st.fragments += nf;
    st.reads += paired ? 2 * nf : nf;
    st.fastq_bytes += n1 + n2;
    if (!(opt.write_files || opt.fetch)) return;
    bool compressed = false;
    if (opt.gzip) {
      auto tc = Clock::now();
      uint64_t g1 = 0, g2 = 0;
      eng.check(sg_compress(eng.ctx, &g1, &g2), "sg_compress");
      st.t_compress += since(tc);
      st.gz_bytes += g1 + g2;
      compressed = true;
    }
    // The text leaves the context as a detached output set and is drained by a worker thread (D2H on the
    // set's own stream + the file writers) while this thread plans and samples the next batch.  One
    // drain at a time: the pinned buffers, the files and their order belong to it.
    drain_wait();
    for (int i = 0; i < (paired ? 4 : 2); i++)   // context calls stay on this thread
      if (!pinned[i]) eng.check(sg_host_alloc(eng.ctx, kChunk, &pinned[i]), "sg_host_alloc");
    sg_outputs* h = nullptr;
    eng.check(sg_detach_outputs(eng.ctx, &h), "sg_detach_outputs");
    pending.handle = h;
    pending.active = true;
    pending.err.clear();
    pending.t_fetch = pending.t_write = 0;
    FILE* f1 = opt.write_files ? sink.f1 : nullptr;
    FILE* f2 = opt.write_files ? sink.f2 : nullptr;
    pending.th = std::thread([this, h, f1, f2, compressed]() {
      try { drain(h, f1, f2, compressed); }
      catch (const std::exception& e) { pending.err = e.what(); }
    });
  }

  struct PendingDrain {
    sg_outputs* handle = nullptr;
    std::thread th;
    bool active = false;
    std::string err;
    double t_fetch = 0, t_write = 0;
  } pending;
  // joins the drain in flight (if any), books its times, returns its buffers to the engine
  void drain_wait() {
    if (!pending.active) return;
    pending.th.join();
    pending.active = false;
    st.t_fetch += pending.t_fetch;
    st.t_write += pending.t_write;
    eng.check(sg_release_outputs(eng.ctx, pending.handle), "sg_release_outputs");
    pending.handle = nullptr;
    if (!pending.err.empty()) throw Error(pending.err, -1);
  }

  // FASTQ sink (the reference's SeqWriter::write, lib/seqwriter/SeqWriter.cpp:41-54): D2H in pinned
  // 64 MB chunks, double buffered per mate, while one writer thread per file appends the previous
  // chunk (an fwrite into the page cache runs at ~10 GB/s per thread, a fifth of the D2H rate, so the
  // two files are written concurrently).  Mate 1 and mate 2 text are independent byte streams into
  // their own files, so pair order is kept.
  static constexpr size_t kChunk = 64u << 20;
  void* pinned[4] = {nullptr, nullptr, nullptr, nullptr};
  void drain(sg_outputs* h, FILE* f1, FILE* f2, bool compressed) {
    const int mates = cfg.paired() ? 2 : 1;
    uint64_t tb[2], gb[2];
    sg_outputs_sizes(h, tb, gb);
    const uint64_t n1 = compressed ? gb[0] : tb[0], n2 = compressed ? gb[1] : tb[1];
    struct Writer {
      std::mutex mu;
      std::condition_variable cv;
      FILE* f = nullptr;
      const char* p = nullptr;
      size_t n = 0;
      bool have = false, stop = false, failed = false;
      double t_write = 0;
      std::thread th;
      void run() {
        for (;;) {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&]() { return have || stop; });
          if (!have && stop) return;
          FILE* jf = f; const char* jp = p; const size_t jn = n;
          lk.unlock();
          auto tw = Clock::now();
          if (jf && jn && fwrite(jp, 1, jn, jf) != jn) failed = true;
          t_write += since(tw);
          lk.lock();
          have = false;
          cv.notify_all();
        }
      }
      void wait_idle() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return !have; }); }
      void post(FILE* jf, const char* jp, size_t jn) {
        std::unique_lock<std::mutex> lk(mu);
        f = jf; p = jp; n = jn; have = true;
        cv.notify_all();
      }
      void finish() {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return !have; }); stop = true; cv.notify_all(); }
        th.join();
      }
    };
    Writer w[2];
    for (int m = 0; m < mates; m++) w[m].th = std::thread([&w, m]() { w[m].run(); });
    auto t0 = Clock::now();
    const uint64_t total[2] = {n1, n2};
    int cur[2] = {0, 0}, rc = SG_OK;
    for (uint64_t off = 0; (off < total[0] || (mates == 2 && off < total[1])) && rc == SG_OK; off += kChunk) {
      for (int m = 0; m < mates && rc == SG_OK; m++) {
        if (off >= total[m]) continue;
        const size_t n = (size_t)std::min<uint64_t>(kChunk, total[m] - off);
        void* buf = pinned[2 * m + cur[m]];
        // this buffer's previous chunk was posted two rounds ago; the copy overlaps both writers' fwrite
        rc = sg_outputs_fetch(h, m, compressed ? 1 : 0, off, n, buf);
        w[m].wait_idle();  // the mate's other buffer is free again
        if (rc == SG_OK) w[m].post(m ? f2 : f1, (const char*)buf, n);
        cur[m] ^= 1;
      }
    }
    double t_write = 0;
    bool failed = false;
    for (int m = 0; m < mates; m++) { w[m].finish(); t_write = std::max(t_write, w[m].t_write); failed |= w[m].failed; }
    if (rc != SG_OK) throw Error(std::string("GPU engine error in sg_outputs_fetch: ") + sg_outputs_last_error(h));
    const double wall = since(t0);
    pending.t_write = t_write;
    pending.t_fetch = wall > t_write ? wall - t_write : 0;  // fetch time not hidden behind the writers
    if (failed) throw Error("Error: short write to fastq file", -1);
  }

  Clock::time_point t_all;
  void open(const std::string& config_path) {
    t_all = Clock::now();
    auto t0 = Clock::now();
    cfg.load(config_path);
    seed = opt.has_seed ? opt.seed : (uint64_t)cfg.num["seed"];
    const int device = opt.device >= 0 ? opt.device : (int)cfg.num["device"];
    auto t1 = Clock::now();
    if (sg_create(&eng.ctx, device, seed) != SG_OK) throw Error(std::string("GPU engine error: ") + sg_last_error(nullptr));
    sg_set_profiling(eng.ctx, 1);
    st.t_engine = since(t1);
    genome.device_haps = !opt.host_haplotypes;
    genome.engine = eng.ctx;
    genome.shard_rank = opt.shard_rank;
    genome.fa.crlf_as_lf = opt.crlf_as_lf != 0;
    genome.fa.unique_contigs = opt.unique_contigs != 0;
    if (sg_set_strict_bases(eng.ctx, opt.strict_bases) != SG_OK) throw Error(std::string("GPU engine error: ") + sg_last_error(eng.ctx));
    genome.shard_world = opt.shard_world;
    genome.shard_contigs = opt.shard_contigs != 0 && opt.shard_world > 1;
    // The profile -- parsing its text (Profile::load / normParas / initCDFs), converting every CDF row into the engine's
    // integer tables (sg_profile_prepare: no device involved) -- on a worker thread while this one streams the reference
    // to the device: ~10 ms of work that every rank of a sharded run would otherwise repeat on its critical path.
    sg_profile_tables* tables = nullptr;
    std::string prof_err;
    int prof_exit = 1;
    const std::string prof_path = cfg.str["profile"];   // (the worker must not touch the config maps: operator[] inserts)
    const bool prof_paired = cfg.paired();
    const int prof_isize = (int)cfg.num["insertSize"];
    std::thread prof_thread([&]() {
      try {
        prof.train(prof_path, prof_paired, prof_isize);
        sg_profile_cdf view = prof.view();
        sg_profile_prepare(&view, &tables);
      } catch (const Error& e) { prof_err = e.what(); prof_exit = e.exit_code ? e.exit_code : 1; }
      catch (const std::exception& e) { prof_err = e.what(); }
    });
    struct Joiner { std::thread& t; sg_profile_tables*& T; ~Joiner() { if (t.joinable()) t.join(); if (T) sg_profile_tables_free(T); T = nullptr; } } joiner{prof_thread, tables};
    genome.load_data();
    st.t_reference = genome.t_reference;
    const std::string out_dir = (opt.output_dir && opt.output_dir[0]) ? opt.output_dir : cfg.str["output"];
    if (opt.write_files) {  // `mkdir -p` (src/simuReads.cpp:56-60)
      for (size_t i = 1; i <= out_dir.size(); i++)
        if (i == out_dir.size() || out_dir[i] == '/') mkdir(out_dir.substr(0, i).c_str(), 0755);
    }
    prof_thread.join();
    if (!prof_err.empty()) throw Error(prof_err, prof_exit);
    log("profile was loaded from file " + cfg.str["profile"] + "\n");
    eng.check(sg_load_prepared_profile(eng.ctx, tables), "sg_load_profile");
    genome.generate_segments();
    st.t_load = since(t0);
    st.planned_reads = (uint64_t)(genome.target_length() * cfg.num["coverage"] / prof.read_length);
  }

  void run(const std::string& config_path) {
    open(config_path);
    const std::string out_dir = (opt.output_dir && opt.output_dir[0]) ? opt.output_dir : cfg.str["output"];
    // Genome::yieldReads
    const std::vector<std::string>& popus = cfg.popu_names;
    const long reads = (long)st.planned_reads;
    if (cfg.verbose()) log("\nNumber of reads to sample: " + std::to_string(reads) + "\n");
    std::map<std::string, double> acn;  // Genome::calculateACNs, Genome.cpp:765-781
    for (auto& pp : genome.plans) {
      long sum = 0;
      for (auto& pc : pp.second)
        for (const Segment& g : pc.second.segs) sum += g.seq_size();
      acn[pp.first] = (double)sum / genome.genome_length();
    }
    log("\n*****Generating samples*****\n");
    const bool paired = cfg.paired();
    const std::string suffix = opt.shard_world > 1 ? ".part" + std::to_string(opt.shard_rank) : "";
    sink.eof_block = !opt.no_eof_block;
    if (genome.mix_props.empty()) {
      if (opt.write_files) sink.open(out_dir, popus[0], paired, suffix, opt.gzip != 0);
      set_read_counts(popus[0], reads);
      for (const std::string& chr : genome.chromosomes) run_batch(popus[0], chr, sink);
    } else {
      for (const std::vector<float>& props : genome.mix_props) {
        double w_acn = 0;
        std::string stem;
        char buf[1000];
        for (size_t i = 0; i < popus.size(); i++) {
          w_acn += props[i] * acn[popus[i]];
          snprintf(buf, sizeof buf, i == 0 ? "%s_%.3f" : "+%s_%.3f", popus[i].c_str(), props[i]);
          stem += buf;
        }
        drain_wait();  // the previous mixture's last batch still writes into the files about to be closed
        if (opt.write_files) sink.open(out_dir, stem, paired, suffix, opt.gzip != 0);
        for (size_t i = 0; i < popus.size(); i++) {
          const long popu_reads = (long)(reads * props[i] * acn[popus[i]] / w_acn);  // long*float is a float product (Genome.cpp:935)
          set_read_counts(popus[i], popu_reads);
          for (const std::string& chr : genome.chromosomes) run_batch(popus[i], chr, sink);
        }
      }
    }
    drain_wait();
    sink.close();
    log("\nReads generation done!\n");
    st.t_total = since(t_all);
  }
};

}  // namespace
}  // namespace simu

extern "C" void simu_default_options(simu_options* o) {
  std::memset(o, 0, sizeof *o);
  o->device = -1;
  o->write_files = 1;
  o->shard_world = 1;
}

extern "C" void simu_assign_contigs(const uint64_t* lengths, int32_t n, int32_t world, int32_t* owner_out) {
  const std::vector<int> o = simu::Genome::assign_contigs(std::vector<uint64_t>(lengths, lengths + n), world);
  for (int32_t i = 0; i < n; i++) owner_out[i] = o[(size_t)i];
}

extern "C" int simu_run(const char* config_path, const simu_options* opt, simu_stats* stats, char* err, size_t err_len) {
  simu_options o;
  if (opt) o = *opt; else simu_default_options(&o);
  if (o.shard_world < 1) o.shard_world = 1;
  auto set_err = [&](const std::string& m) {
    if (err && err_len) { strncpy(err, m.c_str(), err_len - 1); err[err_len - 1] = '\0'; }
  };
  try {
    std::unique_ptr<simu::Driver> d(new simu::Driver());
    d->opt = o;
    d->run(config_path ? config_path : "");
    if (stats) *stats = d->st;
    return 0;
  } catch (const simu::Error& e) {
    set_err(e.what());
    return e.exit_code == 0 ? 1 : e.exit_code;
  } catch (const std::exception& e) {
    set_err(e.what());
    return 1;
  }
}

extern "C" int simu_selftest_haplotypes(const char* config_path, uint64_t seed, char* err, size_t err_len) {
  auto set_err = [&](const std::string& m) {
    if (err && err_len) { strncpy(err, m.c_str(), err_len - 1); err[err_len - 1] = '\0'; }
  };
  try {
    simu::Config cfg;
    cfg.load(config_path ? config_path : "");
    // two genomes over the same inputs: strings and copy lists (the host FASTA parser serves both)
    simu::Genome gs(cfg), gp(cfg);
    for (simu::Genome* g : {&gs, &gp}) { g->load_data(); g->generate_segments(); }
    gp.device_haps = true;  // after load_data: the reference stays on the host, only build_chains changes
    for (const std::string& popu : cfg.popu_names)
      for (const std::string& chr : gs.chromosomes) {
        gs.build_chains(popu, chr, seed);
        gp.build_chains(popu, chr, seed);
        const simu::ChromPlan& a = gs.plans[popu][chr];
        const simu::ChromPlan& b = gp.plans[popu][chr];
        const std::string& contig = gp.fa.seqs.at(chr);
        for (size_t h = 0; h < a.chains.size(); h++) {
          if (a.chains[h].size() != b.chain_len[h])
            throw simu::Error("haplotype " + std::to_string(h) + " of " + popu + "/" + chr + ": lengths " +
                              std::to_string(a.chains[h].size()) + " vs " + std::to_string(b.chain_len[h]));
          std::string m(b.chain_len[h], '?');
          uint64_t covered = 0;
          for (const sg_hap_piece& p : b.pieces) {
            if (p.chain != h) continue;
            const char* src = p.kind ? b.literals.data() + p.src : contig.data() + p.src;
            for (uint32_t i = 0; i < p.len; i++) {
              char c = src[i];
              if (c >= 'a' && c <= 'z') c -= 32;
              m[p.dst + i] = c;
            }
            covered += p.len;
          }
          if (covered != b.chain_len[h]) throw simu::Error("copy list of " + popu + "/" + chr + " does not tile haplotype " + std::to_string(h));
          for (const sg_hap_patch& q : b.patches)
            if (q.chain == h) { char c = (char)q.base; if (c >= 'a' && c <= 'z') c -= 32; m[q.dst] = c; }
          if (m != a.chains[h]) {
            size_t i = 0;
            while (i < m.size() && m[i] == a.chains[h][i]) i++;
            throw simu::Error("haplotype " + std::to_string(h) + " of " + popu + "/" + chr + " differs at " + std::to_string(i));
          }
        }
        // same segment bookkeeping on both routes
        for (size_t k = 0; k < a.segs.size(); k++)
          if (a.segs[k].hap_base != b.segs[k].hap_base || a.segs[k].hap_len != b.segs[k].hap_len)
            throw simu::Error("segment " + std::to_string(k) + " of " + popu + "/" + chr + ": haplotype offsets differ");
      }
    return 0;
  } catch (const std::exception& e) {
    set_err(e.what());
    return 1;
  }
}

// ------------------------------------------------------------------------------------------------
// Session API: the same driver, opened step by step so that a caller (bench.py, a multi-GPU
// launcher) can keep the inputs resident in HBM and drive sg_sample on its own stream.
// ------------------------------------------------------------------------------------------------
struct simu_session { simu::Driver d; };

static int session_guard(char* err, size_t err_len, const std::function<void()>& fn) {
  try { fn(); return 0; }
  catch (const simu::Error& e) { if (err && err_len) { strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; } return e.exit_code ? e.exit_code : 1; }
  catch (const std::exception& e) { if (err && err_len) { strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; } return 1; }
}

extern "C" int simu_open(const char* config_path, const simu_options* opt, simu_session** out, char* err, size_t err_len) {
  if (!out) return 1;
  *out = nullptr;
  std::unique_ptr<simu_session> s(new simu_session());
  if (opt) s->d.opt = *opt; else simu_default_options(&s->d.opt);
  if (s->d.opt.shard_world < 1) s->d.opt.shard_world = 1;
  int rc = session_guard(err, err_len, [&]() { s->d.open(config_path ? config_path : ""); });
  if (rc == 0) *out = s.release();
  return rc;
}
extern "C" void simu_close(simu_session* s) { delete s; }
extern "C" void* simu_engine(simu_session* s) { return s ? (void*)s->d.eng.ctx : nullptr; }
extern "C" uint64_t simu_planned_reads(simu_session* s) { return s->d.st.planned_reads; }
extern "C" int simu_chromosome_count(simu_session* s) { return (int)s->d.genome.chromosomes.size(); }
// Sum over this session's chromosomes of the GC-weighted length of population `popu`
// (Genome::setReadCounts' WL, Genome.cpp:787-797).  Runs pass 1 (haplotypes + GPU GC scan) on first use.
extern "C" int simu_weighted_length(simu_session* s, int popu, double* wl, char* err, size_t err_len) {
  return session_guard(err, err_len, [&]() {
    simu::Driver& d = s->d;
    const std::string& p = d.cfg.popu_names.at(popu);
    double WL = 0;
    for (const std::string& chr : d.genome.chromosomes) {
      d.weigh(p, chr);
      simu::ChromPlan& plan = d.genome.plans[p][chr];
      double c = 0;
      for (const simu::Segment& g : plan.segs) c += simu::Driver::seg_weight(plan, g);
      WL += c;
    }
    *wl = WL;
  });
}
extern "C" int simu_set_reads(simu_session* s, int popu, int64_t reads, char* err, size_t err_len) {
  return session_guard(err, err_len, [&]() { s->d.set_read_counts(s->d.cfg.popu_names.at(popu), (long)reads); });
}
// Upload the chromosome's haplotype chains and hand its sampling plan to the engine.  `has_work`
// is 0 when this shard has nothing to sample there.
extern "C" int simu_prepare_batch(simu_session* s, int popu, int chr, int* has_work, char* err, size_t err_len) {
  return session_guard(err, err_len, [&]() {
    bool w = s->d.prepare_batch(s->d.cfg.popu_names.at(popu), s->d.genome.chromosomes.at(chr));
    if (has_work) *has_work = w ? 1 : 0;
  });
}
extern "C" void simu_get_stats(simu_session* s, simu_stats* st) { if (s && st) *st = s->d.st; }
