// sg_api.cpp -- C ABI (include/simuscop_amd.h) on top of the gfx950 kernels.
//
// Everything here is host plumbing: table conversion, grow-only device buffers (sized for a
// 288 GB HBM3E part: a whole chromosome's haplotypes, plan and FASTQ text stay resident), kernel
// sequencing on one HIP stream.  There is no CPU fallback: without a HIP device sg_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "sg_deflate.h"
#include "sg_train.h"
#include "sg_device.h"
#include "sg_tables.h"

namespace sg {
void launch_plan(const DevProfile& P, const DevBatch& B, hipStream_t s);
void launch_namebase(const DevBatch& B, hipStream_t s);
void launch_indel(const DevProfile& P, const DevBatch& B, hipStream_t s);
uint32_t scan_blocks(uint32_t n);
void launch_scan(const DevBatch& B, hipStream_t s);
void launch_mail(const uint64_t* totals, uint64_t* mail, hipStream_t s);
uint32_t record_seg_shift(uint32_t n_slots);
void launch_header(const DevProfile& P, const DevBatch& B, hipStream_t s);
void launch_emit(const DevProfile& P, const DevBatch& B, hipStream_t s, bool force_generic, hipEvent_t after_main);
bool emit_uses_fast_kernel(const DevProfile& P, const DevBatch& B);
int emit_variant(const DevProfile& P);
void launch_encode(uint8_t* buf, size_t bytes, hipStream_t s);
void launch_gc(const uint8_t* chains, const uint64_t* chain_off, const sg_gc_window* wins, uint64_t n, int32_t* out, hipStream_t s);
void launch_tile(const sg_window_gen* gens, const uint64_t* prefix, uint32_t n_gens, uint64_t n, uint32_t frag, const uint64_t* seg_first,
                 sg_gc_window* out, uint32_t* seg_ord, uint32_t* win_ord, hipStream_t s);
void launch_seg_sum(const double* wt, const uint64_t* seg_first, uint32_t n_segs, double* out, hipStream_t s);
void launch_window_reads(const sg_window_gen* gens, const uint64_t* prefix, uint32_t n_gens, uint64_t n, uint32_t frag, const double* wt,
                         const sg_active_seg* act, const uint32_t* seg_first, uint32_t n_act, sg_window* rows, unsigned long long* seg_sum,
                         int32_t paired, uint32_t* planned, hipStream_t s);
void launch_slot_base(sg_window* rows, uint64_t n, const uint64_t* off, const uint32_t* seg_first, uint32_t n_act, const uint64_t* total,
                      uint64_t* seg_slots, hipStream_t s);
void launch_slice(const sg_window* all, uint64_t w_lo, uint64_t n, uint32_t a0, uint32_t slot_lo, sg_window* out, hipStream_t s);
void launch_gc_weight(const int32_t* gc, const sg_gc_window* wins, const uint32_t* seg_ord, const uint32_t* win_ord, uint64_t n,
                      const double* means, double std, const double* Q, uint32_t lg_cells, uint32_t frag, int32_t full_tile_form,
                      uint32_t ctx24, uint64_t seed, double* out, hipStream_t s);
// sg_haplotypes.hip
struct DevContig { uint64_t raw_off, code_off, length; uint32_t line_bases, line_width; uint64_t first_block; };
struct DevPiece { uint64_t dst, src; uint32_t len, pad; };
struct DevPatch { uint64_t dst; uint32_t base, pad; };
void launch_ref_scan(const uint8_t* raw, uint64_t n, uint64_t* list, uint32_t cap, uint32_t* count, uint32_t* flags, hipStream_t s);
void launch_ref_ingest(const uint8_t* raw, uint8_t* codes, const void* contigs, uint32_t n_contigs, uint64_t n_blocks,
                       uint32_t* flags, hipStream_t s);
void launch_hap_copy(uint8_t* chains, const uint8_t* ref_codes, const uint8_t* literals, const void* pieces, uint64_t n, hipStream_t s);
void launch_hap_patch(uint8_t* chains, const void* patches, uint64_t n, hipStream_t s);
void launch_encode_bytes(uint8_t* buf, uint64_t n, hipStream_t s);
void launch_pack2(const uint8_t* chains, uint64_t bytes, uint32_t* fwd2, uint32_t* rc2, uint16_t* bad, hipStream_t s);
// sg_deflate.hip
struct DevDeflate {
  const uint8_t* text; uint64_t bytes; uint32_t n_chunks;
  const uint32_t* code; const uint32_t* len_tok; const uint32_t* dist_code;
  const uint32_t* prefix; uint32_t prefix_words, prefix_bits;
  const uint32_t* crc_tab; const uint32_t* crc_shift; uint32_t crc_init_full, crc_init_last;
  uint32_t* next; uint32_t* msize; const uint64_t* moff;
  uint4* rec; uint32_t* lbits; unsigned long long* hist; uint8_t* out;
  uint32_t min_run, min_copy;
};
void launch_gz_hist(const void* d, uint32_t n_chunks, hipStream_t s);
void launch_gz_match(const void* d, uint32_t n_chunks, hipStream_t s);
void launch_gz_encode(const void* d, uint32_t n_chunks, uint32_t prefix_bits, hipStream_t s);
void launch_scan_u32(const uint32_t* in, uint32_t n, uint64_t* bsum, uint64_t* out, uint64_t* total, hipStream_t s);
}  // namespace sg

namespace {

// Device blocks released by a context stay with the process and serve the next request of their size on the same
// device.  A whole-genome run allocates ~60 GB in a few dozen blocks and returns them at its end; the next run of the
// process asked the runtime for the same blocks again, and every dozen runs or so ONE such hipMalloc took 2-5 s
// (`SG_TRACE_ALLOC=1 python tools/c3_steps.py`: "hipMalloc 3455.3 MB: 4925.76 ms" in the thirteenth run, 20-40 ms in the
// others) -- seventeen times the run itself.  Best fit within 1.5x; the cache holds at most SG_BLOCK_CACHE_GB (default: a
// third of the device's memory) and gives everything back through sg_release_cached_memory().  SG_BLOCK_CACHE_GB=0 turns it off.
struct BlockCache {
  struct Block { void* p; size_t cap; int dev; };
  std::mutex mu;
  std::vector<Block> blocks;   // oldest first
  size_t held = 0;
  // Default: a third of the device's memory (MI355X: 96 GB -- a whole-genome run's ~60 GB of blocks fit; what other
  // allocators of the process and other processes on the device may need stays free), SG_BLOCK_CACHE_GB overrides.
  static size_t limit() {
    static const size_t v = [] {
      const char* e = getenv("SG_BLOCK_CACHE_GB");
      if (e) return (size_t)(atof(e) * 1073741824.0);
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || !total_b) return (size_t)(64.0 * 1073741824.0);
      return total_b / 3;
    }();
    return v;
  }
  void* take(size_t want, int dev, size_t* cap) {
    std::lock_guard<std::mutex> lk(mu);
    size_t best = blocks.size();
    for (size_t i = 0; i < blocks.size(); i++)
      if (blocks[i].dev == dev && blocks[i].cap >= want && blocks[i].cap <= want + want / 2 + (64u << 20) &&
          (best == blocks.size() || blocks[i].cap < blocks[best].cap))
        best = i;
    if (best == blocks.size()) return nullptr;
    void* p = blocks[best].p;
    *cap = blocks[best].cap;
    held -= blocks[best].cap;
    blocks.erase(blocks.begin() + (long)best);
    return p;
  }
  void give(void* p, size_t cap, int dev) {
    if (cap > limit()) { (void)hipFree(p); return; }
    // whoever gets the block next may write it at once: nothing of this process may still be using it (hipFree waits
    // likewise)
    int cur = dev;
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    (void)hipDeviceSynchronize();
    if (cur != dev) (void)hipSetDevice(cur);
    std::vector<Block> drop;
    {
      std::lock_guard<std::mutex> lk(mu);
      blocks.push_back({p, cap, dev});
      held += cap;
      while (held > limit() && !blocks.empty()) {
        drop.push_back(blocks.front());
        held -= blocks.front().cap;
        blocks.erase(blocks.begin());
      }
    }
    for (const Block& b : drop) (void)hipFree(b.p);
  }
  bool any() {
    std::lock_guard<std::mutex> lk(mu);
    return !blocks.empty();
  }
  void trim() {
    std::vector<Block> drop;
    {
      std::lock_guard<std::mutex> lk(mu);
      drop.swap(blocks);
      held = 0;
    }
    for (const Block& b : drop) (void)hipFree(b.p);
  }
};
BlockCache& block_cache() {
  static BlockCache* c = new BlockCache;  // (never destroyed: contexts may outlive static destructors)
  return *c;
}

// Pinned staging buffers are kept for the process like the device blocks (BlockCache above): pinning 64 MB takes ~5 ms, a
// run's two reference-staging buffers 10 ms of a 0.27 s whole-genome run.  Exact sizes only (the callers use a few fixed
// ones), at most 1 GB kept.
struct HostCache {
  struct Block { void* p; uint64_t bytes; };
  std::mutex mu;
  std::vector<Block> blocks;
  std::map<void*, uint64_t> live;   // what sg_host_alloc handed out: sizes for the way back
  uint64_t held = 0;
  void* take(uint64_t bytes) {
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < blocks.size(); i++)
      if (blocks[i].bytes == bytes) {
        void* p = blocks[i].p;
        held -= bytes;
        blocks.erase(blocks.begin() + (long)i);
        live[p] = bytes;
        return p;
      }
    return nullptr;
  }
  void note(void* p, uint64_t bytes) {
    std::lock_guard<std::mutex> lk(mu);
    live[p] = bytes;
  }
  bool give(void* p) {   // false: not one of ours, or no room
    std::lock_guard<std::mutex> lk(mu);
    auto it = live.find(p);
    if (it == live.end()) return false;
    const uint64_t bytes = it->second;
    live.erase(it);
    if (BlockCache::limit() == 0 || held + bytes > (1ull << 30)) return false;
    blocks.push_back({p, bytes});
    held += bytes;
    return true;
  }
  void trim() {
    std::vector<Block> drop;
    {
      std::lock_guard<std::mutex> lk(mu);
      drop.swap(blocks);
      held = 0;
    }
    for (const Block& b : drop) (void)hipHostFree(b.p);
  }
};
HostCache& host_cache() {
  static HostCache* c = new HostCache;
  return *c;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int dev = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    release();
    size_t want = bytes + bytes / 8 + 256;
    static const bool trace = getenv("SG_TRACE_ALLOC") != nullptr;
    (void)hipGetDevice(&dev);
    if ((p = block_cache().take(want, dev, &cap)) != nullptr) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess && block_cache().any()) {  // the cache may be what is in the way
      block_cache().trim();
      e = hipMalloc(&p, want);
    }
    if (trace)
      fprintf(stderr, "[sg] hipMalloc %.1f MB: %.2f ms\n", want / 1048576.0,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (e != hipSuccess) { p = nullptr; return (int)e; }
    cap = want;
    return 0;
  }
  void release() {
    if (p) block_cache().give(p, cap, dev);
    p = nullptr; cap = 0;
  }
  template <class T> T* as() const { return (T*)p; }
};

thread_local std::string g_create_error;

}  // namespace

struct sg_train_session;

struct sg_outputs {
  int device = 0;
  hipStream_t stream = nullptr;
  DevBuf text[2], gz[2];
  uint64_t text_bytes[2] = {0, 0}, gz_bytes[2] = {0, 0};
  std::string err;
};

struct sg_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  uint64_t seed = 0;
  std::string err;

  bool have_profile = false, have_haps = false, have_plan = false, sampled = false;
  sg::DevProfile P{};
  sg::DevBatch B{};
  DevBuf tab, chains, chains2, chain_meta, windows, segmeta, prefix, pairs, win_actual, win_namebase, rlen, events, reclen,
      recoff, meta, totals, bsum, out1, out2, gcw, gco, gcm, slowq, ref_raw, ref_codes, ref_meta, hap_work, gz1, gz2, gz_work;
  uint64_t gz_bytes[2] = {0, 0};
  bool gz_valid = false;
  std::vector<sg_outputs*> spare;  // released output sets, reused by the next pass
  // device-made sampling plan: window weights per store id (sg_windows_build), the batch table of sg_plan_windows
  std::map<uint32_t, DevBuf> wstore;
  std::map<uint32_t, uint64_t> wstore_n;
  DevBuf wplan, wwork;
  struct PlanInfo {
    bool valid = false;
    uint64_t n_windows = 0;
    uint32_t n_active = 0, batch_id = 0;
    int32_t paired = 0;
    std::string prefix;
    std::vector<uint32_t> seg_first, seg_size;  // per active segment (seg_first has n_active + 1 entries)
    std::vector<uint64_t> slot_first;           // planned fragments before each active segment; [n_active] = total
  } winfo;
  uint64_t ref_raw_bytes = 0;
  struct sg_train_session* train = nullptr;   // profile training in progress (sg_train_begin .. sg_train_finish)
  std::vector<sg::DevContig> ref_contigs;  // host copy of the committed contig table
  uint64_t host_totals[4] = {0, 0, 0, 0};
  uint64_t host_flags[2] = {0, 0};  // totals[3..4] after the emit kernels: flags, slow-queue counts
  uint64_t* mail = nullptr;         // pinned: where a pass's totals[0..4] land (copied to the two arrays above by finish_pass)
  bool pass_pending = false;        // a pass is queued whose totals have not been looked at yet
  bool speculative = false;         // ... and its emit kernels were launched before the text size was known (see run_pass)
  uint64_t slow_items = 0;
  bool slow_overflow = false;
  bool results_valid = false;

  bool profiling = false;
  hipEvent_t evs[8] = {};  // 0-3 starts of plan..scan, 4 end of scan, 5 start of emit, 6 end of emit, 7 between the two emit kernels
  bool evs_created = false;
  float last_ms[SG_K_COUNT] = {0, 0, 0, 0, 0, 0};

  int fail(int code, const std::string& m) { err = m; return code; }
  int hipfail(hipError_t e, const char* what) {
    err = std::string(what) + ": " + hipGetErrorString(e);
    return SG_ERR_HIP;
  }
};

#define SG_HIP(call)                                        \
  do {                                                      \
    hipError_t _e = (call);                                 \
    if (_e != hipSuccess) return ctx->hipfail(_e, #call);   \
  } while (0)
#define SG_ENSURE(buf, bytes)                                                                  \
  do {                                                                                         \
    int _e = (buf).ensure(bytes);                                                              \
    if (_e) return ctx->hipfail((hipError_t)_e, "hipMalloc(" #buf ")");                        \
  } while (0)

// Work buffers and DevBatch fields of a planned batch whose windows / segment arrays are already in ctx->windows /
// ctx->segmeta (put there by sg_plan from host arrays, or by sg_plan_range from the device-made table).
static int finish_plan(sg_ctx* ctx, uint64_t nw, uint32_t n_segs, uint32_t n_slots, uint32_t batch_id, uint32_t first_window,
                       uint32_t first_slot, int32_t paired, const char* name_prefix) {
  const size_t plen = name_prefix ? strlen(name_prefix) : 0;
  const uint32_t nm = paired ? 2 : 1;
  SG_ENSURE(ctx->prefix, plen + 16);
  SG_ENSURE(ctx->pairs, ((size_t)n_slots + 1) * sizeof(sg::PairRec));
  SG_ENSURE(ctx->win_actual, (nw + 1) * 4);
  SG_ENSURE(ctx->win_namebase, (nw + 1) * 4);
  SG_ENSURE(ctx->events, ((size_t)nm * n_slots + 1) * 4 * SG_MAX_EVENTS);
  SG_ENSURE(ctx->recoff, ((size_t)nm * n_slots + 1) * 4);
  SG_ENSURE(ctx->meta, ((size_t)nm * n_slots + 1) * 48);
  SG_ENSURE(ctx->totals, sg::kTotalsBytes);
  SG_ENSURE(ctx->bsum, ((size_t)nm * ((n_slots + 255) / 256) + 1) * 8);
  SG_HIP(hipMemcpyAsync(ctx->prefix.p, name_prefix, plen, hipMemcpyHostToDevice, ctx->stream));
  SG_HIP(hipStreamSynchronize(ctx->stream));

  sg::DevBatch& B = ctx->B;
  B.windows = ctx->windows.as<sg_window>();
  B.n_windows = nw;
  B.seg_size = ctx->segmeta.as<uint32_t>();
  B.seg_first_window = ctx->segmeta.as<uint32_t>() + n_segs;
  B.n_segs = n_segs;
  B.n_slots = n_slots;
  B.batch_id = batch_id;
  B.win_offset = first_window;
  B.slot_offset = first_slot;
  B.paired = paired ? 1 : 0;
  B.prefix = ctx->prefix.as<uint8_t>();
  B.prefix_len = (uint32_t)plen;
  for (int i = 0; i < 4; i++) B.prefix_w[i] = 0;
  for (size_t i = 0; i < plen && i < 16; i++) B.prefix_w[i / 4] |= (uint32_t)(uint8_t)name_prefix[i] << (8 * (i % 4));
  B.pairs = ctx->pairs.as<sg::PairRec>();
  B.win_actual = ctx->win_actual.as<uint32_t>();
  B.win_namebase = ctx->win_namebase.as<uint32_t>();
  B.events = ctx->events.as<uint32_t>();
  B.recloc = ctx->recoff.as<uint32_t>();
  B.blkbase = ctx->bsum.as<uint64_t>();
  B.seg_shift = sg::record_seg_shift((uint32_t)n_slots);
  B.meta = ctx->meta.as<uint4>();
  B.totals = ctx->totals.as<uint64_t>();
  B.slowq_count = (uint32_t*)(B.totals + 4);
  ctx->have_plan = true;
  ctx->sampled = false;
  ctx->results_valid = false;
  return SG_OK;
}

// 2-bit copies of the chains buffer (`total` bytes, a multiple of 1024) for the straight-line emit kernel
static int pack_chains(sg_ctx* ctx, size_t total) {
  const size_t q = total / 4, badb = total / 64 / 8;
  SG_ENSURE(ctx->chains2, 2 * q + badb + 64);
  uint8_t* p = ctx->chains2.as<uint8_t>();
  sg::launch_pack2(ctx->chains.as<uint8_t>(), total, (uint32_t*)p, (uint32_t*)(p + q), (uint16_t*)(p + 2 * q), ctx->stream);
  ctx->B.chains2_fwd = p;
  ctx->B.chains2_rc = p + q;
  ctx->B.chains_bad = (const uint16_t*)(p + 2 * q);
  ctx->B.chains_total = total;
  return SG_OK;
}

extern "C" {

uint64_t sg_cdf_count_le(double c) { return sg::count_le(c); }

int sg_sub_row_identity_first(const double cdf4[4], int cd, uint64_t cum[3], uint8_t order[4]) {
  if (!cdf4 || !cum || !order || cd < 0 || cd > 3) return SG_ERR_INVALID;
  const sg::SubRow r = sg::encode_sub_row_identity_first(cdf4, cd);
  for (int i = 0; i < 3; i++) cum[i] = r.c[i];
  for (int i = 0; i < 4; i++) order[i] = r.order[i];
  return SG_OK;
}

uint32_t sg_row_symbols(const double* cdf, int n) { return (cdf && n > 0) ? sg::symbols_with_mass(sg::row_masses(cdf, n)) : 0u; }

int sg_alias_row(const double* cdf, int n, uint32_t lgW, uint32_t* thr, uint8_t* lo, uint8_t* hi) {
  if (!cdf || n < 1 || n > 256 || lgW < 2 || lgW > 8 || !thr || !lo || !hi) return SG_ERR_INVALID;
  try {
    const sg::AliasRow r = sg::build_alias_row(sg::row_masses(cdf, n), lgW);
    for (uint32_t c = 0; c < (1u << lgW); c++) { thr[c] = r.thr[c]; lo[c] = r.lo[c]; hi[c] = r.hi[c]; }
  } catch (const std::exception&) {
    return SG_ERR_INVALID;
  }
  return SG_OK;
}

const char* sg_last_error(const sg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int sg_create(sg_ctx** out, int device, uint64_t seed) {
  if (!out) { g_create_error = "sg_create: null output pointer"; return SG_ERR_INVALID; }
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = std::string("sg_create: no HIP device available (") + hipGetErrorString(e) +
                     "); this engine has no CPU path";
    return SG_ERR_HIP;
  }
  if (device < 0 || device >= ndev) { g_create_error = "sg_create: device index out of range"; return SG_ERR_INVALID; }
  e = hipSetDevice(device);
  if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return SG_ERR_HIP; }
  sg_ctx* ctx = new sg_ctx();
  ctx->device = device;
  ctx->seed = seed;
  e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete ctx; return SG_ERR_HIP; }
  ctx->stream = ctx->own_stream;
  e = hipHostMalloc((void**)&ctx->mail, 64, hipHostMallocDefault);
  if (e != hipSuccess) { g_create_error = std::string("hipHostMalloc: ") + hipGetErrorString(e); (void)hipStreamDestroy(ctx->own_stream); delete ctx; return SG_ERR_HIP; }
  *out = ctx;
  return SG_OK;
}

void sg_destroy(sg_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  sg_train_end(ctx);
  for (DevBuf* b : {&ctx->tab, &ctx->chains, &ctx->chains2, &ctx->chain_meta, &ctx->windows, &ctx->segmeta, &ctx->prefix, &ctx->pairs,
                    &ctx->win_actual, &ctx->win_namebase, &ctx->rlen, &ctx->events, &ctx->reclen, &ctx->recoff,
                    &ctx->meta, &ctx->totals, &ctx->bsum, &ctx->out1, &ctx->out2, &ctx->gcw, &ctx->gco, &ctx->gcm, &ctx->slowq,
                    &ctx->ref_raw, &ctx->ref_codes, &ctx->ref_meta, &ctx->hap_work, &ctx->gz1, &ctx->gz2, &ctx->gz_work})
    b->release();
  for (auto& kv : ctx->wstore) kv.second.release();
  ctx->wplan.release();
  ctx->wwork.release();
  for (sg_outputs* o : ctx->spare) {
    for (int m = 0; m < 2; m++) { o->text[m].release(); o->gz[m].release(); }
    if (o->stream) (void)hipStreamDestroy(o->stream);
    delete o;
  }
  if (ctx->evs_created)
    for (auto& ev : ctx->evs) (void)hipEventDestroy(ev);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  if (ctx->mail) (void)hipHostFree(ctx->mail);
  delete ctx;
}

int sg_set_stream(sg_ctx* ctx, void* hip_stream) {
  if (!ctx) return SG_ERR_INVALID;
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return SG_OK;
}

int sg_set_seed(sg_ctx* ctx, uint64_t seed) {
  if (!ctx) return SG_ERR_INVALID;
  ctx->seed = seed;
  return SG_OK;
}

int sg_set_strict_bases(sg_ctx* ctx, int on) {
  if (!ctx) return SG_ERR_INVALID;
  ctx->B.strict_bases = on ? 1u : 0u;   // read by the generic item code of every later pass (template_code, sg_kernels.hip)
  return SG_OK;
}

int sg_set_profiling(sg_ctx* ctx, int enable) {
  if (!ctx) return SG_ERR_INVALID;
  SG_HIP(hipSetDevice(ctx->device));
  if (enable && !ctx->evs_created) {
    for (auto& ev : ctx->evs) SG_HIP(hipEventCreate(&ev));
    ctx->evs_created = true;
  }
  ctx->profiling = enable != 0;
  return SG_OK;
}

int sg_kernel_times(sg_ctx* ctx, float ms[SG_K_COUNT]) {
  if (!ctx || !ms) return SG_ERR_INVALID;
  for (int i = 0; i < SG_K_COUNT; i++) ms[i] = ctx->last_ms[i];
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------
// The tables of a profile as the engine wants them, made on the host (sg_profile_prepare: no device, any thread) and
// uploaded by sg_load_prepared_profile.  Offsets are in 32-bit words into `tab`.
struct sg_profile_tables {
  std::vector<uint32_t> tab;
  size_t sub_off = 0, sub_rows = 0, alias_off = 0, fast_lds_off = 0, fast_sub_off = 0, fast_alias_off = 0, ins_off = 0, del_off = 0, isz_off = 0,
         gap_off = 0;
  uint32_t lgW = 0, fast_stride = 0, ins_lg = 0, del_lg = 0, isz_lg = 0, inv_remap = 0, remap = 0, packed = 0;
  bool has2 = false, has_isz = false;
  uint64_t evA = 0, evB = 0;
  int kmer = 0, bins = 0, read_length = 0, min_qual = 0, isize_min = 0, insert_size = 0, n_isize = 0;
  int rc = SG_OK;
  std::string err;
  int fail(int code, const std::string& msg) { rc = code; err = msg; return code; }
};

static int build_tables(const sg_profile_cdf* pr, sg_profile_tables& T) {
  if (!pr) return T.fail(SG_ERR_INVALID, "sg_profile_prepare: null profile");
  if (pr->n_bases != 4) return T.fail(SG_ERR_UNSUPPORTED, "sg_load_profile: only 4-letter base alphabets are supported");
  if (pr->kmer < 1 || pr->kmer > 6) return T.fail(SG_ERR_UNSUPPORTED, "sg_load_profile: kmer must be in 1..6");
  if (pr->bins < 1 || pr->read_length < 1 || pr->read_length > 30000) return T.fail(SG_ERR_INVALID, "sg_load_profile: bad bins/read_length");
  {  // the kernels compute bin = i*bins/n' with a 32-bit reciprocal: exact while i*bins*n' < 2^32
    const uint64_t npmax = (uint64_t)pr->read_length + (uint64_t)SG_MAX_EVENTS * (uint64_t)(pr->n_ins > 0 ? pr->n_ins : 1);
    if (npmax > 0xFFFF || npmax * (uint64_t)pr->bins * npmax >= (1ull << 32))
      return T.fail(SG_ERR_UNSUPPORTED, "sg_load_profile: read_length * bins too large for the 32-bit bin arithmetic");
  }
  if (pr->n_qual < 1 || pr->n_qual > 128) return T.fail(SG_ERR_INVALID, "sg_load_profile: n_qual must be in 1..128 (quality symbols are 7-bit fields)");
  if (!pr->subs_cdf1 || !pr->qual_cdf || !pr->ins_cdf || !pr->del_cdf || pr->n_ins < 1 || pr->n_del < 1)
    return T.fail(SG_ERR_INVALID, "sg_load_profile: missing table");
  // base alphabet must be a permutation of ACGT (the kernels classify haplotype bytes by value)
  uint32_t remap = 0, packed = 0;
  {
    const char nat[4] = {'A', 'C', 'T', 'G'};  // natural index = (byte >> 1) & 3
    for (int n = 0; n < 4; n++) {
      int code = -1;
      for (int k = 0; k < 4; k++) if (pr->bases[k] == nat[n]) code = k;
      if (code < 0) return T.fail(SG_ERR_UNSUPPORTED, "sg_load_profile: bases must be a permutation of ACGT");
      remap |= (uint32_t)code << (2 * n);
    }
    for (int k = 0; k < 4; k++) packed |= (uint32_t)(uint8_t)pr->bases[k] << (8 * k);
  }
  int kmer_count = 0;
  {
    int p = 1;
    for (int m = 1; m <= pr->kmer; m++) { p *= 4; kmer_count += p; }
  }
  const int bins = pr->bins;
  const bool has2 = pr->subs_cdf2 != nullptr;

  // context -> index of its last base (the reference base a substitution row is about): contexts with m real bases are
  // numbered in base-4 counting order (Profile::initKmers, Profile.cpp:70-124), so the last base is the lowest digit
  std::vector<uint8_t> ctx_cd((size_t)kmer_count);
  {
    int off = 0, p4 = 1;
    for (int m = 1; m <= pr->kmer; m++) {
      p4 *= 4;
      for (int v = 0; v < p4; v++) ctx_cd[off + v] = (uint8_t)(v & 3);
      off += p4;
    }
  }
  std::vector<uint32_t>& tab = T.tab;
  // substitution rows, identity first (sg_tables.h): {D0, D1, D2, j0 | o0<<2 | o1<<4 | o2<<6 | o3<<8}, o in profile codes
  const size_t sub_rows = (size_t)kmer_count * bins;
  const size_t sub_off = 0;
  const int n_mates = has2 ? 2 : 1;
  std::vector<sg::SubRow> srows((size_t)n_mates * sub_rows);
  tab.resize((size_t)n_mates * sub_rows * 4);
  for (int t = 0; t < n_mates; t++) {
    const double* src = t == 0 ? pr->subs_cdf1 : pr->subs_cdf2;
    for (size_t r = 0; r < sub_rows; r++) {
      const sg::SubRow sr = sg::encode_sub_row_identity_first(src + r * 4, ctx_cd[r / bins]);
      srows[t * sub_rows + r] = sr;
      uint32_t* o = &tab[(t * sub_rows + r) * 4];
      o[0] = sr.D[0]; o[1] = sr.D[1]; o[2] = sr.D[2];
      o[3] = sr.j0 | (uint32_t)sr.order[0] << 2 | (uint32_t)sr.order[1] << 4 | (uint32_t)sr.order[2] << 6 | (uint32_t)sr.order[3] << 8;
    }
  }
  // quality rows as alias columns (sg_tables.h): [16][bins][W] x {thr, lo | hi << 8}
  const size_t qrows = (size_t)16 * bins;
  std::vector<std::vector<uint64_t>> qmass(qrows);
  uint32_t most = 1;
  for (size_t r = 0; r < qrows; r++) {
    qmass[r] = sg::row_masses(pr->qual_cdf + r * pr->n_qual, pr->n_qual);
    most = std::max(most, sg::symbols_with_mass(qmass[r]));
  }
  uint32_t lgW = 2;
  while ((1u << lgW) < most) lgW++;
  const uint32_t W = 1u << lgW;
  std::vector<sg::AliasRow> arows(qrows);
  try {
    for (size_t r = 0; r < qrows; r++) arows[r] = sg::build_alias_row(qmass[r], lgW);
  } catch (const std::exception& e) {
    return T.fail(SG_ERR_INVALID, std::string("sg_load_profile: ") + e.what());
  }
  while (tab.size() % 4) tab.push_back(0);
  const size_t alias_off = tab.size();
  tab.resize(alias_off + qrows * W * 2);
  for (size_t r = 0; r < qrows; r++)
    for (uint32_t c = 0; c < W; c++) {
      tab[alias_off + (r * W + c) * 2] = arows[r].thr[c];
      tab[alias_off + (r * W + c) * 2 + 1] = (uint32_t)arows[r].lo[c] | (uint32_t)arows[r].hi[c] << 8;
    }
  auto add_row = [&](const double* cdf, int n, size_t& off, uint32_t& lg) {
    sg::Row r = sg::encode_row(cdf, n);
    uint32_t W = sg::pow2_at_least((uint32_t)r.T.size());
    lg = sg::log2u(W);
    off = tab.size();
    tab.resize(off + 1 + W, 0xFFFFFFFFu);
    tab[off] = r.k0;
    for (size_t i = 0; i < r.T.size(); i++) tab[off + 1 + i] = r.T[i];
    while (tab.size() % 4) tab.push_back(0xFFFFFFFFu);
  };
  size_t ins_off, del_off, isz_off = 0;
  uint32_t ins_lg, del_lg, isz_lg = 0;
  while (tab.size() % 4) tab.push_back(0xFFFFFFFFu);
  add_row(pr->ins_cdf, pr->n_ins, ins_off, ins_lg);
  add_row(pr->del_cdf, pr->n_del, del_off, del_lg);
  const bool has_isz = pr->isize_cdf != nullptr && pr->n_isize > 0;
  if (has_isz) add_row(pr->isize_cdf, pr->n_isize, isz_off, isz_lg);
  // Sequencing indels by skipping ahead: getIndelSeq tests every template position with `p <= insertRate`, then
  // `p < delRate/(1-insertRate)`, p = x/2^32 (Profile.cpp:1560-1570).  cI, cD = the numbers of 32-bit draws that pass;
  // a position is a candidate with probability evB / 2^64, evB = cI 2^32 + (2^32 - cI) cD, an insertion with evA / evB,
  // evA = cI 2^32.  gap[k] = floor(gap[k-1] * gap[1] / 2^64), gap[1] = 2^64 - evB: P(no candidate in k positions).
  if (pr->insert_rate < 0) return T.fail(SG_ERR_INVALID, "sg_load_profile: negative insert rate");
  const uint64_t ci = sg::count_unit_le(pr->insert_rate);
  const uint64_t cd = sg::count_unit_lt(pr->del_rate / (1 - pr->insert_rate));
  if (ci >= (1ull << 31) || cd >= (1ull << 31)) return T.fail(SG_ERR_UNSUPPORTED, "sg_load_profile: sequencing indel rate >= 0.5");
  const uint64_t evA = ci << 32, evB = evA + ((1ull << 32) - ci) * cd;
  while (tab.size() % 4) tab.push_back(0);
  const size_t gap_off = tab.size();
  {
    const int L = std::max(pr->read_length, 1);
    std::vector<uint64_t> gap((size_t)L + 1, 0xFFFFFFFFFFFFFFFFull);
    if (evB) {
      gap[1] = 0ull - evB;
      for (int k = 1; k < L; k++) gap[(size_t)k + 1] = (uint64_t)(((unsigned __int128)gap[(size_t)k] * gap[1]) >> 64);
    }
    tab.resize(gap_off + 2 * gap.size());
    memcpy(&tab[gap_off], gap.data(), gap.size() * 8);
    while (tab.size() % 4) tab.push_back(0);
  }
  // Straight-line kernel tables (kmer 3).  The kernel packs base codes 2 bits each in NATURAL order (A0 C1 T2 G3) with
  // the OLDEST base of a context in the lowest digit; the reference numbers a context with its oldest base in the
  // highest digit, in `bases` order (Profile::initKmers).  Context ids of the kernel: [0,4) one base, [4,20) two, [20,84)
  // three.  Everything below is indexed by the kernel's ids and natural base codes.
  uint32_t inv_remap = 0;
  for (int n = 0; n < 4; n++) inv_remap |= (uint32_t)n << (2 * ((remap >> (2 * n)) & 3u));
  size_t fast_lds_off = 0, fast_sub_off = 0, fast_alias_off = 0;
  uint32_t fast_stride = 0;
  if (pr->kmer == 3) {
    std::vector<uint32_t> perm;  // kernel context id -> reference context id
    uint32_t off = 0, p4 = 1;
    for (int m = 1; m <= 3; m++) {
      p4 *= 4;
      for (uint32_t v = 0; v < p4; v++) {
        uint32_t src = 0;
        for (int tt = 0; tt < m; tt++) {
          const uint32_t nat = (v >> (2 * tt)) & 3u;  // base at age position tt (0 = oldest)
          src |= ((remap >> (2 * nat)) & 3u) << (2 * (m - 1 - tt));
        }
        perm.push_back(off + src);
      }
      off += p4;
    }
    auto nat_of = [&](uint32_t prof) { return (inv_remap >> (2u * prof)) & 3u; };
    auto prof_of = [&](uint32_t nat) { return (remap >> (2u * nat)) & 3u; };
    // kernel context ids: [0,4) one base, [4,20) two, [20,84) three, the oldest base in the lowest digit.  Per bin the
    // tables hold 192 context rows addressed by the 6-bit field kv = b[i-2] | b[i-1] << 2 | b[i] << 4 of the packed
    // codes: [0,64) the three-base contexts, [64,128) the two-base contexts (kv >> 2; a read's second base) and
    // [128,192) the one-base contexts (kv >> 4; a read's first base), so that every position uses the same extract.
    auto ctx_of = [&](uint32_t slot) -> uint32_t {
      const uint32_t kv = slot & 63u;
      return slot < 64u ? perm[20u + kv] : slot < 128u ? perm[4u + (kv >> 2)] : perm[kv >> 4];
    };
    fast_stride = 192u + 4u * W;
    while (tab.size() % 4) tab.push_back(0);
    // (1) the LDS image, per mate and bin: [0,192) keep_h - 1 of the context's row, keep_h = c0 >> 16 (a 16-bit head
    //     below keep_h is certainly "no substitution"); [192, 192 + 4W) the diagonal alias columns (reference base ==
    //     called base) in the order [col][natural base] as  col << (16 - lgW) | thr >> 16  in the low half (the draw's
    //     low half minus it is u_head - thr_head: the column bits cancel),  (lo ^ hi) << 16 | hi << 24  above, lo and hi as
    //     characters (symbol + the profile's lowest quality character)
    fast_lds_off = tab.size();
    tab.resize(fast_lds_off + (size_t)n_mates * bins * fast_stride);
    const uint32_t mq = (uint32_t)pr->min_qual;
    for (int t = 0; t < n_mates; t++)
      for (int b = 0; b < bins; b++) {
        uint32_t* blk = &tab[fast_lds_off + ((size_t)t * bins + b) * fast_stride];
        for (uint32_t sl = 0; sl < 192; sl++) blk[sl] = (uint32_t)(srows[t * sub_rows + (size_t)ctx_of(sl) * bins + b].c[0] >> 16) - 1u;
        for (uint32_t cdn = 0; cdn < 4; cdn++) {
          const uint32_t pc = prof_of(cdn);
          const sg::AliasRow& ar = arows[((size_t)pc * 4 + pc) * bins + b];
          for (uint32_t c = 0; c < W; c++)
            blk[192 + c * 4 + cdn] = (c << (16 - lgW)) | (ar.thr[c] >> 16) | (uint32_t)((ar.lo[c] + mq) ^ (ar.hi[c] + mq)) << 16 | (uint32_t)(ar.hi[c] + mq) << 24;
        }
      }
    // (2) full substitution rows for the kernel's fix-up path: [mate][bin][192] x {D0, D1, D2, j0 | n0<<2 | .. | n3<<8}
    while (tab.size() % 4) tab.push_back(0);
    fast_sub_off = tab.size();
    tab.resize(fast_sub_off + (size_t)n_mates * bins * 192 * 4);
    for (int t = 0; t < n_mates; t++)
      for (int b = 0; b < bins; b++)
        for (uint32_t sl = 0; sl < 192; sl++) {
          const sg::SubRow& sr = srows[t * sub_rows + (size_t)ctx_of(sl) * bins + b];
          uint32_t* o = &tab[fast_sub_off + (((size_t)t * bins + b) * 192 + sl) * 4];
          o[0] = sr.D[0]; o[1] = sr.D[1]; o[2] = sr.D[2];
          o[3] = sr.j0 | nat_of(sr.order[0]) << 2 | nat_of(sr.order[1]) << 4 | nat_of(sr.order[2]) << 6 | nat_of(sr.order[3]) << 8;
        }
    // (3) all alias columns in natural order: [cdn][kn][bins][W] x {thr, lo | hi << 8}
    fast_alias_off = tab.size();
    tab.resize(fast_alias_off + qrows * W * 2);
    for (uint32_t cdn = 0; cdn < 4; cdn++)
      for (uint32_t kn = 0; kn < 4; kn++)
        for (int b = 0; b < bins; b++) {
          const sg::AliasRow& ar = arows[((size_t)prof_of(cdn) * 4 + prof_of(kn)) * bins + b];
          uint32_t* o = &tab[fast_alias_off + ((((size_t)cdn * 4 + kn) * bins + b) * W) * 2];
          for (uint32_t c = 0; c < W; c++) { o[2 * c] = ar.thr[c]; o[2 * c + 1] = (uint32_t)ar.lo[c] | (uint32_t)ar.hi[c] << 8; }
        }
    while (tab.size() % 4) tab.push_back(0);
  }

  T.sub_off = sub_off; T.sub_rows = sub_rows; T.has2 = has2; T.alias_off = alias_off; T.lgW = lgW;
  T.fast_lds_off = fast_lds_off; T.fast_sub_off = fast_sub_off; T.fast_alias_off = fast_alias_off; T.fast_stride = fast_stride;
  T.ins_off = ins_off; T.ins_lg = ins_lg; T.del_off = del_off; T.del_lg = del_lg; T.isz_off = isz_off; T.isz_lg = isz_lg; T.has_isz = has_isz;
  T.inv_remap = inv_remap; T.remap = remap; T.packed = packed; T.evA = evA; T.evB = evB; T.gap_off = gap_off;
  T.kmer = pr->kmer; T.bins = bins; T.read_length = pr->read_length; T.min_qual = pr->min_qual;
  T.isize_min = pr->isize_min; T.insert_size = pr->insert_size; T.n_isize = pr->n_isize;
  return SG_OK;
}

int sg_profile_prepare(const sg_profile_cdf* pr, sg_profile_tables** out) {
  if (!out) return SG_ERR_INVALID;
  sg_profile_tables* T = new sg_profile_tables();
  *out = T;
  T->rc = build_tables(pr, *T);
  return T->rc;
}
const char* sg_profile_tables_error(const sg_profile_tables* T) { return T ? T->err.c_str() : "null tables"; }
void sg_profile_tables_free(sg_profile_tables* T) { delete T; }

int sg_load_prepared_profile(sg_ctx* ctx, const sg_profile_tables* Tp) {
  if (!ctx || !Tp) return SG_ERR_INVALID;
  const sg_profile_tables& T = *Tp;
  if (T.rc != SG_OK) return ctx->fail(T.rc, T.err);
  SG_HIP(hipSetDevice(ctx->device));
  const std::vector<uint32_t>& tab = T.tab;
  SG_ENSURE(ctx->tab, tab.size() * 4);
  SG_HIP(hipMemcpyAsync(ctx->tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  SG_HIP(hipStreamSynchronize(ctx->stream));

  sg::DevProfile& P = ctx->P;
  const uint32_t* base = ctx->tab.as<uint32_t>();
  P.sub = (const uint4*)(base + T.sub_off);
  P.sub_mate_rows = T.has2 ? (uint32_t)T.sub_rows : 0u;
  P.alias = (const uint2*)(base + T.alias_off);
  P.lgW = T.lgW;
  P.fast_lds = T.kmer == 3 ? base + T.fast_lds_off : nullptr;
  P.fast_mate_words = T.has2 ? (uint32_t)T.bins * T.fast_stride : 0u;
  P.fast_stride = T.fast_stride;
  P.fast_sub = T.kmer == 3 ? (const uint4*)(base + T.fast_sub_off) : nullptr;
  P.fast_alias = T.kmer == 3 ? (const uint2*)(base + T.fast_alias_off) : nullptr;
  P.ins_row = base + T.ins_off; P.ins_lg = T.ins_lg;
  P.del_row = base + T.del_off; P.del_lg = T.del_lg;
  P.isz_row = T.has_isz ? base + T.isz_off : nullptr; P.isz_lg = T.isz_lg;
  P.inv_remap_packed = T.inv_remap;
  P.isz_min = T.isize_min;
  P.fixed_isz = T.insert_size;
  P.isz_lo = T.has_isz ? T.isize_min : T.insert_size;
  P.isz_hi = T.has_isz ? T.isize_min + T.n_isize - 1 : T.insert_size;
  P.evA = T.evA; P.evB = T.evB;
  P.gap_row = (const uint64_t*)(base + T.gap_off);
  P.L = T.read_length; P.bins = T.bins; P.kmer = T.kmer; P.min_qual = T.min_qual;
  P.remap_packed = T.remap; P.bases_packed = T.packed;
  {
    uint32_t off = 0, p = 1;
    for (int m = 0; m < 8; m++) P.kmer_off[m] = 0;
    for (int m = 1; m <= T.kmer; m++) { P.kmer_off[m] = off; p *= 4; off += p; }
  }
  ctx->have_profile = true;
  ctx->have_plan = false;
  return SG_OK;
}

int sg_load_profile(sg_ctx* ctx, const sg_profile_cdf* pr) {
  if (!ctx || !pr) return SG_ERR_INVALID;
  sg_profile_tables* T = nullptr;
  sg_profile_prepare(pr, &T);
  const int rc = sg_load_prepared_profile(ctx, T);
  sg_profile_tables_free(T);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// profile training (sg_train.hip)
// ------------------------------------------------------------------------------------------------
struct sg_train_session {
  char bases[4] = {0, 0, 0, 0};
  uint32_t kmer = 0, bins = 0, n_isize = 0, n_indel_len = 0, count_gc = 0, window = 1000, wes = 0, remap = 0;
  uint64_t max_reads = 300000000;    // Profile.cpp:236
  bool capped = false;               // the cap was reached: the reference has stopped reading
  uint32_t kc = 0, koff[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  size_t subs_n = 0, kmers_n = 0, qual_n = 0, counters = 0;
  uint32_t n_contigs = 0;
  DevBuf keys, contigs, counts, flags, text[2], line_end, reads, gate, steps, windows, window_rc, carry, scan_work, tgt, known, t_ref,
      t_alt, patch, gc_out;
  // Two text buffers and a copy stream: chunk k travels to text[k & 1] while the kernels of chunk k - 1 read the other one.
  hipStream_t copy_stream = nullptr;
  uint64_t fed = 0;                  // chunks handed to the kernels so far
  bool pending = false;              // a chunk's kernels are queued whose carry / flags have not been looked at yet
  uint64_t pending_lines = 0;
  bool own_codes = false;            // t_ref / t_alt are copies with the SNVs of the VCF in them
  uint64_t code_bytes = 0;
  uint64_t n_tgt = 0, n_ins = 0, n_del = 0;
  int cur = 0;                       // carry[cur] is read by the next chunk, carry[cur ^ 1] written
  uint64_t lines = 0, n_windows = 0; // lines fed / windows opened so far (host copies)
  uint64_t windows_cap = 0;          // rows the window arrays hold
  sg::TrainCarry* mail = nullptr;    // pinned: the carry a chunk left, the flag word behind it
  uint32_t* mail_flags() { return (uint32_t*)(mail + 1); }
  void release() {
    if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); copy_stream = nullptr; }
    for (DevBuf* b : {&keys, &contigs, &counts, &flags, &text[0], &text[1], &line_end, &reads, &gate, &steps, &windows, &window_rc, &carry, &scan_work,
                      &tgt, &known, &t_ref, &t_alt, &patch, &gc_out})
      b->release();
    if (mail) (void)hipHostFree(mail);
    mail = nullptr;
  }
};

void sg_train_end(sg_ctx* ctx) {
  if (!ctx || !ctx->train) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->train->release();
  delete ctx->train;
  ctx->train = nullptr;
}

namespace {
// a device buffer grown with what it holds kept (the window arrays live across chunks)
int grow_keep(sg_ctx* ctx, DevBuf& buf, size_t keep_bytes, size_t want_bytes) {
  if (want_bytes <= buf.cap) return SG_OK;
  DevBuf bigger;
  SG_ENSURE(bigger, want_bytes + want_bytes / 2);
  if (keep_bytes) SG_HIP(hipMemcpyAsync(bigger.p, buf.p, keep_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  SG_HIP(hipStreamSynchronize(ctx->stream));
  buf.release();
  buf = bigger;
  return SG_OK;
}
}  // namespace

int sg_train_begin(sg_ctx* ctx, const sg_train_setup* st) {
  if (!ctx || !st || !st->bases || (st->n_contigs && !st->contig_keys)) return SG_ERR_INVALID;
  if (ctx->ref_contigs.empty() || st->n_contigs != ctx->ref_contigs.size())
    return ctx->fail(SG_ERR_INVALID, "sg_train_begin: name the contigs of sg_reference_commit, in its order");
  if (strlen(st->bases) != 4 || st->kmer < 1 || st->kmer > 6 || st->bins < 1 || st->n_isize < 1 || st->n_indel_len < 1 || st->window < 1)
    return ctx->fail(SG_ERR_UNSUPPORTED, "sg_train_begin: bases must hold four letters, kmer 1..6, bins >= 1");
  if ((st->n_snv && !(st->snv_contig && st->snv_pos && st->snv_alt && st->snv_homo)) || (st->n_ins && !(st->ins_contig && st->ins_pos && st->ins_len)) ||
      (st->n_del && !(st->del_contig && st->del_pos && st->del_len)) || (st->target_first && !(st->target_spos && st->target_epos)))
    return SG_ERR_INVALID;
  uint32_t remap = 0;
  {
    const char nat[4] = {'A', 'C', 'T', 'G'};
    for (int n = 0; n < 4; n++) {
      int code = -1;
      for (int k = 0; k < 4; k++) if (st->bases[k] == nat[n]) code = k;
      if (code < 0) return ctx->fail(SG_ERR_UNSUPPORTED, "sg_train_begin: bases must be a permutation of ACGT");
      remap |= (uint32_t)code << (2 * n);
    }
  }
  SG_HIP(hipSetDevice(ctx->device));
  sg_train_end(ctx);
  sg_train_session* T = new sg_train_session();
  ctx->train = T;
  auto fail = [&](int rc) { sg_train_end(ctx); return rc; };
  memcpy(T->bases, st->bases, 4);
  T->kmer = (uint32_t)st->kmer; T->bins = (uint32_t)st->bins; T->n_isize = st->n_isize; T->n_indel_len = st->n_indel_len;
  T->count_gc = st->count_gc ? 1u : 0u; T->window = st->window; T->remap = remap; T->n_contigs = st->n_contigs;
  if (st->max_reads) T->max_reads = st->max_reads;
  {
    uint32_t p4 = 1;
    for (int m = 1; m <= st->kmer; m++) { T->koff[m] = T->kc; p4 *= 4; T->kc += p4; }
  }
  T->subs_n = (size_t)T->kc * T->bins * 4; T->kmers_n = (size_t)T->bins * T->kc; T->qual_n = (size_t)16 * T->bins * 94;
  T->counters = 2 * T->subs_n + T->kmers_n + T->qual_n + T->n_isize + 2 * (size_t)T->n_indel_len + sg::kTrainScalars;
  const uint32_t nc = st->n_contigs;
  hipStream_t s = ctx->stream;
  // ---- contigs, their targets (countGC's windows of an exome: [spos, epos - 1], Profile.cpp:590-593) ----
  std::vector<char> keys((size_t)nc * sg::kTrainKeyBytes, 0);
  std::vector<sg::TrainContig> tc(nc);
  uint64_t code_bytes = 0;
  for (uint32_t c = 0; c < nc; c++) {
    if (!st->contig_keys[c] || strlen(st->contig_keys[c]) >= sg::kTrainKeyBytes) return fail(ctx->fail(SG_ERR_UNSUPPORTED, "sg_train_begin: contig name too long"));
    strcpy(&keys[(size_t)c * sg::kTrainKeyBytes], st->contig_keys[c]);
    memset(&tc[c], 0, sizeof tc[c]);
    tc[c].code_off = ctx->ref_contigs[c].code_off;
    tc[c].length = ctx->ref_contigs[c].length;
    const std::string k = st->contig_keys[c];
    tc[c].xym = (k == "X" || k == "Y" || k == "M") ? 1u : 0u;
    code_bytes = std::max<uint64_t>(code_bytes, tc[c].code_off + ((tc[c].length + 15) / 16) * 16 + 64);
  }
  T->code_bytes = code_bytes;
  std::vector<int64_t> tgt;   // left[n], right[n], pmax[n]
  if (st->target_first && st->target_first[nc] > 0) {
    const uint64_t n = st->target_first[nc];
    T->n_tgt = n; T->wes = 1;
    tgt.resize(3 * n);
    for (uint32_t c = 0; c < nc; c++) {
      const uint64_t a = st->target_first[c], b = st->target_first[c + 1];
      if (b < a || b > n || b - a > 0xFFFFFFFFull) return fail(ctx->fail(SG_ERR_INVALID, "sg_train_begin: target_first must ascend"));
      tc[c].tgt_first = a; tc[c].tgt_n = (uint32_t)(b - a);
      int64_t pm = INT64_MIN;
      for (uint64_t t = a; t < b; t++) {
        // (loadTargets keeps 1 <= spos and epos <= the contig's length, divideTargets spos <= epos: Genome.cpp:270-279, 690-733)
        if (st->target_spos[t] < 0 || st->target_epos[t] < st->target_spos[t] || (uint64_t)st->target_epos[t] > tc[c].length)
          return fail(ctx->fail(SG_ERR_INVALID, "sg_train_begin: a target leaves its contig"));
        tgt[t] = st->target_spos[t];
        tgt[n + t] = st->target_epos[t] - 1;
        pm = std::max(pm, tgt[n + t]);
        tgt[2 * n + t] = pm;
      }
    }
  }
  // ---- known insertions / deletions per contig: file order (running maximum of the positions: where the reference's loop
  // stops, Profile.cpp:314-316) and (position, length) order (is the event there at all, and how early) ----
  std::vector<int64_t> kn64;   // per kind: pmax[n], pos[n]
  std::vector<int32_t> kn32;   // per kind: len[n], first[n] (as int32)
  auto stage_known = [&](uint64_t n, const uint32_t* contig, const int64_t* pos, const int32_t* len, bool ins) -> bool {
    std::vector<std::vector<uint64_t>> rows(nc);
    for (uint64_t i = 0; i < n; i++) {
      if (contig[i] >= nc) return false;
      rows[contig[i]].push_back(i);
    }
    const size_t b64 = kn64.size(), b32 = kn32.size();
    kn64.resize(b64 + 2 * n);
    kn32.resize(b32 + 2 * n);
    uint64_t at = 0;
    for (uint32_t c = 0; c < nc; c++) {
      const std::vector<uint64_t>& r = rows[c];
      if (r.size() > 0x7FFFFFFFull) return false;
      (ins ? tc[c].ins_first : tc[c].del_first) = at;
      (ins ? tc[c].ins_n : tc[c].del_n) = (uint32_t)r.size();
      int64_t pm = INT64_MIN;
      std::vector<uint32_t> order(r.size());
      for (size_t j = 0; j < r.size(); j++) {
        pm = std::max(pm, pos[r[j]]);
        kn64[b64 + at + j] = pm;
        order[j] = (uint32_t)j;
      }
      std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        if (pos[r[x]] != pos[r[y]]) return pos[r[x]] < pos[r[y]];
        return len[r[x]] < len[r[y]];
      });
      for (size_t j = 0; j < r.size(); j++) {
        kn64[b64 + n + at + j] = pos[r[order[j]]];
        kn32[b32 + at + j] = len[r[order[j]]];
        // the least file-order index of this (position, length): stable_sort keeps file order inside equal keys
        const bool same = j > 0 && pos[r[order[j]]] == pos[r[order[j - 1]]] && len[r[order[j]]] == len[r[order[j - 1]]];
        kn32[b32 + n + at + j] = same ? kn32[b32 + n + at + j - 1] : (int32_t)order[j];
      }
      at += r.size();
    }
    return true;
  };
  T->n_ins = st->n_ins; T->n_del = st->n_del;
  if (!stage_known(st->n_ins, st->ins_contig, st->ins_pos, st->ins_len, true) || !stage_known(st->n_del, st->del_contig, st->del_pos, st->del_len, false))
    return fail(ctx->fail(SG_ERR_INVALID, "sg_train_begin: a known insertion / deletion names no contig"));
  // ---- SNVs: altSequence takes every one, refSequence the homozygous ones, later rows over earlier (Genome.cpp:469-475) ----
  std::map<uint64_t, char> alt_patch, ref_patch;
  for (uint64_t i = 0; i < st->n_snv; i++) {
    if (st->snv_contig[i] >= nc) return fail(ctx->fail(SG_ERR_INVALID, "sg_train_begin: a known SNV names no contig"));
    const sg::TrainContig& C = tc[st->snv_contig[i]];
    // (a position outside the contig writes outside the reference's string: skipped)
    if (st->snv_pos[i] < 1 || (uint64_t)st->snv_pos[i] > C.length) continue;
    const uint64_t off = C.code_off + (uint64_t)(st->snv_pos[i] - 1);
    alt_patch[off] = st->snv_alt[i];
    if (st->snv_homo[i]) ref_patch[off] = st->snv_alt[i];
  }
  // ---- device side ----
  auto up = [&](DevBuf& buf, const void* src, size_t bytes) -> int {
    SG_ENSURE(buf, bytes + 64);
    if (bytes) SG_HIP(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, s));
    return SG_OK;
  };
  int rc;
  if ((rc = up(T->keys, keys.data(), keys.size())) != SG_OK) return fail(rc);
  if ((rc = up(T->contigs, tc.data(), tc.size() * sizeof(sg::TrainContig))) != SG_OK) return fail(rc);
  if ((rc = up(T->tgt, tgt.data(), tgt.size() * 8)) != SG_OK) return fail(rc);
  {
    std::vector<uint8_t> blob(kn64.size() * 8 + kn32.size() * 4);
    if (!kn64.empty()) memcpy(blob.data(), kn64.data(), kn64.size() * 8);
    if (!kn32.empty()) memcpy(blob.data() + kn64.size() * 8, kn32.data(), kn32.size() * 4);
    if ((rc = up(T->known, blob.data(), blob.size())) != SG_OK) return fail(rc);
    SG_HIP(hipStreamSynchronize(s));
  }
  if (!alt_patch.empty()) {
    T->own_codes = true;
    SG_ENSURE(T->t_ref, code_bytes);
    SG_ENSURE(T->t_alt, code_bytes);
    SG_HIP(hipMemcpyAsync(T->t_ref.p, ctx->ref_codes.p, code_bytes, hipMemcpyDeviceToDevice, s));
    SG_HIP(hipMemcpyAsync(T->t_alt.p, ctx->ref_codes.p, code_bytes, hipMemcpyDeviceToDevice, s));
    for (int which = 0; which < 2; which++) {
      const std::map<uint64_t, char>& m = which ? ref_patch : alt_patch;
      if (m.empty()) continue;
      std::vector<uint64_t> off; std::vector<uint8_t> ch;
      off.reserve(m.size()); ch.reserve(m.size());
      for (const auto& kv : m) { off.push_back(kv.first); ch.push_back((uint8_t)kv.second); }
      SG_ENSURE(T->patch, off.size() * 9 + 64);
      uint8_t* d_ch = T->patch.as<uint8_t>() + off.size() * 8;
      SG_HIP(hipMemcpyAsync(T->patch.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, s));
      SG_HIP(hipMemcpyAsync(d_ch, ch.data(), ch.size(), hipMemcpyHostToDevice, s));
      sg::launch_train_patch((which ? T->t_ref : T->t_alt).as<uint8_t>(), T->patch.as<uint64_t>(), d_ch, off.size(), s);
      SG_HIP(hipGetLastError());
      SG_HIP(hipStreamSynchronize(s));
    }
  }
  SG_ENSURE(T->counts, T->counters * 8);
  SG_ENSURE(T->flags, 64);
  SG_ENSURE(T->carry, 2 * sizeof(sg::TrainCarry) + 64);
  SG_HIP(hipMemsetAsync(T->counts.p, 0, T->counters * 8, s));
  SG_HIP(hipMemsetAsync(T->flags.p, 0, 64, s));
  SG_HIP(hipMemsetAsync(T->carry.p, 0, 2 * sizeof(sg::TrainCarry), s));
  SG_HIP(hipHostMalloc((void**)&T->mail, sizeof(sg::TrainCarry) + 64, hipHostMallocDefault));
  SG_HIP(hipStreamCreateWithFlags(&T->copy_stream, hipStreamNonBlocking));
  SG_HIP(hipStreamSynchronize(s));
  return SG_OK;
}

namespace {
void train_job(sg_ctx* ctx, sg_train_session* T, sg::TrainJob& J) {
  memset(&J, 0, sizeof J);
  J.keys = T->keys.as<char>();
  J.contigs = T->contigs.as<sg::TrainContig>();
  J.n_contigs = T->n_contigs;
  J.ref_codes = T->own_codes ? T->t_ref.as<uint8_t>() : ctx->ref_codes.as<uint8_t>();
  J.alt_codes = T->own_codes ? T->t_alt.as<uint8_t>() : ctx->ref_codes.as<uint8_t>();
  memcpy(J.bases, T->bases, 4);
  J.remap = T->remap;
  J.kmer = T->kmer; J.bins = T->bins; J.kmer_count = T->kc; J.n_isize = T->n_isize; J.n_indel_len = T->n_indel_len;
  for (int m = 0; m < 8; m++) J.kmer_off[m] = T->koff[m];
  J.count_gc = T->count_gc; J.wes = T->wes; J.window = T->window; J.max_reads = T->max_reads;
  J.tgt_left = T->tgt.as<int64_t>(); J.tgt_right = J.tgt_left + T->n_tgt; J.tgt_pmax = J.tgt_left + 2 * T->n_tgt;
  const int64_t* k64 = T->known.as<int64_t>();
  const int32_t* k32 = (const int32_t*)(T->known.as<uint8_t>() + (2 * T->n_ins + 2 * T->n_del) * 8);
  J.known_ins = sg::TrainKnown{k64, k64 + T->n_ins, k32, (const uint32_t*)(k32 + T->n_ins)};
  J.known_del = sg::TrainKnown{k64 + 2 * T->n_ins, k64 + 2 * T->n_ins + T->n_del, k32 + 2 * T->n_ins, (const uint32_t*)(k32 + 2 * T->n_ins + T->n_del)};
  unsigned long long* c0 = T->counts.as<unsigned long long>();
  J.subs1 = c0; J.subs2 = c0 + T->subs_n; J.kmers = c0 + 2 * T->subs_n; J.quality = J.kmers + T->kmers_n; J.isize = J.quality + T->qual_n;
  J.ins_len = J.isize + T->n_isize; J.del_len = J.ins_len + T->n_indel_len; J.scalars = J.del_len + T->n_indel_len;
  J.flags = T->flags.as<uint32_t>();
  J.carry_in = T->carry.as<sg::TrainCarry>() + T->cur;
  J.carry_out = T->carry.as<sg::TrainCarry>() + (T->cur ^ 1);
}
}  // namespace

namespace {
// What the chunk whose kernels are queued left behind: the malformed-line flag, the cap, the windows opened so far.
int train_settle(sg_ctx* ctx, sg_train_session* T) {
  if (!T->pending) return SG_OK;
  SG_HIP(hipStreamSynchronize(ctx->stream));
  T->pending = false;
  if (*T->mail_flags() & 1u) return ctx->fail(SG_ERR_INVALID, "Error: malformed read , there should be 11 mandatory fields");   // Profile.cpp:246-251
  if (T->mail->cut_line != ~0ull) { T->capped = true; T->lines += T->mail->cut_line + 1; }
  else T->lines += T->pending_lines;
  if (T->count_gc) T->n_windows = T->mail->n_windows;
  return SG_OK;
}
}  // namespace

// One chunk of lines. The copy to the device runs on its own stream into the text buffer the previous chunk is not using, so it
// overlaps that chunk's kernels; the call returns with its own kernels queued (sg_train_capped / the malformed-line error of
// chunk k are known when chunk k + 1 is fed or sg_train_finish runs).
int sg_train_feed(sg_ctx* ctx, const char* sam_text, uint64_t sam_bytes) {
  if (!ctx || (sam_bytes && !sam_text)) return SG_ERR_INVALID;
  sg_train_session* T = ctx->train;
  if (!T) return ctx->fail(SG_ERR_INVALID, "sg_train_feed: call sg_train_begin first");
  if (!sam_bytes || T->capped) return SG_OK;   // (behind the cap: Profile::train has left its loop, Profile.cpp:1461-1464)
  SG_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const bool open_end = sam_text[sam_bytes - 1] != '\n';   // a last line without a line break gets one in the device copy
  const uint64_t bytes = sam_bytes + (open_end ? 1 : 0);
  DevBuf& text = T->text[T->fed & 1];
  SG_ENSURE(text, bytes + 64);
  SG_HIP(hipMemcpyAsync(text.p, sam_text, sam_bytes, hipMemcpyHostToDevice, T->copy_stream));
  if (open_end) SG_HIP(hipMemsetAsync(text.as<char>() + sam_bytes, '\n', 1, T->copy_stream));
  int rc = train_settle(ctx, T);
  SG_HIP(hipStreamSynchronize(T->copy_stream));   // the caller's buffer is free again when this returns
  if (rc != SG_OK) return rc;
  if (T->capped) return SG_OK;
  SG_ENSURE(T->scan_work, sg::train_scan_work_bytes(bytes / 64 + 1));
  sg::TrainJob J;
  train_job(ctx, T, J);
  J.text = text.as<char>();
  J.bytes = bytes;
  J.scan_work = T->scan_work.p;
  sg::launch_train_lines_count(J, s);
  SG_HIP(hipGetLastError());
  SG_HIP(hipMemcpyAsync(T->mail, J.carry_out, sizeof(sg::TrainCarry), hipMemcpyDeviceToHost, s));
  SG_HIP(hipStreamSynchronize(s));
  const uint64_t n_lines = T->mail->n_lines;
  if (!n_lines) return SG_OK;
  SG_ENSURE(T->line_end, n_lines * 8 + 64);
  SG_ENSURE(T->reads, n_lines * sizeof(sg::TrainRead) + 64);
  if (T->count_gc) {
    SG_ENSURE(T->gate, n_lines * sizeof(sg::TrainGate) + 64);
    SG_ENSURE(T->steps, n_lines * sizeof(sg::TrainStep) + 64);
    const uint64_t want = T->n_windows + n_lines + 1;
    if (want > T->windows_cap) {
      const uint64_t cap = want + want / 2;
      if ((rc = grow_keep(ctx, T->windows, T->n_windows * sizeof(sg::TrainWindow), cap * sizeof(sg::TrainWindow))) != SG_OK) return rc;
      if ((rc = grow_keep(ctx, T->window_rc, T->n_windows * 4, cap * 4)) != SG_OK) return rc;
      SG_HIP(hipMemsetAsync(T->window_rc.as<uint32_t>() + T->n_windows, 0, (cap - T->n_windows) * 4, s));
      T->windows_cap = cap;
    }
  }
  SG_ENSURE(T->scan_work, sg::train_scan_work_bytes(std::max<uint64_t>(bytes / 64 + 1, n_lines)));
  if (T->scan_work.p != J.scan_work) {   // (the tile prefixes of the line scan sit in the old block: count again)
    J.scan_work = T->scan_work.p;
    sg::launch_train_lines_count(J, s);
  }
  J.line_end = T->line_end.as<uint64_t>();
  J.n_lines = n_lines;
  J.reads = T->reads.as<sg::TrainRead>();
  J.gate = T->gate.as<sg::TrainGate>();
  J.steps = T->steps.as<sg::TrainStep>();
  J.windows = T->windows.as<sg::TrainWindow>();
  J.window_rc = T->window_rc.as<uint32_t>();
  sg::launch_train_lines_fill(J, s);
  sg::launch_train_chunk(J, s);
  SG_HIP(hipGetLastError());
  SG_HIP(hipMemcpyAsync(T->mail, J.carry_out, sizeof(sg::TrainCarry), hipMemcpyDeviceToHost, s));
  SG_HIP(hipMemcpyAsync(T->mail_flags(), T->flags.p, 4, hipMemcpyDeviceToHost, s));
  T->pending = true;
  T->pending_lines = n_lines;
  T->cur ^= 1;
  T->fed++;
  return SG_OK;
}

// 1 once the cap on counted reads was reached. A chunk's verdict is known when the next one is fed (or at sg_train_finish):
// a caller that stops feeding on it has handed over at most one chunk the reference would not have read, which is dropped.
int sg_train_capped(sg_ctx* ctx) { return ctx && ctx->train && ctx->train->capped ? 1 : 0; }

int sg_train_finish(sg_ctx* ctx, sg_train_counts* out, double* gc, double* rc, uint64_t gc_cap, uint64_t* n_gc) {
  if (!ctx || !out) return SG_ERR_INVALID;
  sg_train_session* T = ctx->train;
  if (!T) return ctx->fail(SG_ERR_INVALID, "sg_train_finish: call sg_train_begin first");
  SG_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  {
    const int settled = train_settle(ctx, T);
    if (settled != SG_OK) return settled;
  }
  // ---- the windows countGC pushed (Profile.cpp:559-570, 623-634): every window but the last one it was in, if its GC
  // content is above zero and it counted a read; in the order they were opened ----
  std::vector<double> h_gc, h_rc;
  std::vector<uint32_t> h_raw;
  const uint64_t nw = T->n_windows;
  if (nw) {
    SG_ENSURE(T->gc_out, nw * 16 + 64);
    sg::TrainJob J;
    train_job(ctx, T, J);
    double* d_gc = T->gc_out.as<double>();
    sg::launch_train_window_gc(T->windows.as<sg::TrainWindow>(), T->window_rc.as<uint32_t>(), nw, J.contigs, J.ref_codes, T->wes, d_gc, d_gc + nw, s);
    SG_HIP(hipGetLastError());
    h_gc.resize(nw); h_rc.resize(nw); h_raw.resize(nw);
    SG_HIP(hipMemcpyAsync(h_gc.data(), d_gc, nw * 8, hipMemcpyDeviceToHost, s));
    SG_HIP(hipMemcpyAsync(h_rc.data(), d_gc + nw, nw * 8, hipMemcpyDeviceToHost, s));
    SG_HIP(hipMemcpyAsync(h_raw.data(), T->window_rc.p, nw * 4, hipMemcpyDeviceToHost, s));
  }
  std::vector<uint64_t> host(T->counters);
  SG_HIP(hipMemcpyAsync(host.data(), T->counts.p, T->counters * 8, hipMemcpyDeviceToHost, s));
  SG_HIP(hipStreamSynchronize(s));
  uint64_t pushed = 0;
  for (uint64_t w = 0; w + 1 < nw; w++)
    if (h_gc[w] > 0 && h_raw[w] > 0) {
      if (pushed < gc_cap && gc && rc) { gc[pushed] = h_gc[w]; rc[pushed] = h_rc[w]; }
      pushed++;
    }
  if (n_gc) *n_gc = pushed;
  if (gc && rc && pushed > gc_cap) return ctx->fail(SG_ERR_OVERFLOW, "sg_train_finish: more (GC, read count) pairs than gc_cap");
  const uint64_t* h = host.data();
  if (out->subs1) memcpy(out->subs1, h, T->subs_n * 8);
  if (out->subs2) memcpy(out->subs2, h + T->subs_n, T->subs_n * 8);
  if (out->kmers) memcpy(out->kmers, h + 2 * T->subs_n, T->kmers_n * 8);
  if (out->quality) memcpy(out->quality, h + 2 * T->subs_n + T->kmers_n, T->qual_n * 8);
  const uint64_t* is = h + 2 * T->subs_n + T->kmers_n + T->qual_n;
  if (out->isize) memcpy(out->isize, is, (size_t)T->n_isize * 8);
  if (out->ins_len) memcpy(out->ins_len, is + T->n_isize, (size_t)T->n_indel_len * 8);
  if (out->del_len) memcpy(out->del_len, is + T->n_isize + T->n_indel_len, (size_t)T->n_indel_len * 8);
  const uint64_t* sc = is + T->n_isize + 2 * (size_t)T->n_indel_len;
  out->lines = T->lines - sc[sg::kTrainEmptyLines];   // (empty lines are no reads: Profile.cpp:229-231)
  out->reads_counted = sc[sg::kTrainReads];
  out->cigar_chars = sc[sg::kTrainCigarChars];
  out->insert_events = sc[sg::kTrainInsEvents];
  out->delete_events = sc[sg::kTrainDelEvents];
  out->isize_overflow = sc[sg::kTrainIsizeOverflow];
  out->indel_len_overflow = sc[sg::kTrainIndelLenOverflow];
  out->skipped_overhang = sc[sg::kTrainOverhang];
  out->gc_rejected = sc[sg::kTrainGcRejected];
  out->gc_windows = nw;
  out->capped = T->capped ? 1 : 0;
  sg_train_end(ctx);
  return SG_OK;
}

int sg_train_count(sg_ctx* ctx, const char* sam_text, uint64_t sam_bytes, const char* const* contig_keys, uint32_t n_contigs,
                   const char* bases, int32_t kmer, int32_t bins, uint32_t n_isize, uint32_t n_indel_len, sg_train_counts* out) {
  if (!ctx || !out) return SG_ERR_INVALID;
  sg_train_setup st;
  memset(&st, 0, sizeof st);
  st.contig_keys = contig_keys; st.n_contigs = n_contigs; st.bases = bases; st.kmer = kmer; st.bins = bins;
  st.n_isize = n_isize; st.n_indel_len = n_indel_len; st.count_gc = 0; st.window = 1000;
  int rc = sg_train_begin(ctx, &st);
  if (rc == SG_OK) rc = sg_train_feed(ctx, sam_text, sam_bytes);
  if (rc == SG_OK) rc = sg_train_finish(ctx, out, nullptr, nullptr, 0, nullptr);
  if (rc != SG_OK) sg_train_end(ctx);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// detached outputs
// ------------------------------------------------------------------------------------------------
int sg_detach_outputs(sg_ctx* ctx, sg_outputs** out) {
  if (!ctx || !out) return SG_ERR_INVALID;
  if (!ctx->results_valid) return ctx->fail(SG_ERR_INVALID, "sg_detach_outputs: call sg_result first");
  SG_HIP(hipSetDevice(ctx->device));
  sg_outputs* o = nullptr;
  if (!ctx->spare.empty()) { o = ctx->spare.back(); ctx->spare.pop_back(); }
  if (!o) {
    o = new sg_outputs();
    o->device = ctx->device;
    hipError_t e = hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete o; return ctx->hipfail(e, "hipStreamCreate(outputs)"); }
  }
  // whatever the spare still holds goes back to the context (it is empty or smaller than what was just used)
  std::swap(ctx->out1, o->text[0]); std::swap(ctx->out2, o->text[1]);
  std::swap(ctx->gz1, o->gz[0]); std::swap(ctx->gz2, o->gz[1]);
  o->text_bytes[0] = ctx->host_totals[0];
  o->text_bytes[1] = ctx->B.paired ? ctx->host_totals[1] : 0;
  o->gz_bytes[0] = ctx->gz_valid ? ctx->gz_bytes[0] : 0;
  o->gz_bytes[1] = ctx->gz_valid ? ctx->gz_bytes[1] : 0;
  ctx->results_valid = false;
  ctx->sampled = false;
  ctx->gz_valid = false;
  *out = o;
  return SG_OK;
}

int sg_outputs_sizes(const sg_outputs* o, uint64_t text_bytes[2], uint64_t gz_bytes[2]) {
  if (!o) return SG_ERR_INVALID;
  for (int m = 0; m < 2; m++) {
    if (text_bytes) text_bytes[m] = o->text_bytes[m];
    if (gz_bytes) gz_bytes[m] = o->gz_bytes[m];
  }
  return SG_OK;
}

const char* sg_outputs_last_error(const sg_outputs* o) { return o ? o->err.c_str() : ""; }

int sg_outputs_fetch(sg_outputs* o, int mate, int compressed, uint64_t offset, uint64_t bytes, void* host_dst) {
  if (!o || mate < 0 || mate > 1 || (bytes && !host_dst)) return SG_ERR_INVALID;
  const uint64_t have = compressed ? o->gz_bytes[mate] : o->text_bytes[mate];
  if (offset + bytes > have) { o->err = "sg_outputs_fetch: range past the end of the data"; return SG_ERR_INVALID; }
  if (!bytes) return SG_OK;
  hipError_t e = hipSetDevice(o->device);
  const uint8_t* src = (compressed ? o->gz[mate] : o->text[mate]).as<uint8_t>() + offset;
  if (e == hipSuccess) e = hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, o->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(o->stream);
  if (e != hipSuccess) { o->err = std::string("sg_outputs_fetch: ") + hipGetErrorString(e); return SG_ERR_HIP; }
  return SG_OK;
}

int sg_release_outputs(sg_ctx* ctx, sg_outputs* o) {
  if (!ctx || !o) return SG_ERR_INVALID;
  SG_HIP(hipSetDevice(ctx->device));
  SG_HIP(hipStreamSynchronize(o->stream));
  o->text_bytes[0] = o->text_bytes[1] = o->gz_bytes[0] = o->gz_bytes[1] = 0;
  ctx->spare.push_back(o);
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------
// block-gzip sink (kernels: sg_deflate.hip, code construction: sg_deflate.cpp)
// ------------------------------------------------------------------------------------------------
int sg_bgzf_eof(uint8_t out[28]) {
  static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (!out) return SG_ERR_INVALID;
  memcpy(out, eof, 28);
  return SG_OK;
}

uint32_t sg_deflate_plan(const uint64_t lit_counts[286], const uint64_t dist_counts[30], uint8_t lit_lens[286], uint32_t lit_codes[286],
                         uint8_t dist_lens[30], uint32_t dist_codes[30], uint32_t len_tokens[260], uint32_t* prefix_words, uint32_t cap) {
  if (!lit_counts || !dist_counts || !lit_lens || !lit_codes || !dist_lens || !dist_codes || !len_tokens || !prefix_words) return 0;
  sg::DeflatePlan plan;
  sg::deflate_build_plan(lit_counts, dist_counts, &plan);
  if (plan.prefix.size() > cap) return 0;
  memcpy(lit_lens, plan.lit_len, sizeof plan.lit_len);
  memcpy(lit_codes, plan.lit_code, sizeof plan.lit_code);
  memcpy(dist_lens, plan.dist_len, sizeof plan.dist_len);
  memcpy(dist_codes, plan.dist_code, sizeof plan.dist_code);
  memcpy(len_tokens, plan.len_token, sizeof plan.len_token);
  memcpy(prefix_words, plan.prefix.data(), plan.prefix.size() * 4);
  return plan.prefix_bits;
}

int sg_compress(sg_ctx* ctx, uint64_t* gz_bytes_r1, uint64_t* gz_bytes_r2) {
  if (!ctx) return SG_ERR_INVALID;
  if (!ctx->results_valid) return ctx->fail(SG_ERR_INVALID, "sg_compress: call sg_result first");
  SG_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int nm = ctx->B.paired ? 2 : 1;
  ctx->gz_bytes[0] = ctx->gz_bytes[1] = 0;
  for (int m = 0; m < nm; m++) {
    const uint64_t bytes = ctx->host_totals[m];
    if (!bytes) continue;
    if (bytes / sg::kGzChunk >= 0xFFFFFFF0ull) return ctx->fail(SG_ERR_UNSUPPORTED, "sg_compress: text too large");
    const uint8_t* text = m == 0 ? ctx->out1.as<uint8_t>() : ctx->out2.as<uint8_t>();
    const uint32_t n_chunks = (uint32_t)((bytes + sg::kGzChunk - 1) / sg::kGzChunk);
    // work buffer: hist[320] u64 | total u64 | counters | tables | msize[n] u32 | moff[n] u64 | block sums | lane bits | records
    const size_t head = 320 * 8 + 64;
    const size_t tab_words = 288 + sg::kGzLenTokens + 32 + 1024 + sg::kGzLevels * 128 + 256;  // + prefix (<= 256 words)
    const size_t off_tab = head, off_msize = off_tab + tab_words * 4;
    const size_t off_moff = (off_msize + (size_t)n_chunks * 4 + 63) & ~(size_t)63;
    const size_t off_bsum = off_moff + (size_t)n_chunks * 8;
    const size_t off_lbits = (off_bsum + ((size_t)sg::scan_blocks(n_chunks) + 8) * 8 + 63) & ~(size_t)63;
    const size_t off_rec = (off_lbits + (size_t)n_chunks * sg::kGzThreads * 4 + 63) & ~(size_t)63;
    SG_ENSURE(ctx->gz_work, off_rec + (size_t)n_chunks * sg::kGzThreads * 32);
    uint8_t* wk = ctx->gz_work.as<uint8_t>();
    sg::DevDeflate D;
    memset(&D, 0, sizeof D);
    D.text = text; D.bytes = bytes; D.n_chunks = n_chunks;
    D.min_run = sg::kGzMinRun;
    D.min_copy = sg::kGzMinGramMatch;
    if (const char* e = getenv("SG_GZ_MINCOPY")) D.min_copy = (uint32_t)std::max(8, atoi(e));   // (experiments)
    if (const char* e = getenv("SG_GZ_MINRUN")) D.min_run = (uint32_t)std::max(3, atoi(e));   // (experiments)
    D.hist = (unsigned long long*)wk;
    D.next = (uint32_t*)(wk + 320 * 8 + 16);  // inside the zeroed head of the work buffer
    D.msize = (uint32_t*)(wk + off_msize);
    D.moff = (const uint64_t*)(wk + off_moff);
    D.lbits = (uint32_t*)(wk + off_lbits);
    D.rec = (uint4*)(wk + off_rec);
    // 1. token histogram of every 16th member -> the two Huffman codes, member prefix, CRC tables (host)
    SG_HIP(hipMemsetAsync(wk, 0, head, s));
    sg::launch_gz_hist(&D, n_chunks, s);
    SG_HIP(hipGetLastError());
    uint64_t hist[320];
    SG_HIP(hipMemcpyAsync(hist, wk, sizeof hist, hipMemcpyDeviceToHost, s));
    SG_HIP(hipStreamSynchronize(s));
    sg::DeflatePlan plan;
    sg::deflate_build_plan(hist, hist + 288, &plan);
    if (plan.prefix.size() > 256) return ctx->fail(SG_ERR_UNSUPPORTED, "sg_compress: block header longer than expected");
    if (getenv("SG_GZ_TRACE")) {   // where the sampled members' bits go: literals by character, matches by length / distance symbol
      double lit_bits[256], tot_lit = 0, tot_len = 0, tot_dist = 0, n_match = 0, n_lit = 0, match_bytes = 0;
      static const int lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
      static const int lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
      for (int c = 0; c < 256; c++) { lit_bits[c] = (double)hist[c] * plan.lit_len[c]; tot_lit += lit_bits[c]; n_lit += (double)hist[c]; }
      for (int i = 0; i < 29; i++) { tot_len += (double)hist[257 + i] * (plan.lit_len[257 + i] + lext[i]); n_match += (double)hist[257 + i];
                                     match_bytes += (double)hist[257 + i] * (lbase[i] + (i + 1 < 29 ? (lbase[i + 1] - lbase[i] - 1) / 2.0 : 0)); }
      for (int i = 0; i < 30; i++) tot_dist += (double)hist[288 + i] * (plan.dist_len[i] + (i < 4 ? 0 : (i - 2) / 2));
      const double members = (double)hist[256], all = tot_lit + tot_len + tot_dist;
      fprintf(stderr, "[gz] per member: %.0f literals %.0f bits, %.0f matches (~%.0f bytes) %.0f length bits + %.0f distance bits; total %.0f bits = %.0f bytes\n",
              n_lit / members, tot_lit / members, n_match / members, match_bytes / members, tot_len / members, tot_dist / members, all / members, all / members / 8);
      fprintf(stderr, "[gz] literal bits per member by character:");
      for (int c = 0; c < 256; c++)
        if (lit_bits[c] / members >= 20) fprintf(stderr, " '%c'x%.0f(%d b)=%.0f", c >= 32 && c < 127 ? c : '?', hist[c] / members, plan.lit_len[c], lit_bits[c] / members);
      fprintf(stderr, "\n[gz] matches per member by length symbol:");
      for (int i = 0; i < 29; i++) if (hist[257 + i] / members >= 1) fprintf(stderr, " %d+:%0.f(%d b)", lbase[i], hist[257 + i] / members, plan.lit_len[257 + i] + lext[i]);
      fprintf(stderr, "\n[gz] distance symbols:");
      for (int i = 0; i < 30; i++) if (hist[288 + i] / members >= 1) fprintf(stderr, " %d:%.0f(%d b)", i, hist[288 + i] / members, plan.dist_len[i] + (i < 4 ? 0 : (i - 2) / 2));
      fprintf(stderr, "\n");
    }
    std::vector<uint32_t> tab(tab_words, 0);
    for (int i = 0; i < sg::kGzLitSyms; i++) tab[i] = plan.lit_code[i] | ((uint32_t)plan.lit_len[i] << 16);
    memcpy(&tab[288], plan.len_token, sizeof plan.len_token);
    for (int i = 0; i < sg::kGzDistSyms; i++) tab[288 + sg::kGzLenTokens + i] = plan.dist_code[i] | ((uint32_t)plan.dist_len[i] << 16);
    const size_t t_crc = 288 + sg::kGzLenTokens + 32;
    memcpy(&tab[t_crc], plan.crc_table, sizeof plan.crc_table);
    memcpy(&tab[t_crc + 1024], plan.crc_shift, sizeof plan.crc_shift);
    memcpy(&tab[t_crc + 1024 + sg::kGzLevels * 128], plan.prefix.data(), plan.prefix.size() * 4);
    SG_HIP(hipMemcpyAsync(wk + off_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
    D.code = (const uint32_t*)(wk + off_tab);
    D.len_tok = D.code + 288;
    D.dist_code = D.len_tok + sg::kGzLenTokens;
    D.crc_tab = D.dist_code + 32;
    D.crc_shift = D.crc_tab + 1024;
    D.prefix = D.crc_shift + sg::kGzLevels * 128;
    D.prefix_words = (uint32_t)plan.prefix.size();
    D.prefix_bits = plan.prefix_bits;
    D.crc_init_full = plan.crc_init_full;
    const uint64_t last = bytes - (uint64_t)(n_chunks - 1) * sg::kGzChunk;
    D.crc_init_last = sg::crc_advance(plan, 0xFFFFFFFFu, last);
    // 2. tokens, member sizes -> offsets
    sg::launch_gz_match(&D, n_chunks, s);
    sg::launch_scan_u32(D.msize, n_chunks, (uint64_t*)(wk + off_bsum), (uint64_t*)(wk + off_moff), (uint64_t*)(wk + 320 * 8), s);
    SG_HIP(hipGetLastError());
    uint64_t total = 0;
    SG_HIP(hipMemcpyAsync(&total, wk + 320 * 8, 8, hipMemcpyDeviceToHost, s));
    SG_HIP(hipStreamSynchronize(s));
    DevBuf& gz = m == 0 ? ctx->gz1 : ctx->gz2;
    SG_ENSURE(gz, total + 64);
    D.out = gz.as<uint8_t>();
    // 3. encode
    sg::launch_gz_encode(&D, n_chunks, plan.prefix_bits, s);
    SG_HIP(hipGetLastError());
    SG_HIP(hipStreamSynchronize(s));  // tab / plan are host-owned
    ctx->gz_bytes[m] = total;
  }
  ctx->gz_valid = true;
  if (gz_bytes_r1) *gz_bytes_r1 = ctx->gz_bytes[0];
  if (gz_bytes_r2) *gz_bytes_r2 = ctx->gz_bytes[1];
  return SG_OK;
}

int sg_fetch_compressed(sg_ctx* ctx, int mate, uint64_t offset, uint64_t bytes, void* host_dst) {
  if (!ctx || mate < 0 || mate > 1 || (bytes && !host_dst)) return SG_ERR_INVALID;
  if (!ctx->gz_valid || !ctx->results_valid) return ctx->fail(SG_ERR_INVALID, "sg_fetch_compressed: call sg_compress first");
  if (offset + bytes > ctx->gz_bytes[mate]) return ctx->fail(SG_ERR_INVALID, "sg_fetch_compressed: range past the end of the compressed text");
  SG_HIP(hipSetDevice(ctx->device));
  if (bytes) {
    const uint8_t* src = (mate == 0 ? ctx->gz1.as<uint8_t>() : ctx->gz2.as<uint8_t>()) + offset;
    SG_HIP(hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SG_HIP(hipStreamSynchronize(ctx->stream));
  }
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------
// reference ingest + haplotype assembly on the device (kernels: sg_haplotypes.hip)
// ------------------------------------------------------------------------------------------------
int sg_reference_begin(sg_ctx* ctx, uint64_t raw_bytes) {
  if (!ctx) return SG_ERR_INVALID;
  SG_HIP(hipSetDevice(ctx->device));
  SG_ENSURE(ctx->ref_raw, raw_bytes + 64);
  SG_HIP(hipMemsetAsync((uint8_t*)ctx->ref_raw.p + raw_bytes, '\n', 64, ctx->stream));  // the scan reads whole 16-byte words
  ctx->ref_raw_bytes = raw_bytes;
  ctx->ref_contigs.clear();
  return SG_OK;
}

int sg_reference_chunk(sg_ctx* ctx, uint64_t offset, const void* host, uint64_t bytes) {
  if (!ctx || (!host && bytes)) return SG_ERR_INVALID;
  if (!ctx->ref_raw.p || offset + bytes > ctx->ref_raw_bytes) return ctx->fail(SG_ERR_INVALID, "sg_reference_chunk: range outside sg_reference_begin's size");
  SG_HIP(hipSetDevice(ctx->device));
  if (bytes) SG_HIP(hipMemcpyAsync((uint8_t*)ctx->ref_raw.p + offset, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return SG_OK;
}

int sg_sync(sg_ctx* ctx) {
  if (!ctx) return SG_ERR_INVALID;
  SG_HIP(hipSetDevice(ctx->device));
  SG_HIP(hipStreamSynchronize(ctx->stream));
  return SG_OK;
}

int sg_reference_scan(sg_ctx* ctx, uint64_t* header_offsets, uint32_t cap, uint32_t* n_found, uint32_t* flags) {
  if (!ctx || !n_found || (cap && !header_offsets)) return SG_ERR_INVALID;
  if (!ctx->ref_raw.p) return ctx->fail(SG_ERR_INVALID, "sg_reference_scan: call sg_reference_begin first");
  SG_HIP(hipSetDevice(ctx->device));
  SG_ENSURE(ctx->hap_work, (size_t)cap * 8 + 64);
  uint32_t* counters = (uint32_t*)((uint8_t*)ctx->hap_work.p + (size_t)cap * 8);
  SG_HIP(hipMemsetAsync(counters, 0, 8, ctx->stream));
  sg::launch_ref_scan(ctx->ref_raw.as<uint8_t>(), ctx->ref_raw_bytes, ctx->hap_work.as<uint64_t>(), cap, counters, counters + 1, ctx->stream);
  SG_HIP(hipGetLastError());
  uint32_t host_c[2] = {0, 0};
  SG_HIP(hipMemcpyAsync(host_c, counters, 8, hipMemcpyDeviceToHost, ctx->stream));
  SG_HIP(hipStreamSynchronize(ctx->stream));
  *n_found = host_c[0];
  if (flags) *flags = host_c[1];
  const uint32_t n = host_c[0] < cap ? host_c[0] : cap;
  if (n) SG_HIP(hipMemcpy(header_offsets, ctx->hap_work.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return SG_OK;
}

int sg_reference_commit(sg_ctx* ctx, const sg_contig* contigs, uint32_t n_contigs) {
  if (!ctx || (n_contigs && !contigs)) return SG_ERR_INVALID;
  if (!ctx->ref_raw.p) return ctx->fail(SG_ERR_INVALID, "sg_reference_commit: call sg_reference_begin first");
  SG_HIP(hipSetDevice(ctx->device));
  std::vector<sg::DevContig> tab(n_contigs);
  uint64_t code_off = 0, blocks = 0;
  for (uint32_t c = 0; c < n_contigs; c++) {
    const sg_contig& k = contigs[c];
    if (k.length && (k.line_bases == 0 || k.line_width < k.line_bases))
      return ctx->fail(SG_ERR_INVALID, "sg_reference_commit: contig " + std::to_string(c) + " has an impossible line shape");
    if (k.length > 0xFFFFFFF0ull)   // (the ingest kernel's line / column arithmetic is 32-bit; the host says the same, fasta.cpp)
      return ctx->fail(SG_ERR_UNSUPPORTED, "sg_reference_commit: contig " + std::to_string(c) + " is longer than 4 Gbp");
    const uint64_t lines = k.length ? (k.length - 1) / k.line_bases : 0;  // line breaks inside the contig
    if (k.raw_offset + k.length + lines * (k.line_width - k.line_bases) > ctx->ref_raw_bytes)
      return ctx->fail(SG_ERR_INVALID, "sg_reference_commit: contig " + std::to_string(c) + " runs past the end of the file");
    tab[c] = sg::DevContig{k.raw_offset, code_off, k.length, k.line_bases, k.line_width, blocks};
    blocks += (k.length + 15) / 16;
    code_off += ((k.length + 15) / 16) * 16 + 64;
  }
  SG_ENSURE(ctx->ref_codes, code_off + 64);
  SG_ENSURE(ctx->ref_meta, (size_t)n_contigs * sizeof(sg::DevContig) + 64);
  uint32_t* flags = (uint32_t*)((uint8_t*)ctx->ref_meta.p + (size_t)n_contigs * sizeof(sg::DevContig));
  SG_HIP(hipMemsetAsync(flags, 0, 4, ctx->stream));
  if (n_contigs) SG_HIP(hipMemcpyAsync(ctx->ref_meta.p, tab.data(), (size_t)n_contigs * sizeof(sg::DevContig), hipMemcpyHostToDevice, ctx->stream));
  sg::launch_ref_ingest(ctx->ref_raw.as<uint8_t>(), ctx->ref_codes.as<uint8_t>(), ctx->ref_meta.p, n_contigs, blocks, flags, ctx->stream);
  SG_HIP(hipGetLastError());
  uint32_t host_flags = 0;
  SG_HIP(hipMemcpyAsync(&host_flags, flags, 4, hipMemcpyDeviceToHost, ctx->stream));
  SG_HIP(hipStreamSynchronize(ctx->stream));
  if (host_flags & 2u) return ctx->fail(SG_ERR_FORMAT, "sg_reference_commit: a contig's lines are not of one width");
  ctx->ref_contigs = tab;
  ctx->ref_raw.release();  // the file image is not needed again
  ctx->ref_raw_bytes = 0;
  return SG_OK;
}

int sg_build_haplotypes(sg_ctx* ctx, int32_t n_chains, const uint64_t* lens, const sg_hap_piece* pieces, uint64_t n_pieces,
                        const char* literals, uint64_t n_literal_bytes, const sg_hap_patch* patches, uint64_t n_patches) {
  if (!ctx || n_chains < 0 || (n_chains && !lens) || (n_pieces && !pieces) || (n_patches && !patches) ||
      (n_literal_bytes && !literals))
    return SG_ERR_INVALID;
  if (ctx->ref_contigs.empty() && n_pieces) return ctx->fail(SG_ERR_INVALID, "sg_build_haplotypes: call sg_reference_commit first");
  SG_HIP(hipSetDevice(ctx->device));
  const size_t PAD = 256;  // same layout as sg_upload_haplotypes
  std::vector<uint64_t> meta(2 * (size_t)n_chains + 2, 0);
  size_t total = PAD;
  for (int c = 0; c < n_chains; c++) {
    meta[c] = total;
    meta[n_chains + c] = lens[c];
    total += (lens[c] + PAD + 63) & ~(size_t)63;
  }
  total += PAD;
  total = (total + 1023) & ~(size_t)1023;
  // absolute offsets, long pieces split so that every workgroup moves <= 64 KB; coverage is checked
  // by summing the piece lengths per chain after a bounds check of each piece
  const uint32_t kSplit = 1u << 16;
  std::vector<sg::DevPiece> dp;
  dp.reserve((size_t)n_pieces + total / kSplit + 16);
  std::vector<uint64_t> covered((size_t)n_chains, 0);
  for (uint64_t i = 0; i < n_pieces; i++) {
    const sg_hap_piece& p = pieces[i];
    if ((int64_t)p.chain >= n_chains || p.dst + p.len > lens[p.chain])
      return ctx->fail(SG_ERR_INVALID, "sg_build_haplotypes: piece " + std::to_string(i) + " falls outside its chain");
    uint64_t src;
    if (p.kind == 0) {
      if (p.contig >= ctx->ref_contigs.size() || p.src + p.len > ctx->ref_contigs[p.contig].length)
        return ctx->fail(SG_ERR_INVALID, "sg_build_haplotypes: piece " + std::to_string(i) + " falls outside its contig");
      src = ctx->ref_contigs[p.contig].code_off + p.src;
    } else {
      if (p.src + p.len > n_literal_bytes)
        return ctx->fail(SG_ERR_INVALID, "sg_build_haplotypes: piece " + std::to_string(i) + " falls outside the literal bytes");
      src = p.src;
    }
    covered[p.chain] += p.len;
    for (uint32_t o = 0; o < p.len; o += kSplit)
      dp.push_back(sg::DevPiece{meta[p.chain] + p.dst + o, src + o, std::min<uint32_t>(kSplit, p.len - o), p.kind ? 1u : 0u});
  }
  for (int c = 0; c < n_chains; c++)
    if (covered[c] != lens[c]) return ctx->fail(SG_ERR_INVALID, "sg_build_haplotypes: the pieces of chain " + std::to_string(c) + " do not add up to its length");
  std::vector<sg::DevPatch> pt((size_t)n_patches);
  for (uint64_t i = 0; i < n_patches; i++) {
    const sg_hap_patch& q = patches[i];
    if ((int64_t)q.chain >= n_chains || q.dst >= lens[q.chain])
      return ctx->fail(SG_ERR_INVALID, "sg_build_haplotypes: patch " + std::to_string(i) + " falls outside its chain");
    pt[i] = sg::DevPatch{meta[q.chain] + q.dst, q.base, 0};
  }
  SG_ENSURE(ctx->chains, total);
  SG_ENSURE(ctx->chain_meta, meta.size() * 8);
  const size_t pieces_b = (dp.size() * sizeof(sg::DevPiece) + 63) & ~(size_t)63, patches_b = (pt.size() * sizeof(sg::DevPatch) + 63) & ~(size_t)63;
  SG_ENSURE(ctx->hap_work, pieces_b + patches_b + n_literal_bytes + 64);
  uint8_t* wk = ctx->hap_work.as<uint8_t>();
  SG_HIP(hipMemsetAsync(ctx->chains.p, 4, total, ctx->stream));  // guard bytes read as 'N'
  if (!dp.empty()) SG_HIP(hipMemcpyAsync(wk, dp.data(), dp.size() * sizeof(sg::DevPiece), hipMemcpyHostToDevice, ctx->stream));
  if (!pt.empty()) SG_HIP(hipMemcpyAsync(wk + pieces_b, pt.data(), pt.size() * sizeof(sg::DevPatch), hipMemcpyHostToDevice, ctx->stream));
  if (n_literal_bytes) {
    SG_HIP(hipMemcpyAsync(wk + pieces_b + patches_b, literals, n_literal_bytes, hipMemcpyHostToDevice, ctx->stream));
    sg::launch_encode_bytes(wk + pieces_b + patches_b, n_literal_bytes, ctx->stream);
  }
  SG_HIP(hipMemcpyAsync(ctx->chain_meta.p, meta.data(), meta.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  sg::launch_hap_copy(ctx->chains.as<uint8_t>(), ctx->ref_codes.as<uint8_t>(), wk + pieces_b + patches_b, wk, dp.size(), ctx->stream);
  sg::launch_hap_patch(ctx->chains.as<uint8_t>(), wk + pieces_b, pt.size(), ctx->stream);
  if (int rc = pack_chains(ctx, total)) return rc;
  SG_HIP(hipGetLastError());
  SG_HIP(hipStreamSynchronize(ctx->stream));  // dp / pt / meta are stack-owned host memory
  ctx->B.chains = ctx->chains.as<uint8_t>();
  ctx->B.chain_off = ctx->chain_meta.as<uint64_t>();
  ctx->B.chain_len = ctx->chain_meta.as<uint64_t>() + n_chains;
  ctx->have_haps = true;
  ctx->have_plan = false;
  return SG_OK;
}

int sg_haplotype_codes(sg_ctx* ctx, uint32_t chain, uint64_t offset, uint64_t n, uint8_t* codes_out) {
  if (!ctx || (n && !codes_out)) return SG_ERR_INVALID;
  if (!ctx->have_haps) return ctx->fail(SG_ERR_INVALID, "sg_haplotype_codes: no haplotypes on the device");
  SG_HIP(hipSetDevice(ctx->device));
  uint64_t meta[2];
  // chain_meta = [off_0 .. off_{k-1}, len_0 .. len_{k-1}]; k is recovered from the pointers kept in B
  const uint64_t k = (uint64_t)(ctx->B.chain_len - ctx->B.chain_off);
  if (chain >= k) return ctx->fail(SG_ERR_INVALID, "sg_haplotype_codes: chain index out of range");
  SG_HIP(hipMemcpy(&meta[0], ctx->B.chain_off + chain, 8, hipMemcpyDeviceToHost));
  SG_HIP(hipMemcpy(&meta[1], ctx->B.chain_len + chain, 8, hipMemcpyDeviceToHost));
  if (offset + n > meta[1]) return ctx->fail(SG_ERR_INVALID, "sg_haplotype_codes: range past the end of the chain");
  if (n) SG_HIP(hipMemcpy(codes_out, ctx->B.chains + meta[0] + offset, n, hipMemcpyDeviceToHost));
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------
int sg_upload_haplotypes(sg_ctx* ctx, int32_t n_chains, const char* const* chains, const uint64_t* lens) {
  if (!ctx || n_chains < 0 || (n_chains && (!chains || !lens))) return SG_ERR_INVALID;
  SG_HIP(hipSetDevice(ctx->device));
  const size_t PAD = 256;  // front/back guard: kernels read up to 16 bytes around a fragment
  std::vector<uint64_t> meta(2 * (size_t)n_chains + 2, 0);  // [off..., len...]
  size_t total = PAD;  // front pad: the emit kernel reads up to 11 bytes before a fragment start
  for (int c = 0; c < n_chains; c++) {
    meta[c] = total;
    meta[n_chains + c] = lens[c];
    total += (lens[c] + PAD + 63) & ~(size_t)63;
  }
  total += PAD;
  total = (total + 1023) & ~(size_t)1023;
  SG_ENSURE(ctx->chains, total);
  SG_ENSURE(ctx->chain_meta, meta.size() * 8);
  SG_HIP(hipMemsetAsync(ctx->chains.p, 'N', total, ctx->stream));
  for (int c = 0; c < n_chains; c++)
    if (lens[c]) SG_HIP(hipMemcpyAsync((uint8_t*)ctx->chains.p + meta[c], chains[c], lens[c], hipMemcpyHostToDevice, ctx->stream));
  SG_HIP(hipMemcpyAsync(ctx->chain_meta.p, meta.data(), meta.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  // ASCII -> base codes, in place (A0 C1 T2 G3, N=4, other=5): the kernels never see ASCII
  sg::launch_encode((uint8_t*)ctx->chains.p, total, ctx->stream);
  if (int rc = pack_chains(ctx, total)) return rc;
  SG_HIP(hipGetLastError());
  SG_HIP(hipStreamSynchronize(ctx->stream));
  ctx->B.chains = ctx->chains.as<uint8_t>();
  ctx->B.chain_off = ctx->chain_meta.as<uint64_t>();
  ctx->B.chain_len = ctx->chain_meta.as<uint64_t>() + n_chains;
  ctx->have_haps = true;
  ctx->have_plan = false;
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------
int sg_plan(sg_ctx* ctx, const sg_batch* b) {
  if (!ctx || !b) return SG_ERR_INVALID;
  if (!ctx->have_profile) return ctx->fail(SG_ERR_INVALID, "sg_plan: call sg_load_profile first");
  if (!ctx->have_haps) return ctx->fail(SG_ERR_INVALID, "sg_plan: call sg_upload_haplotypes first");
  if (b->n_windows && (!b->windows || !b->seg_size || !b->seg_first_window)) return ctx->fail(SG_ERR_INVALID, "sg_plan: null arrays");
  if (b->n_windows > 0xFFFFFFFFull) return ctx->fail(SG_ERR_INVALID, "sg_plan: more than 2^32 windows in one batch");
  if (b->batch_id > 0xFFFF) return ctx->fail(SG_ERR_INVALID, "sg_plan: batch_id must fit 16 bits");
  SG_HIP(hipSetDevice(ctx->device));
  // validate the plan on the host: every operand shape the kernels assume
  uint64_t slots = 0;
  const uint64_t nw = b->n_windows;
  for (uint64_t w = 0; w < nw; w++) {
    const sg_window& x = b->windows[w];
    if (x.seg >= b->n_segs) return ctx->fail(SG_ERR_INVALID, "sg_plan: window.seg out of range");
    if (x.slot_base != slots) return ctx->fail(SG_ERR_INVALID, "sg_plan: window.slot_base is not the running prefix sum");
    if (x.len == 0) return ctx->fail(SG_ERR_INVALID, "sg_plan: empty window");
    if (b->seg_size[x.seg] == 0) return ctx->fail(SG_ERR_INVALID, "sg_plan: seg_size 0");
    uint64_t planned = x.n_reads <= 0 ? 0 : (b->paired ? ((uint64_t)x.n_reads + 1) / 2 : (uint64_t)x.n_reads);
    slots += planned;
    if (slots > 0xFFFFFFF0ull) return ctx->fail(SG_ERR_INVALID, "sg_plan: more than 2^32 fragments in one batch");
  }
  for (uint32_t s = 0; s < b->n_segs; s++)
    if (b->seg_first_window[s] > b->seg_first_window[s + 1] || b->seg_first_window[s + 1] > nw)
      return ctx->fail(SG_ERR_INVALID, "sg_plan: seg_first_window not monotone");
  if (b->n_segs && (b->seg_first_window[0] != 0 || b->seg_first_window[b->n_segs] != nw))
    return ctx->fail(SG_ERR_INVALID, "sg_plan: seg_first_window must cover all windows");
  const size_t plen = b->name_prefix ? strlen(b->name_prefix) : 0;
  if (plen == 0 || plen > 990) return ctx->fail(SG_ERR_INVALID, "sg_plan: bad name_prefix (1..990 bytes)");  // header length is a 10-bit field
  // chain bounds need the chain lengths: read them back once (tiny)
  {
    const size_t nch = (size_t)(ctx->B.chain_len - ctx->B.chain_off);
    std::vector<uint64_t> meta(2 * nch);
    if (nch) SG_HIP(hipMemcpy(meta.data(), ctx->chain_meta.p, meta.size() * 8, hipMemcpyDeviceToHost));
    for (uint64_t w = 0; w < nw; w++) {
      const sg_window& x = b->windows[w];
      if (x.chain >= nch) return ctx->fail(SG_ERR_INVALID, "sg_plan: window.chain out of range");
      if (x.hap_base + x.spos + x.len > meta[nch + x.chain]) return ctx->fail(SG_ERR_INVALID, "sg_plan: window runs past its chain");
    }
  }
  SG_ENSURE(ctx->windows, (nw + 1) * sizeof(sg_window));
  SG_ENSURE(ctx->segmeta, ((size_t)b->n_segs * 2 + 2) * 4);
  if (nw) SG_HIP(hipMemcpyAsync(ctx->windows.p, b->windows, nw * sizeof(sg_window), hipMemcpyHostToDevice, ctx->stream));
  if (b->n_segs) {
    SG_HIP(hipMemcpyAsync(ctx->segmeta.p, b->seg_size, (size_t)b->n_segs * 4, hipMemcpyHostToDevice, ctx->stream));
    SG_HIP(hipMemcpyAsync(ctx->segmeta.as<uint32_t>() + b->n_segs, b->seg_first_window, ((size_t)b->n_segs + 1) * 4,
                          hipMemcpyHostToDevice, ctx->stream));
  }
  return finish_plan(ctx, nw, b->n_segs, (uint32_t)slots, b->batch_id, b->first_window, b->first_slot, b->paired, b->name_prefix);
}

// ------------------------------------------------------------------------------------------------
// header + emit kernels of the current batch into the context's output buffers
static int launch_text(sg_ctx* ctx, bool prof) {
  sg::DevBatch& B = ctx->B;
  hipStream_t s = ctx->stream;
  B.out[0] = ctx->out1.as<uint8_t>();
  B.out[1] = ctx->out2.as<uint8_t>();
  B.out_cap[0] = ctx->out1.cap;
  B.out_cap[1] = ctx->out2.cap;
  // Queue of the items the fast emit kernel leaves to the generic code (windows with a non-ACGT base,
  // reads with >= 2 sequencing indels): room for every item of ~10 % of the reads.  A batch that
  // needs more is emitted again by the generic kernel (sg_result), so the size is not a correctness
  // matter.  SG_SLOWQ_CAP overrides it (tests force the overflow path with it).
  B.slowq = nullptr;
  B.slowq_cap = 0;
  if (sg::emit_uses_fast_kernel(ctx->P, B)) {
    uint64_t cap = std::max<uint64_t>(1u << 16, 2ull * B.n_slots);
    if (const char* e = getenv("SG_SLOWQ_CAP")) cap = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    cap = std::min<uint64_t>(cap, 1ull << 30);
    SG_ENSURE(ctx->slowq, cap * 8 * (B.paired ? 2 : 1));
    B.slowq = ctx->slowq.as<uint2>();
    B.slowq_cap = (uint32_t)cap;
  }
  if (prof) SG_HIP(hipEventRecord(ctx->evs[5], s));  // after the (first-pass) output allocation
  sg::launch_header(ctx->P, B, s);
  sg::launch_emit(ctx->P, B, s, false, prof ? ctx->evs[7] : nullptr);
  if (prof) SG_HIP(hipEventRecord(ctx->evs[6], s));
  return SG_OK;
}

static int run_pass(sg_ctx* ctx) {
  sg::DevBatch& B = ctx->B;
  B.k0 = (uint32_t)ctx->seed;
  B.k1 = (uint32_t)(ctx->seed >> 32);
  // timing ablations (outputs are wrong when set): SG_DIAG selects the generic emit kernel, SG_FDIAG keeps the straight-line one
  { const char* dg = getenv("SG_DIAG"); B.diag = dg ? (uint32_t)atoi(dg) : 0u; }
  if (const char* fd = getenv("SG_FDIAG")) B.diag = (uint32_t)atoi(fd);
  hipStream_t s = ctx->stream;
  const bool prof = ctx->profiling;
  if (!B.n_windows) SG_HIP(hipMemsetAsync(B.totals, 0, sg::kTotalsBytes, s));  // (otherwise plan_kernel clears them)
  if (prof) SG_HIP(hipEventRecord(ctx->evs[0], s));
  sg::launch_plan(ctx->P, B, s);
  if (prof) SG_HIP(hipEventRecord(ctx->evs[1], s));
  sg::launch_namebase(B, s);
  if (prof) SG_HIP(hipEventRecord(ctx->evs[2], s));
  sg::launch_indel(ctx->P, B, s);
  if (prof) SG_HIP(hipEventRecord(ctx->evs[3], s));
  sg::launch_scan(B, s);
  if (prof) SG_HIP(hipEventRecord(ctx->evs[4], s));
  SG_HIP(hipGetLastError());
  if (!ctx->out1.p && !ctx->out2.p && !ctx->gz1.p && !ctx->gz2.p && !ctx->spare.empty()) {
    // a released output set (sg_release_outputs): its buffers become this pass's
    sg_outputs* o = ctx->spare.back();
    std::swap(ctx->out1, o->text[0]); std::swap(ctx->out2, o->text[1]);
    std::swap(ctx->gz1, o->gz[0]); std::swap(ctx->gz2, o->gz[1]);
  }
  // The FASTQ size is only known now, on the device.  When the context already holds output buffers (every pass
  // but a context's first), the emit kernels are launched at once: they compare the size with the buffers' capacity
  // themselves and do nothing but raise a flag when it does not fit (finish_pass then grows the buffers and launches
  // them again).  Otherwise two u64 are read back first -- one stream sync in the middle of the pass.
  ctx->speculative = ctx->out1.p != nullptr && (!B.paired || ctx->out2.p != nullptr) && getenv("SG_NO_SPECULATION") == nullptr;
  if (!ctx->speculative) {
    sg::launch_mail(B.totals, ctx->mail, s);
    SG_HIP(hipStreamSynchronize(s));
    memcpy(ctx->host_totals, ctx->mail, 4 * 8);
    if (ctx->host_totals[3] & 1) return ctx->fail(SG_ERR_OVERFLOW, "sg_sample: a read drew more than SG_MAX_EVENTS sequencing indels");
    SG_ENSURE(ctx->out1, ctx->host_totals[0] + 64);
    if (B.paired) SG_ENSURE(ctx->out2, ctx->host_totals[1] + 64);
  }
  if (int rc = launch_text(ctx, prof)) return rc;
  sg::launch_mail(B.totals, ctx->mail, s);
  SG_HIP(hipGetLastError());
  ctx->sampled = true;
  ctx->pass_pending = true;
  ctx->results_valid = false;
  ctx->gz_valid = false;
  return SG_OK;
}

// What a queued pass left: sizes, flags; the emit kernels again if the text did not fit the buffers they were given.
static int finish_pass(sg_ctx* ctx) {
  if (!ctx->pass_pending) return SG_OK;
  SG_HIP(hipStreamSynchronize(ctx->stream));
  ctx->pass_pending = false;
  memcpy(ctx->host_totals, ctx->mail, 4 * 8);
  ctx->host_flags[0] = ctx->mail[3];
  ctx->host_flags[1] = ctx->mail[4];
  if (ctx->host_totals[3] & 1) {
    ctx->sampled = false;
    return ctx->fail(SG_ERR_OVERFLOW, "sg_sample: a read drew more than SG_MAX_EVENTS sequencing indels");
  }
  if (ctx->speculative && (ctx->host_flags[0] & 4)) {  // the buffers were too small: grow, emit again
    sg::DevBatch& B = ctx->B;
    SG_ENSURE(ctx->out1, ctx->host_totals[0] + 64);
    if (B.paired) SG_ENSURE(ctx->out2, ctx->host_totals[1] + 64);
    // what the aborted launch left behind: the flags, the slow-queue counts (a mate whose text did fit has appended its
    // items already) and the read-group counters of both emit kernels (exhausted by that mate); the record offsets' segment
    // bases and the sizes stay
    SG_HIP(hipMemsetAsync(B.totals + 3, 0, 2 * 8, ctx->stream));
    SG_HIP(hipMemsetAsync((uint8_t*)B.totals + 128, 0, sg::kTotalsSegBase - 128, ctx->stream));
    if (int rc = launch_text(ctx, ctx->profiling)) return rc;   // (the kernel times are then those of the launch that counted)
    sg::launch_mail(B.totals, ctx->mail, ctx->stream);
    SG_HIP(hipStreamSynchronize(ctx->stream));
    ctx->host_flags[0] = ctx->mail[3];
    ctx->host_flags[1] = ctx->mail[4];
  }
  return SG_OK;
}

int sg_emit_variant(sg_ctx* ctx) {
  if (!ctx || !ctx->have_profile) return -1;
  return sg::emit_variant(ctx->P);
}

int sg_emit_info(sg_ctx* ctx, uint64_t* queued_items, int* requeued) {
  if (!ctx) return SG_ERR_INVALID;
  if (!ctx->results_valid) return ctx->fail(SG_ERR_INVALID, "sg_emit_info: call sg_result first");
  if (queued_items) *queued_items = ctx->slow_items;
  if (requeued) *requeued = ctx->slow_overflow ? 1 : 0;
  return SG_OK;
}

int sg_sample(sg_ctx* ctx) {
  if (!ctx) return SG_ERR_INVALID;
  if (!ctx->have_plan) return ctx->fail(SG_ERR_INVALID, "sg_sample: call sg_plan first");
  SG_HIP(hipSetDevice(ctx->device));
  return run_pass(ctx);
}

int sg_result(sg_ctx* ctx, uint64_t* bytes_r1, uint64_t* bytes_r2, uint64_t* n_fragments) {
  if (!ctx) return SG_ERR_INVALID;
  if (!ctx->sampled) return ctx->fail(SG_ERR_INVALID, "sg_result: call sg_sample first");
  SG_HIP(hipSetDevice(ctx->device));
  if (int rc = finish_pass(ctx)) return rc;
  SG_HIP(hipStreamSynchronize(ctx->stream));
  if (!ctx->results_valid) {
    ctx->slow_items = (ctx->host_flags[1] & 0xFFFFFFFFu) + (ctx->host_flags[1] >> 32);
    ctx->slow_overflow = (ctx->host_flags[0] & 2) != 0;
    if (ctx->slow_overflow) {  // headers are in place; every item again through the generic kernel
      sg::launch_emit(ctx->P, ctx->B, ctx->stream, true, nullptr);
      SG_HIP(hipGetLastError());
      SG_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  if (ctx->profiling) {
    for (int i = 0; i < 4; i++) SG_HIP(hipEventElapsedTime(&ctx->last_ms[i], ctx->evs[i], ctx->evs[i + 1]));
    SG_HIP(hipEventElapsedTime(&ctx->last_ms[SG_K_EMIT], ctx->evs[5], ctx->evs[7]));
    SG_HIP(hipEventElapsedTime(&ctx->last_ms[SG_K_EMIT_SLOW], ctx->evs[7], ctx->evs[6]));
  }
  ctx->results_valid = true;
  if (bytes_r1) *bytes_r1 = ctx->host_totals[0];
  if (bytes_r2) *bytes_r2 = ctx->B.paired ? ctx->host_totals[1] : 0;
  if (n_fragments) *n_fragments = ctx->host_totals[2];
  return SG_OK;
}

int sg_fetch(sg_ctx* ctx, char* host_r1, char* host_r2) {
  if (!ctx) return SG_ERR_INVALID;
  if (!ctx->sampled) return ctx->fail(SG_ERR_INVALID, "sg_fetch: call sg_sample first");
  SG_HIP(hipSetDevice(ctx->device));
  if (int rc = finish_pass(ctx)) return rc;
  if (host_r1 && ctx->host_totals[0])
    SG_HIP(hipMemcpyAsync(host_r1, ctx->out1.p, ctx->host_totals[0], hipMemcpyDeviceToHost, ctx->stream));
  if (host_r2 && ctx->B.paired && ctx->host_totals[1])
    SG_HIP(hipMemcpyAsync(host_r2, ctx->out2.p, ctx->host_totals[1], hipMemcpyDeviceToHost, ctx->stream));
  SG_HIP(hipStreamSynchronize(ctx->stream));
  return SG_OK;
}

int sg_fetch_range(sg_ctx* ctx, int mate, uint64_t offset, uint64_t bytes, char* host_dst) {
  if (!ctx || (bytes && !host_dst) || mate < 0 || mate > 1) return SG_ERR_INVALID;
  if (!ctx->sampled) return ctx->fail(SG_ERR_INVALID, "sg_fetch_range: call sg_sample first");
  if (mate == 1 && !ctx->B.paired) return ctx->fail(SG_ERR_INVALID, "sg_fetch_range: single-end batch has no mate 2");
  SG_HIP(hipSetDevice(ctx->device));
  if (int rc = finish_pass(ctx)) return rc;
  if (offset + bytes > ctx->host_totals[mate]) return ctx->fail(SG_ERR_INVALID, "sg_fetch_range: range past the end of the FASTQ text");
  if (!bytes) return SG_OK;
  SG_HIP(hipSetDevice(ctx->device));
  const char* src = (const char*)(mate ? ctx->out2.p : ctx->out1.p) + offset;
  SG_HIP(hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SG_HIP(hipStreamSynchronize(ctx->stream));
  return SG_OK;
}

int sg_host_alloc(sg_ctx* ctx, uint64_t bytes, void** host_ptr) {
  if (!ctx || !host_ptr) return SG_ERR_INVALID;
  SG_HIP(hipSetDevice(ctx->device));
  const uint64_t want = bytes ? bytes : 1;
  if ((*host_ptr = host_cache().take(want)) != nullptr) return SG_OK;
  SG_HIP(hipHostMalloc(host_ptr, want, hipHostMallocDefault));
  host_cache().note(*host_ptr, want);
  return SG_OK;
}

void sg_release_cached_memory(void) {
  block_cache().trim();
  host_cache().trim();
}

int sg_host_free(sg_ctx* ctx, void* host_ptr) {
  if (!ctx) return SG_ERR_INVALID;
  if (!host_ptr) return SG_OK;
  // hipHostFree waits for the device; a buffer that goes to the cache instead may be handed out again at once, so copies
  // still reading or writing it (sg_reference_chunk's asynchronous uploads on an error path) must have ended
  SG_HIP(hipSetDevice(ctx->device));
  SG_HIP(hipStreamSynchronize(ctx->stream));
  if (!host_cache().give(host_ptr)) SG_HIP(hipHostFree(host_ptr));
  return SG_OK;
}

int sg_device_output(sg_ctx* ctx, void** dev_r1, void** dev_r2) {
  if (!ctx) return SG_ERR_INVALID;
  if (!ctx->sampled) return ctx->fail(SG_ERR_INVALID, "sg_device_output: call sg_sample first");
  SG_HIP(hipSetDevice(ctx->device));
  if (int rc = finish_pass(ctx)) return rc;   // a pass queued without its size may still move to larger buffers
  if (dev_r1) *dev_r1 = ctx->out1.p;
  if (dev_r2) *dev_r2 = ctx->B.paired ? ctx->out2.p : nullptr;
  return SG_OK;
}

int sg_gc_percent(sg_ctx* ctx, const sg_gc_window* windows, uint64_t n, int32_t* gc_out) {
  if (!ctx || (n && (!windows || !gc_out))) return SG_ERR_INVALID;
  if (!ctx->have_haps) return ctx->fail(SG_ERR_INVALID, "sg_gc_percent: call sg_upload_haplotypes first");
  if (!n) return SG_OK;
  SG_HIP(hipSetDevice(ctx->device));
  {
    const size_t nch = (size_t)(ctx->B.chain_len - ctx->B.chain_off);
    std::vector<uint64_t> meta(2 * nch);
    if (nch) SG_HIP(hipMemcpy(meta.data(), ctx->chain_meta.p, meta.size() * 8, hipMemcpyDeviceToHost));
    for (uint64_t w = 0; w < n; w++) {
      if (windows[w].chain >= nch) return ctx->fail(SG_ERR_INVALID, "sg_gc_percent: chain out of range");
      if (windows[w].start + windows[w].len > meta[nch + windows[w].chain]) return ctx->fail(SG_ERR_INVALID, "sg_gc_percent: window runs past its chain");
    }
  }
  SG_ENSURE(ctx->gcw, n * sizeof(sg_gc_window));
  SG_ENSURE(ctx->gco, n * 4);
  SG_HIP(hipMemcpyAsync(ctx->gcw.p, windows, n * sizeof(sg_gc_window), hipMemcpyHostToDevice, ctx->stream));
  sg::launch_gc(ctx->B.chains, ctx->B.chain_off, ctx->gcw.as<sg_gc_window>(), n, ctx->gco.as<int32_t>(), ctx->stream);
  SG_HIP(hipGetLastError());
  SG_HIP(hipMemcpyAsync(gc_out, ctx->gco.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  SG_HIP(hipStreamSynchronize(ctx->stream));
  return SG_OK;
}

// ------------------------------------------------------------------------------------------------
// sampling plan made on the device
// ------------------------------------------------------------------------------------------------
static int check_gens(sg_ctx* ctx, const sg_window_gen* gens, uint64_t n_gens, uint32_t n_segs, uint32_t frag, std::vector<uint64_t>& prefix,
                      const char* who) {
  const size_t nch = (size_t)(ctx->B.chain_len - ctx->B.chain_off);
  std::vector<uint64_t> meta(2 * nch);
  if (nch) SG_HIP(hipMemcpy(meta.data(), ctx->chain_meta.p, meta.size() * 8, hipMemcpyDeviceToHost));
  prefix.assign(n_gens + 1, 0);
  for (uint64_t g = 0; g < n_gens; g++) {
    const sg_window_gen& G = gens[g];
    if (G.chain >= nch || G.hap_len == 0 || G.hap_base + G.hap_len > meta[nch + G.chain])
      return ctx->fail(SG_ERR_INVALID, std::string(who) + ": generator " + std::to_string(g) + " does not lie inside its chain");
    if (G.seg >= n_segs || (g && G.seg < gens[g - 1].seg)) return ctx->fail(SG_ERR_INVALID, std::string(who) + ": generators must be ordered by segment");
    prefix[g + 1] = prefix[g] + (G.hap_len + frag - 1) / frag;
  }
  return SG_OK;
}

int sg_windows_build(sg_ctx* ctx, uint32_t store_id, const sg_window_gen* gens, uint64_t n_gens, uint32_t n_segs, const sg_gc_model* model,
                     double* seg_weight_out, uint64_t* n_windows_out) {
  if (!ctx || !model || !model->means || !model->quantiles || (n_gens && !gens) || (n_segs && !seg_weight_out)) return SG_ERR_INVALID;
  if (model->lg_cells < 1 || model->lg_cells > 20 || model->frag_size == 0) return ctx->fail(SG_ERR_INVALID, "sg_windows_build: bad model");
  if (!ctx->have_haps) return ctx->fail(SG_ERR_INVALID, "sg_windows_build: call sg_upload_haplotypes / sg_build_haplotypes first");
  SG_HIP(hipSetDevice(ctx->device));
  std::vector<uint64_t> prefix;
  if (int rc = check_gens(ctx, gens, n_gens, n_segs, model->frag_size, prefix, "sg_windows_build")) return rc;
  const uint64_t n = prefix[n_gens];
  std::vector<uint64_t> seg_first((size_t)n_segs + 1, n);
  {
    uint32_t k = 0;
    for (uint64_t g = 0; g < n_gens; g++)
      for (; k <= gens[g].seg; k++) seg_first[k] = prefix[g];
  }
  if (n_windows_out) *n_windows_out = n;
  for (uint32_t k = 0; k < n_segs; k++) seg_weight_out[k] = 0.0;
  DevBuf& store = ctx->wstore[store_id];
  ctx->wstore_n[store_id] = n;
  if (!n) return SG_OK;
  SG_ENSURE(store, n * 8);
  // work: gens | prefix | seg_first | windows | seg_ord | win_ord | gc | seg sums
  const size_t cells = (size_t)1 << model->lg_cells;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 63) & ~(size_t)63; return o; };
  const size_t o_gens = take(n_gens * sizeof(sg_window_gen)), o_pre = take((n_gens + 1) * 8), o_sf = take(((size_t)n_segs + 1) * 8);
  const size_t o_win = take(n * sizeof(sg_gc_window)), o_so = take(n * 4), o_wo = take(n * 4), o_gc = take(n * 4), o_ss = take((size_t)n_segs * 8);
  SG_ENSURE(ctx->wwork, off);
  SG_ENSURE(ctx->gcm, (101 + cells + 1) * 8);
  uint8_t* wk = ctx->wwork.as<uint8_t>();
  hipStream_t s = ctx->stream;
  SG_HIP(hipMemcpyAsync(wk + o_gens, gens, n_gens * sizeof(sg_window_gen), hipMemcpyHostToDevice, s));
  SG_HIP(hipMemcpyAsync(wk + o_pre, prefix.data(), (n_gens + 1) * 8, hipMemcpyHostToDevice, s));
  SG_HIP(hipMemcpyAsync(wk + o_sf, seg_first.data(), ((size_t)n_segs + 1) * 8, hipMemcpyHostToDevice, s));
  SG_HIP(hipMemcpyAsync(ctx->gcm.p, model->means, 101 * 8, hipMemcpyHostToDevice, s));
  SG_HIP(hipMemcpyAsync(ctx->gcm.as<double>() + 101, model->quantiles, (cells + 1) * 8, hipMemcpyHostToDevice, s));
  sg::launch_tile((const sg_window_gen*)(wk + o_gens), (const uint64_t*)(wk + o_pre), (uint32_t)n_gens, n, model->frag_size,
                  (const uint64_t*)(wk + o_sf), (sg_gc_window*)(wk + o_win), (uint32_t*)(wk + o_so), (uint32_t*)(wk + o_wo), s);
  sg::launch_gc(ctx->B.chains, ctx->B.chain_off, (const sg_gc_window*)(wk + o_win), n, (int32_t*)(wk + o_gc), s);
  sg::launch_gc_weight((const int32_t*)(wk + o_gc), (const sg_gc_window*)(wk + o_win), (const uint32_t*)(wk + o_so), (const uint32_t*)(wk + o_wo), n,
                       ctx->gcm.as<double>(), model->std, ctx->gcm.as<double>() + 101, model->lg_cells, model->frag_size,
                       model->full_tile_form, model->ctx24, ctx->seed, store.as<double>(), s);
  sg::launch_seg_sum(store.as<double>(), (const uint64_t*)(wk + o_sf), n_segs, (double*)(wk + o_ss), s);
  SG_HIP(hipGetLastError());
  SG_HIP(hipMemcpyAsync(seg_weight_out, wk + o_ss, (size_t)n_segs * 8, hipMemcpyDeviceToHost, s));
  SG_HIP(hipStreamSynchronize(s));
  return SG_OK;
}

void sg_windows_drop(sg_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->wstore) kv.second.release();
  ctx->wstore.clear();
  ctx->wstore_n.clear();
}

int sg_plan_windows(sg_ctx* ctx, uint32_t store_id, const sg_window_gen* gens, uint64_t n_gens, const sg_active_seg* active, uint32_t n_active,
                    uint32_t frag_size, uint32_t batch_id, int32_t paired, const char* name_prefix, uint64_t* slots_out,
                    uint64_t* n_windows_out) {
  if (!ctx || (n_gens && !gens) || (n_active && (!active || !slots_out)) || frag_size == 0) return SG_ERR_INVALID;
  if (!ctx->have_profile) return ctx->fail(SG_ERR_INVALID, "sg_plan_windows: call sg_load_profile first");
  if (!ctx->have_haps) return ctx->fail(SG_ERR_INVALID, "sg_plan_windows: no haplotypes on the device");
  if (batch_id > 0xFFFF) return ctx->fail(SG_ERR_INVALID, "sg_plan_windows: batch_id must fit 16 bits");
  const size_t plen = name_prefix ? strlen(name_prefix) : 0;
  if (plen == 0 || plen > 990) return ctx->fail(SG_ERR_INVALID, "sg_plan_windows: bad name_prefix (1..990 bytes)");
  auto it = ctx->wstore.find(store_id);
  if (it == ctx->wstore.end()) return ctx->fail(SG_ERR_INVALID, "sg_plan_windows: no window weights under this store id (sg_windows_build)");
  SG_HIP(hipSetDevice(ctx->device));
  std::vector<uint64_t> prefix;
  if (int rc = check_gens(ctx, gens, n_gens, n_active, frag_size, prefix, "sg_plan_windows")) return rc;
  const uint64_t n = prefix[n_gens], n_store = ctx->wstore_n[store_id];
  if (n > 0xFFFFFFF0ull) return ctx->fail(SG_ERR_INVALID, "sg_plan_windows: more than 2^32 windows in one batch");
  for (uint64_t g = 0; g < n_gens; g++)
    if (gens[g].first_window + (prefix[g + 1] - prefix[g]) > n_store)
      return ctx->fail(SG_ERR_INVALID, "sg_plan_windows: generator " + std::to_string(g) + " points past the stored weights");
  sg_ctx::PlanInfo& pi = ctx->winfo;
  pi = sg_ctx::PlanInfo();
  pi.seg_first.assign((size_t)n_active + 1, (uint32_t)n);
  {
    uint32_t k = 0;
    for (uint64_t g = 0; g < n_gens; g++)
      for (; k <= gens[g].seg; k++) pi.seg_first[k] = (uint32_t)prefix[g];
  }
  for (uint32_t a = 0; a < n_active; a++) {
    if (active[a].seg_size == 0) return ctx->fail(SG_ERR_INVALID, "sg_plan_windows: seg_size 0");
    if (pi.seg_first[a] == pi.seg_first[a + 1]) return ctx->fail(SG_ERR_INVALID, "sg_plan_windows: active segment without windows");
    pi.seg_size.push_back(active[a].seg_size);
  }
  if (n_windows_out) *n_windows_out = n;
  pi.n_windows = n; pi.n_active = n_active; pi.batch_id = batch_id; pi.paired = paired ? 1 : 0; pi.prefix = name_prefix;
  pi.slot_first.assign((size_t)n_active + 1, 0);
  if (n) {
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 63) & ~(size_t)63; return o; };
    const size_t o_gens = take(n_gens * sizeof(sg_window_gen)), o_pre = take((n_gens + 1) * 8), o_act = take((size_t)n_active * sizeof(sg_active_seg));
    const size_t o_sf = take(((size_t)n_active + 1) * 4), o_sum = take((size_t)n_active * 8), o_pl = take(n * 4), o_off = take(n * 8);
    const size_t o_bs = take(((size_t)sg::scan_blocks((uint32_t)n) + 8) * 8), o_tot = take(8), o_ss = take(((size_t)n_active + 1) * 8);
    SG_ENSURE(ctx->wwork, off);
    SG_ENSURE(ctx->wplan, n * sizeof(sg_window));
    uint8_t* wk = ctx->wwork.as<uint8_t>();
    hipStream_t s = ctx->stream;
    SG_HIP(hipMemcpyAsync(wk + o_gens, gens, n_gens * sizeof(sg_window_gen), hipMemcpyHostToDevice, s));
    SG_HIP(hipMemcpyAsync(wk + o_pre, prefix.data(), (n_gens + 1) * 8, hipMemcpyHostToDevice, s));
    SG_HIP(hipMemcpyAsync(wk + o_act, active, (size_t)n_active * sizeof(sg_active_seg), hipMemcpyHostToDevice, s));
    SG_HIP(hipMemcpyAsync(wk + o_sf, pi.seg_first.data(), ((size_t)n_active + 1) * 4, hipMemcpyHostToDevice, s));
    SG_HIP(hipMemsetAsync(wk + o_sum, 0, (size_t)n_active * 8, s));
    sg::launch_window_reads((const sg_window_gen*)(wk + o_gens), (const uint64_t*)(wk + o_pre), (uint32_t)n_gens, n, frag_size, it->second.as<double>(),
                            (const sg_active_seg*)(wk + o_act), (const uint32_t*)(wk + o_sf), n_active, ctx->wplan.as<sg_window>(),
                            (unsigned long long*)(wk + o_sum), paired, (uint32_t*)(wk + o_pl), s);
    sg::launch_scan_u32((const uint32_t*)(wk + o_pl), (uint32_t)n, (uint64_t*)(wk + o_bs), (uint64_t*)(wk + o_off), (uint64_t*)(wk + o_tot), s);
    sg::launch_slot_base(ctx->wplan.as<sg_window>(), n, (const uint64_t*)(wk + o_off), (const uint32_t*)(wk + o_sf), n_active,
                         (const uint64_t*)(wk + o_tot), (uint64_t*)(wk + o_ss), s);
    SG_HIP(hipGetLastError());
    SG_HIP(hipMemcpyAsync(pi.slot_first.data(), wk + o_ss, ((size_t)n_active + 1) * 8, hipMemcpyDeviceToHost, s));
    SG_HIP(hipStreamSynchronize(s));
    if (pi.slot_first[n_active] > 0xFFFFFFF0ull) return ctx->fail(SG_ERR_INVALID, "sg_plan_windows: more than 2^32 fragments in one batch");
  }
  for (uint32_t a = 0; a < n_active; a++) slots_out[a] = pi.slot_first[a + 1] - pi.slot_first[a];
  pi.valid = true;
  ctx->have_plan = false;
  return SG_OK;
}

int sg_plan_range(sg_ctx* ctx, uint32_t a0, uint32_t a1) {
  if (!ctx) return SG_ERR_INVALID;
  sg_ctx::PlanInfo& pi = ctx->winfo;
  if (!pi.valid) return ctx->fail(SG_ERR_INVALID, "sg_plan_range: call sg_plan_windows first");
  if (a0 >= a1 || a1 > pi.n_active) return ctx->fail(SG_ERR_INVALID, "sg_plan_range: empty or out-of-range run of segments");
  SG_HIP(hipSetDevice(ctx->device));
  const uint32_t w_lo = pi.seg_first[a0], w_hi = pi.seg_first[a1], n_segs = a1 - a0;
  const uint64_t nw = (uint64_t)w_hi - w_lo;
  const uint64_t slot_lo = pi.slot_first[a0], slots = pi.slot_first[a1] - slot_lo;
  SG_ENSURE(ctx->windows, (nw + 1) * sizeof(sg_window));
  SG_ENSURE(ctx->segmeta, ((size_t)n_segs * 2 + 2) * 4);
  std::vector<uint32_t> segmeta;
  for (uint32_t a = a0; a < a1; a++) segmeta.push_back(pi.seg_size[a]);
  for (uint32_t a = a0; a <= a1; a++) segmeta.push_back(pi.seg_first[a] - w_lo);
  SG_HIP(hipMemcpyAsync(ctx->segmeta.p, segmeta.data(), segmeta.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  sg::launch_slice(ctx->wplan.as<sg_window>(), w_lo, nw, a0, (uint32_t)slot_lo, ctx->windows.as<sg_window>(), ctx->stream);
  SG_HIP(hipGetLastError());
  return finish_plan(ctx, nw, n_segs, (uint32_t)slots, pi.batch_id, w_lo, (uint32_t)slot_lo, pi.paired, pi.prefix.c_str());
}

int sg_window_weights(sg_ctx* ctx, const sg_gc_window* windows, const uint32_t* seg_ord, const uint32_t* win_ord, uint64_t n,
                      const sg_gc_model* model, double* weights_out, int32_t* gc_out) {
  if (!ctx || !model || !model->means || !model->quantiles || (n && (!windows || !seg_ord || !win_ord))) return SG_ERR_INVALID;
  if (model->lg_cells < 1 || model->lg_cells > 20 || model->frag_size == 0) return ctx->fail(SG_ERR_INVALID, "sg_window_weights: bad model");
  if (!ctx->have_haps) return ctx->fail(SG_ERR_INVALID, "sg_window_weights: call sg_upload_haplotypes first");
  if (!n) return SG_OK;
  SG_HIP(hipSetDevice(ctx->device));
  {
    const size_t nch = (size_t)(ctx->B.chain_len - ctx->B.chain_off);
    std::vector<uint64_t> meta(2 * nch);
    if (nch) SG_HIP(hipMemcpy(meta.data(), ctx->chain_meta.p, meta.size() * 8, hipMemcpyDeviceToHost));
    for (uint64_t w = 0; w < n; w++) {
      if (windows[w].chain >= nch) return ctx->fail(SG_ERR_INVALID, "sg_window_weights: chain out of range");
      if (windows[w].start + windows[w].len > meta[nch + windows[w].chain]) return ctx->fail(SG_ERR_INVALID, "sg_window_weights: window runs past its chain");
    }
  }
  // device work buffer: windows | seg_ord | win_ord | gc | weights | means[101] + quantile knots
  const size_t cells = (size_t)1 << model->lg_cells;
  const size_t o_seg = n * sizeof(sg_gc_window), o_win = o_seg + n * 4, o_gc = o_win + n * 4, o_wt = (o_gc + n * 4 + 7) & ~(size_t)7;
  SG_ENSURE(ctx->gcw, o_wt + n * 8);
  SG_ENSURE(ctx->gcm, (101 + cells + 1) * 8);
  uint8_t* wk = ctx->gcw.as<uint8_t>();
  hipStream_t s = ctx->stream;
  SG_HIP(hipMemcpyAsync(wk, windows, n * sizeof(sg_gc_window), hipMemcpyHostToDevice, s));
  SG_HIP(hipMemcpyAsync(wk + o_seg, seg_ord, n * 4, hipMemcpyHostToDevice, s));
  SG_HIP(hipMemcpyAsync(wk + o_win, win_ord, n * 4, hipMemcpyHostToDevice, s));
  SG_HIP(hipMemcpyAsync(ctx->gcm.p, model->means, 101 * 8, hipMemcpyHostToDevice, s));
  SG_HIP(hipMemcpyAsync(ctx->gcm.as<double>() + 101, model->quantiles, (cells + 1) * 8, hipMemcpyHostToDevice, s));
  sg::launch_gc(ctx->B.chains, ctx->B.chain_off, (const sg_gc_window*)wk, n, (int32_t*)(wk + o_gc), s);
  sg::launch_gc_weight((const int32_t*)(wk + o_gc), (const sg_gc_window*)wk, (const uint32_t*)(wk + o_seg), (const uint32_t*)(wk + o_win), n,
                       ctx->gcm.as<double>(), model->std, ctx->gcm.as<double>() + 101, model->lg_cells, model->frag_size,
                       model->full_tile_form, model->ctx24, ctx->seed, (double*)(wk + o_wt), s);
  SG_HIP(hipGetLastError());
  if (weights_out) SG_HIP(hipMemcpyAsync(weights_out, wk + o_wt, n * 8, hipMemcpyDeviceToHost, s));
  if (gc_out) SG_HIP(hipMemcpyAsync(gc_out, wk + o_gc, n * 4, hipMemcpyDeviceToHost, s));
  SG_HIP(hipStreamSynchronize(s));
  return SG_OK;
}

}  // extern "C"
