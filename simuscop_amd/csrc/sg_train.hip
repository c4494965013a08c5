// sg_train.hip -- profile training on the device: what Profile::train (lib/profile/Profile.cpp:1442-1484)
// gathers from lines of `samtools view` text, chunk of lines by chunk of lines, in file order.
//
//   train_lines_*         line breaks of the chunk -> offsets (a scan over 64-byte blocks)
//   train_fields_kernel   lane = line: the eleven mandatory fields and the filters of Profile::processRead (:244-279)
//   gate scan             the lines that reach Profile::countGC (:281-287), compacted in file order
//   state scan            Profile::countGC (:512-703) without its sequential loop.  Its state -- the window reads are being
//                         counted in -- only ever moves forward inside a run of reads of one contig, and is a function of
//                         the LARGEST position the run has shown so far: the grid window holding it (whole genome), or the
//                         first target that ends at or behind it (exome; proof in DESIGN.md section 8).  A segmented
//                         running maximum therefore gives every read the state it finds, and with it the verdict the
//                         reference reaches: turned away (behind the window), counted in it, or opening the next one
//   window scan           windows opened so far -> the read's window; read counts by atomics (one per thread and window)
//   train_verdict_kernel  lane = line: will the read be counted (no hard clip, a single nM, inside its contig)?  Then a prefix
//                         count of those reads finds the line at which Profile::processRead's cap ends the run (:497-507)
//   train_effects_kernel  lane = line up to that one: the CIGAR walk of :290-382 (insertion / deletion length counts unless
//                         the VCF knows the event), the counters of lines turned away
//   train_count_lds_kernel  workgroup = a group of bins x a slice of the reads, sixteen lanes = read: subsDist1 / subsDist2 /
//                         kmersDist (:399-442), qualityDist (:453-481) counted in LDS, added to memory once; iSizeDist
//                         (:444-451).  (train_count_kernel: the same straight to memory, for tables that do not fit LDS)
//   train_window_gc_kernel (at the end) wave = window: calculateGCContent (lib/mydefine/MyDefine.cpp:306-331)
// Integer work throughout; results are exact counts.  The checker is oracle/train_oracle.cpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "sg_train.h"

namespace sg {

namespace {

constexpr uint32_t kScanThreads = 256, kScanItems = 8, kScanTile = kScanThreads * kScanItems;

// ---- a device-wide scan in three launches: tile aggregates, the spine (one workgroup, carries the value of the chunk
// before in and the total out), the tiles again with their prefixes.  Op: T, identity(), n(), load(i), combine(a, b)
// (associative, a before b), store(i, exclusive, inclusive), finish(), carry_in(), carry_out(total).
template <class T, class Op>
__device__ __forceinline__ T block_inclusive(T v, T* lds, Op& op) {
  const uint32_t t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (uint32_t off = 1; off < kScanThreads; off <<= 1) {
    T x = lds[t];
    if (t >= off) x = op.combine(lds[t - off], x);
    __syncthreads();
    lds[t] = x;
    __syncthreads();
  }
  return lds[t];
}

template <class Op>
__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(Op op, typename Op::T* tile_sum) {
  using T = typename Op::T;
  __shared__ T lds[kScanThreads];
  const uint64_t n = op.n();
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  T acc = op.identity();
  for (uint32_t k = 0; k < kScanItems; k++)
    if (base + k < n) acc = op.combine(acc, op.load(base + k));
  const T inc = block_inclusive(acc, lds, op);
  if (threadIdx.x == kScanThreads - 1) tile_sum[blockIdx.x] = inc;
}

template <class Op>
__global__ __launch_bounds__(kScanThreads) void scan_spine_kernel(Op op, typename Op::T* tile_sum, uint32_t n_tiles) {
  using T = typename Op::T;
  __shared__ T lds[kScanThreads];
  T carry = op.carry_in();
  for (uint32_t b = 0; b < n_tiles; b += kScanThreads) {
    const uint32_t i = b + threadIdx.x;
    const T v = i < n_tiles ? tile_sum[i] : op.identity();
    const T inc = block_inclusive(v, lds, op);
    const T before = threadIdx.x ? lds[threadIdx.x - 1] : op.identity();
    const T last = lds[kScanThreads - 1];
    __syncthreads();
    if (i < n_tiles) tile_sum[i] = op.combine(carry, before);   // exclusive prefix of tile i
    carry = op.combine(carry, last);
    (void)inc;
  }
  if (threadIdx.x == 0) op.carry_out(carry);
}

template <class Op>
__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(Op op, const typename Op::T* tile_excl) {
  using T = typename Op::T;
  __shared__ T lds[kScanThreads];
  const uint64_t n = op.n();
  const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
  T acc = op.identity();
  for (uint32_t k = 0; k < kScanItems; k++)
    if (base + k < n) acc = op.combine(acc, op.load(base + k));
  block_inclusive(acc, lds, op);
  T run = op.combine(tile_excl[blockIdx.x], threadIdx.x ? lds[threadIdx.x - 1] : op.identity());
  for (uint32_t k = 0; k < kScanItems; k++) {
    if (base + k >= n) break;
    const T inc = op.combine(run, op.load(base + k));
    op.store(base + k, run, inc);
    run = inc;
  }
  op.finish();
}

template <class Op>
void run_scan_count(Op op, uint64_t n_bound, void* work, hipStream_t s) {   // tile aggregates and the spine (total -> carry_out)
  using T = typename Op::T;
  const uint32_t n_tiles = (uint32_t)((n_bound + kScanTile - 1) / kScanTile);
  T* tiles = (T*)work;
  if (n_tiles) hipLaunchKernelGGL(scan_reduce_kernel<Op>, dim3(n_tiles), dim3(kScanThreads), 0, s, op, tiles);
  hipLaunchKernelGGL(scan_spine_kernel<Op>, dim3(1), dim3(kScanThreads), 0, s, op, tiles, n_tiles);
}
template <class Op>
void run_scan_apply(Op op, uint64_t n_bound, void* work, hipStream_t s) {
  using T = typename Op::T;
  const uint32_t n_tiles = (uint32_t)((n_bound + kScanTile - 1) / kScanTile);
  if (n_tiles) hipLaunchKernelGGL(scan_apply_kernel<Op>, dim3(n_tiles), dim3(kScanThreads), 0, s, op, (const T*)work);
}
template <class Op>
void run_scan(Op op, uint64_t n_bound, void* work, hipStream_t s) {
  run_scan_count(op, n_bound, work, s);
  run_scan_apply(op, n_bound, work, s);
}

// ---- line breaks: element = 64 bytes of text, value = its line breaks ----
struct LineOp {
  using T = uint64_t;
  TrainJob J;
  __device__ T identity() const { return 0; }
  __device__ uint64_t n() const { return (J.bytes + 63) / 64; }
  __device__ T combine(T a, T b) const { return a + b; }
  __device__ T load(uint64_t i) const {
    const uint64_t a = i * 64, e = a + 64 < J.bytes ? a + 64 : J.bytes;
    uint32_t c = 0;
    if (e - a == 64) {   // (the text buffer is 16-byte aligned)
      const uint4* p = (const uint4*)(J.text + a);
      for (int q = 0; q < 4; q++) {
        const uint4 v = p[q];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (int k = 0; k < 4; k++) {
          const uint32_t x = w[k] ^ 0x0A0A0A0Au;                       // zero bytes where a line break stands
          const uint32_t z = ~((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) | 0x7F7F7F7Fu);   // 0x80 in exactly those bytes
          c += (uint32_t)__popc(z);
        }
      }
    } else {
      for (uint64_t b = a; b < e; b++) c += J.text[b] == '\n';
    }
    return c;
  }
  __device__ void store(uint64_t i, T excl, T) const {
    const uint64_t a = i * 64, e = a + 64 < J.bytes ? a + 64 : J.bytes;
    uint64_t k = excl;
    for (uint64_t b = a; b < e; b++)
      if (J.text[b] == '\n') J.line_end[k++] = b;
  }
  __device__ void finish() const {}
  __device__ T carry_in() const { return 0; }
  __device__ void carry_out(T total) const { J.carry_out->n_lines = total; J.carry_out->cut_line = ~0ull; }
};

// ---- the reads countGC sees, compacted in file order ----
struct GateOp {
  using T = uint64_t;
  TrainJob J;
  __device__ T identity() const { return 0; }
  __device__ uint64_t n() const { return J.n_lines; }
  __device__ T combine(T a, T b) const { return a + b; }
  __device__ T load(uint64_t i) const { return J.reads[i].flags & 1u; }
  __device__ void store(uint64_t i, T excl, T) const {
    const TrainRead R = J.reads[i];
    if (R.flags & 1u) J.gate[excl] = TrainGate{R.pos0, R.contig, (uint32_t)i};
  }
  __device__ void finish() const {}
  __device__ T carry_in() const { return 0; }
  __device__ void carry_out(T total) const { J.carry_out->n_gated = total; }
};

// ---- countGC's state: segmented running maximum of the positions (a run = consecutive reads of one contig), and the
// least contig length met so far (winSize shrinks to it for good, :645-648) ----
struct StateVal { int64_t max_pos, ref_min; uint32_t head, pad; };

__device__ __forceinline__ uint64_t first_at_or_above(const int64_t* a, uint64_t n, int64_t p) {   // first k with a[k] >= p (a ascending)
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (a[mid] >= p) hi = mid; else lo = mid + 1;
  }
  return lo;
}

struct StateOp {
  using T = StateVal;
  TrainJob J;
  __device__ T identity() const { return StateVal{INT64_MIN, INT64_MAX, 0u, 0u}; }
  __device__ uint64_t n() const { return J.carry_out->n_gated; }
  __device__ T combine(const T& a, const T& b) const {
    T r;
    r.head = a.head | b.head;
    r.max_pos = b.head ? b.max_pos : (a.max_pos > b.max_pos ? a.max_pos : b.max_pos);
    r.ref_min = a.ref_min < b.ref_min ? a.ref_min : b.ref_min;
    r.pad = 0;
    return r;
  }
  __device__ bool head_of(uint64_t d, uint32_t contig) const {
    if (d) return J.gate[d - 1].contig != contig;
    return !J.carry_in->has || J.carry_in->last_contig != contig;
  }
  __device__ T load(uint64_t d) const {
    const TrainGate g = J.gate[d];
    return StateVal{g.pos0, (int64_t)J.contigs[g.contig].length, head_of(d, g.contig) ? 1u : 0u, 0u};
  }
  // the window countGC is in once the largest position of the run is m
  __device__ void window_of(const TrainContig& C, int64_t m, int64_t ws, int64_t* left, int64_t* right, bool* none) const {
    *none = false;
    if (!J.wes) {                                        // :572-577, :650-657
      int64_t r = (m / ws + 1) * ws - 1;
      if (r > (int64_t)C.length - 1) r = (int64_t)C.length - 1;
      *right = r; *left = r - ws + 1;
    } else {                                             // :582-611, :662-690
      const uint64_t t = first_at_or_above(J.tgt_pmax + C.tgt_first, C.tgt_n, m);
      if (t < C.tgt_n) { *left = J.tgt_left[C.tgt_first + t]; *right = J.tgt_right[C.tgt_first + t]; }
      else { *left = *right = (int64_t)C.length; *none = true; }
    }
  }
  __device__ void store(uint64_t d, const T& excl, const T& inc) const {
    const TrainGate g = J.gate[d];
    const TrainContig C = J.contigs[g.contig];
    int64_t ws = (int64_t)J.window;
    if (inc.ref_min < ws) ws = inc.ref_min;
    const bool head = head_of(d, g.contig);
    TrainStep S;
    S.opens = 0; S.counted = 0; S.left = S.right = 0; S.ws = (uint32_t)ws; S.pad = 0;
    int64_t l, r;
    bool none;
    bool open_now = head;
    if (!head) {
      window_of(C, excl.max_pos, ws, &l, &r, &none);
      if (g.pos0 < l) { J.steps[d] = S; return; }        // :552-554
      if (g.pos0 <= r) { S.counted = 1; S.left = l; S.right = r; J.steps[d] = S; J.reads[g.line].flags |= 8u; return; }   // :555-558
      open_now = true;
    }
    if (open_now) {
      window_of(C, g.pos0, ws, &l, &r, &none);
      S.opens = 1; S.left = l; S.right = r;
      S.counted = J.wes ? ((!none && l <= g.pos0) ? 1u : 0u) : 1u;   // :579, :594-599, :659, :673-678
      if (none) S.pad = 1;
    }
    J.steps[d] = S;
    if (S.counted) J.reads[g.line].flags |= 8u;
  }
  __device__ void finish() const {}
  __device__ T carry_in() const {
    return StateVal{J.carry_in->has ? J.carry_in->max_pos : INT64_MIN, J.carry_in->has ? J.carry_in->ref_min : INT64_MAX, 0u, 0u};
  }
  __device__ void carry_out(const T& total) const {   // (the gate list is complete by now: this spine runs after the gate scan)
    J.carry_out->max_pos = total.max_pos;
    J.carry_out->ref_min = total.ref_min;
    const uint64_t n = J.carry_out->n_gated;
    if (n) { J.carry_out->last_contig = J.gate[n - 1].contig; J.carry_out->has = 1u; }
    else { J.carry_out->last_contig = J.carry_in->last_contig; J.carry_out->has = J.carry_in->has; }
  }
};

// ---- windows opened so far -> the window of every read; read counts ----
struct WindowOp {
  using T = uint64_t;
  TrainJob J;
  uint64_t cur = ~0ull;   // (per thread: the window its last reads were counted in)
  uint32_t cnt = 0;
  __device__ T identity() const { return 0; }
  __device__ uint64_t n() const { return J.carry_out->n_gated; }
  __device__ T combine(T a, T b) const { return a + b; }
  __device__ bool seen(uint64_t d) const { return (uint64_t)J.gate[d].line <= J.carry_out->cut_line; }   // (behind the cap: never read)
  __device__ T load(uint64_t d) const { return seen(d) ? J.steps[d].opens : 0u; }
  __device__ void store(uint64_t d, T, T inc) {
    if (!seen(d)) return;
    const TrainStep S = J.steps[d];
    const uint64_t id = inc - 1;     // (carry_in = the windows of the chunks before; a first read that opens nothing rides in
                                     //  their last window)
    if (S.opens) J.windows[id] = TrainWindow{S.left, S.right, J.gate[d].contig, S.pad ? 0u : S.ws};
    if (S.counted) {
      if (id != cur) { finish(); cur = id; }
      cnt++;
    }
  }
  __device__ void finish() {
    if (cnt) atomicAdd(J.window_rc + cur, cnt);
    cnt = 0;
  }
  __device__ T carry_in() const { return J.carry_in->n_windows; }
  __device__ void carry_out(T total) const { J.carry_out->n_windows = total; }
};

// ---- Profile::processRead's cap: the run ends with the read that makes readCount reach maxCount (:497-507, :1461-1464);
// nothing behind that line is ever read.  Prefix count of the reads that will be counted, over the lines of the chunk ----
struct CutOp {
  using T = uint64_t;
  TrainJob J;
  uint64_t limit;
  __device__ T identity() const { return 0; }
  __device__ uint64_t n() const { return J.n_lines; }
  __device__ T combine(T a, T b) const { return a + b; }
  __device__ T load(uint64_t i) const { return (J.reads[i].flags >> 4) & 1u; }
  __device__ void store(uint64_t i, T excl, T inc) const {
    if (inc == limit && excl + 1 == inc) J.carry_out->cut_line = i;
  }
  __device__ void finish() const {}
  __device__ T carry_in() const { return J.carry_in->reads_total; }
  __device__ void carry_out(T total) const { J.carry_out->reads_total = total; }
};

__device__ __forceinline__ bool is_digit(char c) { return c >= '0' && c <= '9'; }

// atoi / atol on a field [p, e): optional sign, digits (fields hold no white space)
__device__ long long field_int(const char* p, const char* e) {
  bool neg = false;
  if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
  long long v = 0;
  while (p < e && is_digit(*p)) { v = v * 10 + (*p - '0'); p++; }
  return neg ? -v : v;
}

// abbrOfChr (lib/mydefine/MyDefine.cpp:212-225): what follows the first "chrom", else the first "chr", else the name
__device__ void abbr_of_chr(const char*& p, const char* e) {
  for (int pass = 0; pass < 2; pass++) {
    const int n = pass == 0 ? 5 : 3;
    for (const char* q = p; q + n <= e; q++) {
      bool hit = q[0] == 'c' && q[1] == 'h' && q[2] == 'r';
      if (pass == 0) hit = hit && q[3] == 'o' && q[4] == 'm';
      if (hit) { p = q + n; return; }
    }
  }
}

__global__ __launch_bounds__(256) void train_fields_kernel(TrainJob J) {
  const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= J.n_lines) return;
  TrainRead R;
  R.len = 0; R.flags = 0; R.tlen = 0; R.seq_off = R.qual_off = R.ref_off = R.cigar_off = 0; R.cigar_len = 0; R.pos0 = 0; R.contig = 0; R.pad = 0;
  const char* p = J.text + (li ? J.line_end[li - 1] + 1 : 0);
  const char* le = J.text + J.line_end[li];  // the line break
  auto done = [&]() { J.reads[li] = R; };
  if (p == le) { R.flags = 128u; done(); return; }   // an empty line: processRead returns at once (:229-231)
  // ---- the first eleven fields (:244-251) ----
  const char* fb[11];
  const char* fe[11];
  int nf = 0;
  const char* a = p;
  for (const char* q = p; q <= le && nf < 11; q++) {
    if (q == le || *q == '\t') {
      fb[nf] = a; fe[nf] = q; nf++;
      a = q + 1;
    }
  }
  if (nf < 11) { R.flags = 256u; done(); return; }   // (an error, once the line is known to lie before the cap: train_effects_kernel)
  const long long position = field_int(fb[3], fe[3]);
  const int mapq = (int)field_int(fb[4], fe[4]);
  const int tlen = (int)field_int(fb[8], fe[8]);
  if (position == 0) { done(); return; }                 // :263
  if (mapq < 15) { done(); return; }                     // :267
  const char* cb = fb[2];
  abbr_of_chr(cb, fe[2]);
  int contig = -1;
  for (uint32_t c = 0; c < J.n_contigs; c++) {           // :271-275
    const char* key = J.keys + (size_t)c * kTrainKeyBytes;
    uint32_t i = 0;
    while (cb + i < fe[2] && key[i] != 0 && key[i] == cb[i]) i++;
    if (cb + i == fe[2] && key[i] == 0) { contig = (int)c; break; }
  }
  if (contig < 0) { done(); return; }
  const uint32_t slen = (uint32_t)(fe[9] - fb[9]);
  if (slen == 1 && fb[9][0] == '*') { done(); return; }  // :277
  const TrainContig C = J.contigs[contig];
  R.len = slen;
  R.tlen = tlen;
  R.seq_off = (uint64_t)(fb[9] - J.text);
  R.qual_off = (uint64_t)(fb[10] - J.text);
  R.cigar_off = (uint64_t)(fb[5] - J.text);
  R.cigar_len = (uint32_t)(fe[5] - fb[5]);
  R.pos0 = position - 1;
  R.contig = (uint32_t)contig;
  R.flags = (tlen < 0 ? 2u : 0u) | ((uint32_t)(fe[10] - fb[10]) == slen ? 4u : 0u);
  if (J.count_gc) {
    if (C.xym) { R.flags |= 32u; done(); return; }   // :532-535
    // a read that starts behind its contig's end makes the reference throw (std::string::substr, Genome.cpp:435); one on an
    // empty contig never counts: both stay out of countGC's sight here
    if (position - 1 < 0 || (uint64_t)(position - 1) >= C.length) { R.flags |= 64u; done(); return; }
    R.flags |= 1u;
  } else {
    R.flags |= 1u | 8u;   // (sg_train_count: every read through the filters is counted)
  }
  done();
}

// is (pos, len) among the known events the reference's loop reaches (Profile.cpp:313-321, :343-351)?  The loop walks the
// contig's list in file order and stops at the first position above pos.
__device__ bool known_event(const TrainKnown& K, uint64_t first, uint32_t n, int64_t pos, int32_t len) {
  if (!n) return false;
  uint64_t lo = 0, hi = n;                                // first file-order index whose running maximum exceeds pos
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (K.pmax[first + mid] > pos) hi = mid; else lo = mid + 1;
  }
  const uint64_t stop = lo;
  lo = 0; hi = n;                                         // first row >= (pos, len)
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    const int64_t p = K.pos[first + mid];
    const int32_t l = K.len[first + mid];
    if (p > pos || (p == pos && l >= len)) hi = mid; else lo = mid + 1;
  }
  if (lo >= n || K.pos[first + lo] != pos || K.len[first + lo] != len) return false;
  return K.first[first + lo] < stop;
}

// Will the read reach Profile::processRead's `readCount++` (:483)?  The CIGAR walk without its counters: no hard clip, a
// single nM, inside its contig.  (No side effect: which lines exist at all is only known once the cap has been applied.)
__global__ __launch_bounds__(256) void train_verdict_kernel(TrainJob J) {
  const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= J.n_lines) return;
  const TrainRead R = J.reads[li];
  if (!(R.flags & 8u)) return;
  const char* cg = J.text + R.cigar_off;
  const int n_c = (int)R.cigar_len;
  int k = 0;
  for (int i = 0; i < n_c; i++) {
    const char c = cg[i];
    if (is_digit(c)) { k++; continue; }
    if (c == 'H') return;
  }
  if (n_c == 0 || k != n_c - 1 || cg[n_c - 1] != 'M') return;
  const TrainContig C = J.contigs[R.contig];
  if ((uint64_t)R.pos0 + R.len > C.length) return;
  J.reads[li].ref_off = C.code_off + (uint64_t)R.pos0;
  J.reads[li].flags = R.flags | 16u;
}

// Everything a line adds besides the count matrices, for the lines up to the cap: the counters of empty / turned-away /
// overhanging lines, the error flag of a line with fewer than eleven fields, the CIGAR walk of :290-382.
__global__ __launch_bounds__(256) void train_effects_kernel(TrainJob J) {
  const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= J.n_lines || li > J.carry_out->cut_line) return;
  const TrainRead R = J.reads[li];
  if (R.flags & 128u) { atomicAdd(J.scalars + kTrainEmptyLines, 1ull); return; }
  if (R.flags & 256u) { atomicOr(J.flags, 1u); return; }
  if (R.flags & 64u) { atomicAdd(J.scalars + kTrainOverhang, 1ull); return; }
  if ((R.flags & 32u) || ((R.flags & 1u) && !(R.flags & 8u))) { atomicAdd(J.scalars + kTrainGcRejected, 1ull); return; }   // :283-285
  if (!(R.flags & 8u)) return;
  const TrainContig C = J.contigs[R.contig];
  // ---- CIGAR (:290-382) ----
  const char* cg = J.text + R.cigar_off;
  const int n_c = (int)R.cigar_len;
  atomicAdd(J.scalars + kTrainCigarChars, (unsigned long long)n_c);   // `baseCount += n`, n = strlen(cigar) (:294)
  int sIndx = 0, k = 0;
  long long refIndx = 0;
  const long long position = R.pos0 + 1;
  for (int i = 0; i < n_c; i++) {
    const char c = cg[i];
    if (is_digit(c)) { k++; continue; }
    if (c == 'H') { atomicAdd(J.scalars + kTrainCigarChars, (unsigned long long)(-(long long)n_c)); return; }   // :300-303
    if (c == 'S') sIndx = i + 1;
    else if (c == 'I' || c == 'D') {                     // :307-367
      const long long len = field_int(cg + sIndx, cg + i);
      const bool ins = c == 'I';
      const long long pos = ins ? position + refIndx - 1 : position + refIndx;
      const bool found = ins ? known_event(J.known_ins, C.ins_first, C.ins_n, pos, (int32_t)len)
                             : known_event(J.known_del, C.del_first, C.del_n, pos, (int32_t)len);
      if (!found) {
        unsigned long long* row = ins ? J.ins_len : J.del_len;
        if (len >= 0 && len < (long long)J.n_indel_len) atomicAdd(row + len, 1ull);
        else atomicAdd(J.scalars + kTrainIndelLenOverflow, 1ull);
        atomicAdd(J.scalars + (ins ? kTrainInsEvents : kTrainDelEvents), 1ull);
      }
      if (!ins) refIndx += len;
      sIndx = i + 1;
    } else if (c == 'M') {
      refIndx += field_int(cg + sIndx, cg + i);
      sIndx = i + 1;
    } else sIndx = i + 1;
  }
  if (n_c == 0 || k != n_c - 1 || cg[n_c - 1] != 'M') return;   // :380-382
  // the reference indexes refSeq past its end when the read hangs over its contig (:458 with n = strlen(readSeq)): skipped
  if ((uint64_t)R.pos0 + R.len > C.length) atomicAdd(J.scalars + kTrainOverhang, 1ull);
}

// ---- the count matrices ----
// What one base of a counted read adds (Profile.cpp:399-481), shared by the two kernels below; `Sink` says where the three
// counters live.
template <class Sink>
__device__ __forceinline__ void count_base(const TrainJob& J, const TrainRead& R, uint32_t i, Sink& sink) {
  const uint32_t K = J.kmer, bins = J.bins;
  const bool has_alt = J.alt_codes != J.ref_codes;
  const bool rev = (R.flags & 2u) != 0u;               // tlen < 0: everything reverse-complemented, mate 2 (:388-397)
  const uint32_t n = R.len;
  const char* seq = J.text + R.seq_off;
  const char* qual = J.text + R.qual_off;
  const uint8_t* ref = J.ref_codes + R.ref_off;
  const uint8_t* alt = J.alt_codes + R.ref_off;
  // index in `bases` of the base the k-mer context holds at read position q (:404-415: the alternative allele where the
  // read shows it, else the reference base), and whether the read shows the alternative allele there
  auto context_idx = [&](uint32_t q, bool* is_alt) -> int {
    const uint32_t w = rev ? n - 1u - q : q;
    uint32_t code = ref[w];
    *is_alt = false;
    if (has_alt) {
      const uint32_t ac = alt[w];
      if (ac != code) {
        const char c = seq[w];
        bool eq;
        if (ac <= 3u) eq = c == "ACTG"[ac];
        else eq = rev ? !(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'a' || c == 'c' || c == 'g' || c == 't')   // (both complement to 'N')
                      : (ac == 4u && c == 'N');
        if (eq) { code = ac; *is_alt = true; }
      }
    }
    if (code > 3u) return -1;
    if (rev) code ^= 2u;
    return (int)((J.remap >> (2u * code)) & 3u);
  };
  const uint32_t j = rev ? n - 1u - i : i;           // index of read position i in the line's strings / the reference window
  // read base -> index in `bases` (getIndexOfBase, MyDefine.cpp:228-236), complemented first on the reverse strand
  // (Segment::getComplementSeq keeps the case, so lower-case bases stay unknown)
  char c = seq[j];
  if (rev) c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'N';
  const int b = c == J.bases[0] ? 0 : c == J.bases[1] ? 1 : c == J.bases[2] ? 2 : c == J.bases[3] ? 3 : -1;
  const uint32_t bin = (uint32_t)(((uint64_t)i * bins) / n);
  bool shows_alt;
  const int s0 = context_idx(i, &shows_alt);
  if (b >= 0) {                                      // :416-442
    const uint32_t m = i + 1u < K ? i + 1u : K;       // real bases of the context, the rest is 'X'
    int kidx = (int)J.kmer_off[m];
    int v = 0;
    for (uint32_t t = 0; t < m; t++) {                // oldest base in the highest digit
      bool dummy;
      const int x = t + 1u == m ? s0 : context_idx(i + 1u - m + t, &dummy);
      if (x < 0) { kidx = -1; break; }
      v = v * 4 + x;
    }
    if (kidx >= 0) sink.sub(rev, (uint32_t)(kidx + v), bin, (uint32_t)b);
  }
  if ((R.flags & 4u) && b >= 0) {                    // :457-480
    uint32_t rc = ref[j];                            // refSeq[i] itself must be a base (:460,463) ...
    int r0 = -1;
    if (rc <= 3u) { if (rev) rc ^= 2u; r0 = (int)((J.remap >> (2u * rc)) & 3u); }
    if (r0 >= 0) {
      if (shows_alt) r0 = s0;                        // ... and gives way to the alternative allele the read shows (:466-468)
      const int q = (int)(signed char)qual[j];
      if (r0 >= 0 && q >= 33 && q <= 126) sink.qual((uint32_t)(r0 * 4 + b), bin, (uint32_t)(q - 33));
    }
  }
}

__device__ __forceinline__ void count_read_scalars(const TrainJob& J, const TrainRead& R) {
  if (R.tlen > 0) {                                  // :446-451
    if ((uint32_t)R.tlen < J.n_isize) atomicAdd(J.isize + R.tlen, 1ull);
    else atomicAdd(J.scalars + kTrainIsizeOverflow, 1ull);
  }
  atomicAdd(J.scalars + kTrainReads, 1ull);          // :483
}

// Straight to the tables in memory: wave = read, lane = base, 64-bit atomics.  For context lengths whose tables do not fit
// LDS (kmer 6); three atomics per base on a few thousand hot cells -- 13 G atomics/s, 5.6 ms per 64 MB of text.
struct GlobalSink {
  const TrainJob& J;
  __device__ void sub(bool rev, uint32_t kidx, uint32_t bin, uint32_t b) {
    atomicAdd((rev ? J.subs2 : J.subs1) + ((size_t)kidx * J.bins + bin) * 4u + b, 1ull);
    atomicAdd(J.kmers + (size_t)bin * J.kmer_count + kidx, 1ull);
  }
  __device__ void qual(uint32_t pair, uint32_t bin, uint32_t q) { atomicAdd(J.quality + ((size_t)pair * J.bins + bin) * 94u + q, 1ull); }
};
__global__ __launch_bounds__(256) void train_count_kernel(TrainJob J) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  GlobalSink sink{J};
  for (uint64_t li = wave; li < J.n_lines; li += n_waves) {
    const TrainRead R = J.reads[li];
    if (!(R.flags & 16u) || li > J.carry_out->cut_line) continue;
    for (uint32_t i = lane; i < R.len; i += 64u) count_base(J, R, i, sink);
    if (lane == 0u) count_read_scalars(J, R);
  }
}

// The tables of a GROUP OF BINS in LDS: a workgroup counts the bases of `bpg` bins of its slice of the chunk's reads with
// 32-bit LDS atomics and adds what it has to the tables in memory once, at its end.  bin = i * bins / n, so a group of bins
// is a run of read positions: sixteen lanes take a read and walk that run (15 to 18 bases of a 151-base read with six of
// fifty bins).  LDS: bpg x (2 mates x kc x 4 + kc + 16 x 94) counters.
struct LdsSink {
  uint32_t* subs;    // [2][kc][bpg][4]
  uint32_t* kmers;   // [bpg][kc]
  uint32_t* quals;   // [16][bpg][94]
  uint32_t kc, bpg, b0;
  __device__ void sub(bool rev, uint32_t kidx, uint32_t bin, uint32_t b) {
    const uint32_t lb = bin - b0;
    atomicAdd(subs + (((rev ? kc : 0u) + kidx) * bpg + lb) * 4u + b, 1u);
    atomicAdd(kmers + lb * kc + kidx, 1u);
  }
  __device__ void qual(uint32_t pair, uint32_t bin, uint32_t q) { atomicAdd(quals + (pair * bpg + (bin - b0)) * 94u + q, 1u); }
};
__global__ __launch_bounds__(256) void train_count_lds_kernel(TrainJob J, uint32_t bpg, uint32_t n_groups, uint32_t reads_per_wg) {
  extern __shared__ uint32_t lds[];
  const uint32_t kc = J.kmer_count, bins = J.bins;
  const uint32_t n_sub = 2u * kc * bpg * 4u, n_km = bpg * kc, n_q = 16u * bpg * 94u, n_all = n_sub + n_km + n_q;
  for (uint32_t i = threadIdx.x; i < n_all; i += blockDim.x) lds[i] = 0u;
  __syncthreads();
  const uint32_t g = blockIdx.x % n_groups;
  const uint64_t r0 = (uint64_t)(blockIdx.x / n_groups) * reads_per_wg;
  const uint64_t r1 = r0 + reads_per_wg < J.n_lines ? r0 + reads_per_wg : J.n_lines;
  const uint32_t b0 = g * bpg, b1 = b0 + bpg < bins ? b0 + bpg : bins;
  LdsSink sink{lds, lds + n_sub, lds + n_sub + n_km, kc, bpg, b0};
  const uint32_t sub = threadIdx.x >> 4, l16 = threadIdx.x & 15u;   // sixteen groups of sixteen lanes
  for (uint64_t li = r0 + sub; li < r1; li += 16u) {
    const TrainRead R = J.reads[li];
    if (!(R.flags & 16u) || li > J.carry_out->cut_line) continue;
    const uint64_t n = R.len;
    // positions whose bin lies in [b0, b1): i * bins / n >= b0  <=>  i >= ceil(b0 n / bins)
    const uint32_t i_lo = (uint32_t)(((uint64_t)b0 * n + bins - 1u) / bins), i_hi = (uint32_t)(((uint64_t)b1 * n + bins - 1u) / bins);
    for (uint32_t i = i_lo + l16; i < i_hi && i < R.len; i += 16u) count_base(J, R, i, sink);
    if (g == 0u && l16 == 0u) count_read_scalars(J, R);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_all; i += blockDim.x) {
    const uint32_t v = lds[i];
    if (!v) continue;
    if (i < n_sub) {
      const uint32_t b = i & 3u, lb = (i >> 2) % bpg, k2 = (i >> 2) / bpg;   // k2 = mate * kc + kidx
      const bool rev = k2 >= kc;
      atomicAdd((rev ? J.subs2 : J.subs1) + ((size_t)(rev ? k2 - kc : k2) * bins + b0 + lb) * 4u + b, (unsigned long long)v);
    } else if (i < n_sub + n_km) {
      const uint32_t e = i - n_sub, lb = e / kc, kidx = e - lb * kc;
      atomicAdd(J.kmers + (size_t)(b0 + lb) * kc + kidx, (unsigned long long)v);
    } else {
      const uint32_t e = i - n_sub - n_km, q = e % 94u, lb = (e / 94u) % bpg, pair = e / (94u * bpg);
      atomicAdd(J.quality + ((size_t)pair * bins + b0 + lb) * 94u + q, (unsigned long long)v);
    }
  }
}

// wave = window: G/C and N bytes of refSequence[left .. right] (calculateGCContent); exome read counts scaled to the
// window size (:565-566)
__global__ __launch_bounds__(256) void train_window_gc_kernel(const TrainWindow* __restrict__ wins, const uint32_t* __restrict__ rc, uint64_t n,
                                                              const TrainContig* __restrict__ contigs, const uint8_t* __restrict__ ref,
                                                              uint32_t wes, double* __restrict__ gc, double* __restrict__ rcs) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t w = wave; w < n; w += n_waves) {
    const TrainWindow W = wins[w];
    double g = -1.0;
    if (W.ws) {                                          // (0: behind the last target -- refSeq = NULL, GC = -1, :606-611)
      const TrainContig C = contigs[W.contig];
      // Genome::getSubRefSequence(chr, left, length): std::string::substr cuts at the contig's end
      const int64_t len_asked = wes ? W.right - W.left + 1 : (int64_t)W.ws;
      int64_t len = len_asked;
      if (W.left + len > (int64_t)C.length) len = (int64_t)C.length - W.left;
      uint32_t gcn = 0, nn = 0;
      for (int64_t i = lane; i < len; i += 64) {
        const uint32_t c = ref[C.code_off + (uint64_t)(W.left + i)];
        gcn += (c == 1u) | (c == 3u);
        nn += c == 4u;
      }
      for (int o = 32; o; o >>= 1) { gcn += __shfl_xor(gcn, o); nn += __shfl_xor(nn, o); }
      if (len <= 0) g = 0.0;
      else if (nn) g = -1.0;
      else g = 1.0 * (double)gcn / (double)(len - nn);
    }
    if (lane == 0u) {
      gc[w] = g;
      uint32_t r = rc[w];
      if (wes && W.ws) {
        const int32_t target = (int32_t)(W.right - W.left + 1);
        r = (uint32_t)((int32_t)(W.ws * r / (uint32_t)target));   // `rc = winSize*rc/targetSize`: unsigned arithmetic, stored in an int
      }
      rcs[w] = (double)(int32_t)r;
    }
  }
}

__global__ __launch_bounds__(256) void train_patch_kernel(uint8_t* codes, const uint64_t* off, const uint8_t* ch, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t b = ch[i];
  if (b >= 'a' && b <= 'z') b -= 32u;   // Genome.cpp:529-530
  const bool acgt = (b == 'A') | (b == 'C') | (b == 'G') | (b == 'T');
  codes[off[i]] = (uint8_t)(acgt ? ((b >> 1) & 3u) : (b == 'N' ? 4u : (b == 'X' ? 6u : 5u)));
}

}  // namespace

size_t train_scan_work_bytes(uint64_t n_elems) { return ((n_elems + kScanTile - 1) / kScanTile + 1) * sizeof(StateVal); }

void launch_train_lines_count(const TrainJob& J, hipStream_t s) { run_scan_count(LineOp{J}, (J.bytes + 63) / 64, J.scan_work, s); }
void launch_train_lines_fill(const TrainJob& J, hipStream_t s) { run_scan_apply(LineOp{J}, (J.bytes + 63) / 64, J.scan_work, s); }

void launch_train_chunk(const TrainJob& J, hipStream_t s) {
  if (!J.n_lines) return;
  const dim3 per_line((uint32_t)((J.n_lines + 255) / 256));
  hipLaunchKernelGGL(train_fields_kernel, per_line, dim3(256), 0, s, J);
  if (J.count_gc) {
    run_scan(GateOp{J}, J.n_lines, J.scan_work, s);
    run_scan(StateOp{J}, J.n_lines, J.scan_work, s);
  }
  hipLaunchKernelGGL(train_verdict_kernel, per_line, dim3(256), 0, s, J);
  {
    // (always run: the carry's read count is what the host watches; without a cap no line can reach the limit)
    const uint64_t limit = J.max_reads ? (J.wes ? 2 * J.max_reads : J.max_reads) : ~0ull;
    run_scan(CutOp{J, limit}, J.n_lines, J.scan_work, s);
  }
  if (J.count_gc) run_scan(WindowOp{J}, J.n_lines, J.scan_work, s);
  hipLaunchKernelGGL(train_effects_kernel, per_line, dim3(256), 0, s, J);
  // the count tables of as many bins as fit 60 KB of LDS; none do for kmer 6
  const uint32_t per_bin = (9u * J.kmer_count + 16u * 94u) * 4u;
  const uint32_t bpg = std::min<uint32_t>(J.bins, (60u << 10) / per_bin);
  static const bool no_lds = getenv("SG_TRAIN_GLOBAL_ATOMICS") != nullptr;   // (tests: the two kernels count alike)
  if (bpg >= 1u && !no_lds) {
    const uint32_t n_groups = (J.bins + bpg - 1u) / bpg;
    // ~8 workgroups per CU, at least 256 reads each
    const uint64_t rpw = std::max<uint64_t>(256, (J.n_lines * n_groups + 2047) / 2048);
    const uint32_t slices = (uint32_t)((J.n_lines + rpw - 1) / rpw);
    hipLaunchKernelGGL(train_count_lds_kernel, dim3(slices * n_groups), dim3(256), bpg * per_bin, s, J, bpg, n_groups, (uint32_t)rpw);
    return;
  }
  const uint64_t waves = J.n_lines;
  const uint32_t grid = (uint32_t)(waves * 64 / 256 + 1 < 256u * 32u ? waves * 64 / 256 + 1 : 256u * 32u);
  hipLaunchKernelGGL(train_count_kernel, dim3(grid), dim3(256), 0, s, J);
}

void launch_train_window_gc(const TrainWindow* w, const uint32_t* rc, uint64_t n, const TrainContig* contigs, const uint8_t* ref_codes,
                            uint32_t wes, double* gc, double* rcs, hipStream_t s) {
  if (!n) return;
  const uint32_t grid = (uint32_t)(n / 4 + 1 < 256u * 16u ? n / 4 + 1 : 256u * 16u);
  hipLaunchKernelGGL(train_window_gc_kernel, dim3(grid), dim3(256), 0, s, w, rc, n, contigs, ref_codes, wes, gc, rcs);
}

void launch_train_patch(uint8_t* codes, const uint64_t* off, const uint8_t* ch, uint64_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(train_patch_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, codes, off, ch, n);
}

}  // namespace sg
