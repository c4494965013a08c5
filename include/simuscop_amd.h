/* simuscop_amd.h -- C ABI of the MI355X read-sampling engine (libsimuscop_amd.so).
 *
 * The reference (SimuSCoP v1.0) has no FFI/plugin interface; its only seam on this path is the C
 * work item handed to the thread pool,
 *     threadPool->pool_add_work(&Segment::yieldReads, &chrSegs[k], n++)   lib/genome/Genome.cpp:881,949
 *     void* (*)(const void* arg)                                          lib/threadpool/ThreadPool.h:195
 * executed once per <=1 Mbp segment between generateSegSequences() (Genome.cpp:876-878) and
 * threadPool->wait() (:883).  This library replaces that whole per-chromosome block -- every
 * Segment::yieldReads call of one (population, chromosome), including Profile::predict,
 * Profile::yieldInsertSize, Segment::getFragSequence / Genome::produceFragment and the FASTQ
 * formatting -- by one batched GPU pass.  What crosses the boundary is exactly the state those
 * functions read: the Profile CDF tables, the haplotype strings, and the per-window sampling plan.
 * INTEGRATION.md shows the binding a SimuSCoP maintainer would add.
 *
 * Plain C, no torch / HIP types in signatures.  One sg_ctx per GPU; a ctx is thread-compatible
 * (use it from one host thread at a time).  All functions return SG_OK (0) or an error code;
 * sg_last_error() gives the message (the reference prints to cerr and exit(1)s instead,
 * e.g. lib/config/Config.cpp:67-70 -- the CLI layer maps a non-zero status to that behaviour).
 */
#ifndef SIMUSCOP_AMD_H
#define SIMUSCOP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_OK 0
#define SG_ERR_INVALID 1     /* bad argument / call order            */
#define SG_ERR_HIP 2         /* HIP runtime error (no GPU, OOM, ...) */
#define SG_ERR_UNSUPPORTED 3 /* profile shape outside the kernels' range (bases != 4, kmer > 6 ...) */
#define SG_ERR_OVERFLOW 4    /* more than SG_MAX_EVENTS sequencing indels in one read             */
#define SG_ERR_FORMAT 5      /* FASTA without fixed-width lines (sg_reference_commit): use the host parser */

#define SG_MAX_EVENTS 32

typedef struct sg_ctx sg_ctx;

/* ---- lifetime ---------------------------------------------------------------------------- */
/* Replaces `threadPool = new ThreadPool(threads); pool_init()` (src/simuReads.cpp:63-64,
 * lib/threadpool/ThreadPool.cpp:8-51): one context per GPU instead of N pinned pthreads, and one
 * 64-bit seed instead of clock-seeded mt19937 pairs (ThreadPool.cpp:41-49).                      */
int sg_create(sg_ctx** ctx, int device, uint64_t seed);
void sg_destroy(sg_ctx* ctx);
/* Device blocks a destroyed context (or a grown buffer) gave up stay with the process and serve the next context's
 * requests of their size (the reference keeps its `new[]` arenas per run likewise: nothing of it outlives main());
 * this returns them to the runtime.  SG_BLOCK_CACHE_GB bounds what is kept (default 160, 0 = keep nothing).        */
void sg_release_cached_memory(void);
const char* sg_last_error(const sg_ctx* ctx); /* ctx may be NULL: error of a failed sg_create */
/* Run on a caller-owned HIP stream (hipStream_t passed as void*); NULL = the ctx's own stream. */
int sg_set_stream(sg_ctx* ctx, void* hip_stream);
int sg_set_seed(sg_ctx* ctx, uint64_t seed);
/* A literal 'X' in the genome.  The reference's k-mer trie files the short contexts of a read's first bases under the
 * character 'X' ("XXb", "Xbb": Profile::initKmers, lib/profile/Profile.cpp:94-101) and Profile::getKmerIndx (:220-226)
 * walks it with whatever the template holds, so an X of the genome acts as that place holder in the middle of a read
 * (on forward reads: Segment::getComplementSeq turns it into 'N', lib/segment/Segment.cpp:99).  Default 0: the same here.
 * 1 (`simuReads --strict-bases`): an X is an unknown base like any other non-ACGT character.                          */
int sg_set_strict_bases(sg_ctx* ctx, int on);

/* ---- profile tables ---------------------------------------------------------------------- */
/* The products of Profile::train(file) (lib/profile/Profile.cpp:1436-1440) exactly as the
 * reference holds them in memory: fp64 cumulative tables (Matrix<double>::getEntrance() rows,
 * Profile.cpp:1367-1434).  The library converts them to exact u32 thresholds of the reference's
 * 32-bit uniform (lib/threadpool/ThreadPool.cpp:203-207, lib/mydefine/MyDefine.cpp:176-184).   */
typedef struct sg_profile_cdf {
  int32_t n_bases;          /* config "bases" length, must be 4        Profile.cpp:1000 */
  char bases[8];            /* e.g. "ACTG"                                               */
  int32_t kmer;             /* Profile.cpp:1001                                           */
  int32_t bins;             /* Profile.cpp:1002 (already clipped to read_length, :185)    */
  int32_t read_length;      /* Profile.cpp:1003                                           */
  int32_t n_qual;           /* maxBaseQuality-minBaseQuality+1 = 94, Profile.cpp:208      */
  int32_t min_qual;         /* 33, Profile.cpp:173                                        */
  double insert_rate;       /* Profile::insertRate                                        */
  double del_rate;          /* Profile::delRate                                           */
  const double* ins_cdf;    /* insCdf  [n_ins]                         Profile.cpp:1375   */
  int32_t n_ins;
  const double* del_cdf;    /* delCdf  [n_del]                         Profile.cpp:1378   */
  int32_t n_del;
  const double* subs_cdf1;  /* subsCdf1[kmer_count][bins][n_bases]     Profile.cpp:1419   */
  const double* subs_cdf2;  /* subsCdf2 or NULL (SE / stdISize<=0)     Profile.cpp:1420-1430 */
  const double* qual_cdf;   /* qualityCdf[n_bases*n_bases][bins][n_qual] Profile.cpp:1398 */
  const double* isize_cdf;  /* iSizeCdf[n_isize] or NULL -> fixed insert size, Profile.cpp:1487 */
  int32_t n_isize;
  int32_t isize_min;        /* iSizeAlphabet[0] (alphabet is consecutive, Profile.cpp:919-922) */
  int32_t insert_size;      /* config insertSize (used when isize_cdf == NULL)            */
} sg_profile_cdf;
int sg_load_profile(sg_ctx* ctx, const sg_profile_cdf* prof);
/* The same in two steps, so that a host can convert the tables (milliseconds of integer arithmetic: draw counts of every
 * CDF entry, alias columns) on a worker thread while it streams the reference to the device: sg_profile_prepare touches no
 * device and no context; sg_load_prepared_profile uploads.  sg_profile_prepare always returns a tables object (free it);
 * on failure its code is also what sg_load_prepared_profile returns, with the message in sg_last_error. */
typedef struct sg_profile_tables sg_profile_tables;
int sg_profile_prepare(const sg_profile_cdf* prof, sg_profile_tables** out);
const char* sg_profile_tables_error(const sg_profile_tables* tables);
int sg_load_prepared_profile(sg_ctx* ctx, const sg_profile_tables* tables);
void sg_profile_tables_free(sg_profile_tables* tables);

/* ---- haplotypes of one (population, chromosome) ------------------------------------------- */
/* chain h = concatenation over the chromosome's segments, in order, of Segment::segSequences[h]
 * (lib/segment/Segment.cpp:448-458; NULL entries contribute nothing).  A fragment that runs past
 * its segment continues into the following segments' same-index haplotype
 * (Segment::getFragSequence :1085-1101 -> Genome::produceFragment, Genome.cpp:599-632): on a chain
 * that is a plain substring, clipped at the chain end.  Bytes are upper-case ASCII.             */
int sg_upload_haplotypes(sg_ctx* ctx, int32_t n_chains, const char* const* chains, const uint64_t* lens);

/* ---- reference ingest and haplotype assembly on the device ------------------ */
/* Replaces the reference's per-segment FASTA reads (lib/fastahack/Fasta.cpp:304-334 through
 * Segment.cpp:137) and its std::string haplotype editing (Segment::generateSegSequences,
 * Segment.cpp:124-460) for callers that do not hold haplotype strings themselves:
 *   sg_reference_begin / _chunk   stream the FASTA file, as it is on disk, to the device;
 *   sg_reference_scan             offsets of the header lines ('>' at a line start), unordered;
 *   sg_reference_commit           contigs -> resident base codes (newlines dropped by index
 *                                 arithmetic, upper-cased as Segment.cpp:143); SG_ERR_FORMAT when a
 *                                 contig's lines are not of one width (fall back to host parsing
 *                                 and pass line_bases = line_width = length);
 *   sg_build_haplotypes           chains of one (population, chromosome) = copies of reference
 *                                 ranges and literal bases, then single-base patches; same device
 *                                 state afterwards as sg_upload_haplotypes.                        */
typedef struct sg_contig {
  uint64_t raw_offset;  /* file offset of the contig's first base                              */
  uint64_t length;      /* bases                                                               */
  uint32_t line_bases;  /* bases per line (.fai LINEBASES)                                     */
  uint32_t line_width;  /* bytes per line incl. the line break (.fai LINEWIDTH)                */
} sg_contig;
int sg_reference_begin(sg_ctx* ctx, uint64_t raw_bytes);
/* asynchronous on the ctx's stream: `host` must stay unchanged until sg_sync() returns */
int sg_reference_chunk(sg_ctx* ctx, uint64_t offset, const void* host, uint64_t bytes);
int sg_sync(sg_ctx* ctx);
/* bit 0 of *flags: the file has ';' comment lines (not handled by sg_reference_commit) */
int sg_reference_scan(sg_ctx* ctx, uint64_t* header_offsets, uint32_t cap, uint32_t* n_found, uint32_t* flags);
int sg_reference_commit(sg_ctx* ctx, const sg_contig* contigs, uint32_t n_contigs);

typedef struct sg_hap_piece {
  uint64_t dst;     /* offset in chain `chain`                                                   */
  uint64_t src;     /* kind 0: 0-based position on contig `contig`; kind 1: offset into literals */
  uint32_t len;
  uint32_t chain;
  uint32_t contig;
  uint32_t kind;
} sg_hap_piece;
typedef struct sg_hap_patch {
  uint64_t dst;     /* offset in chain `chain` */
  uint32_t chain;
  uint32_t base;    /* ASCII, any case         */
} sg_hap_patch;
/* Pieces must tile every chain exactly (no byte of [0, lens[c]) left out); patches apply afterwards. */
int sg_build_haplotypes(sg_ctx* ctx, int32_t n_chains, const uint64_t* lens, const sg_hap_piece* pieces, uint64_t n_pieces,
                        const char* literals, uint64_t n_literal_bytes, const sg_hap_patch* patches, uint64_t n_patches);
/* Diagnostic read-back of the resident chains (either upload route): base codes A0 C1 T2 G3, 'N' = 4,
 * any other character = 5.                                                                        */
int sg_haplotype_codes(sg_ctx* ctx, uint32_t chain, uint64_t offset, uint64_t n, uint8_t* codes_out);

/* ---- sampling plan ------------------------------------------------------------------------ */
/* One entry per sampling window of every processed segment (Segment.cpp:675 skips segments with no
 * sequences or readCount == 0), i.e. Segment::fragStartPos / fragEndPos / hapIndxs / fragRCs
 * (Segment.h:45-49) flattened in segment order, window order.                                   */
typedef struct sg_window {
  uint64_t hap_base;  /* offset of this segment's haplotype string inside chain `chain`        */
  uint32_t chain;     /* hapIndxs[i]                                                            */
  uint32_t spos;      /* fragStartPos[i]                                                        */
  uint32_t len;       /* fragEndPos[i]-fragStartPos[i]+1                                        */
  int32_t n_reads;    /* fragRCs[i]  (PE: ceil(n/2) pairs are produced, Segment.cpp:848)        */
  uint32_t seg;       /* ordinal of the segment inside this batch                               */
  uint32_t slot_base; /* exclusive prefix sum of planned pairs over the batch's windows         */
} sg_window;

typedef struct sg_batch {
  uint32_t batch_id;          /* running index of (mixture, population, chromosome) batches      */
  int32_t paired;             /* Config::isPairedEnd()                                            */
  const char* name_prefix;    /* "@<popu>#<chr>#"  (Segment.cpp:780,809)                          */
  const sg_window* windows;
  uint64_t n_windows;
  const uint32_t* seg_size;   /* per segment: seqSize/CN (Segment.cpp:713-714), for pos%segsize   */
  const uint32_t* seg_first_window; /* per segment: index of its first window; [n_segs] = n_windows */
  uint32_t n_segs;
  /* Sharding: when a batch is split over several GPUs by runs of segments, each shard passes the
   * index of its first window / first fragment slot inside the WHOLE batch, so that every draw keeps
   * the address it has in the unsharded run (the shards' FASTQ concatenate to the 1-GPU output).
   * windows[].slot_base stays shard-local (starts at 0).  Both 0 for an unsharded batch.           */
  uint32_t first_window;
  uint32_t first_slot;
} sg_batch;
int sg_plan(sg_ctx* ctx, const sg_batch* batch);

/* ---- run ----------------------------------------------------------------------------------- */
/* Enqueue the whole pass (plan draws, indel pass, record offsets, base/quality sampling + FASTQ
 * formatting) for the planned batch.  Results stay in device memory.  A context that already holds
 * output buffers (any pass but its first) queues the pass without waiting for anything: what the
 * pass reports -- SG_ERR_OVERFLOW for a read with more than SG_MAX_EVENTS sequencing indels -- then
 * comes from sg_result / sg_fetch instead.                                                       */
int sg_sample(sg_ctx* ctx);
/* Wait for the pass and report sizes: FASTQ bytes for mate 1 / mate 2 (0 for SE) and the number
 * of fragments actually produced (pairs for PE, reads for SE).                                   */
int sg_result(sg_ctx* ctx, uint64_t* bytes_r1, uint64_t* bytes_r2, uint64_t* n_fragments);
/* Copy the FASTQ text to host buffers (the bytes SeqWriter::write(char*,char*) would receive,
 * lib/seqwriter/SeqWriter.cpp:41-54).  host_r2 may be NULL for SE.                               */
int sg_fetch(sg_ctx* ctx, char* host_r1, char* host_r2);
/* Partial copy for pipelined sinks: bytes [offset, offset+bytes) of mate 0/1's FASTQ text.  With a
 * destination from sg_host_alloc (pinned) the copy runs at PCIe speed.                            */
int sg_fetch_range(sg_ctx* ctx, int mate, uint64_t offset, uint64_t bytes, char* host_dst);
/* Page-locked host memory for sg_fetch / sg_fetch_range destinations.                             */
int sg_host_alloc(sg_ctx* ctx, uint64_t bytes, void** host_ptr);
int sg_host_free(sg_ctx* ctx, void* host_ptr);
/* Device pointers of the FASTQ text (valid until the next sg_sample / sg_plan).                  */
int sg_device_output(sg_ctx* ctx, void** dev_r1, void** dev_r2);

/* ---- GC-window scan (Segment::getWeightedLength byte scan, Segment.cpp:563-592) ------------- */
/* For every window [start, start+len) of chain `chain`: integer GC% = 100*GC/len, or -1 if the
 * window holds any 'N' (lib/mydefine/MyDefine.cpp:279-303).  Uses the uploaded haplotypes.       */
typedef struct sg_gc_window { uint64_t start; uint32_t chain; uint32_t len; } sg_gc_window;
int sg_gc_percent(sg_ctx* ctx, const sg_gc_window* windows, uint64_t n, int32_t* gc_out);

/* GC-bias weight of sampling windows, on the device: Segment::getWeightedLength's per-window loop
 * (lib/segment/Segment.cpp:550-641) -- GC% of the window (calculateGCPercent, lib/mydefine/MyDefine.cpp:279-303),
 * GC factor = Normal(means[gc], std) redrawn while negative (Profile::getGCFactor, lib/profile/Profile.cpp:1507-1517),
 * weight = factor / frag_size for a window of exactly frag_size bases when full_tile_form is set (Segment.cpp:576,586),
 * else factor * len / frag_size^2 (:615).  The normal variate is addressed (Philox kind GC: window ordinal, attempt,
 * segment ordinal, ctx24 = population << 16 | chromosome) and read off the quantile table of the model -- n_cells + 1
 * knots of the standard normal's quantile function, cell = draw >> (32 - lg_cells), linear inside the cell -- in three
 * rounded fp64 operations, so the weights are bit for bit those of the host evaluation of the same table.
 * weights_out / gc_out (either may be NULL) receive n values each. */
typedef struct sg_gc_model {
  const double* means;      /* [101] */
  double std;
  const double* quantiles;  /* [2^lg_cells + 1] */
  uint32_t lg_cells;
  uint32_t frag_size;
  int32_t full_tile_form;
  uint32_t ctx24;
} sg_gc_model;
int sg_window_weights(sg_ctx* ctx, const sg_gc_window* windows, const uint32_t* seg_ord, const uint32_t* win_ord, uint64_t n,
                      const sg_gc_model* model, double* weights_out, int32_t* gc_out);

/* ---- sampling plan made on the device (whole-genome runs without targets) ------------------- */
/* The reference tiles every haplotype string of a segment into frag_size windows (Segment::getWeightedLength,
 * Segment.cpp:566-590), weighs them (sg_window_weights), sums the weights per segment, chromosome and genome
 * (Genome::setReadCounts, Genome.cpp:783-825) and hands every window fragWeights[i] * readCount / totalWL reads,
 * the remainder going to the segment's first window (Segment::setReadCount, Segment.cpp:462-476).  Here the host
 * only keeps per-SEGMENT data; the windows exist on the device alone:
 *   sg_windows_build   tiles the generators (one per haplotype string of a segment, in the reference's window order),
 *                      computes GC%, GC factor and weight of every window and the weight sum of every segment (summed
 *                      in window order, like the reference's loop); the weights stay in the context under `store_id`;
 *   sg_plan_windows    for the segments that got reads: per-window read counts, the remainder rule, planned pairs and
 *                      their prefix sum -- the sg_window table of the batch, on the device; reports the planned
 *                      fragments of every active segment (the host cuts pieces / shards by them);
 *   sg_plan_range      makes a run [a0, a1) of those active segments the current batch (what sg_plan does for a
 *                      host-made table); first_window / first_slot keep the draws' batch-wide addresses.        */
typedef struct sg_window_gen {
  uint64_t hap_base;      /* offset of the haplotype string inside chain `chain`                               */
  uint64_t hap_len;       /* its length; windows = ceil(hap_len / frag_size), the last one shorter               */
  uint32_t chain;
  uint32_t seg;           /* sg_windows_build: segment ordinal in the chromosome; sg_plan_windows: index in `active` */
  uint64_t first_window;  /* sg_plan_windows: index of the generator's first window in the numbering of sg_windows_build */
} sg_window_gen;
int sg_windows_build(sg_ctx* ctx, uint32_t store_id, const sg_window_gen* gens, uint64_t n_gens, uint32_t n_segs,
                     const sg_gc_model* model, double* seg_weight_out, uint64_t* n_windows_out);
typedef struct sg_active_seg {
  int64_t reads;          /* Segment::readCount                                                                */
  double weight;          /* the segment's weight sum (seg_weight_out of sg_windows_build)                     */
  uint32_t seg_size;      /* seqSize / CN (Segment.cpp:713-714)                                                */
  uint32_t pad;
} sg_active_seg;
int sg_plan_windows(sg_ctx* ctx, uint32_t store_id, const sg_window_gen* gens, uint64_t n_gens, const sg_active_seg* active,
                    uint32_t n_active, uint32_t frag_size, uint32_t batch_id, int32_t paired, const char* name_prefix,
                    uint64_t* slots_out /* [n_active] */, uint64_t* n_windows_out);
int sg_plan_range(sg_ctx* ctx, uint32_t a0, uint32_t a1);
void sg_windows_drop(sg_ctx* ctx);  /* frees every window-weight store of the context */


/* ---- instrumentation ----------------------------------------------------------------------- */
#define SG_K_PLAN 0
#define SG_K_NAMEBASE 1
#define SG_K_INDEL 2
#define SG_K_SCAN 3
#define SG_K_EMIT 4       /* header + main emit kernel                                   */
#define SG_K_EMIT_SLOW 5  /* emit_slow_kernel: the items queued for the generic item code */
#define SG_K_COUNT 6
/* ---- block-gzip sink ------------------------------------------------------------ */
/* After sg_result: compress the FASTQ text of the pass on the device into BGZF blocks -- independent
 * gzip members of 32 KB of text each (RFC 1952 with the 'BC' extra field), one dynamic-Huffman DEFLATE block
 * of literals and matches (copies of earlier text of the member, runs), one pair of codes per mate and pass.  Concatenated in order (and closed with sg_bgzf_eof) they form
 * a .fq.gz file that `zcat` turns back into exactly the text sg_fetch returns.  SeqWriter::write
 * (lib/seqwriter/SeqWriter.cpp:41-54) has no such mode: additive.                                   */
int sg_compress(sg_ctx* ctx, uint64_t* gz_bytes_r1, uint64_t* gz_bytes_r2);
int sg_fetch_compressed(sg_ctx* ctx, int mate, uint64_t offset, uint64_t bytes, void* host_dst);
/* ---- detached outputs: drain one batch while the next is sampled ------------------------------------- */
/* After sg_result (and sg_compress, if wanted) the FASTQ text -- and its BGZF form -- can be taken out of
 * the context: the handle owns the device buffers and a copy stream of its own, so sg_outputs_fetch may
 * run on another host thread while the context plans and samples the next batch into fresh buffers
 * (the reference's workers flush their 50 MB buffers through one mutex instead, Segment.cpp:834-846,
 * SeqWriter.cpp:49-54).  sg_release_outputs hands the buffers back to the context for reuse; it and
 * sg_detach_outputs are context calls (one thread at a time), sg_outputs_* are not.                  */
typedef struct sg_outputs sg_outputs;
int sg_detach_outputs(sg_ctx* ctx, sg_outputs** out);
int sg_outputs_sizes(const sg_outputs* o, uint64_t text_bytes[2], uint64_t gz_bytes[2]);
int sg_outputs_fetch(sg_outputs* o, int mate, int compressed, uint64_t offset, uint64_t bytes, void* host_dst);
const char* sg_outputs_last_error(const sg_outputs* o);
int sg_release_outputs(sg_ctx* ctx, sg_outputs* o);

/* the 28-byte empty BGZF block that ends a file */
int sg_bgzf_eof(uint8_t out[28]);
/* Host-only view of the code construction, for tests: from the histograms of the literal / length symbols
 * (286; [256] = end-of-block) and of the distance symbols (30), the code lengths and (bit-reversed) codes of both
 * alphabets, the length half of a match token per match length 3..258 (index = length) (code + extra bits in the low 24 bits, bit count
 * in the top 8) and the member prefix (gzip header with BSIZE = 0 + dynamic block header) as LSB-first 32-bit words.
 * Returns the number of prefix bits, 0 if `cap` is too small. */
uint32_t sg_deflate_plan(const uint64_t lit_counts[286], const uint64_t dist_counts[30], uint8_t lit_lens[286], uint32_t lit_codes[286],
                         uint8_t dist_lens[30], uint32_t dist_codes[30], uint32_t len_tokens[260], uint32_t* prefix_words, uint32_t cap);

/* When enabled, HIP events bracket every kernel of sg_sample on the ctx's stream;
 * sg_kernel_times() then returns the last pass's per-kernel milliseconds (after sg_result).     */
int sg_set_profiling(sg_ctx* ctx, int enable);
int sg_kernel_times(sg_ctx* ctx, float ms[SG_K_COUNT]);
/* Diagnostics of the last pass (after sg_result): items the straight-line emit kernel handed to the
 * generic item code (windows holding a non-ACGT base, reads with >= 2 sequencing indels), and whether
 * their queue overflowed so that the whole batch was emitted again by the generic kernel.  Results
 * are identical either way (Profile::predict, Profile.cpp:1520-1650 has one code path).          */
int sg_emit_info(sg_ctx* ctx, uint64_t* queued_items, int* requeued);
/* ---- profile training: what Profile::train gathers from the reads (lib/profile/Profile.cpp:1442-1484) ----
 * Lines of `samtools view` text (the reference reads them through popen, Profile.cpp:1448-1462) go through
 * Profile::processRead (:228-510) in file order, chunk by chunk: the filters of :262-279, Profile::countGC (:512-703: the
 * window whose reads are being counted; it turns reads away and opens windows), the CIGAR walk with its insertion /
 * deletion length counts (:290-382; events the VCF knows are not counted, only a single nM reaches the matrices),
 * subsDist1 / subsDist2 / kmersDist (:399-442), iSizeDist (:444-451), qualityDist (:453-481).  The reference bases come from
 * the contigs committed with sg_reference_commit (upper-cased, Genome.cpp:529); `contig_keys` names them in that order as the
 * FASTA index does (first token, chr / chrom prefix stripped, Fasta.cpp:58-69).
 *   sg_train_begin   tables zeroed, targets and known variants staged (refSequence / altSequence of Genome.cpp:466-475)
 *   sg_train_feed    one chunk of WHOLE lines, in file order; state is carried from chunk to chunk (once the cap of
 *                    Profile::processRead has been reached, further chunks are ignored, as the reference stops reading).
 *                    The text is copied before the call returns (the caller's buffer is free again); the chunk's kernels
 *                    are left running, so its verdict -- SG_ERR_INVALID for a line with fewer than 11 fields
 *                    (Profile.cpp:246-251), the cap -- is reported by the NEXT sg_train_feed or by sg_train_finish
 *   sg_train_finish  the count matrices; the (GC content, read count) pairs countGC pushed, in its order (gcs, readCounts)
 * Where the reference's behaviour is undefined the line is skipped and counted: reads hanging over their contig's end
 * (skipped_overhang).  Arrays are the caller's: subs1 / subs2 [kmer_count][bins][4], kmers [bins][kmer_count], quality
 * [16][bins][94], isize [n_isize], ins_len / del_len [n_indel_len].  Bases must be a permutation of ACGT, kmer 1..6.       */
typedef struct sg_train_setup {
  const char* const* contig_keys;
  uint32_t n_contigs;
  const char* bases;
  int32_t kmer, bins;
  uint32_t n_isize;        /* columns of iSizeDist kept (the reference grows its row; larger TLENs: isize_overflow) */
  uint32_t n_indel_len;    /* columns of insFreqs / delFreqs kept (longer events: indel_len_overflow) */
  int32_t count_gc;        /* 1: Profile::countGC gates and counts as the reference does; 0: every read through the filters counts */
  uint32_t window;         /* Segment::fragSize (1000) */
  uint64_t max_reads;      /* Profile::processRead's maxCount (Profile.cpp:236, 497-507): the run ends with the read that makes the
                            * count of counted reads reach it (twice it with targets); lines behind that one are never read.
                            * 0 = the reference's 300,000,000 */
  /* exome: inTargets[chr] after loadTargets + divideTargets (Genome.cpp:238-299, 684-739), in that order; the rows of contig c
   * are [target_first[c], target_first[c + 1]); spos / epos as the reference holds them.  NULL / no rows: whole genome. */
  const uint64_t* target_first;
  const int64_t* target_spos;
  const int64_t* target_epos;
  /* known variants of the sample (lib/vcfparser/vcfparser.cpp:26-106), in file order; positions as the parser stores them */
  uint64_t n_snv;  const uint32_t* snv_contig;  const int64_t* snv_pos;  const char* snv_alt;  const uint8_t* snv_homo;
  uint64_t n_ins;  const uint32_t* ins_contig;  const int64_t* ins_pos;  const int32_t* ins_len;
  uint64_t n_del;  const uint32_t* del_contig;  const int64_t* del_pos;  const int32_t* del_len;
} sg_train_setup;
typedef struct sg_train_counts {
  uint64_t *subs1, *subs2, *kmers, *quality, *isize, *ins_len, *del_len;
  uint64_t lines, reads_counted, cigar_chars, insert_events, delete_events, isize_overflow, indel_len_overflow, skipped_overhang,
      gc_rejected, gc_windows, capped;   /* capped: 1 when the run ended at max_reads */
} sg_train_counts;
int sg_train_begin(sg_ctx* ctx, const sg_train_setup* setup);
int sg_train_feed(sg_ctx* ctx, const char* sam_text, uint64_t sam_bytes);
/* 1 once the cap on counted reads is known to have been reached (the caller may stop reading its input), else 0; does not
 * wait for the chunk in flight, whose verdict arrives with the next feed / finish (a chunk fed behind the cap is ignored) */
int sg_train_capped(sg_ctx* ctx);
/* gc / rc: room for `gc_cap` pairs (NULL: none wanted); *n_gc = how many there are (SG_ERR_OVERFLOW when gc_cap is too small;
 * call again).  The session ends with a successful call or with sg_train_end. */
int sg_train_finish(sg_ctx* ctx, sg_train_counts* out, double* gc, double* rc, uint64_t gc_cap, uint64_t* n_gc);
void sg_train_end(sg_ctx* ctx);
/* One call for a text that fits memory: begin (no targets, no variants, count_gc = 0) + feed + finish. */
int sg_train_count(sg_ctx* ctx, const char* sam_text, uint64_t sam_bytes, const char* const* contig_keys, uint32_t n_contigs,
                   const char* bases, int32_t kmer, int32_t bins, uint32_t n_isize, uint32_t n_indel_len, sg_train_counts* out);

/* Which emit kernel the loaded profile gets (after sg_load_profile): 0 generic (tables that do not fit
 * LDS, k-mer sizes other than 3), 1 straight-line kernel (table image in LDS). */
int sg_emit_variant(sg_ctx* ctx);

/* Exact u32 form of the reference's inverse-CDF draw, exposed for tests: number of 32-bit draws
 * x for which randIndx's `r <= c` holds (r = 2.2204e-16 + (1-2.2204e-16)*x/2^32).               */
uint64_t sg_cdf_count_le(double c);

/* The per-base sampling tables, exposed for tests (host code, no GPU needed).  A CDF row partitions the 2^32 draws
 * into integer counts per outcome (sg_cdf_count_le differences); the engine samples through two rearrangements that
 * keep every count:
 *   sg_sub_row_identity_first  substitution row (Profile::getSubBaseIndx1/2, Profile.cpp:1527-1554): outcomes in the
 *                              order [cd, the other base indexes ascending], cum[i] = draws of order[0..i];
 *   sg_alias_row               quality row (Profile::getBaseQuality, Profile.cpp:1576-1580) as 2^lgW alias columns:
 *                              draw x -> column x >> (32 - lgW), symbol lo[col] if (x & (2^(32-lgW) - 1)) < thr[col] else
 *                              hi[col];  sg_row_symbols = number of symbols with a non-zero count (columns needed).   */
int sg_sub_row_identity_first(const double cdf4[4], int cd, uint64_t cum[3], uint8_t order[4]);
uint32_t sg_row_symbols(const double* cdf, int n);
int sg_alias_row(const double* cdf, int n, uint32_t lgW, uint32_t* thr, uint8_t* lo, uint8_t* hi);

#ifdef __cplusplus
}
#endif
#endif
