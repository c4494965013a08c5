// host/simulate.h -- `simuReads <config>` as a library call (used by the CLI, tests and bench.py).
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct simu_options {
  int32_t device;          // -1: take `device` from the config (default 0)
  int32_t has_seed;        // 0: take `seed` from the config
  uint64_t seed;
  int32_t write_files;     // 1: write FASTQ like the reference (SeqWriter); 0: keep results on the device only
  int32_t fetch;           // with write_files == 0: still copy the FASTQ text to (pinned) host memory
  int32_t quiet;           // suppress the reference's stderr progress lines
  int32_t shard_rank;      // multi-GPU: this process samples batches with (batch ordinal % shard_world) == shard_rank
  int32_t shard_world;     // 1 = no sharding
  const char* output_dir;  // NULL / "": use the config's `output`
  int32_t repeat_sample;   // >1: re-run sg_sample this many times per batch (kernel timing experiments)
  int32_t host_haplotypes; // 1: parse the FASTA and edit haplotype strings on the host, upload them
                           // (sg_upload_haplotypes); 0 (default): stream the file to the device and
                           // assemble the haplotypes there (sg_reference_*, sg_build_haplotypes)
  int32_t gzip;            // 1: compress the FASTQ text on the device (sg_compress) and write <name>_1.fq.gz ...
                           // as BGZF (block gzip); what `zcat` gives back is byte for byte the plain file
  int32_t shard_contigs;   // multi-GPU, 1: rank shard_rank of shard_world OWNS whole chromosomes (longest-first assignment
                           // by length): it ingests, cuts, scans and samples only those; the per-chromosome GC-weighted
                           // lengths every rank needs for the read apportioning (Genome::setReadCounts) come through
                           // `exchange`.  0: every rank holds the whole genome and samples its run of segments of every batch.
  int32_t no_eof_block;    // gzip part files: leave the BGZF end-of-file block to whoever concatenates the parts
  // all-reduce(sum) of n doubles over the ranks, in place; every element has exactly one non-zero contributor (its
  // owner), so the sum is exact whatever the order.  Called once per population.  Returns 0 on success.
  int (*exchange)(void* user, double* values, int32_t n);
  void* exchange_user;
  // Three reference quirks are kept by default; each flag is additive and turns ONE of them off:
  int32_t crlf_as_lf;      // 1: a FASTA with CR LF line ends reads like its LF twin (default: the carriage returns stay, in the
                           //    contig names and as one unknown base per line, as fastahack keeps them: Fasta.cpp:150-199)
  int32_t strict_bases;    // 1: a literal 'X' in the genome is an unknown base (default: it walks the k-mer trie as the place
                           //    holder of the short contexts, Profile.cpp:94-101, 220-226)
  int32_t unique_contigs;  // 1: refuse a FASTA that holds a contig name twice (default: the name stands twice in the
                           //    chromosome list and both resolve to the first sequence, Fasta.cpp:67,84-97,198)
} simu_options;

typedef struct simu_stats {
  uint64_t reads;          // FASTQ records produced (both mates counted)
  uint64_t fragments;      // pairs (PE) or reads (SE)
  uint64_t fastq_bytes;
  uint64_t planned_reads;  // Genome::yieldReads `reads` (Genome.cpp:831)
  uint64_t windows, segments, batches;
  double t_load;           // config + inputs + profile
  double t_haplotypes;     // haplotype chains: edit lists or strings (host)
  double t_plan;           // GC scan (device) + weights + read counts (host)
  double t_sample;         // sg_plan + sg_sample + sg_result (GPU pass, wall)
  double t_fetch;          // D2H of FASTQ text
  double t_write;          // file output
  double t_total;
  float kernel_ms[8];      // summed per kernel (SG_K_*)
  uint64_t queued_items;   // items the fast emit kernel left to the generic item code (sg_emit_info)
  uint64_t requeued_batches;  // batches emitted again because that queue overflowed
  double t_engine;         // part of t_load: sg_create (HIP context, stream)
  double t_reference;      // part of t_load: reference FASTA to its resident form (host strings or device codes)
  double t_hap_device;     // part of t_plan/t_sample: sg_build_haplotypes / sg_upload_haplotypes calls
  double t_plan_api;       // part of t_sample: sg_plan calls (window upload, work buffers)
  double t_compress;       // sg_compress calls (gzip mode)
  uint64_t gz_bytes;       // compressed bytes produced (gzip mode)
} simu_stats;

// Returns 0 on success.  On failure returns the exit code the reference would use and writes the
// message it would print to `err`.
int simu_run(const char* config_path, const simu_options* opt, simu_stats* stats, char* err, size_t err_len);

void simu_default_options(simu_options* opt);

// Chromosome ownership of the shard_contigs mode: owner_out[i] = rank of contig i (longest first onto the least
// loaded rank; ties by file order).  Exposed for the launchers and tests.
void simu_assign_contigs(const uint64_t* lengths, int32_t n, int32_t world, int32_t* owner_out);

// CPU-only self-test of the haplotype edit lists (no GPU call): every (population, chromosome) of the config is
// built twice -- as strings (Genome::segment_haplotypes, the reference's std::string editing) and as the copy list
// handed to sg_build_haplotypes (Genome::segment_pieces), materialised on the host -- and compared byte for byte.
// Returns 0 when all chains agree; otherwise 1 and a description in `err`.
int simu_selftest_haplotypes(const char* config_path, uint64_t seed, char* err, size_t err_len);

// ---- step-by-step session (bench.py / multi-GPU launcher) ----
typedef struct simu_session simu_session;
int simu_open(const char* config_path, const simu_options* opt, simu_session** out, char* err, size_t err_len);
void simu_close(simu_session* s);
void* simu_engine(simu_session* s);  // the sg_ctx* of this session
uint64_t simu_planned_reads(simu_session* s);
int simu_chromosome_count(simu_session* s);
int simu_weighted_length(simu_session* s, int popu, double* wl, char* err, size_t err_len);
int simu_set_reads(simu_session* s, int popu, int64_t reads, char* err, size_t err_len);
int simu_prepare_batch(simu_session* s, int popu, int chr, int* has_work, char* err, size_t err_len);
void simu_get_stats(simu_session* s, simu_stats* st);

#ifdef __cplusplus
}
#endif
