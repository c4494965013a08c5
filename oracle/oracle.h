// oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of SimuSCoP's simuReads read-sampling path (the reference's
// Genome::yieldReads -> Segment::yieldReads -> Profile::predict -> SeqWriter chain).  It is the
// parity checker for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.  The product (simuscop_amd/) never includes, links or calls it.
//
// Pinning: in RNG mode "mt" the oracle consumes libstdc++ mt19937 / default_random_engine /
// glibc rand() draws in exactly the reference's order, so that for `threads = 1` its FASTQ output
// is byte-identical to the unmodified reference binary (oracle/_ref/simuReads) run under
// oracle/fakeclock.c with the same frozen time.  tests/golden/ holds md5 sums of such reference
// runs (made by tests/golden/make_golden.py).  In RNG mode "philox" the same algorithm code draws
// from counter-addressed Philox4x32-10 streams (oracle/philox.h) -- that is the specification the
// HIP kernels are compared with bit for bit.  Two of its draw -> outcome maps are re-arranged (identity-first
// substitution rows, alias columns for qualities): each outcome keeps exactly the number of 32-bit draws the
// reference's `r <= cdf[k]` scan gives it, so the sampled distributions are identical, not approximated
// (tests/test_integer_tables.py checks the counts row by row).
#pragma once
#include <cstdint>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_RNG_MT = 0, ORC_RNG_PHILOX = 1 };

// Whole-run simulation from a SimuSCoP config file (src/simuReads.cpp:24-87).
//   rng_mode ORC_RNG_MT    : seed_sec/seed_nsec play the frozen wall clock (see fakeclock.c)
//   rng_mode ORC_RNG_PHILOX: seed = ((uint64)seed_sec << 32) | seed_nsec
//   threads  : >1 only allowed in philox mode (results are independent of it)
// Returns 0 on success; on error returns non-zero and orc_last_error() describes it.
int orc_simulate(const char* config_path, int rng_mode, uint64_t seed_sec, uint64_t seed_nsec,
                 const char* output_dir_override, int threads);
const char* orc_last_error(void);
// reads (not pairs) written by the last orc_simulate call
uint64_t orc_last_read_count(void);

// ---- Profile tables (lib/profile/Profile.cpp:934-1238, :836-932, :1367-1434) ----
typedef struct orc_profile orc_profile;
orc_profile* orc_profile_load(const char* path, int paired, int insert_size);
void orc_profile_free(orc_profile*);
// scalar info: 0 n_bases, 1 kmer, 2 bins, 3 read_length, 4 kmer_count, 5 n_qual, 6 n_ins,
//              7 n_del, 8 n_isize (0 = fixed insert size), 9 has_sub2, 10 isize_min, 11 lgW (alias columns = 2^lgW)
int orc_profile_info(const orc_profile*, int what);
double orc_profile_rate(const orc_profile*, int which);  // 0 insertRate 1 delRate 2 stdISize 3 gcStd
// array views (fp64, exactly the reference's in-memory CDFs): 0 insCdf 1 delCdf
// 2 subsCdf1 [kmer_count][bins][N] 3 subsCdf2 4 qualityCdf [N*N][bins][n_qual] 5 iSizeCdf 6 gcMeans[101]
const double* orc_profile_array(const orc_profile*, int which);
// philox-mode integer sampling tables (oracle.cpp "integer sampling tables"):
//   substitution row: cumulative draw counts of the outcomes order[0..2] (order[0] = the reference base itself)
//   quality alias row: per column the threshold inside the column and the symbols below / from it
void orc_profile_sub_row(const orc_profile*, int mate2, int kmer_indx, int bin, uint64_t cum[3], uint8_t order[4]);
void orc_profile_alias_row(const orc_profile*, int base_pair, int bin, uint32_t* thr, uint8_t* lo, uint8_t* hi);
// kmer strings in table order (Profile::initKmers, Profile.cpp:70-124): writes kmer chars of entry i
void orc_profile_kmer(const orc_profile*, int i, char* out);
// indel candidates by skipping ahead: ab = {A, B} (2^-64 units), gaps[k-1] = P(no candidate in k positions) * 2^64; returns L
int orc_profile_indel_gaps(const orc_profile*, uint64_t ab[2], uint64_t* gaps, int n);

// ---- Profile::predict (Profile.cpp:1586-1701) in philox mode, for unit parity ----
// ref: n bytes (not NUL terminated). out: caller buffer >= 2*(n+max_ins)+1. Returns read length n'.
int orc_predict_philox(const orc_profile*, const char* ref, int n, int is_read1, uint64_t seed,
                       uint32_t batch_id, uint32_t pair_slot, char* out_bases, char* out_quals);

// ---- profile training (Profile::train, Profile.cpp:1442-1484): see train_oracle.cpp ----
// Caller-allocated count arrays: subs1/subs2 [kmer_count][bins][N], kmers [bins][kmer_count], quality [N*N][bins][94],
// isize [n_isize], ins_len / del_len [n_indel_len] (same layout as sg_train_counts).  PARITY UNPINNED for the counting
// (no samtools / BAM in this image): restated from the source, not run against it.
typedef struct orc_train_counts {
  uint64_t *subs1, *subs2, *kmers, *quality, *isize, *ins_len, *del_len;
  uint64_t lines, reads_counted, cigar_chars, insert_events, delete_events, isize_overflow, indel_len_overflow, skipped_overhang,
      gc_rejected, gc_windows, capped;
} orc_train_counts;
int orc_train_count(const char* sam_text, uint64_t sam_bytes, const char* fasta_path, const char* bases, int kmer, int bins,
                    uint32_t n_isize, uint32_t n_indel_len, orc_train_counts* out);
int orc_train(const char* sam_text, uint64_t sam_bytes, const char* fasta_path, const char* vcf_path, const char* bed_path,
              const char* bases, int kmer, int bins, uint32_t n_isize, uint32_t n_indel_len, orc_train_counts* out,
              double* gc, double* rc, uint64_t cap, uint64_t* n_gc);
void orc_train_set_max_reads(uint64_t n);   /* Profile.cpp:236 `maxCount` for the following calls; 0 = 300,000,000 */
int orc_train_profile(const char* sam_text, uint64_t sam_bytes, const char* fasta_path, const char* vcf_path, const char* bed_path,
                      const char* bases, int kmer, int bins, const char* out_path, const char* bam_label, const char* stamp);

// Philox known-answer helpers (orc_base_rounds: the round count of the per-base draws, philox.h kBaseRounds)
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_philox4x32_r(int rounds, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
int orc_base_rounds(void);

// Test utility (fastq_digest.cpp): count of the FASTQ records of the files and the sum of their 64-bit hashes -- equal for
// two sets of files that hold the same records in any order.  0 on success, -1 unreadable file, -2 a file ends inside a record.
int orc_fastq_record_digest(const char* const* paths, int n_paths, uint64_t* count, uint64_t* sum);

#ifdef __cplusplus
}
#endif
