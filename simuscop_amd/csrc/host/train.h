// host/train.h -- `seqToProfile` as a library call: the reference's profile training (src/seqToProfile.cpp:19-147,
// Genome::loadTrainData lib/genome/Genome.cpp:32-39, Profile::init / train / saveResults lib/profile/Profile.cpp:172-218,
// 1442-1484, 1240-1365) with the per-read work on the GPU (sg_train_begin / _feed / _finish, include/simuscop_amd.h).
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct simu_train_options {
  const char* bam;        // -b: handed to `samtools view -F 0xD04 -q 20` exactly as the reference does (Profile.cpp:1448)
  const char* sam;        // --sam (additive): lines of `samtools view` text from a file, or "-" for standard input, instead
  const char* target;     // -t
  const char* vcf;        // -v
  const char* ref;        // -r
  const char* output;     // -o ("" / NULL: standard output, as in the reference)
  const char* samtools;   // -s
  int32_t kmer;           // -k (default 3)
  int32_t bins;           // -B (default 50)
  int32_t device;         // --device (additive)
  int32_t threads;        // reader threads of the reference ingest
  int32_t quiet;          // no progress lines on stderr
  const char* stamp;      // NULL: the current time as saveResults prints it; tests pass a fixed line
  uint64_t max_reads;     // --max-reads (additive): Profile::processRead's maxCount; 0 = the reference's 300,000,000 counted reads
                          // (600,000,000 with targets), behind which it stops reading (Profile.cpp:236, 497-507)
} simu_train_options;

typedef struct simu_train_stats {
  uint64_t lines, reads_counted, gc_rejected, gc_windows, gc_pairs, skipped_overhang, sam_bytes;
  int32_t read_length, bins, gc_fitted, capped;   // capped: the run ended at max_reads
  double t_reference, t_reads, t_total;   // seconds: reference to the device; SAM text through the kernels; everything
  double insert_rate, del_rate, std_isize, gc_std;
} simu_train_stats;

void simu_train_default_options(simu_train_options* o);
// 0 on success; otherwise the exit code the reference would use, its message in `err`.
int simu_train(const simu_train_options* opt, simu_train_stats* stats, char* err, size_t err_len);

#ifdef __cplusplus
}
#endif
