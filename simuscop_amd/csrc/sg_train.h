// sg_train.h -- device job of sg_train_count (sg_train.hip), shared with the host API.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sg {

constexpr uint32_t kTrainKeyBytes = 64;  // contig key slots (NUL terminated)
// scalars block (u64 each): [0,256) insertion lengths, [256,512) deletion lengths, then the single counters
enum : uint32_t { kTrainInsLen = 0, kTrainDelLen = 256, kTrainReads = 512, kTrainCigarChars, kTrainInsEvents, kTrainDelEvents,
                  kTrainIsizeOverflow, kTrainOverhang, kTrainScalars };

struct TrainContig { uint64_t code_off, length; };
struct TrainRead {    // one per line, written by train_parse_kernel
  uint64_t seq_off, qual_off, ref_off;
  uint32_t len, flags;  // flags: 1 counted, 2 tlen < 0 (mate 2, reverse-complemented), 4 quality string as long as the read
  int32_t tlen;
  uint32_t pad;
};
struct TrainJob {
  const char* text;            // the lines, every one ended by '\n'
  const uint64_t* line_off;    // [n_lines + 1] offsets of the line starts (the last: one past the final line break)
  uint64_t n_lines;
  const char* keys;            // [n_contigs][kTrainKeyBytes]
  const TrainContig* contigs;
  uint32_t n_contigs;
  const uint8_t* ref_codes;    // resident reference codes (A0 C1 T2 G3, N = 4, other = 5)
  char bases[4];
  uint32_t remap;              // natural code -> index in `bases`, 2 bits each
  uint32_t kmer, bins, kmer_count, n_isize;
  uint32_t kmer_off[8];        // first index of the contexts with m real bases
  TrainRead* reads;
  unsigned long long *subs1, *subs2, *kmers, *quality, *isize, *scalars;
  uint32_t* flags;             // bit 0: a line with fewer than eleven fields
};

void launch_train(const TrainJob& J, hipStream_t s);

}  // namespace sg
