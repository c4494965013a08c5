// sg_train.h -- device job of the profile-training path (sg_train.hip), shared with the host API.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sg {

constexpr uint32_t kTrainKeyBytes = 64;  // contig key slots (NUL terminated)
// single counters behind the count tables (u64 each)
enum : uint32_t { kTrainReads = 0, kTrainCigarChars, kTrainInsEvents, kTrainDelEvents, kTrainIsizeOverflow, kTrainOverhang,
                  kTrainIndelLenOverflow, kTrainGcRejected, kTrainEmptyLines, kTrainScalars };

struct TrainContig {
  uint64_t code_off, length;
  uint64_t tgt_first;      // rows of this contig in the target arrays: [tgt_first, tgt_first + tgt_n)
  uint32_t tgt_n;
  uint32_t xym;            // key is "X", "Y" or "M": Profile::countGC turns such reads away before it looks at its state (:532-535)
  uint64_t ins_first, del_first;   // rows in the known-insertion / known-deletion arrays
  uint32_t ins_n, del_n;
};
struct TrainRead {    // one per line, written by train_fields_kernel
  uint64_t seq_off, qual_off, cigar_off, ref_off;
  int64_t pos0;       // POS - 1
  uint32_t len, cigar_len;
  uint32_t flags;     // 1 through the filters of :262-279 (countGC sees it), 2 tlen < 0 (mate 2, reverse-complemented), 4 quality
                      // string as long as the read, 8 countGC returned non-zero, 16 a single nM inside its contig: counted,
                      // 32 on X / Y / M, 64 starts behind its contig's end, 128 an empty line, 256 fewer than eleven fields
  int32_t tlen;
  uint32_t contig, pad;
};
struct TrainGate { int64_t pos0; uint32_t contig, line; };   // the reads countGC sees, in file order
// what a read does to countGC's state
struct TrainStep { uint32_t opens, counted; int64_t left, right; uint32_t ws, pad; };
struct TrainWindow { int64_t left, right; uint32_t contig, ws; };
// state carried from one chunk of lines to the next (two copies: read / written)
struct TrainCarry {
  int64_t max_pos, ref_min;      // running maximum of pos0 inside the current run of one contig; least contig length met so far
  uint32_t last_contig, has;     // contig of the last read countGC saw
  uint64_t n_windows;            // windows opened so far
  uint64_t n_lines, n_gated;     // of the chunk just processed (for the host)
  uint64_t reads_total;          // reads counted so far (Profile::processRead's readCount)
  uint64_t cut_line;             // the line of this chunk at which readCount reached its cap (:497-507), ~0: none
};
// known insertions / deletions of one kind, all contigs: file order for the prefix maxima, (pos, len) order for the look-up
struct TrainKnown {
  const int64_t* pmax;      // [n] running maximum of the positions in file order, per contig
  const int64_t* pos;       // [n] sorted by (pos, len) per contig
  const int32_t* len;
  const uint32_t* first;    // least file-order index among the rows with this (pos, len)
};

struct TrainJob {
  const char* text;            // one chunk of whole lines
  uint64_t bytes;
  uint64_t* line_end;          // [n_lines] offset of every line break
  uint64_t n_lines;            // (known to the host after the line scan)
  const char* keys;            // [n_contigs][kTrainKeyBytes]
  const TrainContig* contigs;
  uint32_t n_contigs;
  const uint8_t* ref_codes;    // refSequence: the reference with the homozygous SNVs of the VCF (Genome.cpp:469-475)
  const uint8_t* alt_codes;    // altSequence: with every SNV
  char bases[4];
  uint32_t remap;              // natural code -> index in `bases`, 2 bits each
  uint32_t kmer, bins, kmer_count, n_isize, n_indel_len;
  uint32_t kmer_off[8];        // first index of the contexts with m real bases
  uint32_t count_gc, wes, window;
  uint64_t max_reads;          // Profile::processRead stops the run at so many counted reads (0: no cap)
  const int64_t* tgt_left;     // exome targets: first base, last base (0-based), running maximum of the last bases per contig
  const int64_t* tgt_right;
  const int64_t* tgt_pmax;
  TrainKnown known_ins, known_del;
  TrainRead* reads;
  TrainGate* gate;
  TrainStep* steps;
  TrainWindow* windows;        // [windows opened so far + this chunk's]
  uint32_t* window_rc;
  const TrainCarry* carry_in;
  TrainCarry* carry_out;
  void* scan_work;             // tile aggregates of the scans
  unsigned long long *subs1, *subs2, *kmers, *quality, *isize, *ins_len, *del_len, *scalars;
  uint32_t* flags;             // bit 0: a line with fewer than eleven fields
};

size_t train_scan_work_bytes(uint64_t n_elems);
void launch_train_lines_count(const TrainJob& J, hipStream_t s);   // line breaks of the chunk -> carry_out->n_lines
void launch_train_lines_fill(const TrainJob& J, hipStream_t s);    // ... -> line_end[n_lines] (after the count, same job)
void launch_train_chunk(const TrainJob& J, hipStream_t s);      // everything else of one chunk (needs n_lines)
void launch_train_window_gc(const TrainWindow* w, const uint32_t* rc, uint64_t n, const TrainContig* contigs, const uint8_t* ref_codes,
                            uint32_t wes, double* gc, double* rcs, hipStream_t s);
void launch_train_patch(uint8_t* codes, const uint64_t* off, const uint8_t* ch, uint64_t n, hipStream_t s);

}  // namespace sg
