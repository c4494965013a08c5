// host/fasta.h -- bulk reference loader.
//
// The reference opens the FASTA through a .fai index and fseek/freads one <=1 Mbp slice per
// segment, twice per run (lib/fastahack/Fasta.cpp:304-334 via lib/segment/Segment.cpp:137).  Here
// every contig is read once, newline-stripped and upper-cased (Segment.cpp:143) into one contiguous
// byte array that the haplotype builder slices without copying.  Contig keys follow the index
// reader: first whitespace token of the header with the chr/chrom prefix removed (Fasta.cpp:58-69).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../../include/simuscop_amd.h"

namespace simu {

struct FastaContig {  // one row of the index (what a .fai line holds, Fasta.cpp:45-85)
  uint64_t raw_offset = 0, length = 0;
  uint32_t line_bases = 0, line_width = 0;
};

struct Fasta {
  // Two things fastahack does with odd files are kept by default (quirks reproduced, fixes behind additive
  // flags); the oracle restates both and is pinned on the reference binary (tests/edge_inputs.py):
  //  * CR LF line ends: getline() cuts at '\n' only, so the carriage returns stay -- one per line in the sequence, where
  //    they are unknown bases, and at the end of every contig name (Fasta.cpp:150-199).  `crlf_as_lf` (--crlf-as-lf)
  //    reads such a file like its LF twin instead.
  //  * a contig name met again: listed once more, but the index keeps the FIRST entry (std::map::insert, Fasta.cpp:67,198)
  //    and is written and read back sorted by offset (Fasta.cpp:84-97): the name stands twice in the chromosome list,
  //    next to its first place, and both resolve to the first sequence.  `unique_contigs` (--unique-contigs) refuses
  //    such a file.
  bool crlf_as_lf = false;
  bool unique_contigs = false;
  std::vector<std::string> names;            // index order (a repeated name: see above)
  std::map<std::string, std::string> seqs;   // upper-cased bases (host mode only)
  // device mode: the file lives in HBM as base codes (sg_reference_*), the host keeps the index only
  bool on_device = false;
  std::vector<FastaContig> contigs;          // order of the table given to sg_reference_commit
  std::map<std::string, uint32_t> contig_of; // key -> row of `contigs`
  bool streamed = false;                     // false: the general host parser ran and its result was uploaded
  // sharded ingest (multi-GPU, ranks own whole contigs): every contig is in `names` / `contigs` with its length,
  // but only the owned ones are on this device; dev_row = row of the table given to sg_reference_commit, or -1
  std::vector<int32_t> dev_row;

  // a header's key met again: the reference's place for it in `names` (or the refusal); true when it is a repeat
  bool note_name(const std::string& key, bool seen);
  void open(const std::string& path);        // handles the reference's `.gz` convention (Genome.cpp:224-228)
  // Streams the file to the engine (`threads` readers, two pinned staging buffers), finds the headers
  // with the device scan, reads only the header lines on the host.  Files the arithmetic ingest cannot
  // take (lines of several widths, ';' comments) go through open() and are uploaded stripped.
  void open_on_device(const std::string& path, ::sg_ctx* ctx, int threads);
  // The file's index rows without loading it: from `<path>.fai` when that is at least as new as the file
  // (the reference keeps such an index next to the FASTA, lib/fastahack/Fasta.cpp:45-85, 233-260), else from a
  // header scan on `threads` host threads.  False (nothing changed) when the file is not a plain fixed-width FASTA.
  bool load_index(const std::string& path, int threads);
  // Sharded ingest: after load_index(), stream ONLY the contigs with owned[row] != 0 to the engine.
  void open_owned_on_device(const std::string& path, ::sg_ctx* ctx, int threads, const std::vector<char>& owned);
  int32_t device_row(const std::string& chr) const {
    auto it = contig_of.find(chr);
    if (it == contig_of.end()) return -1;
    return dev_row.empty() ? (int32_t)it->second : dev_row[it->second];
  }
  long length(const std::string& chr) const {
    if (on_device) {
      auto it = contig_of.find(chr);
      return it == contig_of.end() ? 0 : (long)contigs[it->second].length;
    }
    auto it = seqs.find(chr);
    return it == seqs.end() ? 0 : (long)it->second.size();
  }
};

}  // namespace simu
