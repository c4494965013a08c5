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
  std::vector<std::string> names;            // file order
  std::map<std::string, std::string> seqs;   // upper-cased bases (host mode only)
  // device mode: the file lives in HBM as base codes (sg_reference_*), the host keeps the index only
  bool on_device = false;
  std::vector<FastaContig> contigs;          // order of the table given to sg_reference_commit
  std::map<std::string, uint32_t> contig_of; // key -> row of `contigs`
  bool streamed = false;                     // false: the general host parser ran and its result was uploaded

  void open(const std::string& path);        // handles the reference's `.gz` convention (Genome.cpp:224-228)
  // Streams the file to the engine (`threads` readers, two pinned staging buffers), finds the headers
  // with the device scan, reads only the header lines on the host.  Files the arithmetic ingest cannot
  // take (lines of several widths, ';' comments) go through open() and are uploaded stripped.
  void open_on_device(const std::string& path, ::sg_ctx* ctx, int threads);
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
