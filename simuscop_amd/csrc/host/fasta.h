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

namespace simu {

struct Fasta {
  std::vector<std::string> names;            // file order
  std::map<std::string, std::string> seqs;   // upper-cased bases
  void open(const std::string& path);        // handles the reference's `.gz` convention (Genome.cpp:224-228)
  long length(const std::string& chr) const {
    auto it = seqs.find(chr);
    return it == seqs.end() ? 0 : (long)it->second.size();
  }
};

}  // namespace simu
