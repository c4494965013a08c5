// host/fasta.cpp -- see fasta.h.
#include "fasta.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace simu {

void Fasta::open(const std::string& ref_file) {
  std::string path = ref_file;
  if (path.empty()) throw Error("genome sequence file not specified!");
  if (path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0) {
    // same side effect as the reference: decompress next to the archive
    std::string plain = path.substr(0, path.size() - 3);
    std::string cmd = "gzip -cd " + path + " > " + plain;
    if (system(cmd.c_str()) != 0) throw Error("could not decompress " + path);
    path = plain;
  }
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) throw Error("could not open " + path);
  names.clear();
  seqs.clear();
  std::string* cur = nullptr;
  std::vector<char> buf(1 << 22);
  std::string header;
  bool in_header = false, line_start = true, skip_line = false;
  size_t got;
  while ((got = fread(buf.data(), 1, buf.size(), fp)) > 0) {
    const char* p = buf.data();
    const char* end = p + got;
    while (p < end) {
      if (in_header || skip_line) {
        const char* nl = (const char*)memchr(p, '\n', end - p);
        if (in_header) header.append(p, (nl ? nl : end) - p);
        if (!nl) break;
        p = nl + 1;
        if (in_header) {
          in_header = false;
          size_t b = header.find_first_not_of(" \t");
          std::string tok;
          if (b != std::string::npos) {
            size_t e = header.find_first_of(" \t", b);
            tok = header.substr(b, e == std::string::npos ? std::string::npos : e - b);
          }
          std::string key = abbr_of_chr(tok);
          if (!seqs.count(key)) names.push_back(key);
          cur = &seqs[key];
          cur->clear();
          header.clear();
        }
        skip_line = false;
        line_start = true;
        continue;
      }
      if (line_start) {
        if (*p == '>' || *p == '@') { in_header = true; p++; line_start = false; continue; }
        if (*p == ';') { skip_line = true; p++; line_start = false; continue; }
      }
      const char* nl = (const char*)memchr(p, '\n', end - p);
      const char* stop = nl ? nl : end;
      if (cur && stop > p) {
        size_t old = cur->size();
        cur->append(p, stop - p);
        for (size_t i = old; i < cur->size(); i++) {
          char& c = (*cur)[i];
          if (c >= 'a' && c <= 'z') c -= 32;
        }
      }
      if (nl) { p = nl + 1; line_start = true; }
      else { p = end; line_start = false; }
    }
  }
  fclose(fp);
  if (names.empty()) throw Error("ERROR: reference sequence cannot be empty!");
}

}  // namespace simu
