// host/common.h -- small shared helpers of the C++ host side (string handling with the reference's
// exact semantics, host Philox for the two host-side draw kinds, error type).
#pragma once
#include <cstdint>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace simu {

// Errors carry the reference's message text and exit code (the reference prints to cerr and calls
// exit(1) / exit(-1), e.g. lib/config/Config.cpp:50-56,67-70); the CLI maps them back.
struct Error : std::runtime_error {
  int exit_code;
  Error(const std::string& m, int code = 1) : std::runtime_error(m), exit_code(code) {}
};

// lib/mydefine/MyDefine.cpp:197-209
inline std::string trim(const std::string& str, const char* charlist = " \t\r\n") {
  size_t b = str.find_first_not_of(charlist);
  if (b == std::string::npos) return std::string();
  size_t e = str.find_last_not_of(charlist);
  return str.substr(b, e - b + 1);
}

// lib/split/split.cpp:3-16: getline semantics (empty fields kept, a trailing empty field dropped)
inline std::vector<std::string> split(const std::string& s, char delim) {
  std::vector<std::string> out;
  size_t b = 0;
  while (b <= s.size()) {
    size_t e = s.find(delim, b);
    if (e == std::string::npos) {
      if (b < s.size()) out.push_back(s.substr(b));
      break;
    }
    out.push_back(s.substr(b, e - b));
    b = e + 1;
  }
  return out;
}

// lib/mydefine/MyDefine.cpp:212-225 (and the identical code in snp.cpp:131-145, Fasta.cpp:59-68):
// the text after the FIRST "chrom", else after the first "chr", wherever it occurs in the name.
inline std::string abbr_of_chr(std::string chr) {
  size_t i = chr.find("chrom");
  if (i == std::string::npos) {
    i = chr.find("chr");
    if (i != std::string::npos) chr = chr.substr(i + 3);
  } else {
    chr = chr.substr(i + 5);
  }
  return chr;
}

// ---- Philox4x32-10 on the host (Salmon et al. SC'11); device twin lives in sg_kernels.hip ----
struct Philox4 { uint32_t v[4]; };
inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}
enum : uint32_t { KIND_HAP = 1, KIND_GC = 2 };

}  // namespace simu
