// oracle/fastq_digest.cpp -- TEST INFRASTRUCTURE ONLY: an order-independent digest of the records of FASTQ files, for the
// tests that compare a sharded run's files with the one-GPU run's (tests/test_gpu_full_size.py).  Every record (four lines)
// is hashed to 64 bits (FNV-1a over its bytes, then a splitmix finaliser); the digest is the count of records and the sum
// of the hashes modulo 2^64.  No part of the reference is restated here.
#include <cstdint>
#include <cstdio>
#include <vector>

extern "C" int orc_fastq_record_digest(const char* const* paths, int n_paths, uint64_t* count, uint64_t* sum) {
  uint64_t n = 0, total = 0;
  std::vector<unsigned char> buf(1u << 24);
  for (int f = 0; f < n_paths; f++) {
    FILE* fp = fopen(paths[f], "rb");
    if (!fp) return -1;
    uint64_t h = 1469598103934665603ull;
    int lines = 0;
    bool open_record = false;
    size_t got;
    while ((got = fread(buf.data(), 1, buf.size(), fp)) > 0) {
      for (size_t i = 0; i < got; i++) {
        const unsigned char c = buf[i];
        h = (h ^ c) * 1099511628211ull;
        open_record = true;
        if (c == '\n' && ++lines == 4) {
          uint64_t z = h + 0x9E3779B97F4A7C15ull;
          z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
          z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
          total += z ^ (z >> 31);
          n++;
          h = 1469598103934665603ull;
          lines = 0;
          open_record = false;
        }
      }
    }
    fclose(fp);
    if (open_record) return -2;   // a file that ends inside a record
  }
  *count = n;
  *sum = total;
  return 0;
}
