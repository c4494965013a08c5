// oracle/oracle_cli.cpp -- TEST INFRASTRUCTURE ONLY.  Command-line front end of the CPU oracle:
//   oracle_cli <config.txt> [--rng mt|philox] [--sec S] [--nsec NS] [--out DIR] [--threads T]
// mirrors `simuReads <config.txt>` (src/simuReads.cpp:24-87).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "oracle.h"

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <config.txt> [--rng mt|philox] [--sec S] [--nsec NS] [--out DIR] [--threads T]\n", argv[0]);
    return 1;
  }
  int mode = ORC_RNG_PHILOX, threads = 1;
  unsigned long long sec = 1500000000ULL, nsec = 123456789ULL;
  std::string out;
  for (int i = 2; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return argv[++i]; };
    if (a == "--rng") { std::string v = next(); mode = (v == "mt") ? ORC_RNG_MT : ORC_RNG_PHILOX; }
    else if (a == "--sec") sec = strtoull(next(), nullptr, 10);
    else if (a == "--nsec") nsec = strtoull(next(), nullptr, 10);
    else if (a == "--out") out = next();
    else if (a == "--threads") threads = atoi(next());
    else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
  }
  auto t0 = std::chrono::steady_clock::now();
  int rc = orc_simulate(argv[1], mode, sec, nsec, out.c_str(), threads);
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rc != 0) { fprintf(stderr, "%s\n", orc_last_error()); return rc; }
  fprintf(stderr, "oracle: %llu reads in %.3f s\n", (unsigned long long)orc_last_read_count(), dt);
  return 0;
}
