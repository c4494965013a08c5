// host/main.cpp -- `simuReads <configuration file>` (src/simuReads.cpp:24-97), GPU-backed.
// Same positional argument, usage text and exit codes; optional flags are additive:
//   --seed N  --device D  --out DIR  --no-write [--fetch]  --quiet  --rank R --world W  --stats  --host-haplotypes  --gzip
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>

#include "simulate.h"

static void usage(const char* app) {
  std::cerr << "\nVersion: 1.0 (MI355X engine)\n\n"
            << "Usage: " << app << " <configuration file> [--seed N] [--device D] [--out DIR] [--no-write] [--quiet]\n\n"
            << "Example:\n    " << app << " /path/to/config.txt\n\n";
}

int main(int argc, char* argv[]) {
  if (argc == 1) {
    std::cerr << "Error: configuration file is required!" << std::endl;
    usage(argv[0]);
    return 1;
  }
  simu_options opt;
  simu_default_options(&opt);
  std::string config, out;
  bool stats = false;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto val = [&]() -> const char* {
      if (i + 1 >= argc) { std::cerr << "Error: missing value for " << a << std::endl; exit(1); }
      return argv[++i];
    };
    if (a == "--seed") { opt.has_seed = 1; opt.seed = strtoull(val(), nullptr, 10); }
    else if (a == "--device") opt.device = atoi(val());
    else if (a == "--out") { out = val(); opt.output_dir = out.c_str(); }
    else if (a == "--no-write") opt.write_files = 0;
    else if (a == "--fetch") opt.fetch = 1;
    else if (a == "--quiet") opt.quiet = 1;
    else if (a == "--rank") opt.shard_rank = atoi(val());
    else if (a == "--world") opt.shard_world = atoi(val());
    else if (a == "--stats") stats = true;
    else if (a == "--host-haplotypes") opt.host_haplotypes = 1;
    else if (a == "--gzip") opt.gzip = 1;
    else if (config.empty()) config = a;
    else {
      std::cerr << "Error: too many input arguments!" << std::endl;
      usage(argv[0]);
      return 1;
    }
  }
  time_t start_t = time(NULL);
  simu_stats st;
  char err[4096] = {0};
  int rc = simu_run(config.c_str(), &opt, &st, err, sizeof err);
  if (rc != 0) {
    std::cerr << err << std::endl;
    return rc;
  }
  long used = (long)(time(NULL) - start_t);
  if (!opt.quiet)
    std::cerr << "\nElapsed time: " << used / 60 << " minutes and " << used % 60 << " seconds!\n" << std::endl;
  if (stats)
    fprintf(stderr,
            "stats: reads=%llu fragments=%llu bytes=%llu windows=%llu segments=%llu batches=%llu | load %.3fs (engine %.3fs reference %.3fs) haplotypes %.3fs "
            "plan %.3fs sample %.3fs (haplotype calls %.3fs sg_plan %.3fs) fetch %.3fs write %.3fs total %.3fs | kernels ms: plan %.3f namebase %.3f indel %.3f scan %.3f emit %.3f emit_slow %.3f | queued_items=%llu requeued_batches=%llu | compress %.3fs gz_bytes=%llu\n",
            (unsigned long long)st.reads, (unsigned long long)st.fragments, (unsigned long long)st.fastq_bytes,
            (unsigned long long)st.windows, (unsigned long long)st.segments, (unsigned long long)st.batches, st.t_load,
            st.t_engine, st.t_reference,
            st.t_haplotypes, st.t_plan, st.t_sample, st.t_hap_device, st.t_plan_api, st.t_fetch, st.t_write, st.t_total, st.kernel_ms[0], st.kernel_ms[1],
            st.kernel_ms[2], st.kernel_ms[3], st.kernel_ms[4], st.kernel_ms[5], (unsigned long long)st.queued_items,
            (unsigned long long)st.requeued_batches, st.t_compress, (unsigned long long)st.gz_bytes);
  return 0;
}
