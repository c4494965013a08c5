// host/main.cpp -- `simuReads <configuration file>` (src/simuReads.cpp:24-97), GPU-backed.
// Same positional argument, usage text and exit codes; optional flags are additive:
//   --seed N  --device D  --out DIR  --no-write [--fetch]  --quiet  --rank R --world W  --stats  --host-haplotypes  --gzip  --gpus N
#include <dirent.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "simulate.h"

static void usage(const char* app) {
  std::cerr << "\nVersion: 1.0 (MI355X engine)\n\n"
            << "Usage: " << app << " <configuration file> [--seed N] [--device D] [--out DIR] [--no-write] [--quiet]\n\n"
            << "Example:\n    " << app << " /path/to/config.txt\n\n";
}

// --gpus N: one child process per GPU (exec'd before anything here touches the GPU), child r samples
// the r-th run of segments of every (population, chromosome) batch (--rank r --world N --device r) into
// <file>.part<r>; the parts of a file are then concatenated in rank order.  Every draw is addressed inside
// the whole batch, so the reads are those of the one-GPU run (record order aside, which the reference
// does not define either).  No collective is needed: each child derives the read counts from the whole
// genome itself.  `python -m simuscop_amd.run` is the torch.distributed (RCCL) front end of the same thing.
static int run_on_gpus(int gpus, int base_device, const std::vector<std::string>& args, const std::string& self,
                       const std::string& config, const std::string& out_override, bool merge) {
  std::vector<pid_t> kids;
  for (int r = 0; r < gpus; r++) {
    pid_t pid = fork();
    if (pid < 0) { std::cerr << "Error: fork failed" << std::endl; return 1; }
    if (pid == 0) {
      std::vector<std::string> a = args;
      // SIMUSCOP_SAME_DEVICE: rehearsal of the sharding on a one-GPU box (all children on the base device)
      const int dev = getenv("SIMUSCOP_SAME_DEVICE") ? base_device : base_device + r;
      a.insert(a.end(), {"--rank", std::to_string(r), "--world", std::to_string(gpus), "--device", std::to_string(dev)});
      if (r > 0) a.push_back("--quiet");
      std::vector<char*> cv;
      cv.push_back(const_cast<char*>(self.c_str()));
      for (std::string& x : a) cv.push_back(const_cast<char*>(x.c_str()));
      cv.push_back(nullptr);
      execv(self.c_str(), cv.data());
      _exit(127);
    }
    kids.push_back(pid);
  }
  int rc = 0;
  for (pid_t pid : kids) {
    int st = 0;
    waitpid(pid, &st, 0);
    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
    if (code != 0 && rc == 0) rc = code;
  }
  if (rc != 0 || !merge) return rc;
  std::string dir = out_override;
  if (dir.empty()) {
    std::ifstream f(config);
    std::string line;
    while (std::getline(f, line)) {
      size_t eq = line.find('=');
      if (eq == std::string::npos) continue;
      std::string k = line.substr(0, eq), v = line.substr(eq + 1);
      auto trim = [](std::string& t) { t.erase(0, t.find_first_not_of(" \t\r")); t.erase(t.find_last_not_of(" \t\r") + 1); };
      trim(k); trim(v);
      if (k == "output") dir = v;
    }
  }
  std::map<std::string, std::vector<std::pair<int, std::string>>> parts;
  if (DIR* d = opendir(dir.c_str())) {
    while (dirent* e = readdir(d)) {
      const std::string n = e->d_name;
      const size_t p = n.rfind(".part");
      if (p == std::string::npos || p + 5 >= n.size()) continue;
      parts[n.substr(0, p)].emplace_back(atoi(n.c_str() + p + 5), n);
    }
    closedir(d);
  }
  std::vector<char> buf(16u << 20);
  for (auto& kv : parts) {
    std::sort(kv.second.begin(), kv.second.end());
    FILE* dst = fopen((dir + "/" + kv.first).c_str(), "wb");
    if (!dst) { std::cerr << "Error: can not open fastq file to save results:\n" << dir + "/" + kv.first << std::endl; return -1; }
    for (auto& pr : kv.second) {
      const std::string path = dir + "/" + pr.second;
      if (FILE* src = fopen(path.c_str(), "rb")) {
        size_t got;
        while ((got = fread(buf.data(), 1, buf.size(), src)) > 0) fwrite(buf.data(), 1, got, dst);
        fclose(src);
        unlink(path.c_str());
      }
    }
    fclose(dst);
  }
  return 0;
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
  int gpus = 1;
  std::vector<std::string> pass;  // arguments handed on to --gpus children
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
    else if (a == "--gpus") { gpus = atoi(val()); continue; }
    else if (config.empty()) config = a;
    else {
      std::cerr << "Error: too many input arguments!" << std::endl;
      usage(argv[0]);
      return 1;
    }
  }
  if (gpus > 1) {
    for (int i = 1; i < argc; i++) {
      const std::string a = argv[i];
      if (a == "--gpus" || a == "--device" || a == "--rank" || a == "--world") { i++; continue; }
      pass.push_back(a);
    }
    char self[4096];
    const ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    if (n <= 0) { std::cerr << "Error: cannot locate the executable for --gpus" << std::endl; return 1; }
    self[n] = 0;
    return run_on_gpus(gpus, opt.device >= 0 ? opt.device : 0, pass, self, config, out, opt.write_files != 0);
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
