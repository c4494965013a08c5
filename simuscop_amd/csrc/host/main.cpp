// host/main.cpp -- `simuReads <configuration file>` (src/simuReads.cpp:24-97), GPU-backed.
// Same positional argument, usage text and exit codes; optional flags are additive:
//   --seed N  --device D  --out DIR  --no-write [--fetch]  --quiet  --rank R --world W  --stats  --host-haplotypes  --gzip
//   --gpus N [--shard-contigs]
//   --crlf-as-lf  --strict-bases  --unique-contigs   (each turns one kept reference quirk off: simulate.h)
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

#include "../../../include/simuscop_amd.h"
#include "simulate.h"

static void usage(const char* app) {
  std::cerr << "\nVersion: 1.0 (MI355X engine)\n\n"
            << "Usage: " << app << " <configuration file> [--seed N] [--device D] [--out DIR] [--no-write] [--quiet]\n\n"
            << "Example:\n    " << app << " /path/to/config.txt\n\n";
}

// --gpus N: one child process per GPU (exec'd before anything here touches the GPU).  Two ways to shard:
//   default            child r samples the r-th run of segments of every (population, chromosome) batch (--rank r --world N
//                      --device r); every child holds the whole genome and derives the read counts itself;
//   --shard-contigs    child r OWNS whole chromosomes (longest first by length): it ingests, scans and samples only those;
//                      the per-chromosome GC-weighted lengths every child needs for the read apportioning
//                      (Genome::setReadCounts) are summed here, in the parent, over a pipe pair per child -- the parent
//                      never touches a GPU.  `python -m simuscop_amd.run` does the same exchange with an RCCL all-reduce.
// Either way child r writes <file>.part<r>; the parts of a file are concatenated in rank order.  Every draw is addressed
// inside its whole batch, so the reads are those of the one-GPU run (record order aside, which the reference does not
// define either).
static std::string config_output_dir(const std::string& config) {
  std::ifstream f(config);
  std::string line, dir;
  while (std::getline(f, line)) {
    size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    std::string k = line.substr(0, eq), v = line.substr(eq + 1);
    auto trim = [](std::string& t) { t.erase(0, t.find_first_not_of(" \t\r")); t.erase(t.find_last_not_of(" \t\r") + 1); };
    trim(k); trim(v);
    if (k == "output") dir = v;
  }
  return dir;
}

static bool read_all(int fd, void* p, size_t n) {
  size_t have = 0;
  while (have < n) {
    const ssize_t got = read(fd, (char*)p + have, n - have);
    if (got <= 0) return false;
    have += (size_t)got;
  }
  return true;
}
static bool write_all(int fd, const void* p, size_t n) {
  size_t have = 0;
  while (have < n) {
    const ssize_t got = write(fd, (const char*)p + have, n - have);
    if (got <= 0) return false;
    have += (size_t)got;
  }
  return true;
}

// child side of the exchange (simu_options.exchange): count + values up, sums down
struct PipeExchange { int up = -1, down = -1; };
static int pipe_exchange(void* user, double* values, int32_t n) {
  PipeExchange* px = (PipeExchange*)user;
  if (!write_all(px->up, &n, sizeof n) || !write_all(px->up, values, sizeof(double) * (size_t)n)) return 1;
  return read_all(px->down, values, sizeof(double) * (size_t)n) ? 0 : 1;
}

static int run_on_gpus(int gpus, int base_device, const std::vector<std::string>& args, const std::string& self,
                       const std::string& config, const std::string& out_override, bool merge, bool shard_contigs, bool gzip) {
  // A profiler's preloaded tool library initialises the GPU before main(): exec'ing the children from here would then be
  // an exec from a GPU-initialised process.  Profile one rank directly instead (`simuReads cfg --rank r --world N`).
  const char* preload = getenv("LD_PRELOAD");
  if (getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || (preload && strstr(preload, "rocprof"))) {
    std::cerr << "Error: --gpus starts one process per GPU and cannot run under a GPU profiler; profile a single rank "
                 "(--rank R --world N --device R) instead" << std::endl;
    return 1;
  }
  const std::string dir = out_override.empty() ? config_output_dir(config) : out_override;
  // parts of an earlier run (another N, a failed run) must not be merged in
  if (merge && !dir.empty()) {
    if (DIR* d = opendir(dir.c_str())) {
      std::vector<std::string> stale;
      while (dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        const size_t p = n.rfind(".part");
        if (p != std::string::npos && p + 5 < n.size() && n.find_first_not_of("0123456789", p + 5) == std::string::npos) stale.push_back(n);
      }
      closedir(d);
      for (const std::string& n : stale) unlink((dir + "/" + n).c_str());
    }
  }
  std::vector<pid_t> kids;
  std::vector<int> from_child((size_t)gpus, -1), to_child((size_t)gpus, -1);
  for (int r = 0; r < gpus; r++) {
    int up[2] = {-1, -1}, down[2] = {-1, -1};
    if (shard_contigs && (pipe(up) != 0 || pipe(down) != 0)) { std::cerr << "Error: pipe failed" << std::endl; return 1; }
    pid_t pid = fork();
    if (pid < 0) { std::cerr << "Error: fork failed" << std::endl; return 1; }
    if (pid == 0) {
      std::vector<std::string> a = args;
      // SIMUSCOP_SAME_DEVICE: rehearsal of the sharding on a one-GPU box (all children on the base device)
      const int dev = getenv("SIMUSCOP_SAME_DEVICE") ? base_device : base_device + r;
      a.insert(a.end(), {"--rank", std::to_string(r), "--world", std::to_string(gpus), "--device", std::to_string(dev)});
      if (shard_contigs) {
        close(up[0]); close(down[1]);
        for (int q = 0; q < r; q++) { close(from_child[(size_t)q]); close(to_child[(size_t)q]); }
        a.insert(a.end(), {"--shard-contigs", "--exchange-fds", std::to_string(up[1]) + "," + std::to_string(down[0])});
      }
      if (gzip && merge) a.push_back("--no-eof-block");
      if (r > 0) a.push_back("--quiet");
      std::vector<char*> cv;
      cv.push_back(const_cast<char*>(self.c_str()));
      for (std::string& x : a) cv.push_back(const_cast<char*>(x.c_str()));
      cv.push_back(nullptr);
      execv(self.c_str(), cv.data());
      _exit(127);
    }
    if (shard_contigs) { close(up[1]); close(down[0]); from_child[(size_t)r] = up[0]; to_child[(size_t)r] = down[1]; }
    kids.push_back(pid);
  }
  // the exchange hub: rounds of (count, values) from every child, element-wise sums back; ends when the children exit
  if (shard_contigs) {
    std::vector<double> sum, part;
    for (;;) {
      bool ok = true, any = false;
      int32_t n0 = -1;
      sum.clear();
      for (int r = 0; r < gpus && ok; r++) {
        int32_t n = 0;
        if (!read_all(from_child[(size_t)r], &n, sizeof n)) { ok = false; break; }
        any = true;
        if (n0 < 0) { n0 = n; sum.assign((size_t)n, 0.0); }
        if (n != n0 || n < 0) { ok = false; break; }
        part.resize((size_t)n);
        if (!read_all(from_child[(size_t)r], part.data(), sizeof(double) * (size_t)n)) { ok = false; break; }
        for (int32_t i = 0; i < n; i++) sum[(size_t)i] += part[(size_t)i];
      }
      if (!ok) { (void)any; break; }   // EOF: the children are done (or one died; its exit code tells)
      for (int r = 0; r < gpus; r++) write_all(to_child[(size_t)r], sum.data(), sizeof(double) * sum.size());
    }
    for (int r = 0; r < gpus; r++) { close(from_child[(size_t)r]); close(to_child[(size_t)r]); }
  }
  int rc = 0;
  for (pid_t pid : kids) {
    int st = 0;
    waitpid(pid, &st, 0);
    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
    if (code != 0 && rc == 0) rc = code;
  }
  if (rc != 0 || !merge) return rc;
  // merge exactly <file>.part0 .. part<N-1> of this run
  std::vector<std::string> bases;
  if (DIR* d = opendir(dir.c_str())) {
    while (dirent* e = readdir(d)) {
      const std::string n = e->d_name;
      if (n.size() > 6 && n.compare(n.size() - 6, 6, ".part0") == 0) bases.push_back(n.substr(0, n.size() - 6));
    }
    closedir(d);
  }
  std::sort(bases.begin(), bases.end());
  std::vector<char> buf(16u << 20);
  for (const std::string& base : bases) {
    FILE* dst = fopen((dir + "/" + base).c_str(), "wb");
    if (!dst) { std::cerr << "Error: can not open fastq file to save results:\n" << dir + "/" + base << std::endl; return -1; }
    for (int r = 0; r < gpus; r++) {
      const std::string path = dir + "/" + base + ".part" + std::to_string(r);
      FILE* src = fopen(path.c_str(), "rb");
      if (!src) { std::cerr << "Error: part file missing:\n" << path << std::endl; fclose(dst); return -1; }
      size_t got;
      while ((got = fread(buf.data(), 1, buf.size(), src)) > 0) fwrite(buf.data(), 1, got, dst);
      fclose(src);
      unlink(path.c_str());
    }
    if (gzip) {  // one BGZF end-of-file block for the whole file (the parts were written without theirs)
      uint8_t eof[28];
      sg_bgzf_eof(eof);
      fwrite(eof, 1, 28, dst);
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
    else if (a == "--shard-contigs") opt.shard_contigs = 1;
    else if (a == "--no-eof-block") opt.no_eof_block = 1;
    else if (a == "--crlf-as-lf") opt.crlf_as_lf = 1;
    else if (a == "--strict-bases") opt.strict_bases = 1;
    else if (a == "--unique-contigs") opt.unique_contigs = 1;
    else if (a == "--exchange-fds") {  // set by the --gpus parent: "<write fd>,<read fd>" of this child's pipe pair
      const char* v = val();
      static PipeExchange px;
      if (sscanf(v, "%d,%d", &px.up, &px.down) != 2) { std::cerr << "Error: bad --exchange-fds" << std::endl; return 1; }
      opt.exchange = pipe_exchange;
      opt.exchange_user = &px;
    }
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
      if (a == "--shard-contigs") continue;
      pass.push_back(a);
    }
    char self[4096];
    const ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    if (n <= 0) { std::cerr << "Error: cannot locate the executable for --gpus" << std::endl; return 1; }
    self[n] = 0;
    return run_on_gpus(gpus, opt.device >= 0 ? opt.device : 0, pass, self, config, out, opt.write_files != 0, opt.shard_contigs != 0,
                       opt.gzip != 0);
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
