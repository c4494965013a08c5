// host/reader_pool.h -- reader threads that live for one stream of a file (reference ingest, SAM text of the trainer).
#pragma once
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace simu {

// Reader threads that live for one ingest: the file goes through two 64 MB staging buffers, ~50 chunks of a human
// genome, and a thread started and joined per slice of every chunk was 700 thread starts -- a fifth of the 55 ms the
// preads of 3.1 GB take.  run() hands every thread its share of one job and returns when all are done.
class ReaderPool {
 public:
  explicit ReaderPool(int n) : n_(std::max(1, n)) {
    for (int i = 0; i < n_; i++) threads_.emplace_back([this, i]() { loop(i); });
  }
  ~ReaderPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_++; }
    go_.notify_all();
    for (std::thread& t : threads_) t.join();
  }
  int size() const { return n_; }
  // job(i) on every thread i; throws the first error a thread met
  void run(const std::function<void(int)>& job) {
    std::unique_lock<std::mutex> lk(mu_);
    job_ = &job; pending_ = n_; error_.clear(); gen_++;
    go_.notify_all();
    done_.wait(lk, [this]() { return pending_ == 0; });
    job_ = nullptr;
    if (!error_.empty()) throw Error(error_);
  }
 private:
  void loop(int i) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        go_.wait(lk, [&]() { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        job = job_;
      }
      std::string err;
      try { (*job)(i); } catch (const std::exception& e) { err = e.what(); }
      std::lock_guard<std::mutex> lk(mu_);
      if (!err.empty() && error_.empty()) error_ = err;
      if (--pending_ == 0) done_.notify_all();
    }
  }
  int n_;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable go_, done_;
  const std::function<void(int)>* job_ = nullptr;
  uint64_t gen_ = 0;
  int pending_ = 0;
  bool stop_ = false;
  std::string error_;
};

// pread of [off, off+n) split over the pool's readers (at least 4 MB each)
inline void parallel_pread(ReaderPool& pool, int fd, uint8_t* dst, uint64_t off, uint64_t n) {
  auto one = [&](uint64_t a, uint64_t b) {
    while (a < b) {
      ssize_t got = pread(fd, dst + (a - off), b - a, (off_t)a);
      if (got <= 0) throw Error("could not read the input file");
      a += (uint64_t)got;
    }
  };
  const uint64_t kMin = 4u << 20;
  const int t = (int)std::min<uint64_t>((uint64_t)pool.size(), (n + kMin - 1) / kMin);
  if (t <= 1) { one(off, off + n); return; }
  const uint64_t per = (n + t - 1) / t;
  pool.run([&](int i) {
    if (i < t) one(off + std::min<uint64_t>(n, per * i), off + std::min<uint64_t>(n, per * (i + 1)));
  });
}

}  // namespace simu
