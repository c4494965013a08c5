// host/profile.h -- sequencing profile (error / quality / insert-size / GC model) loader.
// Follows Profile::train(file) = load + normParas(true) + initCDFs
// (lib/profile/Profile.cpp:934-1238, :836-932, :1367-1434), but keeps everything in flat fp64
// arrays laid out the way include/simuscop_amd.h's sg_profile_cdf expects.
#pragma once
#include <string>
#include <vector>

#include "../../../include/simuscop_amd.h"
#include "common.h"

namespace simu {

struct Profile {
  std::string bases = "ACTG";
  int n_bases = 4, kmer = 0, bins = 0, read_length = 0, kmer_count = 0;
  int min_qual = 33, n_qual = 94;
  double insert_rate = 0, del_rate = 0, std_isize = 0, gc_std = 0;
  double gc_means[101];
  std::vector<double> ins_cdf, del_cdf, subs1, subs2, qual, isize_cdf;
  bool has_sub2 = false;
  int isize_min = 0, insert_size = 350;
  bool paired = false;

  void train(const std::string& file, bool paired, int insert_size);
  int max_insert_size() const { return isize_cdf.empty() ? insert_size : isize_min + (int)isize_cdf.size() - 1; }
  // view for sg_load_profile (pointers stay valid while *this lives)
  sg_profile_cdf view() const;
  // The standard normal behind the GC factor (Profile::getGCFactor, Profile.cpp:1507-1517) as a quantile table the
  // device interpolates (sg_window_weights): 2^14 cells of equal probability, knots Phi^-1(k / 2^14) by bisection on
  // erfc (DESIGN.md section 4 "GC factor").
  std::vector<double> gc_quantiles;
  void build_gc_quantiles();

 private:
  int kmer_index(const std::string& s) const;  // Profile::initKmers order (Profile.cpp:70-124)
  int base_index(char c) const;
};

}  // namespace simu
