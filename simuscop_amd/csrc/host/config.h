// host/config.h -- SimuSCoP's `key = value` configuration surface (lib/config/Config.cpp:14-175).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace simu {

struct Config {
  std::map<std::string, std::string> str;
  std::map<std::string, long long> num;
  std::map<std::string, double> real;
  std::vector<std::string> popu_names;

  Config();
  void load(const std::string& file);  // Config::loadConfig + checkParas
  bool paired() const { return str.at("layout") == "PE"; }
  bool verbose() const { return num.at("verbose") != 0; }
  int ploidy() const { return (int)num.at("ploidy"); }
};

}  // namespace simu
