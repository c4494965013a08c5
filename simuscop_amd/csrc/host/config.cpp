// host/config.cpp -- same keys, defaults, validation order and messages as the reference
// (lib/config/Config.cpp:16-42 defaults, :46-99 parser, :101-175 checks).  Additive keys the
// reference would reject (Config.cpp:90-94): `seed`, `device` -- documented in INTEGRATION.md.
#include "config.h"

#include <cstdlib>
#include <fstream>

namespace simu {

Config::Config() {
  for (const char* k : {"bam", "profile", "ref", "variation", "snp", "vcf", "target", "bases", "output", "abundance",
                        "layout", "samtools"})
    str[k] = "";
  str["layout"] = "SE";
  str["bases"] = "ACTG";
  num = {{"kmer", 0}, {"bins", 0}, {"threads", 1}, {"verbose", 1}, {"readLength", 0}, {"coverage", 0},
         {"ploidy", 2}, {"insertSize", 350}, {"seed", 1500000000LL * 4294967296LL + 123456789LL}, {"device", 0}};
  real["indelRate"] = 0.00025;
}

void Config::load(const std::string& file) {
  if (file.empty()) throw Error("Error: configuration file not specified!", -1);
  std::ifstream ifs(file.c_str());
  if (!ifs.is_open()) throw Error("Error: can not open configuration file" + file, -1);
  std::string line;
  int line_num = 0;
  while (std::getline(ifs, line)) {
    line_num++;
    line = trim(line);
    if (line.empty() || line[0] == '#') continue;
    size_t eq = line.find('=');
    if (eq == std::string::npos)
      throw Error("ERROR: line " + std::to_string(line_num) + " is incorrectly formatted in file " + file + "\n" + line);
    std::string key = trim(line.substr(0, eq)), value = trim(line.substr(eq + 1));
    if (str.count(key)) str[key] = value;
    else if (key == "seed") num[key] = (long long)strtoull(value.c_str(), nullptr, 10);
    else if (num.count(key)) num[key] = atoi(value.c_str());
    else if (real.count(key)) real[key] = atof(value.c_str());
    else if (key == "name") {
      popu_names = split(value, ',');
      for (auto& p : popu_names) p = trim(p);
    } else {
      throw Error("ERROR: unrecognized item \"" + key + "\" @line " + std::to_string(line_num) + " in file " + file + "\n" + line);
    }
  }
  // Config::checkParas
  if (str["profile"].empty()) throw Error("Error: sequencing profile must be specified!");
  if (str["ref"].empty()) throw Error("Error: reference file not specified!");
  if (popu_names.empty()) throw Error("Error: population names not specified!");
  if (popu_names.size() > 1 && str["abundance"].empty()) throw Error("Error: abundance file not specified!");
  if (str["output"].empty()) throw Error("Error: output directory not specified!");
  if (str["layout"].empty()) str["layout"] = "SE";
  else if (str["layout"] != "SE" && str["layout"] != "PE")
    throw Error("Error: sequence layout incorrectly specified!\nshould be SE or PE");
  if (num["threads"] < 1) throw Error("Error: number of threads should be a positive integer!");
  if (num["coverage"] < 1) throw Error("Error: sequence coverage should be a positive integer!");
  if (num["ploidy"] < 1) throw Error("Error: genome ploidy should be a positive integer!");
  if (str["layout"] == "PE" && num["insertSize"] < num["readLength"])
    throw Error("Error: insert size should be not smaller than read length!");
  if (real["indelRate"] < 0 || real["indelRate"] > 0.001)
    throw Error("Error: indel error rate should be a value between 0 to 0.001!");
}

}  // namespace simu
