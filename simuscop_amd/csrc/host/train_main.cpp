// host/train_main.cpp -- `seqToProfile`: the command line of the reference's profile trainer (src/seqToProfile.cpp:19-147;
// same options, same checks, same exit codes) over the GPU path of host/train.cpp.
// Additive options the reference would reject:
//   --sam <file|->   lines of `samtools view -F 0xD04 -q 20` text from a file or standard input (instead of running samtools)
//   --max-reads <n>  Profile::processRead's cap on counted reads (default: the reference's 300,000,000; twice that with targets)
//   --device <n>     GPU to use (default 0)        --quiet   no progress lines        --stats   one JSON line of counts and times
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "train.h"

static void usage(const char* app) {
  std::cerr << "\nUsage: " << app << " [options]\n\n"
            << "Options:\n"
            << "    -h, --help                      give this information\n"
            << "    -b, --bam <string>              normal BAM file\n"
            << "    -t, --target <string>           exome target file (.bed) for whole-exome sequencing[default:null]\n"
            << "    -v, --vcf <string>              the VCF file generated from the normal BAM\n"
            << "    -r, --ref <string>              genome reference file (.fasta) to which the reads were aligned\n"
            << "    -o, --output <string>           output file\n"
            << "    -s, --samtools <string>         the path of samtools [default:samtools]\n"
            << "    -k, --kmer <int>                the length of kmer sequence [default:3]\n"
            << "    -B, --bins <int>                the number of bins into which bases of read are grouped [default:50]\n"
            << "        --sam <file|->              (GPU build) reads as `samtools view` text from a file or standard input\n"
            << "        --device <int>              (GPU build) device to use [default:0]\n"
            << "        --quiet                     (GPU build) no progress lines\n"
            << "        --max-reads <int>           (GPU build) stop at so many counted reads [default: the reference's 300000000,\n"
            << "                                    twice that with targets]\n\n"
            << "Example:\n"
            << "    " << app << " -b normal.bam -v normal.vcf -r ref.fa -o results.model -s /path/to/samtools\n"
            << "    samtools view -F 0xD04 -q 20 normal.bam | " << app << " --sam - -v normal.vcf -r ref.fa > results.model\n\n";
}

int main(int argc, char* argv[]) {
  simu_train_options o;
  simu_train_default_options(&o);
  std::string bam, sam, target, vcf, ref, out, samtools;
  bool stats = false;
  const struct option long_options[] = {
      {"help", no_argument, 0, 'h'},        {"bam", required_argument, 0, 'b'},    {"target", required_argument, 0, 't'},
      {"vcf", required_argument, 0, 'v'},   {"ref", required_argument, 0, 'r'},    {"output", required_argument, 0, 'o'},
      {"samtools", required_argument, 0, 's'}, {"kmer", required_argument, 0, 'k'}, {"bins", required_argument, 0, 'B'},
      {"sam", required_argument, 0, 1000},  {"device", required_argument, 0, 1001}, {"quiet", no_argument, 0, 1002},   {"stats", no_argument, 0, 1003},      {"max-reads", required_argument, 0, 1004},
      {0, 0, 0, 0}};
  int c;
  while ((c = getopt_long(argc, argv, "hb:t:v:r:o:s:k:B:", long_options, NULL)) != -1) {
    switch (c) {
      case 'h': usage(argv[0]); return 0;
      case 'b': bam = optarg; break;
      case 't': target = optarg; break;
      case 'v': vcf = optarg; break;
      case 'r': ref = optarg; break;
      case 'o': out = optarg; break;
      case 's': samtools = optarg; break;
      case 'k': o.kmer = atoi(optarg); break;
      case 'B': o.bins = atoi(optarg); break;
      case 1000: sam = optarg; break;
      case 1001: o.device = atoi(optarg); break;
      case 1002: o.quiet = 1; break;
      case 1003: stats = true; break;
      case 1004: o.max_reads = strtoull(optarg, nullptr, 10); break;
      default: usage(argv[0]); return 1;
    }
  }
  if (bam.empty() && sam.empty()) {
    std::cerr << "Use --bam to specify a normal BAM file." << std::endl;
    usage(argv[0]);
    return 1;
  }
  if (vcf.empty()) {
    std::cerr << "Use --vcf to specify the VCF file generated from the normal BAM." << std::endl;
    usage(argv[0]);
    return 1;
  }
  if (ref.empty()) {
    std::cerr << "Use --ref to specify the reference file(.fasta) to which the reads are aligned." << std::endl;
    usage(argv[0]);
    return 1;
  }
  if (samtools.empty() && sam.empty()) {
    std::cerr << "\nWarning: the path of samtools not specified!" << std::endl;
    std::cerr << "Assume the tool has been installed and included in the system PATH!" << std::endl;
  }
  if (o.kmer < 1 || o.kmer > 5) {
    std::cerr << "Error: parameter \"kmer\" should be a positive integer with maximum value of 5!" << std::endl;
    return 1;
  }
  if (o.bins < 10) {
    std::cerr << "Error: parameter \"bins\" should be a positive integer with minimum value of 10!" << std::endl;
    return 1;
  }
  o.bam = bam.c_str(); o.sam = sam.c_str(); o.target = target.c_str(); o.vcf = vcf.c_str(); o.ref = ref.c_str();
  o.output = out.c_str(); o.samtools = samtools.c_str();
  simu_train_stats st;
  char err[4096] = "";
  const int rc = simu_train(&o, &st, err, sizeof err);
  if (rc != 0) {
    if (err[0]) std::cerr << err << std::endl;
    return rc;
  }
  if (stats)
    fprintf(stderr, "{\"lines\": %llu, \"reads_counted\": %llu, \"gc_rejected\": %llu, \"gc_windows\": %llu, \"gc_pairs\": %llu, \"skipped_overhang\": %llu, "
                    "\"sam_bytes\": %llu, \"read_length\": %d, \"bins\": %d, \"gc_fitted\": %d, \"capped\": %d, \"t_reference\": %.4f, \"t_reads\": %.4f, \"t_total\": %.4f}\n",
            (unsigned long long)st.lines, (unsigned long long)st.reads_counted, (unsigned long long)st.gc_rejected, (unsigned long long)st.gc_windows,
            (unsigned long long)st.gc_pairs, (unsigned long long)st.skipped_overhang, (unsigned long long)st.sam_bytes, st.read_length, st.bins, st.gc_fitted, st.capped,
            st.t_reference, st.t_reads, st.t_total);
  if (!o.quiet) {
    const long secs = (long)st.t_total;
    std::cerr << "\nElapsed time: " << secs / 60 << " minutes and " << secs % 60 << " seconds!\n" << std::endl;
  }
  return 0;
}
