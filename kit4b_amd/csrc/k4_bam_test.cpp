// kit4b_amd/csrc/k4_bam_test.cpp -- include/k4_bam.hpp from the command line, for tests/test_bam_cpu.py (no GPU, no HIP):
//   k4_bam_test <out.bam> <header.txt> <refs.tsv: name \t length per line> <records.bin: uncompressed BAM records> <piece bytes> <threads> <level>
// feeds the record stream to k4bam::Writer in pieces of the given size (records straddle them) and writes out.bam + out.bam.bai
#include <stdio.h>
#include <stdlib.h>
#include <fstream>
#include <iterator>
#include <sstream>
#include "../../include/k4_bam.hpp"

static std::string slurp(const char* p) {
  std::ifstream f(p, std::ios::binary);
  return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc != 8) { fprintf(stderr, "usage: k4_bam_test out.bam header.txt refs.tsv records.bin piece threads level\n"); return 2; }
  const std::string hdr = slurp(argv[2]), recs = slurp(argv[4]);
  std::vector<k4bam::RefSeq> refs;
  std::istringstream rs(slurp(argv[3]));
  std::string line;
  while (std::getline(rs, line)) {
    const size_t t = line.find('\t');
    if (t == std::string::npos) continue;
    refs.push_back({line.substr(0, t), (uint32_t)strtoul(line.c_str() + t + 1, nullptr, 10)});
  }
  const size_t piece = (size_t)strtoull(argv[5], nullptr, 10);
  k4bam::Writer w;
  if (!w.open(argv[1], hdr, refs, atoi(argv[7]), atoi(argv[6]))) { fprintf(stderr, "%s\n", w.error().c_str()); return 1; }
  for (size_t o = 0; o < recs.size(); o += piece)
    if (!w.write(recs.data() + o, std::min(piece, recs.size() - o))) { fprintf(stderr, "%s\n", w.error().c_str()); return 1; }
  if (!w.close()) { fprintf(stderr, "%s\n", w.error().c_str()); return 1; }
  printf("%llu records\n", (unsigned long long)w.n_records());
  return 0;
}
