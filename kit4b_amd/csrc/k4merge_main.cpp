// kit4b_amd/csrc/k4merge_main.cpp -- `k4merge out.sam shard0.sam shard1.sam ...`: merges the coordinate-sorted SAM files
// that N `k4align -S i/N` processes (one per GPU, SURVEY.md 8(e): "ranks write SAM shards and the host merges") wrote
// into one coordinate-sorted file; the rules are in k4_merge.h (`k4align -G` runs the same merge itself).
#include "k4_merge.h"

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "k4merge out.sam shard0.sam [shard1.sam ...]\n"); return 1; }
  std::vector<std::string> shards(argv + 2, argv + argc);
  unsigned long long n = 0;
  const int rc = k4merge::merge_sam(shards, argv[1], 10000, &n, "k4merge");
  if (rc) return rc;
  fprintf(stderr, "k4merge: %llu alignments from %d shards written to %s\n", n, (int)shards.size(), argv[1]);
  return 0;
}
