// kit4b_amd/csrc/k4merge_main.cpp -- `k4merge [-t threads] out.sam shard0.sam shard1.sam ...`: merges the coordinate-sorted SAM files
// that N `k4align -S i/N` processes (one per GPU, SURVEY.md 8(e): "ranks write SAM shards and the host merges") wrote
// into one coordinate-sorted file, on `threads` host threads (default: every hardware thread, at most 32); the rules are in
// k4_merge.h (`k4align -G` runs the same merge itself).
#include <chrono>
#include "k4_merge.h"

int main(int argc, char** argv) {
  int threads = 0, a0 = 1;
  if (argc > 2 && strcmp(argv[1], "-t") == 0) { threads = atoi(argv[2]); a0 = 3; }
  if (argc - a0 < 2) { fprintf(stderr, "k4merge [-t threads] out.sam shard0.sam [shard1.sam ...]\n"); return 1; }
  std::vector<std::string> shards(argv + a0 + 1, argv + argc);
  unsigned long long n = 0;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = k4merge::merge_sam(shards, argv[a0], 10000, &n, "k4merge", threads);
  if (rc) return rc;
  fprintf(stderr, "k4merge: %llu alignments from %d shards written to %s in %.2fs\n", n, (int)shards.size(), argv[a0],
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  return 0;
}
