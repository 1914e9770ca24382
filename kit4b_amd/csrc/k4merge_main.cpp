// kit4b_amd/csrc/k4merge_main.cpp -- `k4merge [-t threads] out.sam shard0.sam shard1.sam ...`: merges the coordinate-sorted SAM files
// that N `k4align -S i/N` processes (one per GPU, SURVEY.md 8(e): "ranks write SAM shards and the host merges") wrote
// into one coordinate-sorted file, on `threads` host threads (default: every hardware thread, at most 32); the rules are in
// k4_merge.h (`k4align -G` runs the same merge itself).
#include <chrono>
#include "k4_merge.h"

int main(int argc, char** argv) {
  int threads = 0, a0 = 1;
  if (argc > 2 && strcmp(argv[1], "-t") == 0) { threads = atoi(argv[2]); a0 = 3; }
  if (argc > 3 && strcmp(argv[1], "--bam-records") == 0) {
    // the merge `k4align -G -o x.bam` runs over its ranks' record streams, on its own: uncompressed BAM records in, the merged
    // stream out (no header, no BGZF)
    std::vector<std::string> in(argv + 3, argv + argc);
    FILE* fo = fopen(argv[2], "wb");
    if (!fo) { fprintf(stderr, "k4merge: unable to create %s\n", argv[2]); return 5; }
    unsigned long long n = 0;
    std::string why;
    const int rc = k4merge::merge_bam_records(in, nullptr, [&](const void* p, size_t len) { return fwrite(p, 1, len, fo) == len; }, &n, &why);
    if (fclose(fo) != 0 || rc) { fprintf(stderr, "k4merge: %s\n", why.empty() ? "write failed" : why.c_str()); remove(argv[2]); return rc ? rc : 5; }
    fprintf(stderr, "k4merge: %llu BAM records from %d streams written to %s\n", n, (int)in.size(), argv[2]);
    return 0;
  }
  if (argc - a0 < 2) { fprintf(stderr, "k4merge [-t threads] out.sam shard0.sam [shard1.sam ...]\n        k4merge --bam-records out.rec rank0.rec [rank1.rec ...]\n"); return 1; }
  std::vector<std::string> shards(argv + a0 + 1, argv + argc);
  unsigned long long n = 0;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = k4merge::merge_sam(shards, argv[a0], 10000, &n, "k4merge", threads);
  if (rc) return rc;
  fprintf(stderr, "k4merge: %llu alignments from %d shards written to %s in %.2fs\n", n, (int)shards.size(), argv[a0],
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  return 0;
}
