// kit4b_amd/csrc/k4merge_main.cpp -- `k4merge out.sam shard0.sam shard1.sam ...`: merges the coordinate-sorted SAM files
// that N `k4align -S i/N` processes (one per GPU, SURVEY.md 8(e): "ranks write SAM shards and the host merges") wrote
// into one coordinate-sorted file.  Header of shard 0 is kept (all shards carry the same @SQ lines); records are merged
// by (RNAME in @SQ order, POS), equal keys in shard order -- i.e. in load order, since shard i holds the i-th slice.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <queue>
#include <string>
#include <vector>

struct Src {
  FILE* f = nullptr;
  std::string line;
  long chrom = 0, pos = 0;
  bool ok = false;
};

static bool read_line(FILE* f, std::string& s) {
  s.clear();
  char buf[1 << 16];
  while (fgets(buf, sizeof(buf), f)) {
    s += buf;
    if (!s.empty() && s.back() == '\n') return true;
  }
  return !s.empty();
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "k4merge out.sam shard0.sam [shard1.sam ...]\n"); return 1; }
  FILE* out = fopen(argv[1], "wb");
  if (!out) { fprintf(stderr, "k4merge: unable to create %s\n", argv[1]); return 2; }
  static char iobuf[1 << 22];
  setvbuf(out, iobuf, _IOFBF, sizeof(iobuf));
  const int ns = argc - 2;
  std::vector<Src> src((size_t)ns);
  std::map<std::string, long> order;
  auto key_of = [&](Src& s) -> bool {  // RNAME (field 3) and POS (field 4)
    size_t a = s.line.find('\t');
    if (a == std::string::npos) return false;
    size_t b = s.line.find('\t', a + 1);
    if (b == std::string::npos) return false;
    size_t c = s.line.find('\t', b + 1);
    if (c == std::string::npos) return false;
    size_t d = s.line.find('\t', c + 1);
    if (d == std::string::npos) return false;
    auto it = order.find(s.line.substr(b + 1, c - b - 1));
    s.chrom = it == order.end() ? (long)order.size() : it->second;
    s.pos = atol(s.line.c_str() + c + 1);
    return true;
  };
  for (int i = 0; i < ns; i++) {
    src[i].f = fopen(argv[2 + i], "rb");
    if (!src[i].f) { fprintf(stderr, "k4merge: unable to open %s\n", argv[2 + i]); return 2; }
    // header: shard 0's is written out and defines the chromosome order
    while ((src[i].ok = read_line(src[i].f, src[i].line)) && src[i].line[0] == '@') {
      if (i == 0) {
        fputs(src[i].line.c_str(), out);
        if (src[i].line.compare(0, 3, "@SQ") == 0) {
          size_t p = src[i].line.find("\tSN:");
          if (p != std::string::npos) {
            size_t e = src[i].line.find_first_of("\t\n", p + 4);
            order.emplace(src[i].line.substr(p + 4, e - p - 4), (long)order.size());
          }
        }
      }
    }
  }
  typedef std::pair<std::pair<long, long>, int> Item;  // ((chrom, pos), shard): smallest first, ties by shard
  std::priority_queue<Item, std::vector<Item>, std::greater<Item>> pq;
  for (int i = 0; i < ns; i++)
    if (src[i].ok && key_of(src[i])) pq.push({{src[i].chrom, src[i].pos}, i});
  unsigned long long n = 0;
  while (!pq.empty()) {
    const int i = pq.top().second;
    pq.pop();
    fputs(src[i].line.c_str(), out);
    n++;
    if ((src[i].ok = read_line(src[i].f, src[i].line)) && key_of(src[i])) pq.push({{src[i].chrom, src[i].pos}, i});
  }
  for (auto& s : src) fclose(s.f);
  fclose(out);
  fprintf(stderr, "k4merge: %llu alignments from %d shards written to %s\n", n, ns, argv[1]);
  return 0;
}
