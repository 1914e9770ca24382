// kit4b_amd/csrc/k4_merge.h -- host-only: merges coordinate-sorted SAM shards (one per GPU rank / read slice) into one
// coordinate-sorted SAM ("ranks write SAM shards and the host merges").  Used by `k4merge` and by the parent
// process of `k4align -G`.
//   * @SQ lines: the union over all shard headers in first-appearance order (every shard of k4align -S / -G carries ALL of
//     them, in index order); with more than `sq_rule` sequences only those that received a record are kept -- the
//     reference's rule for big assemblies (m_MaxRptSAMSeqsThres, KAligner.cpp:5785-5821), which a single run applies too.
//   * order: (RNAME in @SQ order, POS, aligned length of the first CIGAR block group, strand), then shard index -- shard i holds
//     the i-th contiguous slice of the reads, so equal keys stay in load order as far as the text tells (the mismatch count,
//     SortHitMatch's last key, is not part of a SAM line).
//   * a record whose RNAME no header names, or a failed write, is an error.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <queue>
#include <string>
#include <tuple>
#include <vector>

namespace k4merge {

struct Src {
  FILE* f = nullptr;
  std::string line;
  long chrom = 0, pos = 0, len = 0, strand = 0;
  bool ok = false;
};

inline bool read_line(FILE* f, std::string& s) {
  s.clear();
  char buf[1 << 16];
  while (fgets(buf, sizeof(buf), f)) {
    s += buf;
    if (!s.empty() && s.back() == '\n') return true;
  }
  return !s.empty();
}

// returns 0, or an exit code with the message on stderr; *n_out = records written
inline int merge_sam(const std::vector<std::string>& shards, const std::string& out_path, size_t sq_rule, unsigned long long* n_out,
                     const char* who) {
  const int ns = (int)shards.size();
  std::vector<Src> src((size_t)ns);
  std::map<std::string, long> order;
  std::vector<std::string> sq_lines, other_hdr;
  for (int i = 0; i < ns; i++) {
    src[i].f = fopen(shards[i].c_str(), "rb");
    if (!src[i].f) { fprintf(stderr, "%s: unable to open %s\n", who, shards[i].c_str()); return 2; }
    while ((src[i].ok = read_line(src[i].f, src[i].line)) && src[i].line[0] == '@') {
      if (src[i].line.compare(0, 3, "@SQ") == 0) {
        size_t p = src[i].line.find("\tSN:");
        if (p == std::string::npos) continue;
        size_t e = src[i].line.find_first_of("\t\n", p + 4);
        if (order.emplace(src[i].line.substr(p + 4, e - p - 4), (long)order.size()).second) sq_lines.push_back(src[i].line);
      } else if (i == 0)
        other_hdr.push_back(src[i].line);
    }
  }
  int bad = 0;
  auto key_of = [&](Src& s) -> bool {  // FLAG (2), RNAME (3), POS (4), CIGAR (6)
    size_t t[6];
    size_t p = 0;
    for (int k = 0; k < 6; k++) {
      t[k] = s.line.find('\t', p);
      if (t[k] == std::string::npos) { bad = 3; return false; }
      p = t[k] + 1;
    }
    auto it = order.find(s.line.substr(t[1] + 1, t[2] - t[1] - 1));
    if (it == order.end()) {
      fprintf(stderr, "%s: a record names '%s', which no @SQ line declares\n", who, s.line.substr(t[1] + 1, t[2] - t[1] - 1).c_str());
      bad = 3;
      return false;
    }
    s.chrom = it->second;
    s.pos = atol(s.line.c_str() + t[2] + 1);
    s.strand = (atol(s.line.c_str() + t[0] + 1) & 0x10) ? 1 : 0;  // '+' (43) sorts before '-' (45)
    // AdjHitLen(Seg[0]): the M block that follows an optional leading soft clip
    const char* c = s.line.c_str() + t[4] + 1;
    long v = strtol(c, (char**)&c, 10);
    if (*c == 'S') v = strtol(c + 1, (char**)&c, 10);
    s.len = v;
    return true;
  };
  // with the hit-only rule the header depends on the records: a first pass over the shards marks the sequences in use
  std::vector<char> used(order.size(), 1);
  if (order.size() > sq_rule) {
    std::fill(used.begin(), used.end(), 0);
    for (int i = 0; i < ns; i++) {
      Src s;
      s.f = fopen(shards[i].c_str(), "rb");
      if (!s.f) return 2;
      while (read_line(s.f, s.line))
        if (s.line[0] != '@') { if (!key_of(s)) { fclose(s.f); return 3; } used[(size_t)s.chrom] = 1; }
      fclose(s.f);
    }
  }
  FILE* out = fopen(out_path.c_str(), "wb");
  if (!out) { fprintf(stderr, "%s: unable to create %s\n", who, out_path.c_str()); return 2; }
  static char iobuf[1 << 22];
  setvbuf(out, iobuf, _IOFBF, sizeof(iobuf));
  bool werr = false;
  auto put = [&](const std::string& l) { if (fputs(l.c_str(), out) < 0) werr = true; };
  for (const std::string& l : other_hdr)
    if (l.compare(0, 3, "@HD") == 0) put(l);
  for (size_t k = 0; k < sq_lines.size(); k++)
    if (used[k]) put(sq_lines[k]);
  for (const std::string& l : other_hdr)
    if (l.compare(0, 3, "@HD") != 0) put(l);
  typedef std::tuple<long, long, long, long, int> Key;  // chrom, pos, len, strand, shard
  std::priority_queue<Key, std::vector<Key>, std::greater<Key>> pq;
  auto push = [&](int i) { if (src[i].ok && key_of(src[i])) pq.push(Key(src[i].chrom, src[i].pos, src[i].len, src[i].strand, i)); };
  for (int i = 0; i < ns; i++) push(i);
  unsigned long long n = 0;
  while (!pq.empty() && !bad && !werr) {
    const int i = std::get<4>(pq.top());
    pq.pop();
    put(src[i].line);
    n++;
    src[i].ok = read_line(src[i].f, src[i].line);
    push(i);
  }
  for (auto& s : src) fclose(s.f);
  if (fflush(out) != 0 || ferror(out)) werr = true;
  if (fclose(out) != 0) werr = true;
  if (werr) { fprintf(stderr, "%s: write to %s failed\n", who, out_path.c_str()); return 5; }
  if (bad) return bad;
  if (n_out) *n_out = n;
  return 0;
}

}  // namespace k4merge
