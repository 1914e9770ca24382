// kit4b_amd/csrc/k4_merge.h -- host-only: merges coordinate-sorted SAM shards (one per GPU rank / read slice) into one
// coordinate-sorted SAM ("ranks write SAM shards and the host merges").  Used by `k4merge` and by the parent
// process of `k4align -G`.  The reference sorts once globally on all its threads (SortHitMatch through CMTqsort,
// ngskit4b/KAligner.cpp:10870,10969-11014); the shards here are sorted already, so what is left is a k-way merge -- done in
// PARALLEL: the key space is cut at T - 1 splitters (sampled from the shards), every shard is binary-searched for each
// splitter (it is a sorted text: bisect on byte offsets, step to the next line start), which gives every thread one byte range
// per shard and -- summed -- the exact place of its output in the file; the threads then merge their ranges independently and
// write with pwrite().  No byte is read twice, no line is copied into a std::string.
//   * @SQ lines: the union over all shard headers in first-appearance order (every shard of k4align -S / -G carries ALL of
//     them, in index order); with more than `sq_rule` sequences only those that received a record are kept -- the
//     reference's rule for big assemblies (m_MaxRptSAMSeqsThres, KAligner.cpp:5785-5821), which a single run applies too
//     (then a parallel pass over the RNAME fields comes first: the header's size decides where the body starts).
//   * order: (RNAME in @SQ order, POS, aligned length of the first CIGAR block group, strand), then shard index -- shard i holds
//     the i-th contiguous slice of the reads, so equal keys stay in load order as far as the text tells (the mismatch count,
//     SortHitMatch's last key, is not part of a SAM line).  Records with equal keys never straddle a splitter: all of them go
//     to the partition that starts with that key, in every shard.
//   * a record whose RNAME no header names, or a failed write, is an error; the partial output is removed.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace k4merge {

typedef std::tuple<long, long, long, long> Key;  // chrom (header order), pos, len, strand

struct Shard {
  const char* p = nullptr;  // the mapped file
  size_t len = 0, body = 0; // body: offset of the first record
};

struct Names {  // RNAME -> position in the merged header
  std::unordered_map<std::string, long> order;
  std::vector<std::string> sq_lines;
};

// FLAG (2), RNAME (3), POS (4), CIGAR (6) of the record that starts at `l` (ends before `e`).  last_*: the previous lookup of the
// calling thread (consecutive records mostly share their RNAME).  Returns false on a malformed record / unknown RNAME.
static const long kUnplaced = 0x7fffffffffffffffL;  // RNAME '*': behind every sequence
// the NAR abbreviations in the order of their codes (KAligner.h teNAR; k4align's tally lines): -M1 lists the reads without an
// accepted alignment by ascending code, so the code takes the place of the position in such a record's key
inline long nar_code_of(const char* two) {
  static const char* const kAbbr[20] = {"NA", "AA", "EN", "NL", "MH", "ML", "ET", "OJ", "OM", "DP", "DS", "FC", "PR", "UI", "OI", "UP", "IS", "IT", "NP", "LC"};
  for (long k = 0; k < 20; k++)
    if (two[0] == kAbbr[k][0] && two[1] == kAbbr[k][1]) return k;
  return 20;
}
struct KeyReader {
  const Names* nm;
  const char* last_name = nullptr;
  size_t last_len = 0;
  long last_chrom = -1;
  std::string bad_name;
  bool key(const char* l, const char* e, Key& k) {
    const char* t[6];
    const char* p = l;
    for (int q = 0; q < 6; q++) {
      t[q] = (const char*)memchr(p, '\t', (size_t)(e - p));
      if (!t[q]) return false;
      p = t[q] + 1;
    }
    const char* name = t[1] + 1;
    const size_t nl = (size_t)(t[2] - name);
    long chrom;
    if (nl == 1 && name[0] == '*')
      chrom = kUnplaced;  // (-M1: the reads without an accepted alignment follow the alignments, in shard order)
    else if (last_name && nl == last_len && memcmp(name, last_name, nl) == 0)
      chrom = last_chrom;
    else {
      auto it = nm->order.find(std::string(name, nl));
      if (it == nm->order.end()) { bad_name.assign(name, nl); return false; }
      chrom = it->second;
      last_name = name; last_len = nl; last_chrom = chrom;
    }
    long pos = atol(t[2] + 1);
    if (chrom == kUnplaced) {  // "...\tYU:Z:<NAR>\n" ends the record
      const char* z = e;
      while (z > l && (z[-1] == '\n' || z[-1] == '\r')) z--;
      pos = (z - l >= 7 && memcmp(z - 7, "YU:Z:", 5) == 0) ? nar_code_of(z - 2) : 20;
    }
    const long strand = (atol(t[0] + 1) & 0x10) ? 1 : 0;  // '+' (43) sorts before '-' (45)
    // AdjHitLen(Seg[0]): the M block that follows an optional leading soft clip
    const char* c = t[4] + 1;
    long v = strtol(c, (char**)&c, 10);
    if (*c == 'S') v = strtol(c + 1, (char**)&c, 10);
    k = chrom == kUnplaced ? Key(chrom, pos, 0, 0) : Key(chrom, pos, v, strand);  // (within a NAR code: shard order, then order in the shard)
    return true;
  }
};

inline const char* line_end(const char* p, const char* e) {  // one past the record's newline (or e)
  const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
  return q ? q + 1 : e;
}

// offset of the first record of shard s (at or behind `from`) whose key is not below `k`; the body is sorted
inline size_t lower_bound_record(const Shard& s, size_t from, const Key& k, KeyReader& kr, bool& ok) {
  size_t lo = from, hi = s.len;  // invariant: lo is a record start (or len); every record in front of lo is below k; the record at hi (a
                                 // record start or len) is not below k
  while (lo < hi) {
    size_t mid = lo + (hi - lo) / 2;
    // step to the start of the record that holds mid
    const char* q = mid > lo ? (const char*)memrchr(s.p + lo, '\n', mid - lo) : nullptr;
    const size_t st = q ? (size_t)(q - s.p) + 1 : lo;
    Key mk;
    const char* le = line_end(s.p + st, s.p + s.len);
    if (!kr.key(s.p + st, le, mk)) { ok = false; return s.len; }
    if (mk < k) lo = (size_t)(le - s.p);
    else hi = st;
  }
  return lo;
}

inline bool pwrite_all(int fd, const char* p, size_t n, size_t off) {
  while (n) {
    const ssize_t w = pwrite(fd, p, n, (off_t)off);
    if (w <= 0) return false;
    p += w; n -= (size_t)w; off += (size_t)w;
  }
  return true;
}

// returns 0, or an exit code with the message on stderr; *n_out = records written.  threads <= 0: one per hardware thread (<= 32)
inline int merge_sam(const std::vector<std::string>& shards, const std::string& out_path, size_t sq_rule, unsigned long long* n_out,
                     const char* who, int threads = 0) {
  const int ns = (int)shards.size();
  if (threads <= 0) threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
  std::vector<Shard> sh((size_t)ns);
  auto unmap_all = [&]() { for (Shard& s : sh) if (s.p && s.len) munmap((void*)s.p, s.len); };
  Names nm;
  std::vector<std::string> other_hdr;
  for (int i = 0; i < ns; i++) {
    const int fd = open(shards[i].c_str(), O_RDONLY);
    struct stat stt;
    if (fd < 0 || fstat(fd, &stt) != 0) { if (fd >= 0) close(fd); fprintf(stderr, "%s: unable to open %s\n", who, shards[i].c_str()); unmap_all(); return 2; }
    sh[i].len = (size_t)stt.st_size;
    if (sh[i].len) {
      void* m = mmap(nullptr, sh[i].len, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { close(fd); fprintf(stderr, "%s: unable to map %s\n", who, shards[i].c_str()); sh[i].len = 0; unmap_all(); return 2; }
      sh[i].p = (const char*)m;
      madvise(m, sh[i].len, MADV_SEQUENTIAL);
    }
    close(fd);
    size_t o = 0;
    while (o < sh[i].len && sh[i].p[o] == '@') {
      const char* le = line_end(sh[i].p + o, sh[i].p + sh[i].len);
      std::string line(sh[i].p + o, (size_t)(le - (sh[i].p + o)));
      if (line.empty() || line.back() != '\n') line += '\n';
      if (line.compare(0, 3, "@SQ") == 0) {
        const size_t p = line.find("\tSN:");
        if (p != std::string::npos) {
          const size_t e = line.find_first_of("\t\n", p + 4);
          if (nm.order.emplace(line.substr(p + 4, e - p - 4), (long)nm.order.size()).second) nm.sq_lines.push_back(line);
        }
      } else if (i == 0)
        other_hdr.push_back(line);
      o = (size_t)(le - sh[i].p);
    }
    sh[i].body = o;
  }
  size_t body_bytes = 0;
  for (const Shard& s : sh) body_bytes += s.len - s.body;
  // (a shard whose last record lacks its newline would shift every offset behind it: k4align always writes it; checked here)
  for (int i = 0; i < ns; i++)
    if (sh[i].len > sh[i].body && sh[i].p[sh[i].len - 1] != '\n') { fprintf(stderr, "%s: %s does not end with a newline\n", who, shards[i].c_str()); unmap_all(); return 3; }
  const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, body_bytes / (4u << 20) + 1));
  std::atomic<int> bad(0);
  std::string bad_name;
  std::atomic<bool> bad_named(false);
  auto note_bad = [&](KeyReader& kr) {
    bad = 3;
    if (!kr.bad_name.empty() && !bad_named.exchange(true)) bad_name = kr.bad_name;
  };
  // ---- splitters: keys at evenly spaced byte positions of every shard, sorted; every (samples / T)-th becomes a cut
  std::vector<Key> cuts;  // T - 1 ascending keys
  if (T > 1) {
    std::vector<Key> sample;
    const int per = 16 * T;
    KeyReader kr{&nm};
    for (int i = 0; i < ns && !bad; i++) {
      const size_t bl = sh[i].len - sh[i].body;
      if (!bl) continue;
      for (int q = 0; q < per; q++) {
        const size_t at = sh[i].body + bl * (size_t)q / (size_t)per;
        const char* r = at > sh[i].body ? (const char*)memrchr(sh[i].p + sh[i].body, '\n', at - sh[i].body) : nullptr;
        const size_t st = r ? (size_t)(r - sh[i].p) + 1 : sh[i].body;
        Key k;
        if (!kr.key(sh[i].p + st, line_end(sh[i].p + st, sh[i].p + sh[i].len), k)) { note_bad(kr); break; }
        sample.push_back(k);
      }
    }
    std::sort(sample.begin(), sample.end());
    for (int c = 1; c < T && !sample.empty(); c++) cuts.push_back(sample[sample.size() * (size_t)c / (size_t)T]);
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
  }
  const int P = (int)cuts.size() + 1;  // partitions
  // ---- where every shard is cut: bound[i][c] = first record of shard i not below cut c - 1 (bound[i][0] = body, [P] = len)
  std::vector<std::vector<size_t>> bound((size_t)ns, std::vector<size_t>((size_t)P + 1, 0));
  {
    std::vector<std::thread> th;
    std::atomic<int> next(0);
    for (int t = 0; t < std::min(T, ns); t++)
      th.emplace_back([&] {
        KeyReader kr{&nm};
        for (int i; (i = next.fetch_add(1)) < ns;) {
          bound[i][0] = sh[i].body;
          bound[i][(size_t)P] = sh[i].len;
          for (int c = 1; c < P; c++) {
            bool ok = true;
            bound[i][(size_t)c] = lower_bound_record(sh[i], bound[i][(size_t)c - 1], cuts[(size_t)c - 1], kr, ok);
            if (!ok) { note_bad(kr); break; }
          }
        }
      });
    for (std::thread& x : th) x.join();
  }
  // ---- the header; with the hit-only rule it depends on the records: a parallel pass over the RNAME fields marks the sequences in use
  std::vector<char> used(nm.order.size(), 1);
  if (!bad && nm.order.size() > sq_rule) {
    std::fill(used.begin(), used.end(), 0);
    std::vector<std::vector<char>> mine((size_t)P, std::vector<char>(nm.order.size(), 0));
    std::vector<std::thread> th;
    std::atomic<int> next(0);
    for (int t = 0; t < T; t++)
      th.emplace_back([&] {
        KeyReader kr{&nm};
        for (int c; (c = next.fetch_add(1)) < P && !bad;)
          for (int i = 0; i < ns && !bad; i++)
            for (const char *l = sh[i].p + bound[i][(size_t)c], *e = sh[i].p + bound[i][(size_t)c + 1]; l < e;) {
              const char* le = line_end(l, e);
              Key k;
              if (!kr.key(l, le, k)) { note_bad(kr); break; }
              if (std::get<0>(k) != kUnplaced) mine[(size_t)c][(size_t)std::get<0>(k)] = 1;
              l = le;
            }
      });
    for (std::thread& x : th) x.join();
    for (const auto& m : mine)
      for (size_t k = 0; k < used.size(); k++) used[k] |= m[k];
  }
  if (bad) {
    if (bad_named) fprintf(stderr, "%s: a record names '%s', which no @SQ line declares\n", who, bad_name.c_str());
    else fprintf(stderr, "%s: malformed record in a shard\n", who);
    unmap_all();
    return bad;
  }
  std::string header;
  for (const std::string& l : other_hdr)
    if (l.compare(0, 3, "@HD") == 0) header += l;
  for (size_t k = 0; k < nm.sq_lines.size(); k++)
    if (used[k]) header += nm.sq_lines[k];
  for (const std::string& l : other_hdr)
    if (l.compare(0, 3, "@HD") != 0) header += l;
  const int out = open(out_path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (out < 0) { fprintf(stderr, "%s: unable to create %s\n", who, out_path.c_str()); unmap_all(); return 2; }
  std::vector<size_t> part_off((size_t)P + 1, header.size());
  for (int c = 0; c < P; c++) {
    size_t b = 0;
    for (int i = 0; i < ns; i++) b += bound[i][(size_t)c + 1] - bound[i][(size_t)c];
    part_off[(size_t)c + 1] = part_off[(size_t)c] + b;
  }
  std::atomic<bool> werr(false);
  if (!pwrite_all(out, header.data(), header.size(), 0)) werr = true;
  // ---- the partitions, merged side by side
  std::atomic<unsigned long long> n_rec(0);
  {
    std::vector<std::thread> th;
    std::atomic<int> next(0);
    for (int t = 0; t < std::min(T, P); t++)
      th.emplace_back([&] {
        KeyReader kr{&nm};
        std::vector<char> buf;
        buf.reserve((8u << 20) + (1u << 16));
        struct Cur { const char *l, *e, *le; Key k; int i; };
        for (int c; (c = next.fetch_add(1)) < P && !bad && !werr;) {
          size_t off = part_off[(size_t)c];
          std::vector<Cur> cur;
          for (int i = 0; i < ns; i++) {
            Cur u{sh[i].p + bound[i][(size_t)c], sh[i].p + bound[i][(size_t)c + 1], nullptr, Key(), i};
            if (u.l >= u.e) continue;
            u.le = line_end(u.l, u.e);
            if (!kr.key(u.l, u.le, u.k)) { note_bad(kr); break; }
            cur.push_back(u);
          }
          unsigned long long n = 0;
          while (!cur.empty() && !bad) {
            size_t m = 0;  // the lowest (key, shard): a linear scan -- there are as many runs as GPUs
            for (size_t q = 1; q < cur.size(); q++)
              if (cur[q].k < cur[m].k) m = q;  // (cur is in shard order: the first of equal keys wins)
            Cur& u = cur[m];
            // the winner's records are taken while they stay below every other run's head: mostly several per comparison round
            const Key* lim = nullptr;
            bool lim_incl = false;  // may records EQUAL to the limit still go first (the other run is a later shard)?
            for (size_t q = 0; q < cur.size(); q++)
              if (q != m && (!lim || cur[q].k < *lim || (cur[q].k == *lim && q < m && lim_incl))) { lim = &cur[q].k; lim_incl = q > m; }
            for (;;) {
              buf.insert(buf.end(), u.l, u.le);
              n++;
              u.l = u.le;
              if (u.l >= u.e) break;
              u.le = line_end(u.l, u.e);
              if (!kr.key(u.l, u.le, u.k)) { note_bad(kr); break; }
              if (lim && !(u.k < *lim || (lim_incl && u.k == *lim))) break;
            }
            if (u.l >= u.e) cur.erase(cur.begin() + (long)m);
            if (buf.size() >= (8u << 20)) {
              if (!pwrite_all(out, buf.data(), buf.size(), off)) { werr = true; break; }
              off += buf.size();
              buf.clear();
            }
          }
          if (!buf.empty() && !werr) {
            if (!pwrite_all(out, buf.data(), buf.size(), off)) werr = true;
            off += buf.size();
            buf.clear();
          }
          if (!bad && !werr && off != part_off[(size_t)c + 1]) bad = 4;  // (cannot happen: the ranges were measured above)
          n_rec += n;
        }
      });
    for (std::thread& x : th) x.join();
  }
  if (close(out) != 0) werr = true;
  unmap_all();
  if (bad || werr) {
    remove(out_path.c_str());  // no partial file is left behind
    if (werr) { fprintf(stderr, "%s: write to %s failed\n", who, out_path.c_str()); return 5; }
    if (bad_named) fprintf(stderr, "%s: a record names '%s', which no @SQ line declares\n", who, bad_name.c_str());
    else fprintf(stderr, "%s: malformed record in a shard\n", who);
    return bad;
  }
  if (n_out) *n_out = n_rec.load();
  return 0;
}

// ---- BAM record streams (k4align -G -o x.bam) ---------------------------------------------------------------------------------
// Each file: the coordinate-sorted BAM records of one rank, uncompressed, one behind the other (block_size, refID, pos, ...; SAM
// specification 4.2), refID numbering every sequence of the index.  They are merged by the key the ranks sorted by -- sequence,
// position, length of the first aligned block, strand (SortHitMatch, KAligner.cpp:10969-11014) -- equal keys in file order, records
// without coordinates (refID -1) last; ref_map (optional) renumbers refID / next_refID.  `emit(ptr, n)` takes the merged stream
// in pieces of whole records and returns false on an output error.  Sequential: the deflate behind emit is what takes the time.
struct BamKey {
  uint32_t ref; int32_t pos; uint32_t len; uint32_t strand;
  bool operator<(const BamKey& o) const {
    if (ref != o.ref) return ref < o.ref;
    if (pos != o.pos) return pos < o.pos;
    if (len != o.len) return len < o.len;
    return strand < o.strand;
  }
};
inline bool bam_key(const uint8_t* p, size_t avail, BamKey& k, size_t& rec_bytes) {
  if (avail < 36) return false;
  uint32_t bs; int32_t ref, pos; uint16_t n_cig, flag;
  memcpy(&bs, p, 4); memcpy(&ref, p + 4, 4); memcpy(&pos, p + 8, 4); memcpy(&n_cig, p + 16, 2); memcpy(&flag, p + 18, 2);
  const size_t l_name = p[12];
  if (bs < 32 || (size_t)bs + 4 > avail || 36 + l_name + 4 * (size_t)n_cig > (size_t)bs + 4) return false;
  rec_bytes = (size_t)bs + 4;
  uint32_t len = 0;
  if (n_cig) {
    uint32_t op;
    memcpy(&op, p + 36 + l_name, 4);
    if ((op & 15u) == 4u && n_cig > 1) memcpy(&op, p + 36 + l_name + 4, 4);  // behind a leading soft clip
    len = op >> 4;
  }
  k.ref = (uint32_t)ref; k.pos = pos; k.len = len; k.strand = (flag & 0x10) ? 1u : 0u;
  if (ref < 0) {  // -M1's records without coordinates: by NAR code (their last aux field, YU:Z:<two letters>), then file order
    const uint8_t* z = p + rec_bytes;
    k.pos = (rec_bytes >= 42 && z[-1] == 0 && memcmp(z - 6, "YUZ", 3) == 0) ? (int32_t)nar_code_of((const char*)z - 3) : 20;
    k.len = 0; k.strand = 0;
  }
  return true;
}
template <class Emit>
inline int merge_bam_records(const std::vector<std::string>& files, const std::vector<int32_t>* ref_map, Emit&& emit, unsigned long long* n_out, std::string* why) {
  struct Src { const uint8_t* p = nullptr; size_t len = 0, at = 0, rec = 0; BamKey k; int fd = -1; };
  std::vector<Src> src(files.size());
  auto fail = [&](const std::string& m, int rc) {
    if (why) *why = m;
    for (Src& x : src) { if (x.p && x.len) munmap((void*)x.p, x.len); if (x.fd >= 0) close(x.fd); }
    return rc;
  };
  for (size_t f = 0; f < files.size(); f++) {
    Src& x = src[f];
    x.fd = open(files[f].c_str(), O_RDONLY);
    struct stat sb;
    if (x.fd < 0 || fstat(x.fd, &sb) != 0) return fail("unable to open " + files[f], 2);
    x.len = (size_t)sb.st_size;
    if (x.len) {
      void* m = mmap(nullptr, x.len, PROT_READ, MAP_PRIVATE, x.fd, 0);
      if (m == MAP_FAILED) { x.len = 0; return fail("unable to map " + files[f], 2); }
      x.p = (const uint8_t*)m;
      madvise(m, x.len, MADV_SEQUENTIAL);
      if (!bam_key(x.p, x.len, x.k, x.rec)) return fail("malformed BAM record at the start of " + files[f], 4);
    }
  }
  std::vector<uint8_t> buf;
  buf.reserve((size_t)9 << 20);
  unsigned long long n = 0;
  for (;;) {
    int best = -1;
    for (size_t f = 0; f < src.size(); f++)
      if (src[f].at < src[f].len && (best < 0 || src[f].k < src[(size_t)best].k)) best = (int)f;  // (strict: ties stay with the lower file)
    if (best < 0) break;
    Src& x = src[(size_t)best];
    const size_t o = buf.size();
    buf.insert(buf.end(), x.p + x.at, x.p + x.at + x.rec);
    if (ref_map) {
      for (size_t fo : {(size_t)4, (size_t)24}) {
        int32_t v;
        memcpy(&v, &buf[o + fo], 4);
        if (v >= 0) {
          if ((size_t)v >= ref_map->size() || (*ref_map)[(size_t)v] < 0) return fail("a record of " + files[(size_t)best] + " names a sequence no rank reported as hit", 4);
          v = (*ref_map)[(size_t)v];
          memcpy(&buf[o + fo], &v, 4);
        }
      }
    }
    n++;
    x.at += x.rec;
    if (x.at < x.len && !bam_key(x.p + x.at, x.len - x.at, x.k, x.rec)) return fail("malformed BAM record in " + files[(size_t)best], 4);
    if (buf.size() >= ((size_t)8 << 20)) {
      if (!emit(buf.data(), buf.size())) return fail("", 5);
      buf.clear();
    }
  }
  if (!buf.empty() && !emit(buf.data(), buf.size())) return fail("", 5);
  if (n_out) *n_out = n;
  fail("", 0);
  if (why) why->clear();
  return 0;
}

}  // namespace k4merge
