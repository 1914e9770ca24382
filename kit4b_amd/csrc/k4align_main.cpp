// kit4b_amd/csrc/k4align_main.cpp -- `k4align`: the stand-alone counterpart of `ngskit4b kalign` for the accelerated path.
//
// Host C++ over the C ABI of libk4sfx.so only (no HIP here).  Mirrors, for the default SAM report mode:
//   reads loading        CKAligner::LoadRawReads    ngskit4b/KAligner.cpp:11648-12421 (FASTA/FASTQ, .gz, length filter,
//                                                   descriptor = first token, bases to etSeqBase)
//   align phase          CKAligner::ProcCoredApprox ngskit4b/KAligner.cpp:10110-10263  -> k4_kalign_batch / k4_kalign_pe_batch
//   PE pass              CKAligner::ProcessPairedEnds :2944-3596                      -> inside k4_kalign_pe_batch
//   statistics           CKAligner::ReportAlignStats :3600-3830 (NAR histogram, strand counts)
//   SAM                  CKAligner::WriteBAMReadHits :5718-5914, ReportBAMread :5957-6320, SortHitMatch :10969,
//                        CSAMfile::AddAlignment libkit4b/SAMfile.cpp:2194-2377, header :1615,1667-1669,1799
// Options follow kalign's letters: -i -u -I -o -s -e -m -n -U -d -D -E -l -L (plus -g <gpu> -b <batch reads>).
#include <zlib.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/k4sfx.h"

namespace {

struct Opts {
  std::string in1, in2, sfx, out;
  int max_subs = 5, min_edit = 1, pmode = 0, max_ns = 1, pe_mode = 0, pair_min = 100, pair_max = 1000, pair_strand = 0;
  int min_len = 50, max_len = 500;  // cDfltMinAcceptReadLen / cDfltMaxAcceptReadLen, KAligner.h:112-113
  int gpu = 0;
  long batch = 4000000;
};

struct FastxReader {  // FASTA or FASTQ, plain or gzip (CFasta, libkit4b/Fasta.cpp)
  gzFile f = nullptr;
  std::string line, pending;
  bool have_pending = false;
  bool open(const std::string& path) {
    f = gzopen(path.c_str(), "rb");
    if (f) gzbuffer(f, 1 << 20);
    return f != nullptr;
  }
  void close() { if (f) gzclose(f); f = nullptr; }
  bool getline(std::string& s) {
    if (have_pending) { s.swap(pending); have_pending = false; return true; }
    s.clear();
    char buf[1 << 16];
    for (;;) {
      if (!gzgets(f, buf, sizeof(buf))) return !s.empty();
      size_t n = strlen(buf);
      bool eol = n && buf[n - 1] == '\n';
      while (n && (buf[n - 1] == '\n' || buf[n - 1] == '\r')) n--;
      s.append(buf, n);
      if (eol) return true;
    }
  }
  // next record: name = first whitespace-delimited token of the descriptor, seq = etSeqBase codes
  bool next(std::string& name, std::vector<uint8_t>& seq) {
    std::string s;
    do { if (!getline(s)) return false; } while (s.empty());
    if (s[0] != '>' && s[0] != '@') return false;
    const bool fastq = s[0] == '@';
    size_t e = 1;
    while (e < s.size() && !isspace((unsigned char)s[e])) e++;
    name.assign(s, 1, std::min<size_t>(e - 1, 127));
    seq.clear();
    auto add = [&](const std::string& t) {
      for (char c : t) {
        switch (c) {
          case 'a': case 'A': seq.push_back(0); break;
          case 'c': case 'C': seq.push_back(1); break;
          case 'g': case 'G': seq.push_back(2); break;
          case 't': case 'T': case 'u': case 'U': seq.push_back(3); break;
          default: if (!isspace((unsigned char)c)) seq.push_back(4);
        }
      }
    };
    if (fastq) {
      if (!getline(s)) return false;
      add(s);
      if (!getline(s) || !getline(s)) return false;  // '+' line and qualities (ignored: -g default eFQIgnore)
    } else {
      while (getline(s)) {
        if (!s.empty() && s[0] == '>') { pending.swap(s); have_pending = true; break; }
        add(s);
      }
    }
    return true;
  }
};

struct Rec {  // one accepted alignment, enough to sort and print
  uint32_t chrom, loci;
  uint16_t len;
  uint8_t strand, mm;
  uint32_t flag;
  int32_t pnext, tlen;
  uint64_t read;  // index into names / seqs
  bool mate_eq;
};

const char* kNarAbbr[20] = {"NA", "AA", "EN", "NL", "MH", "ML", "ET", "OJ", "OM", "DP", "DS", "FC", "PR", "UI", "OI", "UP", "IS", "IT", "NP", "LC"};

void usage() {
  fprintf(stderr,
          "k4align -i reads.f[aq][.gz] [-u mates] -I index.sfx -o out.sam [-s subs/100bp=5] [-e 1|2] [-m 0..3] [-n maxNs=1]\n"
          "        [-U 0..4 PE mode] [-d minins=100] [-D maxins=1000] [-E] [-l minlen=50] [-L maxlen=500] [-g gpu=0] [-b batch]\n");
}

}  // namespace

int main(int argc, char** argv) {
  Opts o;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if (a.size() < 2 || a[0] != '-') { usage(); return 1; }
    auto val = [&]() -> std::string { return a.size() > 2 ? a.substr(2) : (i + 1 < argc ? std::string(argv[++i]) : std::string()); };
    switch (a[1]) {
      case 'i': o.in1 = val(); break;
      case 'u': o.in2 = val(); break;
      case 'I': o.sfx = val(); break;
      case 'o': o.out = val(); break;
      case 's': o.max_subs = atoi(val().c_str()); break;
      case 'e': o.min_edit = atoi(val().c_str()); break;
      case 'm': o.pmode = atoi(val().c_str()); break;
      case 'n': o.max_ns = atoi(val().c_str()); break;
      case 'U': o.pe_mode = atoi(val().c_str()); break;
      case 'd': o.pair_min = atoi(val().c_str()); break;
      case 'D': o.pair_max = atoi(val().c_str()); break;
      case 'E': o.pair_strand = 1; break;
      case 'l': o.min_len = atoi(val().c_str()); break;
      case 'L': o.max_len = atoi(val().c_str()); break;
      case 'g': o.gpu = atoi(val().c_str()); break;
      case 'b': o.batch = atol(val().c_str()); break;
      case 'T': case 'F': (void)val(); break;  // accepted and ignored (threads, log file)
      default: usage(); return 1;
    }
  }
  if (o.in1.empty() || o.sfx.empty() || o.out.empty()) { usage(); return 1; }
  const bool pe = !o.in2.empty();
  if (pe && o.pe_mode == 0) o.pe_mode = 1;  // kalign: -u without -U defaults to orphan recovery
  if (!pe) o.pe_mode = 0;

  auto t0 = std::chrono::steady_clock::now();
  k4_index* ix = nullptr;
  int rc = k4_open(o.sfx.c_str(), o.gpu, 0, &ix);
  if (rc != K4_OK) { fprintf(stderr, "k4align: unable to load '%s': %s (%d)\n", o.sfx.c_str(), k4_global_error(), rc); return 2; }
  k4_info_t info;
  k4_info(ix, &info);
  // SetMaxIter by sensitivity, KAligner.cpp:373-388
  k4_set_max_iter(ix, o.pmode == 2 ? 20000 : o.pmode == 1 ? 10000 : o.pmode == 0 ? 5000 : 2500);
  int slides = 0;
  const int mcl = k4_min_core_len(ix, o.pmode, &slides);
  fprintf(stderr, "k4align: index '%s' %u sequences, %llu bp; minimum core size %dbp\n", info.dataset, info.n_entries,
          (unsigned long long)info.tot_seqs_len, mcl);

  FastxReader r1, r2;
  if (!r1.open(o.in1) || (pe && !r2.open(o.in2))) { fprintf(stderr, "k4align: unable to open reads\n"); return 2; }

  std::vector<std::string> names;          // kept for accepted reads only would need a second pass: keep all (as kit4b does)
  std::vector<uint64_t> all_off;
  std::vector<uint32_t> all_len;
  std::vector<uint8_t> all_seq;
  std::vector<Rec> recs;
  uint64_t nar_hist[20] = {0}, plus = 0, minus = 0, n_loaded = 0, n_under = 0, n_over = 0;

  k4_kalign_params kp = {o.max_subs, o.min_edit, o.max_ns, o.pmode, K4_STRAND_BOTH, 1, 0, mcl, slides};
  k4_pe_params pp = {o.pe_mode, o.pair_min, o.pair_max, o.pair_strand};

  std::string nm1, nm2;
  std::vector<uint8_t> s1, s2;
  bool more = true;
  while (more) {
    // ---- load one batch -----------------------------------------------------------------------------------
    std::vector<uint8_t> b1, b2;
    std::vector<uint64_t> o1, o2;
    std::vector<uint32_t> l1, l2;
    std::vector<uint64_t> gidx;  // global read index of each batch read (PE: of PE1; PE2 = +1)
    while ((long)l1.size() < o.batch) {
      if (!r1.next(nm1, s1)) { more = false; break; }
      if (pe && !r2.next(nm2, s2)) { fprintf(stderr, "k4align: fewer PE2 than PE1 reads\n"); return 3; }
      bool bad = (int)s1.size() < o.min_len || (pe && (int)s2.size() < o.min_len);
      bool big = (int)s1.size() > o.max_len || (pe && (int)s2.size() > o.max_len);
      if (bad) { n_under++; continue; }  // sloughed, KAligner.cpp:12024-12060
      if (big) { n_over++; continue; }
      gidx.push_back(names.size());
      names.push_back(nm1);
      all_off.push_back(all_seq.size()); all_len.push_back((uint32_t)s1.size());
      all_seq.insert(all_seq.end(), s1.begin(), s1.end());
      o1.push_back(b1.size()); l1.push_back((uint32_t)s1.size());
      b1.insert(b1.end(), s1.begin(), s1.end());
      if (pe) {
        names.push_back(nm2);
        all_off.push_back(all_seq.size()); all_len.push_back((uint32_t)s2.size());
        all_seq.insert(all_seq.end(), s2.begin(), s2.end());
        o2.push_back(b2.size()); l2.push_back((uint32_t)s2.size());
        b2.insert(b2.end(), s2.begin(), s2.end());
      }
    }
    const int64_t n = (int64_t)l1.size();
    if (n == 0) break;
    n_loaded += pe ? 2 * n : n;
    b1.resize(b1.size() + 16);
    b2.resize(b2.size() + 16);
    // ---- align ------------------------------------------------------------------------------------------------
    if (!pe) {
      std::vector<k4_read_result> rr((size_t)n);
      std::vector<k4_hit> hits((size_t)n);
      rc = k4_kalign_batch(ix, &kp, n, b1.data(), o1.data(), l1.data(), rr.data(), hits.data());
      if (rc != K4_OK) { fprintf(stderr, "k4align: %s (%d)\n", k4_last_error(ix), rc); return 4; }
      for (int64_t i = 0; i < n; i++) {
        nar_hist[rr[i].nar < 20 ? rr[i].nar : 0]++;
        if (rr[i].nar != K4_NAR_ACCEPTED) continue;
        const k4_hit& h = hits[i];
        Rec r = {h.chrom_id, h.match_loci, h.match_len, h.strand, h.mismatches, h.strand == '+' ? 0u : 0x10u, -1, 0, gidx[i], false};
        (h.strand == '+' ? plus : minus)++;
        recs.push_back(r);
      }
    } else {
      std::vector<k4_pe_read> out((size_t)2 * n);
      rc = k4_kalign_pe_batch(ix, &kp, &pp, n, b1.data(), o1.data(), l1.data(), b2.data(), o2.data(), l2.data(), out.data());
      if (rc != K4_OK) { fprintf(stderr, "k4align: %s (%d)\n", k4_last_error(ix), rc); return 4; }
      for (int64_t i = 0; i < 2 * n; i++) {
        const k4_pe_read& a = out[i];
        const k4_pe_read& m = out[i ^ 1];
        nar_hist[a.nar < 20 ? a.nar : 0]++;
        if (a.nar != K4_NAR_ACCEPTED) continue;
        const k4_hit& h = a.hit;
        uint32_t flag = 0x1 | 0x2 | ((i & 1) ? 0x80u : 0x40u) | (h.strand == '+' ? 0u : 0x10u);  // ReportBAMread :6041-6114
        int32_t pnext = -1, tlen = 0;
        bool eq = false;
        if (a.pe_aligned && m.pe_aligned && m.nar == K4_NAR_ACCEPTED) {
          if (m.hit.strand != '+') flag |= 0x20;
          eq = true;
          pnext = (int32_t)m.hit.match_loci;
          const int64_t s = h.match_loci, e = m.hit.match_loci;
          tlen = (int32_t)(s <= e ? (e - s) + m.hit.match_len : (s - e) + h.match_len);
        } else
          flag |= 0x8;
        Rec r = {h.chrom_id, h.match_loci, h.match_len, h.strand, h.mismatches, flag, pnext, tlen, gidx[i / 2] + (i & 1), eq};
        (h.strand == '+' ? plus : minus)++;
        recs.push_back(r);
      }
    }
  }
  r1.close();
  if (pe) r2.close();
  auto t1 = std::chrono::steady_clock::now();

  // ---- statistics (ReportAlignStats) ------------------------------------------------------------------------------
  fprintf(stderr, "k4align: From %llu source reads there are %llu accepted alignments, %llu on '+' strand, %llu on '-' strand\n",
          (unsigned long long)n_loaded, (unsigned long long)nar_hist[1], (unsigned long long)plus, (unsigned long long)minus);
  if (n_under || n_over)
    fprintf(stderr, "k4align: %llu under length and %llu over length reads were sloughed\n", (unsigned long long)n_under,
            (unsigned long long)n_over);
  for (int k = 0; k < 20; k++) fprintf(stderr, "k4align:    %llu (%s)\n", (unsigned long long)nar_hist[k], kNarAbbr[k]);

  // ---- SAM ---------------------------------------------------------------------------------------------------------------
  std::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) {  // SortHitMatch: chrom, start, len, strand, mismatches
    if (a.chrom != b.chrom) return a.chrom < b.chrom;
    if (a.loci != b.loci) return a.loci < b.loci;
    if (a.len != b.len) return a.len < b.len;
    if (a.strand != b.strand) return a.strand < b.strand;
    if (a.mm != b.mm) return a.mm < b.mm;
    return a.read < b.read;
  });
  FILE* fp = fopen(o.out.c_str(), "wb");
  if (!fp) { fprintf(stderr, "k4align: unable to create %s\n", o.out.c_str()); return 5; }
  static char iobuf[1 << 22];
  setvbuf(fp, iobuf, _IOFBF, sizeof(iobuf));
  fprintf(fp, "@HD\tVN:1.4\tSO:coordinate\n");
  std::vector<char> hit_chrom(info.n_entries + 1, 0);
  for (const Rec& r : recs) hit_chrom[r.chrom] = 1;
  const bool all_chroms = info.n_entries <= 10000;  // m_MaxRptSAMSeqsThres, KAligner.cpp:5785-5821
  std::vector<std::string> cname(info.n_entries + 1);
  for (uint32_t c = 1; c <= info.n_entries; c++) {
    k4_entry e;
    k4_get_entry(ix, c, &e);
    cname[c] = e.name;
    if (all_chroms || hit_chrom[c]) fprintf(fp, "@SQ\tAS:%s\tSN:%s\tLN:%u\n", info.dataset, e.name, e.seq_len);
  }
  fprintf(fp, "@PG\tID:k4align\tVN:1.0\n");
  std::string seq;
  for (const Rec& r : recs) {
    const uint8_t* s = all_seq.data() + all_off[r.read];
    const uint32_t len = all_len[r.read];
    seq.resize(len);
    if (r.strand == '+')
      for (uint32_t j = 0; j < len; j++) seq[j] = "ACGTN"[s[j] > 4 ? 4 : s[j]];
    else
      for (uint32_t j = 0; j < len; j++) { uint8_t b = s[len - 1 - j]; seq[j] = b <= 3 ? "TGCA"[b] : 'N'; }  // :6279
    // MAPQ = max(1, 254 * hitlen / readlen) (:6146,6231); CIGAR <len>M; RNEXT '=' / '*'; PNEXT 1-based or 0; QUAL '*'
    int mapq = std::max(1, (int)(254 * ((double)r.len / len)));
    if (mapq > 254) mapq = 254;
    fprintf(fp, "%s\t%u\t%s\t%u\t%d\t%uM\t%c\t%d\t%d\t%s\t*\n", names[r.read].c_str(), r.flag, cname[r.chrom].c_str(),
            r.loci + 1, mapq, (unsigned)r.len, r.mate_eq ? '=' : '*', r.mate_eq ? r.pnext + 1 : 0, r.tlen, seq.c_str());
  }
  fclose(fp);
  auto t2 = std::chrono::steady_clock::now();
  fprintf(stderr, "k4align: %zu alignments written to %s; load+align %.2fs, sort+write %.2fs\n", recs.size(), o.out.c_str(),
          std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
  k4_close(ix);
  return 0;
}
