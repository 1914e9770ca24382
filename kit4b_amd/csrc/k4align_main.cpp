// kit4b_amd/csrc/k4align_main.cpp -- `k4align`: the stand-alone counterpart of `ngskit4b kalign` for the accelerated path.
//
// Host C++ over the C ABI of libk4sfx.so only (no HIP here): the files are read (and inflated) on the host, everything
// between the raw text and the SAM body runs on the device.  Mirrors, for the default SAM report mode:
//   reads loading        CKAligner::LoadRawReads    ngskit4b/KAligner.cpp:11648-12421 (FASTA/FASTQ, .gz, length filter,
//                                                   descriptor = first token, bases to etSeqBase) -> k4_parse_fastx_dev,
//                                                   k4_prepare_reads_dev
//   align phase          CKAligner::ProcCoredApprox ngskit4b/KAligner.cpp:10110-10263  -> k4_kalign_batch_dev
//   PE pass              CKAligner::ProcessPairedEnds :2944-3596                      -> k4_kalign_pe_batch_dev
//   statistics           CKAligner::ReportAlignStats :3600-3830 (NAR histogram, strand counts)
//   SAM                  CKAligner::WriteBAMReadHits :5718-5914, ReportBAMread :5957-6320, SortHitMatch :10969,
//                        CSAMfile::AddAlignment libkit4b/SAMfile.cpp:2194-2377 -> k4_format_sam_dev; header :1615,1667-1669,1799
// Options follow kalign's letters: -i -u -I -o -s -e -m -n -U -d -D -E -l -L -r -R -X -N -c -a -A -x (plus -g <gpu>, -S <i/N> read slice).
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <map>
#include <queue>
#include <memory>
#include <string>
#include <vector>
#include <sys/mman.h>
#include <sys/wait.h>
#include "../../include/k4comm.h"
#include "../../include/k4sfx.h"
#include "../../include/k4_bam.hpp"
#include "k4_merge.h"

namespace {

struct Opts {
  std::vector<std::string> in1, in2;  // -i / -u may repeat: the files are read one after the other (KAlignerCL.cpp: up to cRRMaxInFiles)
  std::string sfx, out;
  int max_subs = 5, min_edit = 1, pmode = 0, max_ns = 1, pe_mode = 0, pair_min = 100, pair_max = -1 /* not given */, pair_strand = 0;
  int min_len = 50, max_len = 500;  // cDfltMinAcceptReadLen / cDfltMaxAcceptReadLen, KAligner.h:112-113
  int ml_mode = 0, max_multi = 0;   // -r / -R (etMLMode, KAligner.h:250-258)
  bool clamp = false, best = false; // -X / -N (KAlignerCL.cpp:278-280)
  int q_method = 3;  // -g (etFQMethod, KAlignerCL.cpp:241,499): 3 = the quality lines are ignored
  int min_chimeric = 0, micro_indel = 0, splice_junct = 0, min_flank_exacts = 0;  // -c / -a / -A / -x (KAlignerCL.cpp:237,245,246,267)
  double batch_mb = 0;              // -b <MB>: stream the input, this much text per file per batch (0: the whole input at once)
  int shard = 0, n_shards = 1;      // -S i/N: this process aligns the i-th of N contiguous slices of the reads (one process per GPU)
  int gpu = 0;
  std::vector<int> gpus;            // -G g0,g1,...: one rank process per listed GPU (RCCL over xGMI, libk4comm.so)
  int rank = 0, n_ranks = 1;        // this process's place in a -G run
  bool rank_bam = false;            // ... whose output is a BAM file: the rank leaves its sorted records (all sequences numbered) + a dictionary beside them
  struct MultiShared* shared = nullptr;
  int print_slices = 0;             // -W <n>: print the byte offsets -G would cut the reads files at for n ranks, and stop (no GPU needed)
  int rpt_sq_thres = 10000;         // -4 <n>: with more reference sequences than this only those with alignments are declared in the header (KAlignerCL.cpp:289,868-870)
  std::string none_file, multi_file; // -j / -J <file>: the reads without an alignment / the multi-aligned reads as FASTA (KAlignerCL.cpp:254-255)
  int sample_nth = 1;               // -# <n>: every n-th read / pair of the input is processed (KAlignerCL.cpp:244,484-489)
  int trim5 = 0, trim3 = 0;         // -y / -Y <n>: bases taken off the 5' / 3' end of every read when loading (KAlignerCL.cpp:763-774)
  int align_strand = 0;             // -Q <0|1|2>: align to either strand, the sense or the antisense strand only (KAlignerCL.cpp:241,491)
  int fmode = 0;                    // -M <0|1>: 0 SAM / BAM with the accepted alignments, 1 SAM with every loaded read (KAlignerCL.cpp:217)
  bool legacy = false;              // -Z: the serial whole-input path of round 1 (one batch, no overlap), kept for comparison
  int chunk_mb = 256;               // -B <MB>: size of one pinned upload buffer of the pipeline
  int io_threads = 8;               // -t <n>: concurrent pread / pwrite calls per buffer; BAM: deflate threads
  int min_snp_reads = 0;            // -p <n> (0: no SNP calling), -P <qvalue>, -1 <pct>, -S <file> (KAlignerCL.cpp:259,284-291,877-932)
  double qvalue = 0.0, snp_nonref_pcnt = 25.0;
  std::string snp_file;
  int bam_level = 6;                // -z <0..9>: BGZF deflate level of a .bam output (WriteBAMReadHits is called with 6, KAligner.cpp:759)
};

struct Parsed {  // one reads file after k4_parse_fastx_dev: everything lives in HBM
  void* d_text = nullptr;
  void* d_offs = nullptr;
  void* d_lens = nullptr;
  void* d_noff = nullptr;
  void* d_nlen = nullptr;
  uint64_t n = 0, bases = 0;
  uint32_t max_len = 0;
};

#define CK(call)                                                          \
  do {                                                                    \
    int _rc = (call);                                                     \
    if (_rc != K4_OK) {                                                   \
      fprintf(stderr, "k4align: %s (%d)\n", k4_last_error(ix), _rc);      \
      return _rc;                                                         \
    }                                                                     \
  } while (0)

// upload the text and parse it on the device, in pieces below the 4 GiB limit of one call; at most max_records records;
// *consumed = text bytes that belonged to them (the rest is resubmitted in front of the next batch)
int load_reads(k4_index* ix, const uint8_t* text, uint64_t T, int final_text, int64_t max_records, void* d_reads, uint64_t reads_base,
               Parsed& P, uint64_t* consumed) {
  *consumed = 0;
  if (T == 0) return K4_OK;
  const bool fastq = text[0] == '@';
  const uint64_t nl = (uint64_t)std::count(text, text + T, (uint8_t)'\n');
  const int64_t cap = std::min<int64_t>((int64_t)((nl + 1) / (fastq ? 4 : 2) + 4), max_records > 0 ? max_records : INT64_MAX);
  CK(k4_alloc_device(ix, T + 16, &P.d_text));
  CK(k4_copy_to_device(ix, P.d_text, text, T));
  CK(k4_alloc_device(ix, (uint64_t)cap * 8, &P.d_offs));
  CK(k4_alloc_device(ix, (uint64_t)cap * 4, &P.d_lens));
  CK(k4_alloc_device(ix, (uint64_t)cap * 8, &P.d_noff));
  CK(k4_alloc_device(ix, (uint64_t)cap * 4, &P.d_nlen));
  uint64_t pos = 0;
  const uint64_t piece = 3ull << 30;
  int fmt = 0;
  while (pos < T && (int64_t)P.n < cap) {
    const uint64_t len = std::min(piece, T - pos);
    const int final_chunk = final_text && pos + len == T;
    k4_parse_info info;
    CK(k4_parse_fastx_dev(ix, (const uint8_t*)P.d_text + pos, len, pos, final_chunk, fmt, cap - (int64_t)P.n, d_reads,
                          reads_base + P.bases, (uint8_t*)P.d_offs + 8 * P.n, (uint8_t*)P.d_lens + 4 * P.n,
                          (uint8_t*)P.d_noff + 8 * P.n, (uint8_t*)P.d_nlen + 4 * P.n, &info, nullptr));
    if (info.format) fmt = (int)info.format;
    if (info.consumed == 0) break;  // nothing but an incomplete tail
    P.n += info.n_records;
    P.bases += info.n_bases;
    P.max_len = std::max(P.max_len, info.max_len);
    pos += info.consumed;
  }
  *consumed = pos;
  return K4_OK;
}
void free_parsed(Parsed& P) {
  for (void* q : {P.d_text, P.d_offs, P.d_lens, P.d_noff, P.d_nlen}) k4_free_device(q);
  P = Parsed();
}

// the reads files of one end, read in portions as one text (a file that lacks its last newline gets one): buf holds the
// text not yet consumed
struct Stream {
  gzFile f = nullptr;
  std::vector<std::string> paths;
  size_t next = 0;
  std::vector<uint8_t> buf;
  bool eof = false;
  bool open_next() {
    if (f) gzclose(f);
    f = gzopen(paths[next++].c_str(), "rb");
    if (f) gzbuffer(f, 1 << 20);
    return f != nullptr;
  }
  bool open(const std::vector<std::string>& files) {
    paths = files;
    for (const std::string& q : paths) {  // all of them must be readable before anything is aligned
      FILE* t = fopen(q.c_str(), "rb");
      if (!t) { fprintf(stderr, "k4align: unable to open '%s'\n", q.c_str()); return false; }
      fclose(t);
    }
    return !paths.empty() && open_next();
  }
  bool fill(uint64_t want) {  // until buf holds `want` bytes or the file ends
    while (!eof && buf.size() < want) {
      const size_t old = buf.size();
      const size_t step = (size_t)std::min<uint64_t>(std::max<uint64_t>(want - old, 1 << 20), 256u << 20);
      buf.resize(old + step);
      const int got = gzread(f, buf.data() + old, (unsigned)step);
      if (got < 0) return false;
      buf.resize(old + (size_t)got);
      if (got == 0) {
        if (next == paths.size()) { eof = true; break; }
        if (!buf.empty() && buf.back() != '\n') buf.push_back('\n');
        if (!open_next()) return false;
      }
    }
    return true;
  }
  void drop(uint64_t n) { buf.erase(buf.begin(), buf.begin() + (ptrdiff_t)std::min<uint64_t>(n, buf.size())); }
  void close() { if (f) gzclose(f); f = nullptr; }
};

// ---- the pipelined (default) mode: one reader thread per end fills the library's pinned ring buffers ----------------------
bool is_gzip(const std::string& path) {
  unsigned char m[2] = {0, 0};
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  const size_t got = fread(m, 1, 2, f);
  fclose(f);
  return got == 2 && m[0] == 0x1f && m[1] == 0x8b;
}
uint64_t file_size(const std::string& path) {
  struct stat st;
  return stat(path.c_str(), &st) == 0 ? (uint64_t)st.st_size : 0;
}
// plain files: the range [off, off+len) by `nt` concurrent pread()s (tmpfs / page cache deliver more than one thread can take)
bool pread_parallel(int fd, uint64_t off, uint8_t* dst, uint64_t len, int nt) {
  if (len < (8u << 20) || nt <= 1) {
    uint64_t done = 0;
    while (done < len) {
      const ssize_t g = pread(fd, dst + done, len - done, (off_t)(off + done));
      if (g <= 0) return false;
      done += (uint64_t)g;
    }
    return true;
  }
  std::atomic<bool> ok(true);
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++)
    th.emplace_back([&, t] {
      const uint64_t a = len * t / nt, b = len * (t + 1) / nt;
      uint64_t done = a;
      while (done < b) {
        const ssize_t g = pread(fd, dst + done, b - done, (off_t)(off + done));
        if (g <= 0) { ok = false; return; }
        done += (uint64_t)g;
      }
    });
  for (std::thread& x : th) x.join();
  return ok;
}
// every file of one end, in order, into the pipeline; a file that lacks its last newline gets one (as Stream does)
int feed_end(k4_pipeline* pl, int end, const std::vector<std::string>& files, int io_threads, double* secs_read) {
  void* buf = nullptr;
  uint64_t cap = 0, used = 0;
  uint8_t last = '\n';
  auto t0 = std::chrono::steady_clock::now();
  double busy = 0;
  auto flush = [&](int fin) -> int {
    busy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int rc = k4_pipeline_submit(pl, end, used, fin);
    buf = nullptr; used = 0;
    t0 = std::chrono::steady_clock::now();
    return rc;
  };
  for (size_t f = 0; f < files.size(); f++) {
    const bool gz = is_gzip(files[f]);
    gzFile g = nullptr;
    int fd = -1;
    uint64_t fsz = 0, fpos = 0;
    if (gz) {
      g = gzopen(files[f].c_str(), "rb");
      if (!g) return K4_ERR_OPEN_FILE;
      gzbuffer(g, 4u << 20);
    } else {
      fd = open(files[f].c_str(), O_RDONLY);
      if (fd < 0) return K4_ERR_OPEN_FILE;
      fsz = file_size(files[f]);
    }
    for (;;) {
      if (!buf) {
        busy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        int rc = k4_pipeline_acquire(pl, end, &buf, &cap);  // (waiting for a free buffer is not reading time)
        t0 = std::chrono::steady_clock::now();
        if (rc != K4_OK) return rc;
        used = 0;
      }
      uint64_t got = 0;
      if (gz) {
        const int r = gzread(g, (uint8_t*)buf + used, (unsigned)std::min<uint64_t>(cap - used, 1u << 30));
        if (r < 0) return K4_ERR_FILE_ACCESS;
        got = (uint64_t)r;
      } else {
        got = std::min<uint64_t>(cap - used, fsz - fpos);
        if (got && !pread_parallel(fd, fpos, (uint8_t*)buf + used, got, io_threads)) return K4_ERR_FILE_ACCESS;
        fpos += got;
      }
      if (got == 0) break;  // end of this file
      used += got;
      last = ((uint8_t*)buf)[used - 1];
      if (used == cap) { int rc = flush(0); if (rc != K4_OK) return rc; }
    }
    if (gz) gzclose(g); else close(fd);
    if (last != '\n' && f + 1 < files.size()) {  // the next file starts on a line of its own
      if (!buf) { int rc = k4_pipeline_acquire(pl, end, &buf, &cap); if (rc != K4_OK) return rc; used = 0; }
      ((uint8_t*)buf)[used++] = '\n';
      last = '\n';
      if (used == cap) { int rc = flush(0); if (rc != K4_OK) return rc; }
    }
  }
  if (!buf) {  // the final (possibly empty) chunk still has to be announced
    int rc = k4_pipeline_acquire(pl, end, &buf, &cap);
    if (rc != K4_OK) return rc;
    used = 0;
  }
  int rc = flush(1);
  if (secs_read) *secs_read = busy;
  return rc;
}


// ---- k4align -G: one rank process per GPU ---------------------------------------------------------------------------------------
#define K4_MAX_RANKS 64
#define K4_MAX_FILES 64  // reads files per end a -G run may name
struct MultiShared {  // an anonymous shared mapping the parent creates before it forks the ranks
  uint8_t id[K4_COMM_ID_BYTES];
  std::atomic<int> id_ready, slices_ready, failed;
  // byte offsets of the ranks' record slices in every reads file of each end: rank r reads [off[r], off[r + 1]) of file f
  uint64_t slice_off[2][K4_MAX_FILES][K4_MAX_RANKS + 1];
  uint64_t n_records;
};

// One uncompressed FASTA / FASTQ file mapped, its record starts counted per range (pass 1, parallel); byte offsets of given
// record numbers come from a second look at the one range each falls into (pass 2).
struct RecIndex {
  const uint8_t* t = nullptr;
  uint64_t len = 0, R = 0;
  bool fastq = false;
  int nt = 1;
  std::vector<uint64_t> part;  // marks in front of range k (a mark starts a record: FASTA -- '>' at a line start; FASTQ -- every line)
  ~RecIndex() { if (t && len) munmap((void*)t, len); }
  uint64_t count(uint64_t a, uint64_t b) const {  // FASTA: header lines starting in [a, b); FASTQ: newlines in [a, b)
    uint64_t c = 0;
    if (fastq) {
      const uint8_t* p = t + a;
      while (p < t + b) { const void* q = memchr(p, '\n', (size_t)(t + b - p)); if (!q) break; c++; p = (const uint8_t*)q + 1; }
    } else {
      for (uint64_t p = a; p < b;) {
        const void* q = memchr(t + p, '>', (size_t)(b - p));
        if (!q) break;
        const uint64_t at = (uint64_t)((const uint8_t*)q - t);
        if (at == 0 || t[at - 1] == '\n') c++;
        p = at + 1;
      }
    }
    return c;
  }
  bool open_file(const std::string& path, int threads) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    len = file_size(path);
    nt = std::max(threads, 1);
    part.assign((size_t)nt + 1, 0);
    if (len == 0) { close(fd); R = 0; return true; }
    t = (const uint8_t*)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (t == MAP_FAILED) { t = nullptr; return false; }
    fastq = t[0] == '@';
    std::vector<std::thread> th;
    for (int k = 0; k < nt; k++) th.emplace_back([this, k] { part[(size_t)k + 1] = count(len * k / nt, len * (k + 1) / nt); });
    for (std::thread& x : th) x.join();
    for (int k = 0; k < nt; k++) part[(size_t)k + 1] += part[(size_t)k];
    if (fastq) { const uint64_t lines = part[(size_t)nt] + (t[len - 1] != '\n' ? 1 : 0); R = lines / 4; }
    else R = part[(size_t)nt];
    return true;
  }
  uint64_t offset_of(uint64_t rec) const {  // where record `rec` (0-based) starts; R and beyond: the end of the file
    if (rec == 0) return 0;
    if (rec >= R) return len;
    // FASTQ: the byte behind newline number 4 * rec; FASTA: the position of header number rec (0-based)
    const uint64_t target = fastq ? 4 * rec : rec + 1;  // the target-th mark (1-based) of the file
    int k = 0;
    while (k + 1 < nt && part[(size_t)k + 1] < target) k++;
    uint64_t seen = part[(size_t)k], p = len * k / nt;
    const uint64_t end = len * (k + 1) / nt;
    if (fastq) {
      while (p < end) { const void* q = memchr(t + p, '\n', (size_t)(end - p)); if (!q) break; p = (uint64_t)((const uint8_t*)q - t) + 1; if (++seen == target) return p; }
    } else {
      while (p < end) {
        const void* q = memchr(t + p, '>', (size_t)(end - p));
        if (!q) break;
        const uint64_t a2 = (uint64_t)((const uint8_t*)q - t);
        if ((a2 == 0 || t[a2 - 1] == '\n') && ++seen == target) return a2;
        p = a2 + 1;
      }
    }
    return len;
  }
};

// The records of all files of one end, taken as one sequence, cut into n_ranks contiguous slices: rank r gets the records
// [R r / N, R (r + 1) / N) (pairs stay together: both ends are cut at the same record numbers, file by file -- the two ends'
// files must hold the same numbers of records, as kalign demands of them).  offs[f][r]: where rank r starts in file f.
// counts: records per file (filled for end 0, compared for end 1).
bool slice_files(const std::vector<std::string>& files, int n_ranks, std::vector<uint64_t>& counts, bool compare_counts,
                 uint64_t (*offs)[K4_MAX_RANKS + 1], uint64_t* n_records, int nt, std::string* why) {
  std::vector<std::unique_ptr<RecIndex>> ri;
  uint64_t R = 0;
  for (size_t f = 0; f < files.size(); f++) {
    ri.emplace_back(new RecIndex);
    if (!ri.back()->open_file(files[f], nt)) { *why = "unable to read " + files[f]; return false; }
    if (compare_counts && (f >= counts.size() || counts[f] != ri.back()->R)) {
      *why = "the PE1 and PE2 files hold different numbers of reads (" + std::to_string(f < counts.size() ? counts[f] : 0) + ", " + std::to_string(ri.back()->R) + ")";
      return false;
    }
    if (!compare_counts) counts.push_back(ri.back()->R);
    R += ri.back()->R;
  }
  *n_records = R;
  uint64_t start = 0;  // records in front of file f
  for (size_t f = 0; f < files.size(); f++) {
    for (int r = 0; r <= n_ranks; r++) {
      const uint64_t g = r == n_ranks ? R : R * (uint64_t)r / (uint64_t)n_ranks;  // first record of rank r, over all files
      const uint64_t local = g <= start ? 0 : std::min<uint64_t>(g - start, ri[f]->R);
      offs[f][r] = ri[f]->offset_of(local);  // (behind the file's last record: its length, a missing final newline included)
    }
    start += ri[f]->R;
  }
  return true;
}

// the byte ranges [a, b) of uncompressed files, one after the other, into the pipeline; a range that ends a file which lacks its
// last newline gets one (the next file's first record starts on a line of its own)
struct FileRange { std::string path; uint64_t a, b, file_len; };
int feed_ranges(k4_pipeline* pl, int end, const std::vector<FileRange>& ranges, int io_threads, double* secs_read) {
  double busy = 0;
  void* buf = nullptr;
  uint64_t cap = 0, used = 0;
  int rc = K4_OK;
  auto flush = [&](int fin) { const int r = k4_pipeline_submit(pl, end, used, fin); buf = nullptr; used = 0; return r; };
  for (size_t k = 0; k < ranges.size() && rc == K4_OK; k++) {
    const FileRange& fr = ranges[k];
    if (fr.b <= fr.a) continue;
    const int fd = open(fr.path.c_str(), O_RDONLY);
    if (fd < 0) return K4_ERR_OPEN_FILE;
    uint64_t pos = fr.a;
    uint8_t last = '\n';
    while (pos < fr.b && rc == K4_OK) {
      if (!buf) { if ((rc = k4_pipeline_acquire(pl, end, &buf, &cap)) != K4_OK) break; used = 0; }
      const uint64_t len = std::min<uint64_t>(cap - used, fr.b - pos);
      auto t0 = std::chrono::steady_clock::now();
      if (len && !pread_parallel(fd, pos, (uint8_t*)buf + used, len, io_threads)) { rc = K4_ERR_FILE_ACCESS; break; }
      busy += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      pos += len;
      used += len;
      if (len) last = ((uint8_t*)buf)[used - 1];
      if (used == cap) rc = flush(0);
    }
    close(fd);
    if (rc == K4_OK && fr.b == fr.file_len && last != '\n') {
      bool more = false;
      for (size_t q = k + 1; q < ranges.size(); q++) more |= ranges[q].b > ranges[q].a;
      if (more) {
        if (!buf) { if ((rc = k4_pipeline_acquire(pl, end, &buf, &cap)) != K4_OK) break; used = 0; }
        ((uint8_t*)buf)[used++] = '\n';
        if (used == cap) rc = flush(0);
      }
    }
  }
  if (rc == K4_OK) {
    if (!buf) { if ((rc = k4_pipeline_acquire(pl, end, &buf, &cap)) != K4_OK) return rc; used = 0; }
    rc = flush(1);  // the final (possibly empty) chunk
  }
  if (secs_read) *secs_read = busy;
  return rc;
}

// Compressed input cannot be cut by byte offsets: every rank inflates the whole stream and keeps the record blocks dealt to it --
// block number (record number / K4_DEAL_RECORDS) modulo the rank count; both ends are dealt by the same record numbers, so
// mates stay together.  (Equal sort keys then come out in block order, not in load order: the reference leaves that order open.)
#define K4_DEAL_RECORDS (1u << 18)
int feed_dealt(k4_pipeline* pl, int end, const std::vector<std::string>& files, int rank, int n_ranks, double* secs_read) {
  void* buf = nullptr;
  uint64_t cap = 0, used = 0;
  int rc = K4_OK;
  auto t00 = std::chrono::steady_clock::now();
  double waited = 0;
  auto put = [&](const uint8_t* p, uint64_t n) {  // bytes of this rank's records into the pipeline
    while (n && rc == K4_OK) {
      if (!buf) {
        auto w0 = std::chrono::steady_clock::now();
        rc = k4_pipeline_acquire(pl, end, &buf, &cap);
        waited += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        used = 0;
        if (rc != K4_OK) return;
      }
      const uint64_t take = std::min<uint64_t>(n, cap - used);
      memcpy((uint8_t*)buf + used, p, take);
      used += take; p += take; n -= take;
      if (used == cap) { rc = k4_pipeline_submit(pl, end, used, 0); buf = nullptr; used = 0; }
    }
  };
  std::vector<uint8_t> chunk((size_t)16 << 20);
  uint64_t rec = 0;      // records begun so far, over all files
  uint32_t lines = 0;    // FASTQ: lines of the current record seen so far
  int fastq = -1;
  bool at_line_start = true;
  uint8_t last = '\n';
  for (size_t f = 0; f < files.size() && rc == K4_OK; f++) {
    gzFile g = gzopen(files[f].c_str(), "rb");  // (reads plain files as they are)
    if (!g) return K4_ERR_OPEN_FILE;
    gzbuffer(g, 4u << 20);
    for (;;) {
      const int got = gzread(g, chunk.data(), (unsigned)chunk.size());
      if (got < 0) { gzclose(g); return K4_ERR_FILE_ACCESS; }
      if (got == 0) break;
      if (fastq < 0) fastq = chunk[0] == '@' ? 1 : 0;
      // runs of bytes that belong to one record block go out (or not) together
      int run_start = 0;
      bool mine = rec ? ((rec - 1) / K4_DEAL_RECORDS) % (uint64_t)n_ranks == (uint64_t)rank : false;  // the record we are inside
      for (int q = 0; q < got; q++) {
        const uint8_t ch = chunk[(size_t)q];
        bool starts = false;
        if (at_line_start) {
          if (fastq) { if (lines == 0) starts = true; }
          else if (ch == '>') starts = true;
        }
        if (starts) {
          const bool m2 = (rec / K4_DEAL_RECORDS) % (uint64_t)n_ranks == (uint64_t)rank;
          if (m2 != mine) {
            if (mine) put(chunk.data() + run_start, (uint64_t)(q - run_start));
            run_start = q;
            mine = m2;
          }
          rec++;
        }
        at_line_start = ch == '\n';
        if (fastq && at_line_start) lines = (lines + 1) & 3u;
      }
      if (mine) put(chunk.data() + run_start, (uint64_t)(got - run_start));
      last = chunk[(size_t)got - 1];
      if (rc != K4_OK) break;
    }
    gzclose(g);
    if (last != '\n' && f + 1 < files.size()) {  // the next file starts on a line of its own
      const bool mine = rec ? ((rec - 1) / K4_DEAL_RECORDS) % (uint64_t)n_ranks == (uint64_t)rank : false;
      const uint8_t nl = '\n';
      if (mine) put(&nl, 1);
      at_line_start = true;
      if (fastq == 1) lines = (lines + 1) & 3u;
      last = '\n';
    }
  }
  if (rc == K4_OK) {
    if (!buf) { if ((rc = k4_pipeline_acquire(pl, end, &buf, &cap)) != K4_OK) return rc; used = 0; }
    rc = k4_pipeline_submit(pl, end, used, 1);
  }
  if (secs_read) *secs_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t00).count() - waited;
  return rc;
}

const char* kNarAbbr[20] = {"NA", "AA", "EN", "NL", "MH", "ML", "ET", "OJ", "OM", "DP", "DS", "FC", "PR", "UI", "OI", "UP", "IS", "IT", "NP", "LC"};

void usage() {
  fprintf(stderr,
          "k4align -i reads.f[aq][.gz] [-i more ...] [-u mates ...] -I index.sfx -o out.sam|out.bam [-z bgzf level=6] [-s subs/100bp=5] [-e 1|2] [-m 0..3] [-n maxNs=1]\n"
          "        [-U 0..4 PE mode] [-d minins=100] [-D maxins=1000] [-E] [-l minlen=50] [-L maxlen=500] [-r 0..5] [-R maxmulti=5] [-X] [-N] [-j unaligned.fa] [-J multialigned.fa] [-# every nth read] [-4 all @SQ up to n=10000] [-y trim5] [-Y trim3] [-Q 0|1|2 strand] [-M 0|1 all reads] [-c minchimeric%%] [-a microindel] [-A splicejunct] [-x flankexacts] [-p minsnpreads [-P qvalue=0.05] [-1 nonref%%=25] [-S snps.csv]] [-S i/N] [-b MB per batch] [-B MB per upload=256] [-t io threads=8] [-Z] [-g 0..3 FASTQ qualities: Sanger | Illumina 1.3+ | Solexa | ignore=3] [-@ gpu=0] [-G gpu,gpu,... one rank per GPU]\n");
}

}  // namespace

// The C library's rand() as the reference binary sees it on Linux -- glibc's default TYPE_3 additive feedback generator,
// never seeded by kalign (seed 1): r[i] = r[i-3] + r[i-31] over 32-bit words, 310 values discarded, output = r[i] >> 1.
struct GlibcRand {
  uint32_t r[34];
  int k = 0;
  GlibcRand() {
    r[0] = 1;
    for (int i = 1; i < 31; i++) {
      int64_t v = (16807ll * (int32_t)r[i - 1]) % 2147483647ll;
      if (v < 0) v += 2147483647ll;
      r[i] = (uint32_t)v;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    for (int i = 0; i < 310; i++) step();
  }
  uint32_t step() {  // ring of 34: r[k] holds x[i-34]; x[i] = x[i-3] + x[i-31]
    const uint32_t v = r[(k + 31) % 34] + r[(k + 3) % 34];
    r[k] = v;
    k = (k + 1) % 34;
    return v;
  }
  int next() { return (int)(step() >> 1); }
};
static GlibcRand draws;

static int run_multi_gpu(Opts& o, bool pe, int max_ml);

// A run that fails after it created its output leaves no artefact behind (a SAM sized in advance and padded with NULs, a BAM
// without its end-of-file block), and the pipeline's threads and pinned buffers are gone before main returns.
struct RunGuard {
  k4_pipeline** pl;
  std::vector<std::string> made;
  bool ok = false;
  ~RunGuard() {
    if (ok) return;
    if (pl && *pl) { k4_pipeline_close(*pl); *pl = nullptr; }
    for (const std::string& f : made) {
      struct stat sb;
      if (lstat(f.c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) remove(f.c_str());  // (never a device node or /dev/stdout's link)
    }
  }
};

// one process, one GPU: the whole run, or rank o.rank of a -G run
static int run_rank(Opts& o, const bool pe, const int max_ml) {
  auto t0 = std::chrono::steady_clock::now();
  k4_index* ix = nullptr;
  k4_comm* comm = nullptr;
  int rc;
  const bool multi = o.n_ranks > 1 || o.shared;
  const bool chatty = o.rank == 0;  // only rank 0 of a -G run reports
  const char* fault = multi ? getenv("K4ALIGN_FAULT") : nullptr;  // tests of run_multi_gpu: "<rank>:<stage>"
  const int fault_rank = fault ? atoi(fault) : -1;
  const char* fault_stage = fault && strchr(fault, ':') ? strchr(fault, ':') + 1 : "";
  if (multi && fault) {
    if (fault_rank == o.rank && strcmp(fault_stage, "start") == 0) { fprintf(stderr, "k4align: rank %d: injected fault at start\n", o.rank); return 3; }
    const char* peers = getenv("K4ALIGN_FAULT_PEERS");
    if (fault_rank != o.rank && peers && strcmp(peers, "block") == 0)
      for (;;) pause();  // what a rank blocked in a collective looks like to the parent
  }
  if (multi) {
    // the communicator id: rank 0 makes it, the others pick it up from the shared mapping
    MultiShared* sh = o.shared;
    if (o.rank == 0) {
      if ((rc = k4_comm_unique_id(sh->id)) != K4_OK) { fprintf(stderr, "k4align: ncclGetUniqueId failed\n"); sh->failed = 1; return 2; }
      sh->id_ready = 1;
    } else
      while (!sh->id_ready.load()) { if (sh->failed.load()) return 2; usleep(1000); }
    if ((rc = k4_comm_init(o.gpu, o.rank, o.n_ranks, sh->id, &comm)) != K4_OK) { fprintf(stderr, "k4align: rank %d: RCCL communicator failed (%d)\n", o.rank, rc); sh->failed = 1; return 2; }
    // rank 0 reads the .sfx once; sequence + suffix array reach the peers over xGMI; every rank builds its own tables
    rc = k4_comm_open_index(comm, o.rank == 0 ? o.sfx.c_str() : nullptr, 0, &ix);
    if (rc != K4_OK) { fprintf(stderr, "k4align: rank %d: index broadcast failed: %s (%d)\n", o.rank, k4_comm_last_error(comm), rc); sh->failed = 1; return 2; }
    if (fault_rank == o.rank && strcmp(fault_stage, "index") == 0) { fprintf(stderr, "k4align: rank %d: injected fault behind the index broadcast\n", o.rank); return 3; }
  } else {
    // (the arrays load on a thread of the library while the reads are read: k4_open_wait below, or the pipeline's own)
    rc = k4_open_async(o.sfx.c_str(), o.gpu, 0, &ix);
    if (rc != K4_OK) { fprintf(stderr, "k4align: unable to load '%s': %s (%d)\n", o.sfx.c_str(), k4_global_error(), rc); return 2; }
  }
  k4_info_t info;
  k4_info(ix, &info);
  if (o.q_method != 3) CK(k4_set_fastq_quality(ix, o.q_method));
  // SetMaxIter by sensitivity, KAligner.cpp:373-388
  k4_set_max_iter(ix, o.pmode == 2 ? 20000 : o.pmode == 1 ? 10000 : o.pmode == 0 ? 5000 : 2500);
  int slides = 0;
  const int mcl = k4_min_core_len(ix, o.pmode, &slides);
  if (chatty)
    fprintf(stderr, "k4align: index '%s' %u sequences, %llu bp; minimum core size %dbp%s\n", info.dataset, info.n_entries,
            (unsigned long long)info.tot_seqs_len, mcl, multi ? " (read once, sent to the other GPUs over xGMI)" : "");
  auto t_open = std::chrono::steady_clock::now();

  if ((o.ml_mode == 3 || o.ml_mode == 4) && (o.batch_mb > 0 || o.n_shards > 1)) {
    fprintf(stderr, "k4align: -r3 / -r4 cluster over all reads of the run; they cannot be combined with -b or -S\n");
    return 1;
  }
  if (o.batch_mb > 0 && o.n_shards > 1) { fprintf(stderr, "k4align: -S slices the whole input; it cannot be combined with -b\n"); return 1; }
  if ((o.micro_indel || o.splice_junct) && (o.batch_mb > 0 || o.n_shards > 1)) {
    fprintf(stderr, "k4align: -a / -A drop junctions no second read of the RUN supports; they cannot be combined with -b or -S\n");
    return 1;
  }
  k4_kalign_params kp = {o.max_subs, o.min_edit, o.max_ns, o.pmode, o.align_strand /* K4_STRAND_*: the same codes as eALStrand */, max_ml,
                         o.ml_mode == 5 ? (o.best ? 4 : o.clamp ? 3 : 2) : o.ml_mode == 2 ? 2 : o.ml_mode != 0 ? 1 : 0,
                         mcl, slides, o.min_chimeric, o.micro_indel, o.splice_junct};
  const bool two_seg = o.micro_indel > 0 || o.splice_junct > 0;
  k4_pe_params pp = {o.pe_mode, o.pair_min, o.pair_max, o.pair_strand};
  k4_sam_stats tot;
  memset(&tot, 0, sizeof(tot));
  std::vector<uint8_t> hit_chrom(info.n_entries + 1, 0);
  uint64_t n_under = 0, n_over = 0, n_units = 0;
  double s_read = 0, s_parse = 0, s_align = 0, s_write = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };

  // the stages between alignment and report, over ALL reads of the run (CKAligner::Align, KAligner.cpp:615-686)
  auto global_stages = [&](int64_t n, uint32_t max_len, void* d_rr, void* d_hits, void* d_seg2, void* d_pe, void* d_reads, void* d_offs,
                           void* d_lens) -> int {
    if (n <= 0 || max_len == 0) return K4_OK;
    if (!pe) {
      if (o.ml_mode == 3 || o.ml_mode == 4) {  // AssignMultiMatches (KAligner.cpp:5092): clusters over all reads of the run
        int64_t n_assigned = 0;
        CK(k4_assign_multi_dev(ix, o.ml_mode, (int32_t)max_len, n, max_ml, d_rr, d_hits, &n_assigned, nullptr));
        fprintf(stderr, "k4align: %lld multi-aligned reads assigned to one locus by clustering\n", (long long)n_assigned);
      }
      if (o.ml_mode == 2) {
        // eMLrand (KAligner.cpp:9945-9962): rand() once per read within the instance limit, in load order -- what the
        // reference does when it runs one thread (with more its draws depend on thread timing).  The draws are a private
        // restatement of the C library's generator: other code in this process (the HIP runtime) may call rand() too
        std::vector<k4_read_result> rr((size_t)n);
        std::vector<uint32_t> choice((size_t)n, 0);
        CK(k4_copy_to_host(ix, rr.data(), d_rr, (uint64_t)n * sizeof(k4_read_result)));
        for (int64_t i = 0; i < n; i++)
          if (rr[i].nar == K4_NAR_ACCEPTED && rr[i].num_hits >= 1) choice[i] = (uint32_t)(draws.next() % rr[i].num_hits);
        void* d_choice = nullptr;
        CK(k4_alloc_device(ix, (uint64_t)n * 4, &d_choice));
        CK(k4_copy_to_device(ix, d_choice, choice.data(), (uint64_t)n * 4));
        CK(k4_select_hits_dev(ix, n, max_ml, d_rr, d_hits, d_choice, nullptr));
        k4_free_device(d_choice);
      }
    }
    // the filters, in CKAligner::Align's order (KAligner.cpp:653-686)
    int64_t cnt = 0;
    if (o.min_flank_exacts > 0) {
      CK(k4_auto_trim_flanks_dev(ix, o.min_flank_exacts, pe ? 1 : 0, pe ? 2 * n : n, max_ml, pe ? d_pe : d_rr, d_hits, d_reads, d_offs,
                                 d_lens, &cnt, nullptr));
      fprintf(stderr, "k4align: flank autotrim to %d exact bases: %lld aligned reads removed\n", o.min_flank_exacts, (long long)cnt);
    }
    if (!pe && o.splice_junct > 0) {
      CK(k4_remove_orphan_juncts_dev(ix, K4_EXT_SPLICE, n, max_ml, d_rr, d_hits, d_seg2, &cnt, nullptr));
      fprintf(stderr, "k4align: %lld orphan splice junction reads removed\n", (long long)cnt);
    }
    if (!pe && o.micro_indel > 0) {
      CK(k4_remove_orphan_juncts_dev(ix, K4_EXT_INDEL, n, max_ml, d_rr, d_hits, d_seg2, &cnt, nullptr));
      fprintf(stderr, "k4align: %lld orphan microInDel reads removed\n", (long long)cnt);
    }
    return K4_OK;
  };

  // One batch: text holding whole records (the last one possibly cut when more text follows) -> records -> alignments ->
  // SAM body.  The body goes to `body` (a part file) or, for the single batch of the whole-input mode, stays on the device.
  void* keep_sam = nullptr;
  uint64_t keep_bytes = 0;
  auto run_batch = [&](const uint8_t* t1, uint64_t T1, int fin1, const uint8_t* t2, uint64_t T2, int fin2, FILE* body,
                       uint64_t* used1, uint64_t* used2) -> int {
    *used1 = *used2 = 0;
    auto ta = now();
    void* d_reads = nullptr;
    CK(k4_alloc_device(ix, T1 + T2 + 64, &d_reads));
    Parsed p1, p2;
    CK(load_reads(ix, t1, T1, fin1, 0, d_reads, 0, p1, used1));
    if (pe) {
      CK(load_reads(ix, t2, T2, fin2, (int64_t)std::max<uint64_t>(p1.n, 1), d_reads, p1.bases, p2, used2));
      if (p2.n < p1.n) {  // this portion of the mates' file holds fewer records: keep the pairs both files delivered
        if (fin2 && fin1) { fprintf(stderr, "k4align: fewer PE2 than PE1 reads\n"); return 3; }
        const int64_t n2 = (int64_t)p2.n;
        free_parsed(p1);
        free_parsed(p2);
        if (n2 == 0) { k4_free_device(d_reads); *used1 = *used2 = 0; return K4_OK; }
        CK(load_reads(ix, t1, T1, fin1, n2, d_reads, 0, p1, used1));
        CK(load_reads(ix, t2, T2, fin2, n2, d_reads, p1.bases, p2, used2));
      }
    }
    const Parsed a1 = p1, a2 = p2;  // the allocations (p1 / p2 are narrowed to this process's slice below)
    // -S i/N: reads (pairs) are independent units; every process parses the whole input -- that is cheap
    // on the device -- and keeps its contiguous slice, so that the shards' SAM files merge back into load order (k4merge)
    const int64_t r0 = (int64_t)p1.n * o.shard / o.n_shards, r1 = (int64_t)p1.n * (o.shard + 1) / o.n_shards;
    for (Parsed* q : {&p1, &p2}) {
      if (!q->d_offs) continue;
      q->d_offs = (uint8_t*)q->d_offs + 8 * r0; q->d_lens = (uint8_t*)q->d_lens + 4 * r0;
      q->d_noff = (uint8_t*)q->d_noff + 8 * r0; q->d_nlen = (uint8_t*)q->d_nlen + 4 * r0;
    }
    const int64_t n = r1 - r0;
    const int64_t n_reads = pe ? 2 * n : n;
    uint64_t under = 0, over = 0;
    uint32_t max_len = 0;
    void *d_offs = nullptr, *d_lens = nullptr;
    CK(k4_alloc_device(ix, (uint64_t)(n_reads + 1) * 8, &d_offs));
    CK(k4_alloc_device(ix, (uint64_t)(n_reads + 1) * 4, &d_lens));
    CK(k4_prepare_reads_trim_dev(ix, pe ? 1 : 0, n, o.min_len, o.max_len, o.trim5, o.trim3, 1, 0, p1.d_offs, p1.d_lens, p2.d_offs, p2.d_lens, 0, d_offs, d_lens,
                            &under, &over, &max_len, nullptr));
    auto tb = now();
    // ---- align (ProcCoredApprox / ProcessPairedEnds) ---------------------------------------------------------------
    void *d_rr = nullptr, *d_hits = nullptr, *d_pe = nullptr, *d_seg2 = nullptr;
    if (n > 0 && max_len > 0) {
      if (!pe) {
        CK(k4_alloc_device(ix, (uint64_t)n * sizeof(k4_read_result), &d_rr));
        CK(k4_alloc_device(ix, (uint64_t)n * max_ml * sizeof(k4_hit), &d_hits));
        if (two_seg) CK(k4_alloc_device(ix, (uint64_t)n * sizeof(k4_seg2), &d_seg2));
        CK(k4_reserve(ix, n, (int32_t)max_len, max_ml));
        CK(k4_kalign_ext_batch_dev(ix, &kp, n, (int32_t)max_len, d_reads, d_offs, d_lens, d_rr, d_hits, d_seg2, nullptr));
      } else {
        CK(k4_alloc_device(ix, (uint64_t)2 * n * sizeof(k4_pe_read), &d_pe));
        CK(k4_kalign_pe_batch_dev(ix, &kp, &pp, n, (int32_t)max_len, d_reads, d_offs, d_lens, d_pe, nullptr));
      }
      CK(global_stages(n, max_len, d_rr, d_hits, d_seg2, d_pe, d_reads, d_offs, d_lens));
    }
    // ---- SAM body on the device (k4_format_sam_dev) ------------------------------------------------------------------
    k4_sam_names nm;
    memset(&nm, 0, sizeof(nm));
    nm.d_text[0] = p1.d_text; nm.d_name_off[0] = p1.d_noff; nm.d_name_len[0] = p1.d_nlen;
    nm.d_text[1] = p2.d_text; nm.d_name_off[1] = p2.d_noff; nm.d_name_len[1] = p2.d_nlen;
    void* d_sam = nullptr;
    uint64_t sam_bytes = 0;
    k4_sam_stats stt;
    memset(&stt, 0, sizeof(stt));
    std::vector<uint8_t> hc(info.n_entries + 1, 0);
    if (n > 0 && max_len > 0)
      CK(k4_format_sam_ext_dev(ix, pe ? 1 : 0, n, d_rr, d_hits, max_ml, d_pe, d_seg2, d_reads, d_offs, d_lens, &nm, &d_sam, &sam_bytes,
                               &stt, hc.data(), nullptr));
    auto tc = now();
    for (int k = 0; k < 20; k++) tot.nar[k] += stt.nar[k];
    tot.plus += stt.plus; tot.minus += stt.minus; tot.n_lines += stt.n_lines;
    for (size_t c = 0; c < hc.size(); c++) hit_chrom[c] |= hc[c];
    n_under += under; n_over += over; n_units += (uint64_t)n;
    if (body) {
      std::vector<char> piece((size_t)std::min<uint64_t>(sam_bytes, 256ull << 20));
      for (uint64_t off = 0; off < sam_bytes; off += piece.size()) {
        const uint64_t len = std::min<uint64_t>(piece.size(), sam_bytes - off);
        CK(k4_copy_to_host(ix, piece.data(), (const char*)d_sam + off, len));
        if (fwrite(piece.data(), 1, len, body) != len) { fprintf(stderr, "k4align: write failed\n"); return 5; }
      }
      k4_free_device(d_sam);
    } else {
      keep_sam = d_sam;
      keep_bytes = sam_bytes;
    }
    for (void* q : {d_reads, d_offs, d_lens, d_rr, d_hits, d_pe, d_seg2, a1.d_text, a1.d_offs, a1.d_lens, a1.d_noff, a1.d_nlen, a2.d_text,
                    a2.d_offs, a2.d_lens, a2.d_noff, a2.d_nlen})
      k4_free_device(q);
    auto td = now();
    s_parse += secs(ta, tb); s_align += secs(tb, tc); s_write += secs(tc, td);
    return K4_OK;
  };

  // ---- the input ---------------------------------------------------------------------------------------------------------------
  // default: the overlapped pipeline (k4_pipeline_*): reader threads -> pinned ring -> copy stream || parse + align on the
  // compute stream; one global sort + SAM body at the end, handed out while its next pieces come down.
  // -b <MB>: bounded memory, coordinate-sorted parts merged on the host; -S i/N: one slice of the reads (one process per GPU)
  k4_pipeline* pl = nullptr;
  RunGuard guard{&pl};
  uint64_t pl_sam_bytes = 0;
  const bool pipelined = o.batch_mb <= 0 && o.n_shards == 1 && !o.legacy;
  // "-o x.bam": BGZF compressed BAM, any other extension SAM (KAlignerCL.cpp:857-866)
  const bool bam_out = o.rank_bam || (o.out.size() >= 4 && strcasecmp(o.out.c_str() + o.out.size() - 4, ".bam") == 0);
  if (!pipelined && !multi && (rc = k4_open_wait(ix)) != K4_OK) { fprintf(stderr, "k4align: unable to load '%s': %s (%d)\n", o.sfx.c_str(), k4_global_error(), rc); return 2; }
  if (bam_out && (!pipelined || (multi && !o.rank_bam))) { fprintf(stderr, "k4align: BAM output is written by the pipelined modes (not with -b, -S, -Z)\n"); return 3; }
  if (pipelined) {
    for (const std::vector<std::string>* fs : {&o.in1, &o.in2})
      for (const std::string& q : *fs) {
        FILE* t = fopen(q.c_str(), "rb");
        if (!t) { fprintf(stderr, "k4align: unable to open '%s'\n", q.c_str()); return 2; }
        fclose(t);
      }
    k4_pipeline_params pp2;
    memset(&pp2, 0, sizeof(pp2));
    pp2.paired = pe ? 1 : 0; pp2.kp = kp; pp2.pe = pp; pp2.min_len = o.min_len; pp2.max_len = o.max_len;
    pp2.chunk_bytes = (uint64_t)o.chunk_mb << 20;
    for (int e = 0; e < (pe ? 2 : 1); e++) {
      bool plain = true;
      uint64_t tot_sz = 0;
      for (const std::string& q : e ? o.in2 : o.in1) { plain &= !is_gzip(q); tot_sz += file_size(q) + 1; }
      pp2.expect_text_bytes[e] = plain ? tot_sz : 0;
    }
    CK(k4_pipeline_open(ix, &pp2, &pl));
    CK(k4_pipeline_set_trims(pl, o.trim5, o.trim3));
    CK(k4_pipeline_set_sampling(pl, o.sample_nth));
    auto tr = now();
    int rc_end[2] = {K4_OK, K4_OK};
    double rd[2] = {0, 0};
    std::thread t2;
    std::vector<FileRange> fr[2];  // (outlive the reader thread of the second end)
    if (multi) {
      // reads are independent units: rank r aligns the r-th contiguous slice of the records (pairs stay together).  Rank 0
      // finds the slice boundaries (one pass over the files), every rank then reads only its own byte range
      MultiShared* sh = o.shared;
      bool any_gz = false;
      for (const std::string& q : o.in1) any_gz |= is_gzip(q);
      for (const std::string& q : o.in2) any_gz |= is_gzip(q);
      if (any_gz) {  // no byte offsets into a compressed stream: record blocks are dealt to the ranks
        if (pe) t2 = std::thread([&] { rc_end[1] = feed_dealt(pl, 1, o.in2, o.rank, o.n_ranks, &rd[1]); });
        rc_end[0] = feed_dealt(pl, 0, o.in1, o.rank, o.n_ranks, &rd[0]);
      } else {
        if (o.rank == 0) {
          uint64_t R = 0, R2 = 0;
          std::vector<uint64_t> counts;
          std::string why;
          bool ok = slice_files(o.in1, o.n_ranks, counts, false, sh->slice_off[0], &R, o.io_threads, &why);
          if (ok && pe) ok = o.in2.size() == o.in1.size() ? slice_files(o.in2, o.n_ranks, counts, true, sh->slice_off[1], &R2, o.io_threads, &why)
                                                          : (why = "as many -u files as -i files are needed", false);
          if (!ok) { fprintf(stderr, "k4align: unable to cut the reads into slices: %s\n", why.c_str()); sh->failed = 1; return 2; }
          sh->n_records = R;
          sh->slices_ready = 1;
        } else
          while (!sh->slices_ready.load()) { if (sh->failed.load()) return 2; usleep(1000); }
        for (int e = 0; e < (pe ? 2 : 1); e++) {
          const std::vector<std::string>& fl = e ? o.in2 : o.in1;
          for (size_t f = 0; f < fl.size(); f++)
            fr[e].push_back(FileRange{fl[f], sh->slice_off[e][f][o.rank], sh->slice_off[e][f][o.rank + 1], file_size(fl[f])});
        }
        if (pe) t2 = std::thread([&] { rc_end[1] = feed_ranges(pl, 1, fr[1], o.io_threads, &rd[1]); });
        rc_end[0] = feed_ranges(pl, 0, fr[0], o.io_threads, &rd[0]);
      }
    } else {
      if (pe) t2 = std::thread([&] { rc_end[1] = feed_end(pl, 1, o.in2, o.io_threads, &rd[1]); });
      rc_end[0] = feed_end(pl, 0, o.in1, o.io_threads, &rd[0]);
    }
    if (pe) t2.join();
    s_read = std::max(rd[0], rd[1]);
    for (int e = 0; e < 2; e++)
      if (rc_end[e] != K4_OK) { fprintf(stderr, "k4align: error reading the input (%d): %s\n", rc_end[e], k4_last_error(ix)); return 2; }
    k4_pipeline_view v;
    CK(k4_pipeline_wait_aligned(pl, &v));
    s_parse = secs(tr, now()) - s_read;  // what the device side added behind the reading
    auto tg = now();
    CK(global_stages(v.n_units, v.max_read_len, v.d_rr, v.d_hits, v.d_seg2, v.d_pe, v.d_reads, v.d_offs, v.d_lens));
    for (int which = 0; which < 2; which++) {  // ReportNoneAligned / ReportMultiAlign (KAligner.cpp:709-733): before the alignments are reported
      const std::string& fn = which ? o.multi_file : o.none_file;
      if (fn.empty()) continue;
      char* txt = nullptr;
      uint64_t nb = 0, nl = 0;
      CK(k4_unaligned_fasta_dev(ix, pe ? 1 : 0, v.n_units, v.d_rr, v.d_pe, v.d_reads, v.d_offs, v.d_lens, &v.names, which, &txt, &nb, &nl, nullptr));
      FILE* fp = fopen(fn.c_str(), "wb");
      bool ok = fp && fwrite(txt, 1, nb, fp) == nb;
      if (fp && fclose(fp) != 0) ok = false;
      k4_free_host(txt);
      if (!ok) { fprintf(stderr, "k4align: unable to write %s\n", fn.c_str()); return 5; }
      if (chatty) fprintf(stderr, "k4align: %llu %s reads written to %s\n", (unsigned long long)nl, which ? "multi-aligned" : "unalignable", fn.c_str());
    }
    if (o.min_snp_reads > 0) {  // ProcessSNPs (KAligner.cpp:768-790 calls it behind the alignment report): the SNP file and its side files
      auto ts = now();
      // a file name ending in .vcf: VCF instead of the CSV (KAligner.cpp:186-187)
      const bool vcf = o.snp_file.size() >= 4 && strcasecmp(o.snp_file.c_str() + o.snp_file.size() - 4, ".vcf") == 0;
      k4_snp_files sf;
      CK(k4_snp_run_dev(ix, vcf ? 1 : 0, pe ? 1 : 0, v.n_units, v.d_rr, v.d_hits, v.max_ml, v.d_pe, v.d_reads, v.d_offs, v.d_lens, o.min_snp_reads,
                        o.qvalue, o.snp_nonref_pcnt, &sf, nullptr));
      // side files: <snp file cut at its last '.'> + suffix (CUtility::AppendFileNameSuffix, KAligner.cpp:4512, 4553-4554)
      std::string stem = o.snp_file;
      for (size_t q = stem.size(); q > 0; q--) {
        if (stem[q - 1] == '.') { stem.resize(q - 1); break; }
        if (stem[q - 1] == '/' || stem[q - 1] == '\\') break;
      }
      const struct { std::string name; const char* p; uint64_t n; } files[4] = {{stem + ".covsegs.wig", sf.wig, sf.wig_bytes},
                                                                                {stem + ".disnp.csv", sf.disnp, sf.disnp_bytes},
                                                                                {stem + ".trisnp.csv", sf.trisnp, sf.trisnp_bytes},
                                                                                {o.snp_file, sf.snp, sf.snp_bytes}};
      std::string failed;
      for (const auto& f : files) {  // (the coverage WIG of a genome at low coverage runs to gigabytes: pwrite()s side by side)
        FILE* fp = fopen(f.name.c_str(), "wb");
        bool ok = fp != nullptr;
        if (ok && f.n >= (64u << 20)) {
          const int fd = fileno(fp);
          const int nt = std::max(o.io_threads, 1);
          std::atomic<bool> good(true);
          std::vector<std::thread> th;
          for (int t = 0; t < nt; t++)
            th.emplace_back([&, t] {
              const uint64_t a = f.n * (uint64_t)t / nt, b = f.n * (uint64_t)(t + 1) / nt;
              uint64_t done = a;
              while (done < b) {
                const ssize_t g = pwrite(fd, f.p + done, b - done, (off_t)done);
                if (g <= 0) { good = false; return; }
                done += (uint64_t)g;
              }
            });
          for (std::thread& x : th) x.join();
          ok = good;
        } else if (ok)
          ok = fwrite(f.p, 1, f.n, fp) == f.n;
        if (fp && fclose(fp) != 0) ok = false;
        if (!ok && failed.empty()) failed = f.name;
      }
      k4_free_host(sf.snp); k4_free_host(sf.wig); k4_free_host(sf.disnp); k4_free_host(sf.trisnp);
      if (!failed.empty()) { fprintf(stderr, "k4align: unable to write %s\n", failed.c_str()); return 5; }
      if (chatty) fprintf(stderr, "k4align: SNP processing completed with %llu putative SNPs discovered, written to %s in %.2fs\n",
                          (unsigned long long)sf.n_snps, o.snp_file.c_str(), secs(ts, now()));
    }
    // (a rank of a -G run numbers every sequence: the parent renumbers when it knows which ones any rank has hit)
    const int bam_all_sq = (info.n_entries <= (uint32_t)o.rpt_sq_thres || o.rank_bam) ? 1 : 0;
    if (bam_out && o.fmode == 1) CK(k4_pipeline_format_bam_all(pl, bam_all_sq, &tot, hit_chrom.data(), &pl_sam_bytes));
    else if (bam_out) CK(k4_pipeline_format_bam(pl, bam_all_sq, &tot, hit_chrom.data(), &pl_sam_bytes));
    else if (o.fmode == 1) CK(k4_pipeline_format_all(pl, &tot, hit_chrom.data(), &pl_sam_bytes));
    else CK(k4_pipeline_format(pl, &tot, hit_chrom.data(), &pl_sam_bytes));
    s_align = secs(tg, now());
    n_under = v.n_under; n_over = v.n_over; n_units = (uint64_t)v.n_units;
  }
  Stream f1, f2;
  if (!pipelined && (!f1.open(o.in1) || (pe && !f2.open(o.in2)))) { fprintf(stderr, "k4align: unable to open reads\n"); return 2; }
  std::vector<std::string> parts;
  const uint64_t want0 = o.batch_mb > 0 ? std::max<uint64_t>((uint64_t)(o.batch_mb * 1048576.0), 4096) : UINT64_MAX;
  uint64_t want = want0;
  for (; !pipelined;) {
    auto tr = now();
    if (!f1.fill(want) || (pe && !f2.fill(want))) { fprintf(stderr, "k4align: error reading the input\n"); return 2; }
    s_read += secs(tr, now());
    if (f1.buf.empty()) break;
    const bool whole = o.batch_mb <= 0;
    FILE* body = nullptr;
    if (!whole) {
      parts.push_back(o.out + ".part" + std::to_string(parts.size()));
      body = fopen(parts.back().c_str(), "wb");
      if (!body) { fprintf(stderr, "k4align: unable to create %s\n", parts.back().c_str()); return 5; }
    }
    uint64_t u1 = 0, u2 = 0;
    rc = run_batch(f1.buf.data(), f1.buf.size(), f1.eof, f2.buf.data(), f2.buf.size(), f2.eof, body, &u1, &u2);
    if (body) fclose(body);
    if (rc != K4_OK) return rc;
    if (u1 == 0) {  // not one whole record (pair) in this much text
      if (!whole) { remove(parts.back().c_str()); parts.pop_back(); }
      if (f1.eof && (!pe || f2.eof)) break;
      want *= 2;
      continue;
    }
    want = want0;
    f1.drop(u1);
    f2.drop(u2);
    if (whole) break;
  }
  f1.close();
  f2.close();

  // ---- statistics (ReportAlignStats) ------------------------------------------------------------------------------
  const uint64_t my_lines = tot.n_lines;
  if (comm) {  // the final aligned-read count/merge: one all-reduce of the tallies (north_star: the only collective on the path)
    uint64_t t[26];
    for (int k = 0; k < 20; k++) t[k] = tot.nar[k];
    t[20] = tot.plus; t[21] = tot.minus; t[22] = tot.n_lines; t[23] = n_under; t[24] = n_over; t[25] = n_units;
    if ((rc = k4_comm_allreduce_sum_u64(comm, t, 26)) != K4_OK) { fprintf(stderr, "k4align: rank %d: all-reduce failed: %s\n", o.rank, k4_comm_last_error(comm)); return 2; }
    for (int k = 0; k < 20; k++) tot.nar[k] = t[k];
    tot.plus = t[20]; tot.minus = t[21]; tot.n_lines = t[22]; n_under = t[23]; n_over = t[24]; n_units = t[25];
  }
  const uint64_t n_loaded = (uint64_t)(pe ? 2 : 1) * (n_units - n_under - n_over);
  if (chatty) {
    fprintf(stderr, "k4align: From %llu source reads there are %llu accepted alignments, %llu on '+' strand, %llu on '-' strand\n",
            (unsigned long long)n_loaded, (unsigned long long)tot.nar[1], (unsigned long long)tot.plus, (unsigned long long)tot.minus);
    if (n_under || n_over)
      fprintf(stderr, "k4align: %llu under length and %llu over length reads were sloughed\n", (unsigned long long)n_under,
              (unsigned long long)n_over);
    for (int k = 0; k < 20; k++) fprintf(stderr, "k4align:    %llu (%s)\n", (unsigned long long)tot.nar[k], kNarAbbr[k]);
  }

  // ---- SAM file: header here, body as formatted on the device ----------------------------------------------------------
  auto tw = now();
  if (bam_out && o.rank_bam) {
    // ---- a rank of a -G run: its coordinate-sorted BAM records as they are (refID = sequence number - 1), and beside them the
    // dictionary with this rank's hit flags; the parent merges the ranks' streams into the one BAM file (run_multi_gpu) ----------
    FILE* fp = fopen(o.out.c_str(), "wb");
    if (!fp) { fprintf(stderr, "k4align: unable to create %s\n", o.out.c_str()); return 5; }
    guard.made.push_back(o.out);
    guard.made.push_back(o.out + ".sq");
    uint64_t nbytes = 0;
    for (;;) {
      const void* ptr = nullptr;
      uint64_t len = 0;
      CK(k4_pipeline_next_sam(pl, &ptr, &len));
      if (len == 0) break;
      if (fwrite(ptr, 1, (size_t)len, fp) != (size_t)len) { fprintf(stderr, "k4align: unable to write %s\n", o.out.c_str()); fclose(fp); return 5; }
      nbytes += len;
    }
    if (fclose(fp) != 0) { fprintf(stderr, "k4align: unable to write %s\n", o.out.c_str()); return 5; }
    FILE* fd = fopen((o.out + ".sq").c_str(), "wb");
    if (!fd) { fprintf(stderr, "k4align: unable to create %s.sq\n", o.out.c_str()); return 5; }
    fprintf(fd, "%s\n", info.dataset);
    for (uint32_t c = 1; c <= info.n_entries; c++) {
      k4_entry e;
      k4_get_entry(ix, c, &e);
      fprintf(fd, "%s\t%u\t%d\n", e.name, e.seq_len, hit_chrom[c] ? 1 : 0);
    }
    if (fclose(fd) != 0) { fprintf(stderr, "k4align: unable to write %s.sq\n", o.out.c_str()); return 5; }
    k4_pipeline_close(pl);
    pl = nullptr;
    guard.ok = true;
    if (chatty) fprintf(stderr, "k4align: rank %d: %llu alignments as BAM records (%llu bytes) for the merge\n", o.rank, (unsigned long long)my_lines, (unsigned long long)nbytes);
    k4_close(ix);
    return 0;
  }
  if (bam_out) {
    // ---- BAM (+ .bai): dictionary and BGZF blocks here (include/k4_bam.hpp), the records as packed on the device ------------
    std::string hdr = "@HD\tVN:1.4\tSO:coordinate\n";
    std::vector<k4bam::RefSeq> refs;
    const bool all_sq = info.n_entries <= (uint32_t)o.rpt_sq_thres;  // m_MaxRptSAMSeqsThres, KAligner.cpp:5785-5821
    for (uint32_t c = 1; c <= info.n_entries; c++) {
      k4_entry e;
      k4_get_entry(ix, c, &e);
      if (!(all_sq || hit_chrom[c])) continue;
      hdr += std::string("@SQ\tAS:") + info.dataset + "\tSN:" + e.name + "\tLN:" + std::to_string(e.seq_len) + "\n";
      refs.push_back({e.name, e.seq_len});
    }
    hdr += "@PG\tID:k4align\tVN:1.0\n";
    k4bam::Writer bw;
    if (!bw.open(o.out, hdr, refs, o.bam_level, std::max(o.io_threads, 1))) { fprintf(stderr, "k4align: %s\n", bw.error().c_str()); return 5; }
    guard.made.push_back(o.out);
    guard.made.push_back(o.out + ".bai");
    for (;;) {  // the records come down piece by piece; the previous piece is deflated and written meanwhile
      const void* ptr = nullptr;
      uint64_t len = 0;
      CK(k4_pipeline_next_sam(pl, &ptr, &len));
      if (len == 0) break;
      if (!bw.write(ptr, (size_t)len)) { fprintf(stderr, "k4align: %s\n", bw.error().c_str()); return 5; }
    }
    if (!bw.close()) { fprintf(stderr, "k4align: %s\n", bw.error().c_str()); return 5; }
    if (!bw.indexed() && chatty) fprintf(stderr, "k4align: a sequence of 512 Mbp or more: no .bai written (the reference writes a CSI index there)\n");
    if (bw.n_records() != my_lines) { fprintf(stderr, "k4align: internal error: %llu BAM records for %llu alignments\n", (unsigned long long)bw.n_records(), (unsigned long long)my_lines); return 5; }
    k4_pipeline_close(pl);
    pl = nullptr;
    guard.ok = true;
    const double s_write_bam = secs(tw, now());
    if (chatty)
      fprintf(stderr, "k4align: %llu alignments reported to %s (%llu bytes) + .bai; index %.2fs, reads %.2fs, device side behind the reads %.2fs, "
                      "global stages + sort + records %.2fs, deflate + write %.2fs\n", (unsigned long long)bw.n_records(), o.out.c_str(),
              (unsigned long long)bw.compressed_bytes() + 28, std::max(secs(t0, t_open), k4_open_seconds(ix)), s_read, s_parse, s_align, s_write_bam);
    k4_close(ix);
    return 0;
  }
  FILE* fp = fopen(o.out.c_str(), "wb");
  if (!fp) { fprintf(stderr, "k4align: unable to create %s\n", o.out.c_str()); return 5; }
  guard.made.push_back(o.out);
  static char iobuf[1 << 22];
  setvbuf(fp, iobuf, _IOFBF, sizeof(iobuf));
  fprintf(fp, "@HD\tVN:1.4\tSO:coordinate\n");
  // m_MaxRptSAMSeqsThres, KAligner.cpp:5785-5821: with more sequences only those that were hit are declared -- a shard (-S, -G)
  // declares all of them and leaves that rule to the merge, which sees every shard's records
  const bool all_chroms = info.n_entries <= (uint32_t)o.rpt_sq_thres || o.n_shards > 1 || multi;
  std::map<std::string, long> order;
  for (uint32_t c = 1; c <= info.n_entries; c++) {
    k4_entry e;
    k4_get_entry(ix, c, &e);
    order.emplace(e.name, (long)c);
    if (all_chroms || hit_chrom[c]) fprintf(fp, "@SQ\tAS:%s\tSN:%s\tLN:%u\n", info.dataset, e.name, e.seq_len);
  }
  fprintf(fp, "@PG\tID:k4align\tVN:1.0\n");
  if (pl) {  // the body comes down piece by piece while the previous piece is written
    fflush(fp);
    const int fd = fileno(fp);
    const bool seekable = lseek(fd, 0, SEEK_CUR) != (off_t)-1;  // (-o /dev/stdout, a pipe: written in order by one thread)
    uint64_t fpos = seekable ? (uint64_t)ftello(fp) : 0;
    if (seekable && ftruncate(fd, (off_t)(fpos + pl_sam_bytes)) != 0) { /* (the size is only a hint for the file system) */ }
    for (;;) {
      const void* ptr = nullptr;
      uint64_t len = 0;
      CK(k4_pipeline_next_sam(pl, &ptr, &len));
      if (len == 0) break;
      if (!seekable) {
        if (fwrite(ptr, 1, (size_t)len, fp) != (size_t)len) { fprintf(stderr, "k4align: write to %s failed\n", o.out.c_str()); return 5; }
        continue;
      }
      // (several pwrite()s side by side: one thread does not saturate tmpfs / the page cache)
      const int nt = len >= (8u << 20) ? std::max(o.io_threads, 1) : 1;
      std::atomic<bool> ok(true);
      std::vector<std::thread> th;
      for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] {
          const uint64_t a = len * t / nt, b = len * (t + 1) / nt;
          uint64_t done = a;
          while (done < b) {
            const ssize_t g = pwrite(fd, (const char*)ptr + done, b - done, (off_t)(fpos + done));
            if (g <= 0) { ok = false; return; }
            done += (uint64_t)g;
          }
        });
      for (std::thread& x : th) x.join();
      if (!ok) { fprintf(stderr, "k4align: write to %s failed\n", o.out.c_str()); return 5; }
      fpos += len;
    }
    if (seekable) fseeko(fp, (off_t)fpos, SEEK_SET);
    {
      auto tc0 = now();
      k4_pipeline_close(pl);
      pl = nullptr;
      if (getenv("K4_TRACE")) fprintf(stderr, "[k4 trace] pipeline closed in %.2fs\n", secs(tc0, now()));
    }
  } else if (keep_sam) {
    std::vector<char> piece((size_t)std::min<uint64_t>(keep_bytes, 256ull << 20));
    for (uint64_t off = 0; off < keep_bytes; off += piece.size()) {
      const uint64_t len = std::min<uint64_t>(piece.size(), keep_bytes - off);
      CK(k4_copy_to_host(ix, piece.data(), (const char*)keep_sam + off, len));
      if (fwrite(piece.data(), 1, len, fp) != len) { fprintf(stderr, "k4align: write to %s failed\n", o.out.c_str()); return 5; }
    }
    k4_free_device(keep_sam);
  } else if (!parts.empty()) {
    // every part is coordinate sorted; merge by (chromosome, position), equal keys in batch order = load order
    struct Src { FILE* f; std::string line; long chrom, pos; };
    std::vector<Src> src(parts.size());
    auto next = [&](Src& x) -> bool {
      x.line.clear();
      char buf[1 << 16];
      while (fgets(buf, sizeof(buf), x.f)) {
        x.line += buf;
        if (!x.line.empty() && x.line.back() == '\n') break;
      }
      if (x.line.empty()) return false;
      size_t a = x.line.find('\t'), b = a == std::string::npos ? a : x.line.find('\t', a + 1);
      size_t c = b == std::string::npos ? b : x.line.find('\t', b + 1);
      if (c == std::string::npos) return false;
      auto it = order.find(x.line.substr(b + 1, c - b - 1));
      x.chrom = it == order.end() ? LONG_MAX : it->second;
      x.pos = atol(x.line.c_str() + c + 1);
      return true;
    };
    typedef std::pair<std::pair<long, long>, int> Item;
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> pq;
    for (size_t i = 0; i < parts.size(); i++) {
      src[i].f = fopen(parts[i].c_str(), "rb");
      if (!src[i].f) { fprintf(stderr, "k4align: unable to reopen %s\n", parts[i].c_str()); return 5; }
      if (next(src[i])) pq.push({{src[i].chrom, src[i].pos}, (int)i});
    }
    while (!pq.empty()) {
      const int i = pq.top().second;
      pq.pop();
      fputs(src[i].line.c_str(), fp);
      if (next(src[i])) pq.push({{src[i].chrom, src[i].pos}, i});
    }
    for (size_t i = 0; i < parts.size(); i++) { fclose(src[i].f); remove(parts[i].c_str()); }
  }
  fflush(fp);
  const bool wfail = ferror(fp) != 0;
  if (fclose(fp) != 0 || wfail) { fprintf(stderr, "k4align: write to %s failed\n", o.out.c_str()); return 5; }
  s_write += secs(tw, now());
  fprintf(stderr, "k4align: %s%llu alignments written to %s (%zu batch%s); index %.2fs, read files %.2fs, upload+parse %.2fs, align+format %.2fs, write %.2fs\n",
          multi ? ("rank " + std::to_string(o.rank) + ": ").c_str() : "", (unsigned long long)my_lines, o.out.c_str(),
          parts.empty() ? (size_t)1 : parts.size(), parts.size() > 1 ? "es" : "", std::max(secs(t0, t_open), k4_open_seconds(ix)), s_read, s_parse, s_align, s_write);
  guard.ok = true;
  if (comm) { k4_comm_barrier(comm); }
  k4_close(ix);
  if (comm) k4_comm_close(comm);
  return 0;
}

// k4align -G g0,g1,...: the parent forks one rank per GPU BEFORE anything touches HIP and only waits and merges; the ranks
// form an RCCL communicator (the id travels through a shared anonymous mapping), rank 0 reads the .sfx once and every rank
// receives sequence + suffix array over xGMI (k4_comm_open_index), each aligns its contiguous slice of the reads, the NAR
// tallies are summed with one all-reduce, and the parent merges the ranks' coordinate-sorted shards.
// The ranks' record streams + dictionaries -> the one BAM file: header by kalign's @SQ rule over the union of the ranks' hit
// flags (m_MaxRptSAMSeqsThres, KAligner.cpp:5785-5821), records merged by (refID, pos) with equal keys in rank order -- the order
// the reads were loaded in -- renumbered when not every sequence is declared, deflated and indexed by k4bam::Writer.
static int merge_rank_bams(const std::vector<std::string>& shards, const std::string& out, const Opts& o, unsigned long long* n_out) {
  std::string dataset;
  std::vector<k4bam::RefSeq> all;
  std::vector<char> hit;
  for (size_t r = 0; r < shards.size(); r++) {
    FILE* fd = fopen((shards[r] + ".sq").c_str(), "rb");
    if (!fd) { fprintf(stderr, "k4align: unable to open %s.sq\n", shards[r].c_str()); return 5; }
    char line[512];
    size_t c = 0;
    bool first = true, ok = true;
    while (ok && fgets(line, sizeof(line), fd)) {
      size_t L = strlen(line);
      if (L && line[L - 1] == '\n') line[--L] = 0;
      if (first) { if (r == 0) dataset = line; first = false; continue; }
      char* t1 = strchr(line, '\t');
      char* t2 = t1 ? strchr(t1 + 1, '\t') : nullptr;
      if (!t2) { ok = false; break; }
      *t1 = 0; *t2 = 0;
      if (r == 0) { all.push_back({line, (uint32_t)strtoul(t1 + 1, nullptr, 10)}); hit.push_back(0); }
      else if (c >= all.size() || all[c].name != line) { ok = false; break; }
      if (t2[1] == '1') hit[c] = 1;
      c++;
    }
    fclose(fd);
    if (!ok || c != all.size()) { fprintf(stderr, "k4align: %s.sq does not match the other ranks' dictionaries\n", shards[r].c_str()); return 5; }
  }
  const bool all_sq = all.size() <= (size_t)o.rpt_sq_thres;
  std::string hdr = "@HD\tVN:1.4\tSO:coordinate\n";
  std::vector<k4bam::RefSeq> refs;
  std::vector<int32_t> ref_map(all.size(), -1);
  for (size_t c = 0; c < all.size(); c++) {
    if (!(all_sq || hit[c])) continue;
    ref_map[c] = (int32_t)refs.size();
    hdr += "@SQ\tAS:" + dataset + "\tSN:" + all[c].name + "\tLN:" + std::to_string(all[c].len) + "\n";
    refs.push_back(all[c]);
  }
  hdr += "@PG\tID:k4align\tVN:1.0\n";
  k4bam::Writer bw;
  if (!bw.open(out, hdr, refs, o.bam_level, std::max(o.io_threads, 1) * (int)shards.size())) { fprintf(stderr, "k4align: %s\n", bw.error().c_str()); return 5; }
  std::string why;
  const int rc = k4merge::merge_bam_records(shards, all_sq ? nullptr : &ref_map, [&](const void* p, size_t n) { return bw.write(p, n); }, n_out, &why);
  if (rc) { fprintf(stderr, "k4align: %s\n", why.empty() ? bw.error().c_str() : why.c_str()); return rc; }
  if (!bw.close()) { fprintf(stderr, "k4align: %s\n", bw.error().c_str()); return 5; }
  if (bw.n_records() != *n_out) { fprintf(stderr, "k4align: internal error: %llu BAM records written, %llu merged\n", (unsigned long long)bw.n_records(), *n_out); return 5; }
  return 0;
}

static int run_multi_gpu(Opts& o, bool pe, int max_ml) {
  const int N = (int)o.gpus.size();
  if (N > K4_MAX_RANKS) { fprintf(stderr, "k4align: at most %d GPUs\n", K4_MAX_RANKS); return 1; }
  if (o.batch_mb > 0 || o.n_shards > 1 || o.legacy) { fprintf(stderr, "k4align: -G cannot be combined with -b, -S or -Z\n"); return 1; }
  if (o.ml_mode == 2 || o.ml_mode == 3 || o.ml_mode == 4 || o.micro_indel || o.splice_junct) {
    fprintf(stderr, "k4align: -r2 / -r3 / -r4 / -a / -A look at all reads of the run; they cannot be combined with -G\n");
    return 1;
  }
  if (o.in1.size() > K4_MAX_FILES || o.in2.size() > K4_MAX_FILES) { fprintf(stderr, "k4align: -G takes at most %d reads files per end\n", K4_MAX_FILES); return 1; }
  MultiShared* sh = (MultiShared*)mmap(nullptr, sizeof(MultiShared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (sh == MAP_FAILED) { fprintf(stderr, "k4align: unable to map shared memory\n"); return 2; }
  memset((void*)sh, 0, sizeof(*sh));
  const std::string final_out = o.out;
  const bool bam_final = final_out.size() >= 4 && strcasecmp(final_out.c_str() + final_out.size() - 4, ".bam") == 0;
  auto t0 = std::chrono::steady_clock::now();
  // fault injection for the tests of this function (K4ALIGN_FAULT=<rank>:<stage>): the named rank leaves with exit code 3 at
  // "start" (before the communicator exists) / "index" (after the broadcast of the index); with K4ALIGN_FAULT_PEERS=block the
  // other ranks then behave like ranks blocked inside a collective -- they wait for ever -- so that a machine without GPUs can
  // show that the parent ends a run one of whose ranks died
  std::vector<pid_t> kids;
  bool fork_failed = false;
  for (int r = 0; r < N; r++) {
    const pid_t pid = fork();
    if (pid < 0) { fprintf(stderr, "k4align: fork failed\n"); sh->failed = 1; fork_failed = true; break; }
    if (pid == 0) {
      o.rank = r; o.n_ranks = N; o.gpu = o.gpus[(size_t)r]; o.shared = sh;
      o.out = final_out + ".rank" + std::to_string(r);
      o.rank_bam = bam_final;
      const int rc = run_rank(o, pe, max_ml);
      if (rc != 0) sh->failed = 1;
      fflush(nullptr);
      _exit(rc);
    }
    kids.push_back(pid);
  }
  // The ranks meet in collectives: one that dies (or never started) leaves its peers waiting inside RCCL for ever.  So the
  // parent reaps whichever child ends first; at the first failure -- a non-zero exit, a signal, a fork that failed -- the
  // others are killed and reaped, the shards removed, and the run ends with that failure's code within moments.
  int worst = fork_failed ? 2 : 0;
  size_t alive = kids.size();
  auto kill_rest = [&]() {
    for (pid_t k : kids)
      if (k > 0) kill(k, SIGKILL);
  };
  if (fork_failed) kill_rest();
  while (alive) {
    int st = 0;
    const pid_t done = waitpid(-1, &st, 0);
    if (done < 0) { if (errno == EINTR) continue; break; }
    bool ours = false;
    for (pid_t& k : kids)
      if (k == done) { k = -1; ours = true; }
    if (!ours) continue;
    alive--;
    const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 9;
    if (rc && !worst) {
      worst = rc;
      fprintf(stderr, "k4align: a rank %s (%d): ending the others\n", WIFEXITED(st) ? "failed" : "was killed", WIFEXITED(st) ? rc : WTERMSIG(st));
      kill_rest();
    }
  }
  std::vector<std::string> shards;
  for (int r = 0; r < N; r++) shards.push_back(final_out + ".rank" + std::to_string(r));
  if (worst || sh->failed) {
    for (const std::string& q : shards) { remove(q.c_str()); remove((q + ".sq").c_str()); }
    munmap((void*)sh, sizeof(*sh));
    return worst ? worst : 2;
  }
  auto t1 = std::chrono::steady_clock::now();
  unsigned long long n = 0;
  if (bam_final) {
    const int rc = merge_rank_bams(shards, final_out, o, &n);
    for (const std::string& q : shards) { remove(q.c_str()); remove((q + ".sq").c_str()); }
    munmap((void*)sh, sizeof(*sh));
    if (rc) return rc;
    fprintf(stderr, "k4align: %llu alignments from %d GPUs written to %s (+ .bai); ranks %.2fs, merge + deflate %.2fs\n", n, N, final_out.c_str(),
            std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    return 0;
  }
  const int rc = k4merge::merge_sam(shards, final_out, o.rpt_sq_thres, &n, "k4align", std::max(o.io_threads, 1) * N);  // (the ranks' reader threads are idle now)
  for (const std::string& q : shards) remove(q.c_str());
  if (rc) return rc;
  fprintf(stderr, "k4align: %llu alignments from %d GPUs written to %s; ranks %.2fs, merge %.2fs\n", n, N, final_out.c_str(),
          std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
  munmap((void*)sh, sizeof(*sh));
  return 0;
}

int main(int argc, char** argv) {
  Opts o;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if (a.size() < 2 || a[0] != '-') { usage(); return 1; }
    auto val = [&]() -> std::string { return a.size() > 2 ? a.substr(2) : (i + 1 < argc ? std::string(argv[++i]) : std::string()); };
    switch (a[1]) {
      case 'i': o.in1.push_back(val()); break;
      case 'u': o.in2.push_back(val()); break;
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
      case 'r': o.ml_mode = atoi(val().c_str()); break;
      case 'R': o.max_multi = atoi(val().c_str()); break;
      case 'c': o.min_chimeric = atoi(val().c_str()); break;
      case 'a': o.micro_indel = atoi(val().c_str()); break;
      case 'A': o.splice_junct = atoi(val().c_str()); break;
      case 'x': o.min_flank_exacts = atoi(val().c_str()); break;
      case 'p': o.min_snp_reads = atoi(val().c_str()); break;
      case 'P': o.qvalue = atof(val().c_str()); break;
      case '1': o.snp_nonref_pcnt = atof(val().c_str()); break;
      case 'X': o.clamp = true; break;
      case 'N': o.best = true; break;
      case 'S': {  // "i/N": this process's slice of the reads; anything else: kalign's -S, the SNP file
        std::string v = val();
        int sa = 0, sb = 0;
        char tail = 0;
        if (sscanf(v.c_str(), "%d/%d%c", &sa, &sb, &tail) == 2) { o.shard = sa; o.n_shards = sb; }
        else if (v.empty()) { usage(); return 1; }
        else o.snp_file = v;
        break;
      }
      case 'g': o.q_method = atoi(val().c_str()); break;  // kalign's -g: FASTQ quality scoring (0 Sanger, 1 Illumina 1.3+, 2 Solexa, 3 ignore)
      case '@': o.gpu = atoi(val().c_str()); break;
      case 'G': { std::string v = val(); for (size_t q = 0; q < v.size();) { size_t e = v.find(',', q); if (e == std::string::npos) e = v.size(); o.gpus.push_back(atoi(v.substr(q, e - q).c_str())); q = e + 1; } break; }
      case 'b': o.batch_mb = atof(val().c_str()); break;
      case 'B': o.chunk_mb = std::max(1, atoi(val().c_str())); break;
      case 't': o.io_threads = std::max(1, atoi(val().c_str())); break;
      case 'M': o.fmode = atoi(val().c_str()); break;
      case 'Q': o.align_strand = atoi(val().c_str()); break;
      case '4': o.rpt_sq_thres = std::max(1, atoi(val().c_str())); break;
      case 'j': o.none_file = val(); break;
      case 'J': o.multi_file = val(); break;
      case '#': o.sample_nth = std::min(10000, std::max(1, atoi(val().c_str()))); break;
      case 'y': o.trim5 = atoi(val().c_str()); break;
      case 'Y': o.trim3 = atoi(val().c_str()); break;
      case 'Z': o.legacy = true; break;
      case 'z': o.bam_level = std::min(9, std::max(0, atoi(val().c_str()))); break;
      case 'W': o.print_slices = atoi(val().c_str()); break;
      case 'T': case 'F': (void)val(); break;  // accepted and ignored (threads, log file)
      default: usage(); return 1;
    }
  }
  if (o.print_slices > 0 && !o.in1.empty()) {  // the slice table of a -G run (host only): per end and file, where each rank starts
    if (o.print_slices > K4_MAX_RANKS || o.in1.size() > K4_MAX_FILES || o.in2.size() > K4_MAX_FILES) return 1;
    std::vector<std::vector<uint64_t>> tab[2];
    uint64_t R = 0, R2 = 0;
    std::vector<uint64_t> counts;
    std::string why;
    auto offs = std::unique_ptr<uint64_t[][K4_MAX_RANKS + 1]>(new uint64_t[K4_MAX_FILES][K4_MAX_RANKS + 1]);
    for (int e = 0; e < (o.in2.empty() ? 1 : 2); e++) {
      const std::vector<std::string>& fl = e ? o.in2 : o.in1;
      if (e && o.in2.size() != o.in1.size()) { fprintf(stderr, "k4align: as many -u files as -i files are needed\n"); return 3; }
      if (!slice_files(fl, o.print_slices, counts, e == 1, offs.get(), e ? &R2 : &R, o.io_threads, &why)) { fprintf(stderr, "k4align: %s\n", why.c_str()); return e ? 3 : 2; }
      if (e == 0) printf("records %llu\n", (unsigned long long)R);
      for (size_t f = 0; f < fl.size(); f++) {
        if (fl.size() == 1) printf("end %d", e); else printf("end %d file %d", e, (int)f);
        for (int r = 0; r <= o.print_slices; r++) printf(" %llu", (unsigned long long)offs[f][r]);
        printf("\n");
      }
    }
    return 0;
  }
  if (o.in1.empty() || o.sfx.empty() || o.out.empty()) { usage(); return 1; }
  const bool pe = !o.in2.empty();
  // multi-loci modes (KAlignerCL.cpp:686-707): 0 slough, 1 statistics only, 5 report every locus up to -R; the modes that
  // pick or cluster one locus (2 random, 3/4 AssignMultiMatches) are not part of this path
  if (o.ml_mode < 0 || o.ml_mode > 5) { fprintf(stderr, "k4align: -r%d is not supported (0..5 are)\n", o.ml_mode); return 1; }
  if (o.ml_mode != 0 && pe) { fprintf(stderr, "k4align: multiloci processing '-r%d' not supported in paired end processing\n", o.ml_mode); return 1; }
  if (o.n_shards < 1 || o.shard < 0 || o.shard >= o.n_shards) { fprintf(stderr, "k4align: -S i/N needs 0 <= i < N\n"); return 1; }
  if ((o.clamp || o.best) && o.ml_mode != 5) { fprintf(stderr, "k4align: -X / -N are supported together with -r5 only\n"); return 1; }
  int max_ml = 1;
  if (o.ml_mode != 0) {
    max_ml = o.max_multi ? o.max_multi : 5;  // cDfltMaxMultiHits
    const int lim = o.ml_mode == 5 ? 100000 : 500;  // cMaxAllHits / cMaxMultiHits
    if (max_ml < 2 || max_ml > lim) { fprintf(stderr, "k4align: -R%d outside of range 2..%d\n", max_ml, lim); return 1; }
  }
  // the optional AlignReads phases and the stages they bring with them: the same argument rules as kalign (KAlignerCL.cpp:545-569,
  // 667-761,823-830)
  o.min_chimeric = abs(o.min_chimeric);
  if (o.min_chimeric != 0 && (o.min_chimeric < 15 || o.min_chimeric > 99)) { fprintf(stderr, "k4align: minimum chimeric length percentage '-c%d' specified outside of range 15..99\n", o.min_chimeric); return 1; }
  if (o.micro_indel < 0 || o.micro_indel > 20) { fprintf(stderr, "k4align: microInDel length maximum '-a%d' specified outside of range 0..20\n", o.micro_indel); return 1; }
  if (o.splice_junct != 0 && (o.splice_junct < 25 || o.splice_junct > 100000)) { fprintf(stderr, "k4align: RNAseq maximum splice junction separation '-A%d' must be either 0 or in the range 25..100000\n", o.splice_junct); return 1; }
  if (o.q_method < 0 || o.q_method > 3) { fprintf(stderr, "k4align: fastq quality '-g%d' specified outside of range 0..3\n", o.q_method); return 1; }
  if (o.min_flank_exacts < 0 || o.min_flank_exacts > 7) { fprintf(stderr, "k4align: max flank trimming '-x%d' specified outside of range 0..7\n", o.min_flank_exacts); return 1; }
  if (pe && (o.micro_indel || o.splice_junct)) { fprintf(stderr, "k4align: microInDel '-a' / splice junction '-A' processing not supported in paired end processing\n"); return 1; }
  if (o.ml_mode == 5 && (o.micro_indel || o.splice_junct)) { fprintf(stderr, "k4align: microInDels / splice junctions not supported when reporting multiloci alignments '-r5'\n"); return 1; }
  if (o.min_chimeric && (o.best || o.ml_mode == 3 || o.ml_mode == 4)) { fprintf(stderr, "k4align: chimeric read processing cannot be combined with -N / -r3 / -r4\n"); return 1; }
  // output format (KAlignerCL.cpp:217,514-519): -M0 the accepted alignments, -M1 every loaded read; -M2 (BED) and -M3 (packed base
  // alleles) are not built
  if (o.fmode < 0 || o.fmode > 3) { fprintf(stderr, "k4align: output format mode '-M%d' specified outside of range 0..3\n", o.fmode); return 1; }
  if (o.fmode >= 2) { fprintf(stderr, "k4align: output format -M%d (BED / packed base alleles) is not built\n", o.fmode); return 3; }
  if (o.fmode == 1) {
    if (o.batch_mb > 0 || o.n_shards > 1 || o.legacy || o.ml_mode == 5) {
      fprintf(stderr, "k4align: -M1 is written by the pipelined modes (not with -b, -S i/N, -Z, -r5)\n");
      return 3;
    }
    if (o.min_snp_reads > 0 || !o.snp_file.empty()) { fprintf(stderr, "k4align: SNP calling is not available in '-M1' output mode\n"); return 1; }  // KAlignerCL.cpp:935
  }
  // SNP calling (KAlignerCL.cpp:877-945): -S alone means -p20; -p alone writes <out>.snp; -P defaults to 0.05
  if (!o.snp_file.empty() && o.min_snp_reads == 0) o.min_snp_reads = 20;
  if (o.min_snp_reads != 0 && (o.min_snp_reads < 1 || o.min_snp_reads > 100)) { fprintf(stderr, "k4align: minimum read coverage at any loci '-p%d' must be in range 1..100\n", o.min_snp_reads); return 1; }
  if (o.min_snp_reads > 0) {
    if (o.snp_file.empty()) o.snp_file = o.out + ".snp";
    if (o.qvalue < 0.0 || o.qvalue > 0.40) { fprintf(stderr, "k4align: QValue '-P%1.5f' for controlling SNP FDR (Benjamini-Hochberg) must be in range 0.0 to 0.4\n", o.qvalue); return 1; }
    if (o.qvalue == 0.0) o.qvalue = 0.05;
    if (o.snp_nonref_pcnt < 0.1 || o.snp_nonref_pcnt > 35.0) { fprintf(stderr, "k4align: SNP minimum non-ref '-1%f' must be in range 0.1 to 35.0\n", o.snp_nonref_pcnt); return 1; }
    if (o.ml_mode == 5) { fprintf(stderr, "k4align: SNP processing is not supported when reporting all multiloci alignments '-r5'\n"); return 1; }
    if (o.batch_mb > 0 || o.n_shards > 1 || !o.gpus.empty() || o.legacy) { fprintf(stderr, "k4align: SNP calling runs over the whole run's alignments: not with -b, -S i/N, -G, -Z\n"); return 3; }
  }
  if (o.splice_junct > 0 && o.min_chimeric == 0 && o.min_flank_exacts == 0) o.min_flank_exacts = o.max_subs;  // "force flank trim", :829-830
  if (o.min_flank_exacts > 7) o.min_flank_exacts = 7;
  if ((!o.none_file.empty() || !o.multi_file.empty()) && (o.batch_mb > 0 || o.n_shards > 1 || !o.gpus.empty() || o.legacy)) {
    fprintf(stderr, "k4align: -j / -J are written by the pipelined single-GPU mode (not with -b, -S i/N, -G, -Z)\n");
    return 3;
  }
  if (o.sample_nth > 1 && (o.in1.size() > 1 || o.batch_mb > 0 || o.n_shards > 1 || !o.gpus.empty() || o.legacy)) {
    fprintf(stderr, "k4align: -#%d samples the reads of ONE input file per end in the pipelined single-GPU mode (not with several -i files, -b, -S i/N, -G, -Z)\n", o.sample_nth);
    return 3;
  }
  if (o.trim5 < 0 || o.trim5 > 50) { fprintf(stderr, "k4align: Trim 5' raw reads '-y%d' specified outside of range 0..50\n", o.trim5); return 1; }
  if (o.trim3 < 0 || o.trim3 > 50) { fprintf(stderr, "k4align: Trim 3' raw reads '-Y%d' specified outside of range 0..50\n", o.trim3); return 1; }
  if (o.align_strand < 0 || o.align_strand > 2) { fprintf(stderr, "k4align: Aligned to strand '-Q%d' specified outside of range 0..2\n", o.align_strand); return 1; }
  // kalign's argument rules for the alignment proper (KAlignerCL.cpp:546-553,645-657,789-821)
  if (o.min_len < 15 || o.min_len > 2000) { fprintf(stderr, "k4align: minimum accepted read length '-l%d' specified outside of range 15..2000\n", o.min_len); return 1; }  // cMinSeqLen..cMaxSeqLen, :776-786
  if (o.max_len < o.min_len || o.max_len > 2000) { fprintf(stderr, "k4align: maximum accepted read length '-L%d' specified outside of range %d..2000\n", o.max_len, o.min_len); return 1; }
  if (o.pmode < 0 || o.pmode > 3) { fprintf(stderr, "k4align: Processing mode '-m%d' specified outside of range 0..3\n", o.pmode); return 1; }
  if (o.min_edit < 1 || o.min_edit > 2) { fprintf(stderr, "k4align: Minimum edit distance '-e%d' specified outside of range 1..2\n", o.min_edit); return 1; }
  if (o.max_subs < 0 || o.max_subs > 15) { fprintf(stderr, "k4align: Max allowed substitutions per 100bp read length '-s%d' specified outside of range 0..15\n", o.max_subs); return 1; }
  if (o.max_ns < 0 || o.max_ns > 5) { fprintf(stderr, "k4align: Allowed number of indeterminate bases in reads '-n%d' specified outside of range 0..5\n", o.max_ns); return 1; }
  if (pe && o.pe_mode == 0) {  // "-u" without "-U": unique alignments only
    o.pe_mode = 2;
    fprintf(stderr, "k4align: PE2 file(s) with '-u<files>' specified, defaulting PE processing mode to unique alignments only '-U2'\n");
  }
  if (!pe) o.pe_mode = 0;
  if (pe) {
    if (o.pe_mode < 1 || o.pe_mode > 4) { fprintf(stderr, "k4align: PE processing mode '-U%d' specified outside of range 0..4\n", o.pe_mode); return 1; }
    if (o.pair_min < 25 || o.pair_min > 100000) { fprintf(stderr, "k4align: paired end apparent min length '-d%d' must be in range 25..100000\n", o.pair_min); return 1; }
    if (o.pair_max < 0) o.pair_max = std::max(1000, o.pair_min);  // max(cDfltPairMaxLen, PairMinLen)
    if (o.pair_max < std::max(1, o.pair_min) || o.pair_max > 100000) { fprintf(stderr, "k4align: paired end apparent max length '-D%d' must be in range %d..100000\n", o.pair_max, std::max(1, o.pair_min)); return 1; }
  } else if (o.pair_max < 0) o.pair_max = 1000;

  if (!o.gpus.empty()) return run_multi_gpu(o, pe, max_ml);
  return run_rank(o, pe, max_ml);
}
