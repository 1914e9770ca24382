// kit4b_amd/csrc/k4index_main.cpp -- `k4index`: the stand-alone counterpart of `ngskit4b index` (standard mode) with the
// suffix sort on the GPU.  Host C++ over the C ABI of libk4sfx.so only.  Mirrors:
//   sequences   ProcessFastaFile   ngskit4b/kit4bax.cpp:478-627 over CFasta::ReadSequence (libkit4b/Fasta.cpp:1040-1210):
//               entry name = first token of the descriptor, letters a/c/g/t/u -> bases (case = soft masking, dropped),
//               '-' -> InDel, other letters -> N, anything else sloughed; sequences under -l are not indexed; deep inside
//               runs of N every 13th N becomes rand() % 4 (kit4bax.cpp:556-580: so that the sort can sort)
//   entries     CSfxArray::AddEntry  libkit4b/SfxArray.cpp:1628-1753 (one EOS after every sequence, GenHash16 of the name)
//   sort+write  CSfxArray::Finalise  :1758 -> k4_build_sa_device + k4_open_device + k4_write_sfx
// Options follow `ngskit4b index`: -i <fasta[.gz]> (repeatable) -o <out.sfx> -r <ref species> [-d descr] [-t title]
// [-l minseqlen=50] (plus -g <gpu>).
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

bool slurp(const std::string& path, std::vector<uint8_t>& buf) {
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) return false;
  gzbuffer(f, 1 << 20);
  size_t used = 0;
  buf.resize(64 << 20);
  for (;;) {
    if (buf.size() - used < (16u << 20)) buf.resize(buf.size() * 2);
    const int got = gzread(f, buf.data() + used, (unsigned)std::min<size_t>(buf.size() - used, 1u << 30));
    if (got < 0) { gzclose(f); return false; }
    if (got == 0) break;
    used += (size_t)got;
  }
  gzclose(f);
  buf.resize(used);
  return true;
}

uint16_t gen_hash16(const char* name) {  // CUtility::GenHash16, libkit4b/Utility.cpp:402-420
  if (!name || !name[0]) return 0;
  int h = 19937;
  for (char c; (c = *name++);) {
    h = (h ^ (int)tolower((unsigned char)c)) * 3119;
    h ^= (h >> 13);
    h &= 0x0ffff;
  }
  return (uint16_t)(h ? h : 19937);
}

struct Builder {
  std::vector<uint8_t> seq;        // concatenation: bases, one EOS (7) after every entry
  std::vector<k4_entry> entries;
  uint64_t n_under = 0;
  int min_len = 50;

  void add(const std::string& name, std::vector<uint8_t>& s) {
    if ((int64_t)s.size() < (int64_t)min_len) { n_under++; return; }
    // kit4bax.cpp:556-580: after 25 Ns, while at least 5 more follow, every 13th N of the run is replaced by a random base
    int run = 0;
    const size_t len = s.size();
    for (size_t i = 0; i < len; i++) {
      if (s[i] == 4 && i + 5 < len) {
        if (++run > 25 && s[i + 1] == 4 && s[i + 2] == 4 && s[i + 3] == 4 && s[i + 4] == 4) {
          if (!(run % 13)) s[i] = (uint8_t)(rand() % 4);
        }
      } else
        run = 0;
    }
    k4_entry e;
    memset(&e, 0, sizeof(e));
    e.entry_id = (uint32_t)entries.size() + 1;
    e.fblock_id = 1;
    strncpy(e.name, name.c_str(), 80);
    e.name_hash = gen_hash16(e.name);
    e.seq_len = (uint32_t)len;
    e.start_ofs = seq.size();
    e.end_ofs = seq.size() + len - 1;
    entries.push_back(e);
    seq.insert(seq.end(), s.begin(), s.end());
    seq.push_back(7);
  }
};

// CFasta::ReadSequence as ProcessFastaFile drives it: '>' opens a descriptor (to the end of the line), everything else is
// sequence; text before the first descriptor is an entry named <file>.<n>
bool add_file(const std::string& path, Builder& b) {
  std::vector<uint8_t> t;
  if (!slurp(path, t)) return false;
  std::vector<uint8_t> cur;
  std::string name, descr;
  bool have = false, in_descr = false;
  int seq_id = 0;
  auto flush = [&]() {
    if (have && !cur.empty()) b.add(name, cur);
    else if (have) b.n_under++;  // (a descriptor with no sequence: under any minimum length)
    cur.clear();
  };
  // 256-entry translation of a sequence byte: 0..6 = code, 0xFF = sloughed
  static uint8_t lut[256];
  static bool lut_ready = false;
  if (!lut_ready) {
    for (int c = 0; c < 256; c++) lut[c] = (isalpha(c) || c == '-') ? 4 : 0xFF;
    lut['a'] = lut['A'] = 0; lut['c'] = lut['C'] = 1; lut['g'] = lut['G'] = 2;
    lut['t'] = lut['T'] = lut['u'] = lut['U'] = 3; lut['-'] = 6;
    lut_ready = true;
  }
  auto end_descr = [&]() {
    in_descr = false;
    size_t p = 0;
    while (p < descr.size() && isspace((unsigned char)descr[p])) p++;
    size_t e = p;
    while (e < descr.size() && !isspace((unsigned char)descr[e])) e++;
    seq_id++;
    name = e > p ? descr.substr(p, e - p) : path + "." + std::to_string(++seq_id);  // kit4bax.cpp:540-541
    have = true;
  };
  const uint8_t* base = t.data();
  const size_t T = t.size();
  size_t i = 0;
  while (i < T) {
    if (in_descr) {  // to the end of the line
      const uint8_t c = base[i++];
      if (c == '\n' || c == '\r') end_descr();
      else if (!(descr.empty() && (c == ' ' || c == '\t'))) descr.push_back((char)c);
      continue;
    }
    // a stretch of sequence text: up to the next '>' (wherever it stands, as CFasta takes it) or the end
    const uint8_t* gt = (const uint8_t*)memchr(base + i, '>', T - i);
    const size_t stop = gt ? (size_t)(gt - base) : T;
    if (stop > i) {
      const size_t old = cur.size();
      cur.resize(old + (stop - i));
      uint8_t* dst = cur.data() + old;
      size_t k = 0;
      for (size_t q = i; q < stop; q++) {
        const uint8_t v = lut[base[q]];
        dst[k] = v;
        k += v != 0xFF;
      }
      cur.resize(old + k);
      if (k && !have) {  // sequence without a descriptor, kit4bax.cpp:549-556
        seq_id++;
        name = path + "." + std::to_string(seq_id);
        have = true;
      }
      i = stop;
    }
    if (gt) {
      flush();
      have = false;
      in_descr = true;
      descr.clear();
      i++;
    }
  }
  if (in_descr) end_descr();
  flush();
  return true;
}

}  // namespace

int main(int argc, char** argv) {
  std::vector<std::string> in;
  std::string out, ref, descr, title;
  int gpu = 0;
  Builder b;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if (a.size() < 2 || a[0] != '-') { fprintf(stderr, "k4index -i genome.fa[.gz] [-i ...] -o out.sfx -r refname [-d descr] [-t title] [-l minseqlen=50] [-g gpu=0]\n"); return 1; }
    auto val = [&]() -> std::string { return a.size() > 2 ? a.substr(2) : (i + 1 < argc ? std::string(argv[++i]) : std::string()); };
    switch (a[1]) {
      case 'i': in.push_back(val()); break;
      case 'o': out = val(); break;
      case 'r': ref = val(); break;
      case 'd': descr = val(); break;
      case 't': title = val(); break;
      case 'l': b.min_len = std::max(1, atoi(val().c_str())); break;
      case 'g': gpu = atoi(val().c_str()); break;
      case 'T': case 'F': case 'm': case 'k': (void)val(); break;  // accepted and ignored (threads, log, mode 0, maxkmers)
      default: fprintf(stderr, "k4index: unknown option %s\n", a.c_str()); return 1;
    }
  }
  if (in.empty() || out.empty() || ref.empty()) { fprintf(stderr, "k4index: -i, -o and -r are required\n"); return 1; }
  if (descr.empty()) descr = ref;  // kit4bax.cpp:350-363
  if (title.empty()) title = ref;
  auto t0 = std::chrono::steady_clock::now();
  for (const std::string& f : in)
    if (!add_file(f, b)) { fprintf(stderr, "k4index: unable to read %s\n", f.c_str()); return 2; }
  if (b.entries.empty()) { fprintf(stderr, "k4index: no sequence of at least %d bp\n", b.min_len); return 2; }
  const uint64_t n = b.seq.size();
  const uint32_t el = n >= 4000000000ull ? 5 : 4;  // cThres8ByteSfxEls, libkit4b/SfxArray.h:184
  if (b.n_under) fprintf(stderr, "k4index: %llu sequences not accepted for indexing as length under %dbp\n", (unsigned long long)b.n_under, b.min_len);
  auto t1 = std::chrono::steady_clock::now();
  // the C ABI allocates through an index handle; a first, tiny one serves that purpose on the chosen device
  void *d_seq = nullptr, *d_sa = nullptr;
  k4_index* ix = nullptr;
  {
    const uint8_t tiny_seq[5] = {0, 1, 2, 3, 7};
    const uint8_t tiny_sa[20] = {0, 0, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 4, 0, 0, 0};
    k4_entry te;
    memset(&te, 0, sizeof(te));
    te.entry_id = 1; te.fblock_id = 1; strcpy(te.name, "t"); te.seq_len = 4; te.start_ofs = 0; te.end_ofs = 3;
    k4_index* tmp = nullptr;
    int rc = k4_open_host(5, 4, tiny_seq, tiny_sa, 1, &te, "t", gpu, 4, &tmp);
    if (rc != K4_OK) { fprintf(stderr, "k4index: no usable GPU: %s (%d)\n", k4_global_error(), rc); return 3; }
    rc = k4_alloc_device(tmp, n + 64, &d_seq);
    if (rc == K4_OK) rc = k4_alloc_device(tmp, n * el + 64, &d_sa);
    if (rc == K4_OK) rc = k4_copy_to_device(tmp, d_seq, b.seq.data(), n);
    if (rc != K4_OK) { fprintf(stderr, "k4index: %s (%d)\n", k4_last_error(tmp), rc); return 3; }
    k4_close(tmp);
  }
  int rc = k4_build_sa_device(n, el, d_seq, d_sa, gpu);
  if (rc != K4_OK) { fprintf(stderr, "k4index: suffix sort failed: %s (%d)\n", k4_global_error(), rc); return 4; }
  auto t2 = std::chrono::steady_clock::now();
  rc = k4_open_device(n, el, d_seq, d_sa, 1, (uint32_t)b.entries.size(), b.entries.data(), ref.c_str(), gpu, 0, &ix);
  if (rc != K4_OK) { fprintf(stderr, "k4index: %s (%d)\n", k4_global_error(), rc); return 4; }
  k4_set_description(ix, descr.c_str(), title.c_str());
  rc = k4_write_sfx(ix, out.c_str());
  if (rc != K4_OK) { fprintf(stderr, "k4index: %s (%d)\n", k4_last_error(ix), rc); return 5; }
  auto t3 = std::chrono::steady_clock::now();
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double>(c - a).count(); };
  fprintf(stderr, "k4index: %zu sequences, %llu bp (%u-byte suffix elements) -> %s; read %.2fs, upload+sort %.2fs, pack+write %.2fs\n",
          b.entries.size(), (unsigned long long)(n - b.entries.size()), el, out.c_str(), secs(t0, t1), secs(t1, t2), secs(t2, t3));
  k4_close(ix);
  k4_free_device(d_seq);
  return 0;
}
