// kit4b_amd/csrc/k4_snp.hip -- kalign's SNP calling (main CSV) over device-resident alignments.
//
//   CKAligner::ProcessSNPs  ngskit4b/KAligner.cpp:8168-8590   per-locus base counts over the accepted alignments of a chromosome
//   CKAligner::OutputSNPs   ngskit4b/KAligner.cpp:7098-7760   background rate in a 51-base window, binomial p-value, Benjamini-
//                                                             Hochberg cut, one "SNP_ID",... CSV line per call
//   CStats::Binomial / ProbKeqlk / Calc_nCk   libkit4b/Stats.cpp:489-564
//
// Device: the pile-up (one wave per alignment, lanes over its bases, atomic adds into seven per-locus arrays of the chromosome),
// two prefix sums and one kernel that evaluates the window sums and the integer / proportion tests at every locus and appends the
// few loci that pass.  Host (this file, plain C++): error rates, p-values, ranks and text for those loci -- a few hundred per
// chromosome -- with the reference's own floating-point operations in its order (long double n-choose-k included).
// The files kalign writes beside it: the coverage WIG (host threads, one per chromosome) and the DiSNP / TriSNP haplotype files
// (:7767-8101; one thread per alignment finds the called loci it covers and counts its base combination for every run of two /
// three of them).  Not built: marker sequences, centroids, the BED form, packed base alleles.  Equal p-values keep locus order in the ranking (the reference's multi-threaded quicksort
// leaves them in no defined order).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <future>
#include <memory>
#include <string>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "k4_device.h"
#include "k4_pool.h"

namespace {

struct SnpArgs {
  K4DevIndex ix;
  int pe;
  int64_t n_reads;
  const k4_read_result* rr;
  const k4_hit* hits;
  int max_ml;
  const k4_pe_read* pr;
  const uint8_t* reads;
  const uint64_t* offs;
  const uint32_t* lens;
  uint32_t chrom_id;
  uint64_t cs;       // concat offset of the chromosome
  uint32_t clen;
  uint32_t* cnt;     // seven arrays of clen + 16: ref, nonref, A, C, G, T, N
  unsigned long long* tot;  // [0] TotMatch [1] TotMismatch [2] reads piled up [3] their aligned bases
};
#define K4_SNP_STRIDE(a) ((size_t)(a).clen + 16)

// ProcessSNPs' inner loop (:8468-8557, base space), one wave per accepted alignment on this chromosome
__global__ void __launch_bounds__(256) k4k_snp_pileup(SnpArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * 256) >> 6;
  unsigned long long m = 0, mm = 0, nr = 0, nb = 0;
  for (int64_t i = wave0; i < a.n_reads; i += n_waves) {
    const int nar = a.pe ? a.pr[i].nar : a.rr[i].nar;
    if (nar != K4_NAR_ACCEPTED) continue;
    const k4_hit h = a.pe ? a.pr[i].hit : a.hits[i * a.max_ml];
    if (h.chrom_id != a.chrom_id || (h.ext & (K4_EXT_INDEL | K4_EXT_SPLICE))) continue;
    const uint32_t tl = K4_HIT_TRIM_LEFT(h), tr = K4_HIT_TRIM_RIGHT(h);
    uint32_t match_len = (uint32_t)h.match_len - tl - tr;                 // AdjHitLen
    const uint32_t loci0 = h.match_loci + (h.strand == '+' ? tl : tr);   // AdjStartLoci
    if ((uint64_t)loci0 + match_len > a.clen) continue;                  // (GetSeq comes back short: the read is skipped, :8420)
    const uint8_t* src = a.reads + a.offs[i] + tl;
    if (lane == 0) { nr++; nb += match_len; }
    for (uint32_t q = lane; q < match_len; q += 64) {
      const uint32_t ref = k4d_ref_base(a.ix, a.cs + loci0 + q);
      uint32_t r = h.strand == '+' ? (src[q] & 7u) : (src[match_len - 1 - q] & 7u);
      if (h.strand != '+' && r <= 3) r = 3 - r;
      if (ref >= 4 || r > 4) continue;
      uint32_t* c = a.cnt + loci0 + q;
      if (ref == r) { atomicAdd(c, 1u); m++; }
      else {
        atomicAdd(c + K4_SNP_STRIDE(a), 1u);
        atomicAdd(c + (2 + r) * K4_SNP_STRIDE(a), 1u);
        mm++;
      }
    }
  }
  for (int d = 32; d > 0; d >>= 1) { m += __shfl_down(m, d, 64); mm += __shfl_down(mm, d, 64); }
  if (lane == 0) {
    if (m) atomicAdd(&a.tot[0], m);
    if (mm) atomicAdd(&a.tot[1], mm);
    if (nr) { atomicAdd(&a.tot[2], nr); atomicAdd(&a.tot[3], nb); }
  }
}

// which sequences hold an alignment the pile-up would take at all: one pass over the reads before the per-sequence work, so that
// an assembly of 10^5 contigs costs its hit contigs, not its contigs (the reference walks its sorted reads once)
__global__ void __launch_bounds__(256) k4k_snp_mark(SnpArgs a, uint8_t* __restrict__ flags, uint32_t n_entries) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_reads; i += stride) {
    const int nar = a.pe ? a.pr[i].nar : a.rr[i].nar;
    if (nar != K4_NAR_ACCEPTED) continue;
    const k4_hit h = a.pe ? a.pr[i].hit : a.hits[i * a.max_ml];
    if ((h.ext & (K4_EXT_INDEL | K4_EXT_SPLICE)) || h.chrom_id < 1 || h.chrom_id > n_entries) continue;
    flags[h.chrom_id] = 1;
  }
}

struct Cand { uint32_t loci, n_ref, n_nonref, by_base[5], loc_mm, loc_m, ref_base, pad; };  // 48 bytes

// OutputSNPs' per-locus tests that need no error rate (:7375-7438): coverage, non-reference count and proportion; the window
// [l - 25, l + 26) clamped into the chromosome as the reference's sliding sums have it
__global__ void __launch_bounds__(256) k4k_snp_candidates(SnpArgs a, const uint64_t* __restrict__ p_ref, const uint64_t* __restrict__ p_non,
                                                          int min_snp_reads, double nonref_frac, Cand* __restrict__ out, uint32_t cap,
                                                          uint32_t* __restrict__ n_out) {
  const uint32_t l = blockIdx.x * 256u + threadIdx.x;
  if (l >= a.clen) return;
  const size_t S = K4_SNP_STRIDE(a);
  const uint32_t n_ref = a.cnt[l], n_non = a.cnt[S + l];
  const int tot = (int)(n_ref + n_non);
  if (tot < min_snp_reads || n_non < 1) return;
  if ((double)n_non / tot < nonref_frac) return;
  const uint32_t flank = 25, win = 51;
  uint32_t lo = 0, hi = min(win, a.clen);
  if (a.clen > win) {  // slid once for every locus in (flank, clen - flank): it stays where the first / last of them left it
    lo = l <= flank ? 0u : (l + flank < a.clen ? l - flank : a.clen - win);
    hi = lo + win;
  }
  const uint32_t loc_mm = (uint32_t)(p_non[hi] - p_non[lo]), loc_m = (uint32_t)(p_ref[hi] - p_ref[lo]);
  const uint32_t slot = atomicAdd(n_out, 1u);
  if (slot >= cap) return;
  Cand c;
  c.loci = l; c.n_ref = n_ref; c.n_nonref = n_non;
  for (int b = 0; b < 5; b++) c.by_base[b] = a.cnt[(2 + b) * S + l];
  c.loc_mm = loc_mm; c.loc_m = loc_m;
  c.ref_base = k4d_ref_base(a.ix, a.cs + l);  // pSNP->RefBase: the target symbol (a covered locus: the reference has set it)
  c.pad = 0;
  out[slot] = c;
}

// coverage per locus (TotBases) for the WIG: ref + nonref.  First its maximum, then the array in the narrowest of 1 / 2 / 4 bytes
// per locus that holds it (whole-genome coverage fits a byte: a quarter of the bytes to bring down and to walk through)
__global__ void __launch_bounds__(256) k4k_snp_coverage_max(const uint32_t* __restrict__ ref, const uint32_t* __restrict__ non, uint32_t n,
                                                            uint32_t* __restrict__ mx) {
  uint32_t m = 0;
  for (uint32_t l = blockIdx.x * 256u + threadIdx.x; l < n; l += gridDim.x * 256u) m = max(m, ref[l] + non[l]);
  for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_down(m, d, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(mx, m);
}
template <typename T>
__global__ void __launch_bounds__(256) k4k_snp_coverage(const uint32_t* __restrict__ ref, const uint32_t* __restrict__ non, uint32_t n,
                                                        T* __restrict__ cov) {
  const uint32_t l = blockIdx.x * 256u + threadIdx.x;
  if (l < n) cov[l] = (T)(ref[l] + non[l]);
}

// DiSNPs / TriSNPs (OutputSNPs :7767-8101 with IterateReadsOverlapping :10475-10546 and AdjAlignSNPBase :1581-1632): for two / three
// called loci following each other closely, the alignments that cover all of them and the bases they show there.  The reference
// walks the sorted alignments once per locus pair; here every alignment looks up the called loci inside its span (they are sorted)
// and adds itself to the pair ending at each of them (slot from the host: -1 = loci too far apart) -- 16 combination counters + the
// antisense count per pair, 64 + 1 per triple.
struct HapArgs {
  const uint32_t* loci;     // called SNP loci of the chromosome, ascending
  const int32_t* di_slot;   // per locus k: slot of the pair (k-1, k), or -1
  const int32_t* tri_slot;  // per locus k: slot of the triple (k-2, k-1, k), or -1
  uint32_t n_loci;
  uint32_t* di;             // [n_di][17]
  uint32_t* tri;            // [n_tri][65]
};
__global__ void __launch_bounds__(256) k4k_snp_haplotypes(SnpArgs a, HapArgs hp) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_reads; i += (int64_t)gridDim.x * 256) {
    const int nar = a.pe ? a.pr[i].nar : a.rr[i].nar;
    if (nar != K4_NAR_ACCEPTED) continue;
    const k4_hit h = a.pe ? a.pr[i].hit : a.hits[i * a.max_ml];
    if (h.chrom_id != a.chrom_id || (h.ext & (K4_EXT_INDEL | K4_EXT_SPLICE))) continue;
    const uint32_t tl = K4_HIT_TRIM_LEFT(h), tr = K4_HIT_TRIM_RIGHT(h);
    const uint32_t match_len = (uint32_t)h.match_len - tl - tr;
    const uint32_t start = h.match_loci + (h.strand == '+' ? tl : tr);  // AdjStartLoci .. AdjEndLoci
    if (match_len == 0 || (uint64_t)start + match_len > a.clen) continue;
    const uint32_t end = start + match_len - 1;
    uint32_t lo = 0, hi = hp.n_loci;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (hp.loci[mid] < start) lo = mid + 1; else hi = mid;
    }
    const uint8_t* bases = a.reads + a.offs[i];
    const bool anti = h.strand != '+';
    uint32_t b1 = 7, b2 = 7;  // the read's bases at the two loci before this one
    for (uint32_t k = lo; k < hp.n_loci; k++) {
      const uint32_t l = hp.loci[k];
      if (l > end) break;
      uint32_t b = anti ? (bases[h.match_loci + (uint32_t)h.match_len - l - 1] & 7u) : (bases[l - h.match_loci] & 7u);
      if (anti && b <= 3) b = 3 - b;
      if (k >= lo + 1 && b <= 3 && b1 <= 3) {
        const int32_t sd = hp.di_slot[k];
        if (sd >= 0) {
          atomicAdd(&hp.di[(size_t)sd * 17 + ((b1 << 2) | b)], 1u);
          if (anti) atomicAdd(&hp.di[(size_t)sd * 17 + 16], 1u);
        }
        if (k >= lo + 2 && b2 <= 3) {
          const int32_t st = hp.tri_slot[k];
          if (st >= 0) {
            atomicAdd(&hp.tri[(size_t)st * 65 + ((b2 << 4) | (b1 << 2) | b)], 1u);
            if (anti) atomicAdd(&hp.tri[(size_t)st * 65 + 64], 1u);
          }
        }
      }
      b2 = b1; b1 = b;
    }
  }
}

// one line of the .disnp.csv / .trisnp.csv file (:7836-7916, :7997-8089) from a slot's counters; false = not reported (too few
// reads or haplotypes).  A combination counts as a haplotype from max(5, a tenth of the covering reads) reads on.
static bool hap_line(std::string& out, const char* type, int id, const std::string& species, const char* chrom, const uint32_t* loci,
                     const uint32_t* ref, int n, const uint32_t* c, int min_snp_reads) {
  const int nc = n == 2 ? 16 : 64;
  int by_base[3][4] = {{0}}, depth = 0, cnts[64];
  for (int q = 0; q < nc; q++) {
    cnts[q] = (int)c[q];
    depth += cnts[q];
    for (int k = 0; k < n; k++) by_base[k][(q >> (2 * (n - 1 - k))) & 3] += cnts[q];
  }
  if (depth < min_snp_reads) return false;
  const int thres = std::max(5, (depth + 5) / 10);
  int n_hap = 0;
  for (int q = 0; q < nc; q++) { if (cnts[q] >= thres) n_hap++; else cnts[q] = 0; }
  if (n_hap < n) return false;
  char line[1200];
  int m = snprintf(line, sizeof(line), "%d,\"%s\",\"%s\",\"%s\",", id, type, species.c_str(), chrom);
  for (int k = 0; k < n; k++)
    m += snprintf(line + m, sizeof(line) - m, "%d,\"%c\",%d,%d,%d,%d,0,", (int)loci[k], "acgtn"[ref[k] > 4 ? 4 : ref[k]], by_base[k][0], by_base[k][1],
                  by_base[k][2], by_base[k][3]);
  m += snprintf(line + m, sizeof(line) - m, "%d,%d,%d", depth, (int)c[nc], n_hap);
  for (int q = 0; q < nc; q++) m += snprintf(line + m, sizeof(line) - m, ",%d", cnts[q]);
  line[m++] = '\n';
  out.append(line, (size_t)m);
  return true;
}
static std::string hap_header(int n) {  // :8252-8330
  std::string s = n == 2 ? "\"DiSNPs_ID\"" : "\"TriSNPs_ID\"";
  s += ",\"ElType\",\"Species\",\"Chrom\"";
  for (int k = 1; k <= n; k++) {
    const std::string p = "\"SNP" + std::to_string(k);
    s += "," + p + "Loci\"," + p + "RefBase\"," + p + "BaseAcnt\"," + p + "BaseCcnt\"," + p + "BaseGcnt\"," + p + "BaseTcnt\"," + p + "BaseNcnt\"";
  }
  s += ",\"Depth\",\"Antisense\",\"Haplotypes\"";
  for (int q = 0; q < (n == 2 ? 16 : 64); q++) {
    s += ",\"";
    for (int j = n - 1; j >= 0; j--) s += "acgt"[(q >> (2 * j)) & 3];
    s += "\"";
  }
  return s + "\n";
}

// The coverage WIG kalign writes beside the SNP file: variableStep spans of roughly equal coverage (AccumWIGCnts / CompleteWIGSpan,
// KAligner.cpp:6993-7085), fed with the locus counted from 0 (:7375) -- so a span that starts at locus 0 is never written.  The
// walk is sequential by nature (where a span ends depends on its running mean): one host thread per chromosome, running while the
// device piles up the next chromosomes.  Returns the text of the chromosome's closed spans; `tail` = what closing the last open
// span adds (the reference does that only for chromosomes with at least one candidate locus, :7582-7608 / :8135).
struct WigOut { std::string body, tail; };
// The running mean cnts / len is kept as quotient and remainder (a span grows by one locus at a time), so the walk has no division:
// 100 * (cnts / len) against 100 c, 75 c and 125 c is q against c, 4 q against 3 c and 5 c.
template <typename T>
static WigOut wig_walk(const T* cov, uint32_t clen, const std::string& name) {
  WigOut o;
  uint32_t loci = 0, len = 0, rptd_len = 0;
  bool started = false, rptd = false;  // m_WIGChromID != 0, m_WIGRptdChromID == this chromosome
  uint64_t cnts = 0, q = 0;            // q = cnts / len
  int64_t r = 0;                       // cnts - q * len
  // (a genome at low coverage makes a span of nearly every run of equal coverage -- hundreds of millions of lines: own digits, no printf)
  const std::string head = "variableStep chrom=" + name + " span=";
  char line[64];
  auto put = [](char* p, uint32_t v) {
    char t[10];
    int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = t[--n];
    return p;
  };
  auto complete = [&](std::string& dst) {
    if (started && len > 0 && loci > 0 && cnts > 0) {
      if (!rptd || len != rptd_len) {
        dst += head;
        char* p = put(line, len);
        *p++ = '\n';
        dst.append(line, (size_t)(p - line));
        rptd = true; rptd_len = len;
      }
      char* p = put(line, loci);
      *p++ = ' ';
      p = put(p, (uint32_t)(q + (r > 0 ? 1 : 0)));  // (cnts + len - 1) / len
      *p++ = '\n';
      dst.append(line, (size_t)(p - line));
    }
    loci = 0; len = 0; cnts = 0;
  };
  for (uint32_t l = 0; l < clen; l++) {
    const uint64_t c = cov[l];
    if (!started || len >= 100000u || c == 0) {
      if (started) complete(o.body);
      if (c > 0) { started = true; loci = l; len = 1; cnts = c; q = c; r = 0; }
      continue;
    }
    if (len == 0 || cnts == 0) { loci = l; len = 1; cnts = c; q = c; r = 0; continue; }
    if ((c <= 5 && c != q) || 4 * q < 3 * c || 4 * q >= 5 * c) {
      complete(o.body);
      loci = l; len = 1; cnts = c; q = c; r = 0;
      continue;
    }
    cnts += c;
    len = l - loci + 1;  // (= len + 1: the loci of a span follow each other)
    r += (int64_t)c - (int64_t)q;
    while (r >= (int64_t)len) { q++; r -= len; }
    while (r < 0) { q--; r += len; }
  }
  complete(o.tail);
  return o;
}
// `width` bytes per locus (k4k_snp_coverage)
static WigOut wig_chromosome(std::unique_ptr<uint8_t[]> cov, int width, uint32_t clen, std::string name) {
  if (width == 1) return wig_walk<uint8_t>(cov.get(), clen, name);
  if (width == 2) return wig_walk<uint16_t>((const uint16_t*)cov.get(), clen, name);
  return wig_walk<uint32_t>((const uint32_t*)cov.get(), clen, name);
}

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) hipFree(p); }
  hipError_t alloc(size_t bytes) { return k4_malloc_retry(&p, bytes ? bytes : 1); }
  template <typename T> T* as() { return (T*)p; }
};

// ---- CStats (libkit4b/Stats.cpp:489-564), operation for operation ------------------------------------------------------------
double calc_nck(uint32_t n, uint32_t k) {
  if (k > n) return 0.0;
  if (k > n / 2) k = n - k;
  long double accum = 1;
  for (uint32_t i = 1; i <= k; i++) accum = accum * (n - k + i) / i;
  return (double)accum;
}
double prob_k_eql_k(uint32_t n, uint32_t k, double p) {
  if (p < 0 || p > 1) return -1;
  const double nck = calc_nck(n, k);
  const double p2 = pow(p, (int32_t)k);
  const double q2 = pow(1 - p, (int32_t)(n - k));
  return nck * p2 * q2;
}
double binomial(int n, int k, double p) {
  if (k > n) return 0.0;
  if (n > 5000) { k = (int)((1000.0 / n) * k); n = 5000; }
  double sum = 0;
  for (int i = 0; i <= k; i++) {
    sum += prob_k_eql_k((uint32_t)n, (uint32_t)i, p);
    if (sum >= 1.0) break;
  }
  return std::min(sum, 1.0);
}

struct LociPV {
  uint32_t loci, rank, num_reads, num_subs, local_reads, local_subs, ref_base;
  uint32_t by_base[5];
  double pvalue, bkgnd;
};

}  // namespace

static int snp_text_dev(k4_index* ix, int vcf, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                        const void* d_pe, const void* d_reads, const void* d_offs, const void* d_lens, int32_t min_snp_reads,
                        double qvalue, double snp_nonref_pcnt, char** csv, uint64_t* csv_bytes, uint64_t* n_snps, void* stream,
                        char** wig = nullptr, uint64_t* wig_bytes = nullptr, std::string* di_text = nullptr, std::string* tri_text = nullptr);
// both files of a kalign SNP run: the SNP file (CSV, or VCF when vcf != 0) and the coverage WIG (<snp file>.covsegs.wig)
extern "C" int k4_snp_files_dev(k4_index* ix, int vcf, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                                const void* d_pe, const void* d_reads, const void* d_offs, const void* d_lens, int32_t min_snp_reads,
                                double qvalue, double snp_nonref_pcnt, char** snp, uint64_t* snp_bytes, uint64_t* n_snps, char** wig,
                                uint64_t* wig_bytes, void* stream) {
  if (!wig || !wig_bytes) return K4_ERR_PARAMS;
  return snp_text_dev(ix, vcf, pe, n_units, d_rr, d_hits, max_ml, d_pe, d_reads, d_offs, d_lens, min_snp_reads, qvalue, snp_nonref_pcnt, snp, snp_bytes,
                      n_snps, stream, wig, wig_bytes);
}
// every file of a kalign SNP run: the SNP file, the coverage WIG and the two haplotype files (.disnp.csv, .trisnp.csv)
extern "C" int k4_snp_run_dev(k4_index* ix, int vcf, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                              const void* d_pe, const void* d_reads, const void* d_offs, const void* d_lens, int32_t min_snp_reads,
                              double qvalue, double snp_nonref_pcnt, k4_snp_files* out, void* stream) {
  if (!out) return K4_ERR_PARAMS;
  memset(out, 0, sizeof(*out));
  std::string di = hap_header(2), tri = hap_header(3);
  int rc = snp_text_dev(ix, vcf, pe, n_units, d_rr, d_hits, max_ml, d_pe, d_reads, d_offs, d_lens, min_snp_reads, qvalue, snp_nonref_pcnt, &out->snp,
                        &out->snp_bytes, &out->n_snps, stream, &out->wig, &out->wig_bytes, &di, &tri);
  if (rc == K4_OK) {
    out->disnp = (char*)malloc(di.size() + 1);
    out->trisnp = (char*)malloc(tri.size() + 1);
    if (!out->disnp || !out->trisnp) rc = k4_fail(ix, K4_ERR_MEM, "out of memory");
    else {
      memcpy(out->disnp, di.c_str(), di.size() + 1); out->disnp_bytes = di.size();
      memcpy(out->trisnp, tri.c_str(), tri.size() + 1); out->trisnp_bytes = tri.size();
    }
  }
  if (rc != K4_OK) {
    free(out->snp); free(out->wig); free(out->disnp); free(out->trisnp);
    memset(out, 0, sizeof(*out));
  }
  return rc;
}
extern "C" int k4_snp_csv_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                              const void* d_pe, const void* d_reads, const void* d_offs, const void* d_lens, int32_t min_snp_reads,
                              double qvalue, double snp_nonref_pcnt, char** csv, uint64_t* csv_bytes, uint64_t* n_snps, void* stream) {
  return snp_text_dev(ix, 0, pe, n_units, d_rr, d_hits, max_ml, d_pe, d_reads, d_offs, d_lens, min_snp_reads, qvalue, snp_nonref_pcnt, csv, csv_bytes,
                      n_snps, stream);
}
// the VCF form (kalign: a SNP file name ending in .vcf, KAligner.cpp:186-187; header :8196-8199, records :7650-7696)
extern "C" int k4_snp_vcf_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                              const void* d_pe, const void* d_reads, const void* d_offs, const void* d_lens, int32_t min_snp_reads,
                              double qvalue, double snp_nonref_pcnt, char** vcf, uint64_t* vcf_bytes, uint64_t* n_snps, void* stream) {
  return snp_text_dev(ix, 1, pe, n_units, d_rr, d_hits, max_ml, d_pe, d_reads, d_offs, d_lens, min_snp_reads, qvalue, snp_nonref_pcnt, vcf, vcf_bytes,
                      n_snps, stream);
}
static int snp_text_dev(k4_index* ix, int vcf, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                        const void* d_pe, const void* d_reads, const void* d_offs, const void* d_lens, int32_t min_snp_reads,
                        double qvalue, double snp_nonref_pcnt, char** csv, uint64_t* csv_bytes, uint64_t* n_snps, void* stream,
                        char** wig, uint64_t* wig_bytes, std::string* di_text, std::string* tri_text) {
  if (!ix || !csv || !csv_bytes) return K4_ERR_PARAMS;
  *csv = nullptr;
  *csv_bytes = 0;
  if (wig) { *wig = nullptr; *wig_bytes = 0; }
  struct WigJob { std::future<WigOut> f; bool close_tail; };
  std::vector<WigJob> wig_jobs;  // one per chromosome with alignments, in chromosome order
  Buf covb, covmax;
  if (n_snps) *n_snps = 0;
  if (n_units < 0 || min_snp_reads < 1 || qvalue < 0.0 || snp_nonref_pcnt < 0.0) return k4_fail(ix, K4_ERR_PARAMS, "SNP parameters out of range");
  if (n_units > 0 && ((pe && !d_pe) || (!pe && (!d_rr || !d_hits || max_ml < 1)) || !d_reads || !d_offs || !d_lens))
    return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  K4_HIP(ix, hipSetDevice(ix->device));
  hipStream_t st = (hipStream_t)stream;
  std::string text =
      "\"SNP_ID\",\"ElType\",\"Species\",\"Chrom\",\"StartLoci\",\"EndLoci\",\"Len\",\"Strand\",\"Rank\",\"PValue\",\"Bases\",\"Mismatches\",\"RefBase\","
      "\"MMBaseA\",\"MMBaseC\",\"MMBaseG\",\"MMBaseT\",\"MMBaseN\",\"BackgroundSubRate\",\"TotWinBases\",\"TotWinMismatches\",\"MarkerID\",\"NumPolymorphicSites\"\n";
  if (vcf)
    text = "##fileformat=VCFv4.1\n##source=k4align1.0\n##reference=" + ix->dataset +
           "\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency\">\n##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read Depth\">\n"
           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n";
  SnpArgs a;
  memset(&a, 0, sizeof(a));
  a.ix = ix->d; a.pe = pe ? 1 : 0; a.n_reads = pe ? 2 * n_units : n_units;
  a.rr = (const k4_read_result*)d_rr; a.hits = (const k4_hit*)d_hits; a.max_ml = max_ml; a.pr = (const k4_pe_read*)d_pe;
  a.reads = (const uint8_t*)d_reads; a.offs = (const uint64_t*)d_offs; a.lens = (const uint32_t*)d_lens;
  uint32_t max_len = 0;
  for (const k4_entry& e : ix->entries) max_len = std::max(max_len, e.seq_len);
  const size_t S = (size_t)max_len + 16;
  Buf cnt, tot, pref, pnon, cands, ncand, tmp;
  const uint32_t cap = 1u << 22;  // candidate loci per chromosome kept on the device (more: the call fails loudly)
  K4_HIP(ix, cnt.alloc(7 * S * 4));
  K4_HIP(ix, tot.alloc(4 * 8));
  K4_HIP(ix, pref.alloc((S + 1) * 8));
  K4_HIP(ix, pnon.alloc((S + 1) * 8));
  K4_HIP(ix, cands.alloc((size_t)cap * sizeof(Cand)));
  K4_HIP(ix, ncand.alloc(4));
  size_t tb = 0;
  K4_HIP(ix, rocprim::exclusive_scan(nullptr, tb, cnt.as<uint32_t>(), pref.as<uint64_t>(), (uint64_t)0, S + 1, rocprim::plus<uint64_t>(), st));
  K4_HIP(ix, tmp.alloc(tb));
  uint64_t tot_snps = 0;
  const double nonref_frac = snp_nonref_pcnt / 100.0;  // m_SNPNonRefPcnt, KAligner.cpp:256
  std::vector<Cand> hc;
  std::vector<LociPV> pv;
  char alts[100] = "", freq[100] = "";  // VCF: ALT and AF of the last SNP that had any (see below)
  std::vector<uint8_t> chrom_hit((size_t)ix->d.n_entries + 1, 0);
  std::vector<uint64_t> ent_start_h((size_t)ix->d.n_entries, 0);
  if (a.n_reads > 0 && ix->d.n_entries) {
    Buf flags;
    K4_HIP(ix, flags.alloc(chrom_hit.size()));
    K4_HIP(ix, hipMemsetAsync(flags.p, 0, chrom_hit.size(), st));
    hipLaunchKernelGGL(k4k_snp_mark, dim3((unsigned)std::min<int64_t>((a.n_reads + 255) / 256, 2048)), dim3(256), 0, st, a, flags.as<uint8_t>(), ix->d.n_entries);
    K4_HIP(ix, hipMemcpyAsync(chrom_hit.data(), flags.p, chrom_hit.size(), hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipMemcpyAsync(ent_start_h.data(), ix->ent_start, ent_start_h.size() * 8, hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipStreamSynchronize(st));
  }
  for (uint32_t chrom = 1; chrom <= ix->d.n_entries && a.n_reads > 0; chrom++) {  // the sorted reads: one chromosome after the other
    if (!chrom_hit[chrom]) continue;  // (nothing the pile-up would take: no device work, no synchronisation for it)
    const k4_entry& e = ix->entries[chrom - 1];
    a.chrom_id = chrom; a.clen = e.seq_len; a.cnt = cnt.as<uint32_t>(); a.tot = tot.as<unsigned long long>();
    a.cs = ent_start_h[chrom - 1];
    const size_t Sc = K4_SNP_STRIDE(a);
    K4_HIP(ix, hipMemsetAsync(cnt.p, 0, 7 * Sc * 4, st));
    K4_HIP(ix, hipMemsetAsync(tot.p, 0, 32, st));
    K4_HIP(ix, hipMemsetAsync(ncand.p, 0, 4, st));
    hipLaunchKernelGGL(k4k_snp_pileup, dim3(2048), dim3(256), 0, st, a);
    unsigned long long t3[4] = {0, 0, 0, 0};
    K4_HIP(ix, hipMemcpyAsync(t3, tot.p, 32, hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipStreamSynchronize(st));
    if (t3[2] == 0) continue;  // no alignment on this chromosome
    size_t wig_slot = 0;
    if (wig) {  // coverage down to the host, its walk on a thread of its own (at most twelve chromosomes in flight)
      if (!covb.p) { K4_HIP(ix, covb.alloc(S * 4)); K4_HIP(ix, covmax.alloc(4)); }
      K4_HIP(ix, hipMemsetAsync(covmax.p, 0, 4, st));
      hipLaunchKernelGGL(k4k_snp_coverage_max, dim3(1024), dim3(256), 0, st, a.cnt, a.cnt + K4_SNP_STRIDE(a), a.clen, covmax.as<uint32_t>());
      uint32_t mx = 0;
      K4_HIP(ix, hipMemcpyAsync(&mx, covmax.p, 4, hipMemcpyDeviceToHost, st));
      K4_HIP(ix, hipStreamSynchronize(st));
      const int width = mx < 256 ? 1 : mx < 65536 ? 2 : 4;
      const dim3 cg((a.clen + 255) / 256);
      if (width == 1) hipLaunchKernelGGL(k4k_snp_coverage<uint8_t>, cg, dim3(256), 0, st, a.cnt, a.cnt + K4_SNP_STRIDE(a), a.clen, covb.as<uint8_t>());
      else if (width == 2) hipLaunchKernelGGL(k4k_snp_coverage<uint16_t>, cg, dim3(256), 0, st, a.cnt, a.cnt + K4_SNP_STRIDE(a), a.clen, covb.as<uint16_t>());
      else hipLaunchKernelGGL(k4k_snp_coverage<uint32_t>, cg, dim3(256), 0, st, a.cnt, a.cnt + K4_SNP_STRIDE(a), a.clen, covb.as<uint32_t>());
      std::unique_ptr<uint8_t[]> cov(new uint8_t[((size_t)a.clen + 1) * (size_t)width]);
      K4_HIP(ix, hipMemcpyAsync(cov.get(), covb.p, (size_t)a.clen * (size_t)width, hipMemcpyDeviceToHost, st));
      K4_HIP(ix, hipStreamSynchronize(st));
      if (wig_jobs.size() >= 12) wig_jobs[wig_jobs.size() - 12].f.wait();
      wig_slot = wig_jobs.size();
      wig_jobs.push_back({std::async(std::launch::async, wig_chromosome, std::move(cov), width, a.clen, std::string(e.name)), false});
    }
    // prefix sums over [0, clen]: element l = sum of the loci below l (the arrays are zero behind clen)
    K4_HIP(ix, rocprim::exclusive_scan(tmp.p, tb, a.cnt, pref.as<uint64_t>(), (uint64_t)0, (size_t)a.clen + 1, rocprim::plus<uint64_t>(), st));
    K4_HIP(ix, rocprim::exclusive_scan(tmp.p, tb, a.cnt + Sc, pnon.as<uint64_t>(), (uint64_t)0, (size_t)a.clen + 1, rocprim::plus<uint64_t>(), st));
    hipLaunchKernelGGL(k4k_snp_candidates, dim3((a.clen + 255) / 256), dim3(256), 0, st, a, pref.as<uint64_t>(), pnon.as<uint64_t>(), (int)min_snp_reads,
                       nonref_frac, cands.as<Cand>(), cap, ncand.as<uint32_t>());
    uint32_t nc = 0;
    K4_HIP(ix, hipMemcpyAsync(&nc, ncand.p, 4, hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipStreamSynchronize(st));
    if (nc > cap) return k4_fail(ix, K4_ERR_MEM, "more than %u candidate SNP loci on %s", cap, e.name);
    hc.resize(nc);
    if (nc) K4_HIP(ix, hipMemcpy(hc.data(), cands.p, (size_t)nc * sizeof(Cand), hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end(), [](const Cand& x, const Cand& y) { return x.loci < y.loci; });
    // ---- OutputSNPs from here on (:7320, :7425-7450, :7567-7640) -------------------------------------------------------------
    double global_rate = (double)t3[1] / (double)(1 + t3[0] + t3[1]);
    global_rate = std::max(0.005, global_rate);  // cMinSeqErrRate
    pv.clear();
    for (const Cand& c : hc) {
      const uint32_t ltmm = c.n_nonref <= c.loc_mm ? c.loc_mm - c.n_nonref : 0;
      const uint32_t ltm = c.n_ref < c.loc_m ? c.loc_m - c.n_ref : 0;
      double local_rate;
      if ((ltmm + ltm) == 0) local_rate = global_rate;
      else {
        local_rate = (double)ltmm / (double)(ltmm + ltm);
        if (local_rate < global_rate) local_rate = global_rate;
      }
      if (local_rate > 0.20) continue;  // cMaxBkgdNoiseThres
      LociPV p;
      const int tot_bases = (int)(c.n_ref + c.n_nonref);
      p.pvalue = 1.0 - binomial(tot_bases, (int)c.n_nonref, local_rate);
      p.loci = c.loci; p.rank = 0; p.bkgnd = local_rate; p.local_reads = ltmm + ltm; p.local_subs = ltmm;
      p.num_reads = (uint32_t)tot_bases; p.num_subs = c.n_nonref;
      for (int b = 0; b < 5; b++) p.by_base[b] = c.by_base[b];
      p.ref_base = c.ref_base > 4 ? 4 : c.ref_base;
      pv.push_back(p);
    }
    if (wig && !pv.empty()) wig_jobs[wig_slot].close_tail = true;  // (a chromosome without a candidate never closes its last span)
    if (pv.empty()) continue;
    std::stable_sort(pv.begin(), pv.end(), [](const LociPV& x, const LociPV& y) { return x.pvalue < y.pvalue; });
    size_t n_acc = 0;
    for (size_t k = 0; k < pv.size(); k++) {  // Benjamini-Hochberg
      const double adj = ((k + 1) / (double)pv.size()) * qvalue;
      if (pv[k].pvalue >= adj) break;
      pv[k].rank = (uint32_t)(k + 1);
      n_acc++;
    }
    pv.resize(n_acc);
    std::sort(pv.begin(), pv.end(), [](const LociPV& x, const LociPV& y) { return x.loci < y.loci; });
    if ((di_text || tri_text) && n_acc >= 2) {  // ---- DiSNPs / TriSNPs of this chromosome (:7767-8101) ------------------------------
      const int mean_len = (int)(uint32_t)((t3[3] + t3[2] - 1) / t3[2]);
      const int max_sep = std::min(300, mean_len);  // m_MaxDiSNPSep = min(cDfltMaxDiSNPSep, MeanReadLen), :7346
      std::vector<uint32_t> loci(n_acc);
      std::vector<int32_t> slot(2 * n_acc, -1);  // [0, n_acc): pair ending at k, [n_acc, 2 n_acc): triple ending at k
      uint32_t n_di = 0, n_tri = 0;
      for (size_t k = 0; k < n_acc; k++) {
        loci[k] = pv[k].loci;
        const int cur = (int)pv[k].loci;
        if (k >= 1 && cur > 0 && cur - (int)pv[k - 1].loci <= max_sep) slot[k] = (int32_t)n_di++;
        if (k >= 2 && pv[k - 1].loci > 0 && cur > 0 && cur - (int)pv[k - 2].loci <= max_sep) slot[n_acc + k] = (int32_t)n_tri++;
      }
      if (n_di) {  // (a triple holds two pairs: no pair, no triple)
        Buf dl, ds, dd, dt;
        K4_HIP(ix, dl.alloc(n_acc * 4));
        K4_HIP(ix, ds.alloc(2 * n_acc * 4));
        K4_HIP(ix, dd.alloc((size_t)n_di * 17 * 4));
        K4_HIP(ix, dt.alloc((size_t)n_tri * 65 * 4));
        K4_HIP(ix, hipMemcpyAsync(dl.p, loci.data(), n_acc * 4, hipMemcpyHostToDevice, st));
        K4_HIP(ix, hipMemcpyAsync(ds.p, slot.data(), 2 * n_acc * 4, hipMemcpyHostToDevice, st));
        K4_HIP(ix, hipMemsetAsync(dd.p, 0, (size_t)n_di * 17 * 4, st));
        if (n_tri) K4_HIP(ix, hipMemsetAsync(dt.p, 0, (size_t)n_tri * 65 * 4, st));
        HapArgs hp;
        hp.loci = dl.as<uint32_t>(); hp.di_slot = ds.as<int32_t>(); hp.tri_slot = ds.as<int32_t>() + n_acc; hp.n_loci = (uint32_t)n_acc;
        hp.di = dd.as<uint32_t>(); hp.tri = dt.as<uint32_t>();
        const int64_t nb = std::min<int64_t>((a.n_reads + 255) / 256, 8192);
        hipLaunchKernelGGL(k4k_snp_haplotypes, dim3((unsigned)nb), dim3(256), 0, st, a, hp);
        std::vector<uint32_t> hd((size_t)n_di * 17), ht((size_t)n_tri * 65);
        K4_HIP(ix, hipMemcpyAsync(hd.data(), dd.p, hd.size() * 4, hipMemcpyDeviceToHost, st));
        if (n_tri) K4_HIP(ix, hipMemcpyAsync(ht.data(), dt.p, ht.size() * 4, hipMemcpyDeviceToHost, st));
        K4_HIP(ix, hipStreamSynchronize(st));
        int tot_di = 0, tot_tri = 0;  // (the ids restart with every chromosome, :7634-7635)
        for (size_t k = 1; k < n_acc; k++) {
          if (di_text && slot[k] >= 0) {
            const uint32_t l2[2] = {pv[k - 1].loci, pv[k].loci}, r2[2] = {pv[k - 1].ref_base, pv[k].ref_base};
            if (hap_line(*di_text, "DiSNPs", tot_di + 1, ix->dataset, e.name, l2, r2, 2, &hd[(size_t)slot[k] * 17], min_snp_reads)) tot_di++;
          }
          if (tri_text && k >= 2 && slot[n_acc + k] >= 0) {
            const uint32_t l3[3] = {pv[k - 2].loci, pv[k - 1].loci, pv[k].loci}, r3[3] = {pv[k - 2].ref_base, pv[k - 1].ref_base, pv[k].ref_base};
            if (hap_line(*tri_text, "TriSNPs", tot_tri + 1, ix->dataset, e.name, l3, r3, 3, &ht[(size_t)slot[n_acc + k] * 65], min_snp_reads)) tot_tri++;
          }
        }
      }
    }
    for (LociPV& p : pv) {
      tot_snps++;
      int rel = (int)(999 - ((999 * (int64_t)p.rank) / (int64_t)n_acc));
      if (rel < 1) rel = 1;
      char line[512];
      if (vcf) {  // alternative alleles with at least a tenth of the strongest one's count, their frequencies, phred of the p-value
        uint32_t thres = 0;
        for (uint32_t b = 0; b < 4; b++)
          if (b != p.ref_base && p.by_base[b] > thres) thres = p.by_base[b];
        thres = std::max((thres + 5) / 10, 1u);
        // (the reference never clears its two strings: a SNP whose mismatches are all N prints what the SNP before it left there)
        int ao = 0, fo = 0;
        for (uint32_t b = 0; b < 4; b++) {
          if (b == p.ref_base || p.by_base[b] < thres) continue;
          if (ao > 0) { alts[ao++] = ','; freq[fo++] = ','; }
          alts[ao++] = "ACGT"[b]; alts[ao] = 0;
          fo += sprintf(&freq[fo], "%1.4f", (double)p.by_base[b] / p.num_reads);
        }
        const int phred = p.pvalue < 0.0000000001 ? 100 : (int)(0.5 + (10.0 * log10(1.0 / p.pvalue)));
        const int n = snprintf(line, sizeof(line), "%s\t%u\tSNP%d\t%c\t%s\t%d\tPASS\tAF=%s;DP=%d\n", e.name, p.loci + 1, (int)tot_snps, "ACGTN"[p.ref_base],
                               alts, phred, freq, (int)p.num_reads);
        text.append(line, (size_t)n);
        continue;
      }
      p.by_base[p.ref_base] = p.num_reads - p.num_subs;  // :7698
      const int n = snprintf(line, sizeof(line), "%d,\"SNP\",\"%s\",\"%s\",%d,%d,1,\"+\",%d,%f,%d,%d,\"%c\",%d,%d,%d,%d,%d,%f,%d,%d,%d,%d\n", (int)tot_snps,
                             ix->dataset.c_str(), e.name, (int)p.loci, (int)p.loci, rel, p.pvalue, (int)p.num_reads, (int)p.num_subs, "ACGTN"[p.ref_base],
                             (int)p.by_base[0], (int)p.by_base[1], (int)p.by_base[2], (int)p.by_base[3], (int)p.by_base[4], p.bkgnd, (int)p.local_reads,
                             (int)p.local_subs, 0, 0);
      text.append(line, (size_t)n);
    }
  }
  if (wig) {  // header + the chromosomes' texts, copied side by side into the one block the caller gets (gigabytes at low coverage)
    const std::string hdr = "track type=wiggle_0 name=\"Coverage\" description=\"Alignment Segment Coverage\" useScore=1\n";  // :8235
    std::vector<WigOut> parts;
    parts.reserve(wig_jobs.size());
    std::vector<size_t> at;
    size_t total = hdr.size();
    for (WigJob& j : wig_jobs) {
      parts.push_back(j.f.get());
      if (!j.close_tail) parts.back().tail.clear();
      at.push_back(total);
      total += parts.back().body.size() + parts.back().tail.size();
    }
    char* wo = (char*)malloc(total + 1);
    if (!wo) return k4_fail(ix, K4_ERR_MEM, "out of memory");
    memcpy(wo, hdr.data(), hdr.size());
    std::vector<std::future<void>> cp;
    for (size_t k = 0; k < parts.size(); k++)
      cp.push_back(std::async(std::launch::async, [&, k] {
        memcpy(wo + at[k], parts[k].body.data(), parts[k].body.size());
        memcpy(wo + at[k] + parts[k].body.size(), parts[k].tail.data(), parts[k].tail.size());
      }));
    for (std::future<void>& f : cp) f.get();
    wo[total] = 0;
    *wig = wo;
    *wig_bytes = total;
  }
  char* out = (char*)malloc(text.size() + 1);
  if (!out) return k4_fail(ix, K4_ERR_MEM, "out of memory");
  memcpy(out, text.c_str(), text.size() + 1);
  *csv = out;
  *csv_bytes = text.size();
  if (n_snps) *n_snps = tot_snps;
  return K4_OK;
}
extern "C" void k4_free_host(void* p) { free(p); }
