/* oracle/k4oracle_snp.c -- TEST INFRASTRUCTURE ONLY (see k4oracle.h): CPU restatement of kalign's SNP calling, main CSV only.
 *
 *   CKAligner::ProcessSNPs  ngskit4b/KAligner.cpp:8168-8590  per-locus base counts over the accepted alignments of a chromosome
 *   CKAligner::OutputSNPs   ngskit4b/KAligner.cpp:7098-7760  background rate in a 51-base window, binomial p-value, Benjamini-
 *                                                            Hochberg cut, the "SNP_ID",... CSV line
 *   CStats::Binomial / ProbKeqlk / Calc_nCk  libkit4b/Stats.cpp:489-564
 *
 * Not restated (files the reference writes on request): marker
 * sequences, SNP centroids, VCF / BED forms, packed base alleles, SOLiD colourspace.
 * Ranks: the reference orders the candidates by p-value with a multi-threaded quicksort (equal p-values -- most are 0 -- in no
 * defined order); here equal p-values keep locus order, which is what the reference produces on inputs small enough for its
 * sort to stay sequential.  Parity pinned by tests/golden/snp_*.csv (written by `ngskit4b kalign -p -P -S`). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "k4oracle.h"
#include "k4oracle_priv.h"

typedef struct { uint32_t ref, nonref, by_base[5]; uint8_t ref_base; } snp_cnts;
typedef struct {
  uint32_t loci, rank, num_reads, num_subs, local_reads, local_subs;
  double pvalue, bkgnd;
  snp_cnts c;
} loci_pv;

static double calc_nck(uint32_t n, uint32_t k) { /* Stats.cpp:489-524 (what is left of it after `accum = 1`) */
  if (k > n) return 0.0;
  if (k > n / 2) k = n - k;
  long double accum = 1;
  for (uint32_t i = 1; i <= k; i++) accum = accum * (n - k + i) / i;
  return (double)accum;
}
static double prob_k_eql_k(uint32_t n, uint32_t k, double p) { /* :530-538 */
  if (p < 0 || p > 1) return -1;
  return calc_nck(n, k) * pow(p, (double)(int32_t)k) * pow(1 - p, (double)(int32_t)(n - k));
}
static double binomial(int n, int k, double p) { /* :543-564 */
  if (k > n) return 0.0;
  if (n > 5000) { k = (int)((1000.0 / n) * k); n = 5000; }
  double sum = 0;
  for (int i = 0; i <= k; i++) {
    sum += prob_k_eql_k((uint32_t)n, (uint32_t)i, p);
    if (sum >= 1.0) break;
  }
  return sum < 1.0 ? sum : 1.0;
}

static int cmp_pv(const void* a, const void* b) { /* SortLociPValues :11153, ties: locus order (see the header) */
  const loci_pv* x = (const loci_pv*)a; const loci_pv* y = (const loci_pv*)b;
  if (x->pvalue < y->pvalue) return -1;
  if (x->pvalue > y->pvalue) return 1;
  return x->loci < y->loci ? -1 : x->loci > y->loci;
}
static int cmp_loci(const void* a, const void* b) { /* SortPValuesLoci :11165 */
  const loci_pv* x = (const loci_pv*)a; const loci_pv* y = (const loci_pv*)b;
  return x->loci < y->loci ? -1 : x->loci > y->loci;
}

typedef struct { char* p; size_t len, cap; } sbuf;
static void sb_put(sbuf* b, const char* s, size_t n);
/* the coverage WIG kalign writes beside the SNP file (<snp file minus extension>.covsegs.wig): variableStep spans of roughly equal
 * coverage (AccumWIGCnts / CompleteWIGSpan, KAligner.cpp:6993-7085), fed with the LOCUS AS COUNTED FROM 0 (:7375), so a span that
 * starts at locus 0 is never written (CompleteWIGSpan wants a start above 0) */
typedef struct { uint32_t chrom, rptd_chrom, loci, len, rptd_len; uint64_t cnts; sbuf* out; const char* name; } wig_state;
static void wig_complete(wig_state* w) {
  if (w->chrom != 0 && w->len > 0 && w->loci > 0 && w->cnts > 0) {
    char line[256];
    if (w->chrom != w->rptd_chrom || w->len != w->rptd_len) {
      const int n = snprintf(line, sizeof(line), "variableStep chrom=%s span=%d\n", w->name, (int)w->len);
      sb_put(w->out, line, (size_t)n);
      w->rptd_chrom = w->chrom; w->rptd_len = w->len;
    }
    const int n = snprintf(line, sizeof(line), "%d %d\n", (int)w->loci, (int)(uint32_t)((w->cnts + w->len - 1) / w->len));
    sb_put(w->out, line, (size_t)n);
  }
  w->loci = 0; w->len = 0; w->cnts = 0;
}
static void wig_accum(wig_state* w, uint32_t chrom, uint32_t loci, uint32_t cnts) {
  const uint32_t max_span = 100000;
  if (chrom != w->chrom || w->len >= max_span || cnts == 0) {
    if (w->chrom != 0) wig_complete(w);
    if (cnts > 0) { w->chrom = chrom; w->loci = loci; w->len = 1; w->cnts = cnts; }
    return;
  }
  if (w->len == 0 || w->cnts == 0) { w->loci = loci; w->len = 1; w->cnts = cnts; return; }
  const uint32_t mean100 = 100 * (uint32_t)(w->cnts / (uint64_t)w->len);
  if ((cnts <= 5 && (cnts * 100) != mean100) || (mean100 < (cnts * 75) || mean100 >= (cnts * 125))) {
    wig_complete(w);
    w->loci = loci; w->len = 1; w->cnts = cnts;
    return;
  }
  w->cnts += cnts;
  w->len = loci - w->loci + 1;
}
static void sb_put(sbuf* b, const char* s, size_t n) {
  if (b->len + n + 1 > b->cap) { b->cap = (b->len + n + 1) * 2 + 4096; b->p = (char*)realloc(b->p, b->cap); }
  memcpy(b->p + b->len, s, n); b->len += n; b->p[b->len] = 0;
}

/* ---- DiSNPs / TriSNPs: the haplotype files kalign writes beside the SNP file (<snp file minus extension>.disnp.csv / .trisnp.csv,
 * KAligner.cpp:4553-4554; headers :8252-8330; lines :7767-8101) --------------------------------------------------------------------
 * For two (three) called SNP loci following each other within min(300, mean aligned length) bases: the reads that cover all of them,
 * the base each shows at each locus, and how many reads carry each of the 16 (64) combinations. */
typedef struct { uint32_t start, end, hit_len, match_loci, match_len; int64_t read; char strand; } hap_read;
static int cmp_hap_read(const void* a, const void* b) { /* SortHitMatch :10969-11014 within one chromosome: AdjStartLoci, AdjHitLen, strand */
  const hap_read* x = (const hap_read*)a; const hap_read* y = (const hap_read*)b;
  if (x->start != y->start) return x->start < y->start ? -1 : 1;
  if (x->hit_len != y->hit_len) return x->hit_len < y->hit_len ? -1 : 1;
  if (x->strand != y->strand) return x->strand < y->strand ? -1 : 1;
  return x->read < y->read ? -1 : x->read > y->read;
}
static int adj_align_snp_base(const hap_read* r, const uint8_t* bases, uint32_t loci) { /* AdjAlignSNPBase :1581-1632 */
  if (r->start > loci || r->end < loci) return 7;
  if (r->strand == '+') return bases[loci - r->match_loci] & 0x07;
  int b = bases[r->match_loci + r->match_len - loci - 1] & 0x07;
  return b <= 3 ? 3 - b : b;
}
/* IterateReadsOverlapping (:10475-10546) over the chromosome's sorted alignments: every accepted alignment without InDel / splice
 * that starts at or before the first locus and ends at or after the last one; the walk stops at the first that starts behind the
 * first locus.  (The reference resumes later walks from where an earlier one found its first read; the reads it skips that way can
 * neither cover the new loci nor stop the walk, so the walk from the chromosome's first read sees the same reads.) */
static void hap_counts(const hap_read* rd, size_t n_rd, const uint8_t* reads, const uint64_t* offs, const uint32_t* loci, int n_loci,
                       int* cnts, int by_base[3][4], int* depth, int* antisense) {
  memset(cnts, 0, 64 * sizeof(int)); memset(by_base, 0, 12 * sizeof(int));
  *depth = 0; *antisense = 0;
  for (size_t q = 0; q < n_rd; q++) {
    const hap_read* r = &rd[q];
    if (!(r->start <= loci[0] && r->end >= loci[n_loci - 1])) {
      if (r->start > loci[0]) break;
      continue;
    }
    int b[3], ok = 1;
    for (int k = 0; k < n_loci && ok; k++) { b[k] = adj_align_snp_base(r, reads + offs[r->read], loci[k]); ok = b[k] <= 3; }
    if (!ok) continue; /* an N in the read at one of the loci (:7801-7805, :7950-7957) */
    int idx = 0;
    for (int k = 0; k < n_loci; k++) { by_base[k][b[k]]++; idx = (idx << 2) | b[k]; }
    cnts[idx]++; (*depth)++;
    if (r->strand == '-') (*antisense)++;
  }
}
static void hap_line(sbuf* out, const char* type, int id, const char* species, const char* chrom, const uint32_t* loci, const uint8_t* ref,
                     int n_loci, int by_base[3][4], int depth, int antisense, int n_hap, const int* cnts) {
  char line[1200];
  int n = snprintf(line, sizeof(line), "%d,\"%s\",\"%s\",\"%s\",", id, type, species, chrom);
  for (int k = 0; k < n_loci; k++)
    n += snprintf(line + n, sizeof(line) - n, "%d,\"%c\",%d,%d,%d,%d,0,", (int)loci[k], "acgtn"[ref[k] > 4 ? 4 : ref[k]], by_base[k][0], by_base[k][1],
                  by_base[k][2], by_base[k][3]);
  n += snprintf(line + n, sizeof(line) - n, "%d,%d,%d", depth, antisense, n_hap);
  for (int k = 0; k < (n_loci == 2 ? 16 : 64); k++) n += snprintf(line + n, sizeof(line) - n, ",%d", cnts[k]);
  line[n++] = '\n';
  sb_put(out, line, (size_t)n);
}
static void hap_header(sbuf* out, int n_loci) { /* :8252-8330 */
  char line[1600];
  int n = snprintf(line, sizeof(line), "\"%s_ID\",\"ElType\",\"Species\",\"Chrom\"", n_loci == 2 ? "DiSNPs" : "TriSNPs");
  for (int k = 1; k <= n_loci; k++)
    n += snprintf(line + n, sizeof(line) - n, ",\"SNP%dLoci\",\"SNP%dRefBase\",\"SNP%dBaseAcnt\",\"SNP%dBaseCcnt\",\"SNP%dBaseGcnt\",\"SNP%dBaseTcnt\",\"SNP%dBaseNcnt\"",
                  k, k, k, k, k, k, k);
  n += snprintf(line + n, sizeof(line) - n, ",\"Depth\",\"Antisense\",\"Haplotypes\"");
  for (int k = 0; k < (n_loci == 2 ? 16 : 64); k++) {
    line[n++] = ','; line[n++] = '"';
    for (int j = n_loci - 1; j >= 0; j--) line[n++] = "acgt"[(k >> (2 * j)) & 3];
    line[n++] = '"';
  }
  line[n++] = '\n';
  sb_put(out, line, (size_t)n);
}

/* nar / hits: one per read (hits[i * hit_stride] = the reported alignment).  Returns the CSV text (malloc'd; caller frees) and
 * the number of SNPs through *n_snps; NULL on bad arguments. */
char* k4o_snp_text(const k4o_index* ix, int vcf, int64_t n_reads, const int32_t* nar, const k4o_hit* hits, int hit_stride,
                   const uint8_t* reads, const uint64_t* offs, const uint32_t* lens, int min_snp_reads, double qvalue, double snp_nonref_pcnt,
                   int64_t* n_snps);
char* k4o_snp_csv(const k4o_index* ix, int64_t n_reads, const int32_t* nar, const k4o_hit* hits, int hit_stride, const uint8_t* reads,
                  const uint64_t* offs, const uint32_t* lens, int min_snp_reads, double qvalue, double snp_nonref_pcnt,
                  int64_t* n_snps) {
  return k4o_snp_text(ix, 0, n_reads, nar, hits, hit_stride, reads, offs, lens, min_snp_reads, qvalue, snp_nonref_pcnt, n_snps);
}
/* vcf: the VCF form (file name ending in .vcf, KAligner.cpp:186-187; lines :7650-7696) -- the records only: the reference's header
 * names its own version and the path of the index file */
static char* snp_run(const k4o_index* ix, int vcf, int64_t n_reads, const int32_t* nar, const k4o_hit* hits, int hit_stride,
                     const uint8_t* reads, const uint64_t* offs, const uint32_t* lens, int min_snp_reads, double qvalue, double snp_nonref_pcnt,
                     int64_t* n_snps, sbuf* wig, sbuf* di, sbuf* tri);
char* k4o_snp_text(const k4o_index* ix, int vcf, int64_t n_reads, const int32_t* nar, const k4o_hit* hits, int hit_stride,
                   const uint8_t* reads, const uint64_t* offs, const uint32_t* lens, int min_snp_reads, double qvalue, double snp_nonref_pcnt,
                   int64_t* n_snps) {
  return snp_run(ix, vcf, n_reads, nar, hits, hit_stride, reads, offs, lens, min_snp_reads, qvalue, snp_nonref_pcnt, n_snps, NULL, NULL, NULL);
}
/* the coverage WIG of the same run (see wig_state above); release with k4o_free */
char* k4o_snp_wig(const k4o_index* ix, int64_t n_reads, const int32_t* nar, const k4o_hit* hits, int hit_stride, const uint8_t* reads,
                  const uint64_t* offs, const uint32_t* lens, int min_snp_reads, double qvalue, double snp_nonref_pcnt) {
  sbuf wig = {0, 0, 0};
  const char* hdr = "track type=wiggle_0 name=\"Coverage\" description=\"Alignment Segment Coverage\" useScore=1\n"; /* :8235 */
  sb_put(&wig, hdr, strlen(hdr));
  char* t = snp_run(ix, 0, n_reads, nar, hits, hit_stride, reads, offs, lens, min_snp_reads, qvalue, snp_nonref_pcnt, NULL, &wig, NULL, NULL);
  if (!t) { free(wig.p); return NULL; }
  free(t);
  return wig.p;
}
/* the haplotype file of the same run: n_loci 2 = .disnp.csv, 3 = .trisnp.csv; release with k4o_free */
char* k4o_snp_haplotypes(const k4o_index* ix, int n_loci, int64_t n_reads, const int32_t* nar, const k4o_hit* hits, int hit_stride,
                         const uint8_t* reads, const uint64_t* offs, const uint32_t* lens, int min_snp_reads, double qvalue,
                         double snp_nonref_pcnt) {
  if (n_loci != 2 && n_loci != 3) return NULL;
  sbuf hap = {0, 0, 0};
  hap_header(&hap, n_loci);
  char* t = snp_run(ix, 0, n_reads, nar, hits, hit_stride, reads, offs, lens, min_snp_reads, qvalue, snp_nonref_pcnt, NULL, NULL,
                    n_loci == 2 ? &hap : NULL, n_loci == 3 ? &hap : NULL);
  if (!t) { free(hap.p); return NULL; }
  free(t);
  return hap.p;
}
static char* snp_run(const k4o_index* ix, int vcf, int64_t n_reads, const int32_t* nar, const k4o_hit* hits, int hit_stride,
                     const uint8_t* reads, const uint64_t* offs, const uint32_t* lens, int min_snp_reads, double qvalue, double snp_nonref_pcnt,
                     int64_t* n_snps, sbuf* wig, sbuf* di, sbuf* tri) {
  if (!ix || n_reads < 0 || min_snp_reads < 1) return NULL;
  const double nonref_frac = snp_nonref_pcnt / 100.0; /* m_SNPNonRefPcnt, KAligner.cpp:256 */
  sbuf out = {0, 0, 0};
  const char* hdr = "\"SNP_ID\",\"ElType\",\"Species\",\"Chrom\",\"StartLoci\",\"EndLoci\",\"Len\",\"Strand\",\"Rank\",\"PValue\",\"Bases\",\"Mismatches\",\"RefBase\",\"MMBaseA\",\"MMBaseC\",\"MMBaseG\",\"MMBaseT\",\"MMBaseN\",\"BackgroundSubRate\",\"TotWinBases\",\"TotWinMismatches\",\"MarkerID\",\"NumPolymorphicSites\"\n";
  if (!vcf) sb_put(&out, hdr, strlen(hdr));
  else sb_put(&out, "", 0);
  int64_t tot_snps = 0;
  char alts[100] = "", freq[100] = "";
  uint8_t* rs = (uint8_t*)malloc(1 << 16);
  for (uint32_t chrom = 1; chrom <= ix->n_entries; chrom++) { /* the reads come sorted: one chromosome after the other (:8340-8400) */
    const k4o_entry* e = &ix->entries[chrom - 1];
    const uint32_t clen = e->seq_len;
    snp_cnts* cnt = NULL;
    uint64_t tot_match = 0, tot_mismatch = 0, tot_read_len = 0, num_reads = 0;
    hap_read* hr = NULL; /* the chromosome's alignments as IterateReadsOverlapping sees them (DiSNPs / TriSNPs only) */
    size_t n_hr = 0, cap_hr = 0;
    for (int64_t i = 0; i < n_reads; i++) {
      if (nar[i] != K4O_NAR_ACCEPTED) continue;
      const k4o_hit* h = &hits[i * (int64_t)hit_stride];
      if (h->chrom_id != chrom) continue;
      if (h->ext & (K4O_EXT_INDEL | K4O_EXT_SPLICE)) continue; /* :8345 */
      if (!cnt) cnt = (snp_cnts*)calloc((size_t)clen + 16, sizeof(snp_cnts));
      const uint32_t tl = h->ext & 0xFFF, tr = (h->ext >> 12) & 0xFFF;
      uint32_t match_len = (uint32_t)h->match_len - tl - tr;                 /* AdjHitLen */
      const uint32_t hit_loci = h->match_loci + (h->strand == '+' ? tl : tr); /* AdjStartLoci */
      if ((uint64_t)hit_loci + match_len > clen) continue; /* GetSeq returns less than asked for: the read is skipped (:8420) */
      const uint8_t* src = reads + offs[i] + tl; /* pSeg->ReadOfs (0) + TrimLeft */
      for (uint32_t q = 0; q < match_len; q++) rs[q] = src[q] & 0x07;
      if (h->strand == '-') k4o_revcomp(rs, (int)match_len);
      if (hit_loci + match_len > clen) { if ((match_len = clen - hit_loci) < 10) continue; }
      tot_read_len += match_len; num_reads++; /* :8466-8467 */
      if (di || tri) {
        if (n_hr == cap_hr) { cap_hr = cap_hr ? cap_hr * 2 : 4096; hr = (hap_read*)realloc(hr, cap_hr * sizeof(hap_read)); }
        hap_read* r = &hr[n_hr++];
        r->start = hit_loci; r->end = hit_loci + match_len - 1; r->hit_len = match_len; /* AdjStartLoci / AdjEndLoci / AdjHitLen */
        r->match_loci = h->match_loci; r->match_len = (uint32_t)h->match_len; r->read = i; r->strand = (char)h->strand;
      }
      const uint8_t* ref = ix->seq + e->start_ofs + hit_loci;
      snp_cnts* s = cnt + hit_loci;
      for (uint32_t q = 0; q < match_len; q++, s++) { /* :8468-8557, base space */
        const uint8_t a = ref[q] & 0x07;
        uint8_t r = rs[q];
        if (a >= K4O_N || r > K4O_N) continue;
        s->ref_base = a;
        if (a == r) { s->ref++; tot_match++; }
        else {
          if (r > 3) r = K4O_N;
          s->by_base[r]++; s->nonref++; tot_mismatch++;
        }
      }
    }
    if (!cnt) { free(hr); continue; }
    if (hr) qsort(hr, n_hr, sizeof(hap_read), cmp_hap_read);
    /* ---- OutputSNPs for this chromosome --------------------------------------------------------------------------------- */
    double global_rate = (double)tot_mismatch / (double)(1 + tot_match + tot_mismatch);
    if (global_rate < 0.005) global_rate = 0.005; /* cMinSeqErrRate */
    const uint32_t flank = 51 / 2, win = flank * 2 + 1; /* cSNPBkgndRateWindow */
    uint32_t loc_mm = 0, loc_m = 0;
    const snp_cnts* win_r = cnt;
    const snp_cnts* win_l = cnt;
    for (uint32_t l = 0; l < (win < clen ? win : clen); l++, win_r++) { loc_mm += win_r->nonref; loc_m += win_r->ref; }
    loci_pv* pv = NULL;
    size_t n_pv = 0, cap_pv = 0;
    wig_state ws = {0, 0, 0, 0, 0, 0, wig, e->name}; /* InitialiseWIGSpan, :7347 */
    for (uint32_t l = 0; l < clen; l++) {
      const snp_cnts* s = cnt + l;
      if (wig) wig_accum(&ws, chrom, l, s->nonref + s->ref); /* :7375, before any test */
      if (l > flank && (l + flank) < clen) { /* :7358-7374 slide the window */
        loc_mm = loc_mm >= win_l->nonref ? loc_mm - win_l->nonref : 0;
        loc_m = loc_m >= win_l->ref ? loc_m - win_l->ref : 0;
        loc_mm += win_r->nonref; loc_m += win_r->ref;
        win_l++; win_r++;
      }
      const int tot_bases = (int)(s->nonref + s->ref);
      if (tot_bases < min_snp_reads) continue;
      if (s->nonref < 1) continue; /* cMinSNPreads */
      const double proportion = (double)s->nonref / tot_bases;
      if (proportion < nonref_frac) continue;
      const uint32_t ltmm = s->nonref <= loc_mm ? loc_mm - s->nonref : 0;
      const uint32_t ltm = s->ref < loc_m ? loc_m - s->ref : 0;
      double local_rate;
      if ((ltmm + ltm) == 0) local_rate = global_rate;
      else {
        local_rate = (double)ltmm / (double)(ltmm + ltm);
        if (local_rate < global_rate) local_rate = global_rate;
      }
      if (local_rate > 0.20) continue; /* cMaxBkgdNoiseThres */
      if (n_pv == cap_pv) { cap_pv = cap_pv ? cap_pv * 2 : 1024; pv = (loci_pv*)realloc(pv, cap_pv * sizeof(loci_pv)); }
      loci_pv* p = &pv[n_pv++];
      p->pvalue = 1.0 - binomial(tot_bases, (int)s->nonref, local_rate);
      p->loci = l; p->rank = 0; p->bkgnd = local_rate; p->local_reads = ltmm + ltm; p->local_subs = ltmm;
      p->num_reads = (uint32_t)tot_bases; p->num_subs = s->nonref; p->c = *s;
    }
    /* a chromosome without a single candidate returns early (:7582-7608): its last open span is never closed -- lost */
    if (wig && n_pv) wig_complete(&ws); /* CompleteWIGSpan(true), :8135 */
    if (n_pv) {
      qsort(pv, n_pv, sizeof(loci_pv), cmp_pv);
      size_t n_acc = 0;
      for (size_t k = 0; k < n_pv; k++) { /* Benjamini-Hochberg, :7614-7622 */
        const double adj = ((k + 1) / (double)n_pv) * qvalue;
        if (pv[k].pvalue >= adj) break;
        pv[k].rank = (uint32_t)(k + 1);
        n_acc++;
      }
      qsort(pv, n_acc, sizeof(loci_pv), cmp_loci);
      const int mean_len = (int)(uint32_t)((tot_read_len + num_reads - 1) / num_reads);
      const int max_sep = mean_len < 300 ? mean_len : 300; /* m_MaxDiSNPSep = min(cDfltMaxDiSNPSep, MeanReadLen), :7346 */
      int prev_di = -1, prev_tri = -1, first_tri = -1, tot_di = 0, tot_tri = 0; /* :7627-7635: the ids restart with every chromosome */
      for (size_t k = 0; k < n_acc; k++) {
        loci_pv* p = &pv[k];
        tot_snps++;
        if (di || tri) { /* :7767-8101; after the SNP's own line in the reference, but into files of their own */
          const int cur = (int)p->loci;
          int cnts[64], by_base[3][4], depth, anti, n_hap;
          if (di && prev_di != -1 && cur > 0 && (cur - prev_di) <= max_sep) {
            const uint32_t l2[2] = {(uint32_t)prev_di, (uint32_t)cur};
            const uint8_t r2[2] = {cnt[prev_di].ref_base, cnt[cur].ref_base};
            hap_counts(hr, n_hr, reads, offs, l2, 2, cnts, by_base, &depth, &anti);
            n_hap = 0;
            if (depth >= min_snp_reads) {
              const int thres = (depth + 5) / 10 > 5 ? (depth + 5) / 10 : 5;
              for (int q = 0; q < 16; q++) { if (cnts[q] >= thres) n_hap++; else cnts[q] = 0; }
            }
            if (n_hap >= 2) hap_line(di, "DiSNPs", ++tot_di, ix->dataset, e->name, l2, r2, 2, by_base, depth, anti, n_hap, cnts);
          }
          if (tri && first_tri != -1 && prev_tri > 0 && cur > 0 && (cur - first_tri) <= max_sep) {
            const uint32_t l3[3] = {(uint32_t)first_tri, (uint32_t)prev_tri, (uint32_t)cur};
            const uint8_t r3[3] = {cnt[first_tri].ref_base, cnt[prev_tri].ref_base, cnt[cur].ref_base};
            hap_counts(hr, n_hr, reads, offs, l3, 3, cnts, by_base, &depth, &anti);
            n_hap = 0;
            if (depth >= min_snp_reads) {
              const int thres = (depth + 5) / 10 > 5 ? (depth + 5) / 10 : 5;
              for (int q = 0; q < 64; q++) { if (cnts[q] >= thres) n_hap++; else cnts[q] = 0; }
            }
            if (n_hap >= 3) hap_line(tri, "TriSNPs", ++tot_tri, ix->dataset, e->name, l3, r3, 3, by_base, depth, anti, n_hap, cnts);
          }
          prev_di = cur; first_tri = prev_tri; prev_tri = cur;
        }
        int rel = (int)(999 - ((999 * (int64_t)p->rank) / (int64_t)n_acc));
        if (rel < 1) rel = 1;
        char line[512];
        if (vcf) { /* :7650-7696 */
          uint32_t thres = 0;
          for (int b = 0; b < 4; b++)
            if (b != p->c.ref_base && p->c.by_base[b] > thres) thres = p->c.by_base[b];
          thres = (thres + 5) / 10;
          if (thres < 1) thres = 1;
          /* (szALTs / szAltFreq are locals of the loop body that nothing clears: a SNP whose mismatches are all N -- no allele
           * reaches the threshold -- prints what the SNP before it left there; kept, :7652-7683) */
          int ao = 0, fo = 0;
          for (int b = 0; b < 4; b++) {
            if (b == p->c.ref_base || p->c.by_base[b] < thres) continue;
            if (ao > 0) { alts[ao++] = ','; freq[fo++] = ','; }
            alts[ao++] = "ACGT"[b]; alts[ao] = 0;
            fo += sprintf(&freq[fo], "%1.4f", (double)p->c.by_base[b] / p->num_reads);
          }
          const int phred = p->pvalue < 0.0000000001 ? 100 : (int)(0.5 + (10.0 * log10(1.0 / p->pvalue)));
          const int n = snprintf(line, sizeof(line), "%s\t%u\tSNP%d\t%c\t%s\t%d\tPASS\tAF=%s;DP=%d\n", e->name, p->loci + 1, (int)tot_snps,
                                 "ACGTN"[p->c.ref_base > 4 ? 4 : p->c.ref_base], alts, phred, freq, (int)p->num_reads);
          sb_put(&out, line, (size_t)n);
          continue;
        }
        p->c.by_base[p->c.ref_base] = p->num_reads - p->num_subs; /* :7698 */
        const int n = snprintf(line, sizeof(line), "%d,\"SNP\",\"%s\",\"%s\",%d,%d,1,\"+\",%d,%f,%d,%d,\"%c\",%d,%d,%d,%d,%d,%f,%d,%d,%d,%d\n",
                               (int)tot_snps, ix->dataset, e->name, (int)p->loci, (int)p->loci, rel, p->pvalue, (int)p->num_reads, (int)p->num_subs,
                               "ACGTN"[p->c.ref_base > 4 ? 4 : p->c.ref_base], (int)p->c.by_base[0], (int)p->c.by_base[1], (int)p->c.by_base[2],
                               (int)p->c.by_base[3], (int)p->c.by_base[4], p->bkgnd, (int)p->local_reads, (int)p->local_subs, 0, 0);
        sb_put(&out, line, (size_t)n);
      }
    }
    free(pv);
    free(cnt);
    free(hr);
  }
  free(rs);
  if (n_snps) *n_snps = tot_snps;
  return out.p;
}
void k4o_free(void* p) { free(p); }
