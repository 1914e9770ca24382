/* oracle/k4oracle_ext.c -- TEST INFRASTRUCTURE ONLY (see k4oracle.h).
 *
 * Plain-C restatement of the OPTIONAL phases of CSfxArray::AlignReads and of the post-alignment stages
 * that `kalign -c / -a / -A` bring with them.  Every function cites the reference lines it follows (paths relative to
 * /root/reference); quirks are kept as they are and named where they matter.
 *   CSfxArray::AdaptiveTrim            libkit4b/SfxArray.cpp:5561-5795
 *   CSfxArray::LocateCoreMultiples     libkit4b/SfxArray.cpp:5806-6369, chimeric branch :6064-6189
 *   CSfxArray::LocateInDels            libkit4b/SfxArray.cpp:7526-7832
 *   CSfxArray::ExploreInDelMatchRight  libkit4b/SfxArray.cpp:9277-9495
 *   CSfxArray::ExploreInDelMatchLeft   libkit4b/SfxArray.cpp:9506-9735
 *   CSfxArray::LocateSpliceJuncts      libkit4b/SfxArray.cpp:7208-7523
 *   CSfxArray::ExploreSpliceRight      libkit4b/SfxArray.cpp:8771-9011
 *   CSfxArray::ExploreSpliceLeft       libkit4b/SfxArray.cpp:9022-9265
 *   CSfxArray::AlignReads              libkit4b/SfxArray.cpp:7838-7933 (all three optional phases)
 *   CKAligner::AlignRead               ngskit4b/KAligner.cpp:9583-10105 (flag / trim handling :9866-9887)
 *   CKAligner::AutoTrimFlanks          ngskit4b/KAligner.cpp:1714-1917
 *   CKAligner::RemoveOrphanSpliceJuncts / RemoveOrphanMicroInDels   ngskit4b/KAligner.cpp:2406-2594
 * Parity status: PINNED by tests/golden/align_chim_*.npz, align_indel_*.npz, align_splice_*.npz (vectors captured from
 * oracle/_ref/libk4ref.so by tests/golden/make_golden.py) and the `kalign -c / -a / -A` SAM files of tests/golden/sam_*.
 */
#define _GNU_SOURCE
#include "k4oracle_priv.h"
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* constants, libkit4b/SfxArray.h:15-69 */
#define C_MAX_MM_EXPLORE_INDEL 7   /* cMaxMMExploreInDel */
#define C_MIN_INDEL_SEQ_LEN 7      /* cMinInDelSeqLen */
#define C_MAX_MICRO_INDEL_MM 2     /* cMaxMicroInDelMM */
#define C_MIN_JUNCT_ALIGN_SEP 25   /* cMinJunctAlignSep */
#define C_MAX_JUNCT_ALIGN_MM 2     /* cMaxJunctAlignMM */
#define C_MIN_JUNCT_SEG_LEN 10     /* cMinJunctSegLen */
#define C_MAX_PUT_INDEL_OFSS 80    /* cMaxPutInDelOfss */
#define C_BASE_SCORE 500
#define C_MAX_SCORE 1000
#define C_SCORE_MATCH 3
#define C_SCORE_MISMATCH 5
#define C_SPLICE_DONOR_ACCEPT 50
#define C_SPLICE_LEN 10
#define C_SCORE_INDEL_OPN 20
#define C_SCORE_INDEL_EXTN 1
#define C_MIN_AT_SEQ_LEN 25u
#define C_MAX_AT_SEQ_LEN 2048u
#define C_MIN_AT_TRIMMED_LEN 15u
#define C_MAX_AT_MM 15u
#define C_MAX_AT_MAX_FLANK 10u
#define C_MIN_AT_EXACT_LEN 8u
#define ERR_PARAMS (-100) /* eBSFerrParams */

#define IMAX(a, b) ((a) > (b) ? (a) : (b))
#define IMIN(a, b) ((a) < (b) ? (a) : (b))

/* the subset of tsHitLoci (libkit4b/SfxArray.h:239-260) the explore functions fill; MatchLoci are CONCAT offsets until
 * LocateInDels / LocateSpliceJuncts convert them */
typedef struct {
  uint16_t read_ofs; uint8_t strand; uint32_t chrom_id; uint64_t match_loci; uint16_t match_len; uint8_t mismatches;
} xseg;
typedef struct { uint8_t f_indel, f_insert, f_splice; uint16_t score; xseg seg[2]; } xhit;

static inline uint8_t tb(const k4o_index* ix, int64_t pos) { /* target symbol; beyond the block: a separator */
  return (pos < 0 || (uint64_t)pos >= ix->n) ? (uint8_t)K4O_EOS : (uint8_t)(ix->seq[pos] & 0x07);
}

/* ---- AdaptiveTrim, SfxArray.cpp:5561-5795 ------------------------------------------------------------------------------ */
typedef struct { uint16_t ofs, len; uint8_t mm, trim5, trim3; } at_region;

int k4o_adaptive_trim(uint32_t seq_len, const uint8_t* probe, const uint8_t* targ, uint32_t min_trim_len, uint32_t max_mm,
                      uint32_t min_flank, uint32_t* p_len, uint32_t* p_start, uint32_t* p_end, uint32_t* p_mms) {
  if (p_len) *p_len = 0;
  if (p_start) *p_start = 0;
  if (p_end) *p_end = 0;
  if (p_mms) *p_mms = 0;
  if (seq_len < C_MIN_AT_SEQ_LEN || seq_len > C_MAX_AT_SEQ_LEN || !probe || !targ || min_trim_len < C_MIN_AT_TRIMMED_LEN ||
      min_trim_len > seq_len || max_mm > (((C_MAX_AT_MM * seq_len) + 99) / 100) || min_flank > C_MAX_AT_MAX_FLANK)
    return ERR_PARAMS; /* :5601-5605 */
  if (min_flank == 0) min_flank = 1;
  uint32_t ofs;
  if (min_trim_len == seq_len) { /* :5612-5639 only the mismatch total matters */
    uint32_t allowed = ((seq_len * max_mm) + 99) / 100, mms = 0;
    for (ofs = 0; ofs < seq_len; ofs++) {
      if ((probe[ofs] & 0x0f) != (targ[ofs] & 0x0f)) {
        if (++mms > allowed) break;
        if (ofs < min_flank || (seq_len - ofs) < min_flank) { mms = allowed + 1; break; }
      }
    }
    if (mms <= allowed) {
      if (p_len) *p_len = seq_len;
      if (p_mms) *p_mms = mms;
      return (int)seq_len;
    }
    return 0;
  }
  /* :5641-5667 runs of matches / mismatches */
  static __thread at_region reg[C_MAX_AT_SEQ_LEN];
  uint32_t n_reg = 0, n_min_exact = 0;
  at_region* cur = NULL;
  for (ofs = 0; ofs < seq_len; ofs++) {
    const int mm = (probe[ofs] & 0x0f) != (targ[ofs] & 0x0f);
    if (cur == NULL || mm != cur->mm) {
      cur = &reg[n_reg++];
      cur->mm = (uint8_t)mm; cur->len = 1; cur->ofs = (uint16_t)ofs; cur->trim5 = cur->trim3 = 0;
    } else {
      cur->len += 1;
      if (cur->len == C_MIN_AT_EXACT_LEN && !mm) n_min_exact += 1;
    }
  }
  if (!n_min_exact) return 0;
  /* :5672-5710 which regions may start / end a trimmed sequence */
  uint32_t first_start = 0, last_start = 0, first_end = 0, last_end = 0, r;
  for (r = 0; r < n_reg; r++) {
    cur = &reg[r];
    if (cur->mm == 0 && cur->len >= min_flank) {
      if (cur->ofs <= seq_len - min_trim_len) {
        cur->trim5 = 1;
        last_start = r + 1;
        if (first_start == 0) first_start = last_start;
      } else
        cur->trim5 = 0;
      if ((uint32_t)cur->ofs + cur->len >= (uint32_t)(uint16_t)min_trim_len) {
        cur->trim3 = 1;
        last_end = r + 1;
        if (first_end == 0) first_end = last_end;
      } else
        cur->trim3 = 0;
    } else {
      cur->trim5 = 0;
      cur->trim3 = 0;
    }
  }
  if (first_start == 0 || first_end == 0) return 0;
  /* :5712-5780 from every start region extend over the following regions while the mismatch rate allows */
  uint32_t best_len = 0, best_mm = 0, best_start = 0, best_end = 0;
  for (uint32_t s = first_start - 1; s < last_start; s++) {
    const at_region* st = &reg[s];
    if (st->trim5 == 0) continue;
    uint32_t cur_len = 0, cur_mm = 0, idx = s;
    const at_region* p = st;
    while (idx++ < last_end) {
      cur_len += p->len;
      if (p->mm) {
        if (max_mm == 0) break;
        cur_mm += p->len;
        if ((max_mm + 1.0) / 100.0 <= (double)cur_mm / (seq_len - st->ofs)) break;
      } else if (best_len == 0) {
        best_start = st->ofs;
        best_end = seq_len - (best_start + cur_len);
        best_len = cur_len;
        best_mm = 0;
        p += 1;
        continue;
      }
      if (cur_len < min_trim_len || !p->trim3) { p += 1; continue; }
      p += 1;
      if ((max_mm + 1.0) / 100.0 <= (double)cur_mm / cur_len) continue;
      if (best_len < cur_len || (best_len == cur_len && (best_mm == 0 || cur_mm < best_mm))) {
        best_start = st->ofs;
        best_end = seq_len - (best_start + cur_len);
        best_len = cur_len;
        best_mm = cur_mm;
      }
    }
  }
  if (best_len >= min_trim_len) {
    if (p_len) *p_len = best_len;
    if (p_start) *p_start = best_start;
    if (p_end) *p_end = best_end;
    if (p_mms) *p_mms = best_mm;
    return (int)best_len;
  }
  return 0;
}

/* result code of one LocateCoreMultiples call, SfxArray.cpp:6345-6368 */
static int lcm_result(int* p_inst, int* p_low, int* p_nxt, int inst, int low, int nxt, int mm_delta, int max_hits) {
  if (*p_low == low && *p_inst == inst) {
    if (*p_nxt > nxt) {
      *p_nxt = nxt;
      if (nxt - *p_low < mm_delta) return K4O_HR_MMDELTA;
      return K4O_HR_RMMDELTA;
    }
    return K4O_HR_NONE;
  }
  *p_low = low; *p_inst = inst; *p_nxt = nxt;
  if (inst >= 1 && (nxt - low) < mm_delta) return K4O_HR_MMDELTA;
  if (inst > max_hits) return K4O_HR_HITINSTS;
  return K4O_HR_HITS;
}

static void store_chimeric(k4o_hit* h, const k4o_entry* e, int64_t left, char strand, int probe_len, int mm, uint32_t trim5,
                           uint32_t trim3) { /* :6123-6146 */
  memset(h, 0, sizeof(*h));
  h->chrom_id = e->entry_id;
  h->match_loci = (uint32_t)((uint64_t)left - e->start_ofs);
  h->match_len = (uint16_t)probe_len;
  h->strand = (uint8_t)strand;
  h->mismatches = (uint8_t)mm;
  const uint32_t tl = strand == '+' ? trim5 : trim3, tr = strand == '+' ? trim3 : trim5;
  h->ext = K4O_EXT_CHIMERIC | (tl & 0xFFF) | ((tr & 0xFFF) << 12);
}

/* ---- LocateCoreMultiples with MinChimericLen in 15..99 (any other value: the default branch), :5806-6369 ------------------ */
int k4o_locate_core_multiples_chimeric(const k4o_index* ix, int min_chimeric_len, int max_tot_mm, int core_len, int core_delta,
                                       int max_slides, int mm_delta, int strand, int* p_inst, int* p_low, int* p_nxt,
                                       uint8_t* probe, int probe_len, int max_hits, k4o_hit* hits, k4o_counters* ctr) {
  if (!(min_chimeric_len >= 15 && min_chimeric_len <= 99)) /* :5878-5883 */
    return k4o_locate_core_multiples(ix, max_tot_mm, core_len, core_delta, max_slides, mm_delta, strand, p_inst, p_low, p_nxt,
                                     probe, probe_len, max_hits, hits, ctr);
  const uint32_t min_probe_chim = (uint32_t)IMAX(core_len, (min_chimeric_len * probe_len) / 100);
  if (ix->n == 0) return -1;
  if (*p_inst > max_hits && *p_low == 0) return K4O_HR_HITINSTS;
  if (*p_inst >= 1 && *p_low == 0 && (*p_nxt - *p_low) < mm_delta) return K4O_HR_MMDELTA;
  int inst, low, nxt;
  if (*p_inst <= 0 || *p_low < 0 || *p_nxt < 0) {
    inst = *p_inst = 0;
    low = *p_low = max_tot_mm + mm_delta + 1;
    nxt = *p_nxt = low;
  } else {
    inst = *p_inst; low = *p_low; nxt = *p_nxt;
  }
  int cur_hit = inst < max_hits ? inst : -1;
  const int max_iter = ix->max_iter;
  const int64_t n = (int64_t)ix->n;
  char cur_strand = '+';
  if (strand == K4O_STRAND_CRICK) { k4o_revcomp(probe, probe_len); cur_strand = '-'; }
  int best_len = 0, best_mms = 0; /* BestChimericLen / BestMaxChimericMMs: NOT reset between the strands (:5936-5940) */
  k4oi_idset ids;
  k4oi_idset_init(&ids);
  do {
    int cur_delta = core_delta, slides = 0;
    uint32_t n_nodes = 0;
    k4oi_idset_clear(&ids);
    for (int ofs = 0; slides < max_slides && ofs <= probe_len - core_len && cur_delta > core_len / 3 && n_nodes < 1024000u;
         slides++, ofs += cur_delta) {
      if (ofs + core_len + cur_delta > probe_len) cur_delta = probe_len - (ofs + core_len);
      int64_t t = k4o_locate_first_exact(ix, probe + ofs, core_len, 0, n - 1, ctr);
      if (t == 0) continue;
      t -= 1;
      int iter = 0, first = 1;
      while (!max_iter || iter < max_iter) {
        if (n_nodes >= 1024000u) break;
        if (!first) {
          if (t + 1 >= n || k4o_sa_at(ix, t + 1) + core_len > n) break;
          if (k4oi_cmp_probe_targ(probe + ofs, ix->seq + k4o_sa_at(ix, t + 1), core_len) != 0) break;
          t += 1;
        }
        first = 0;
        const int64_t pos = k4o_sa_at(ix, t);
        if (pos < (int64_t)(uint32_t)ofs) continue;
        const int64_t left = pos - ofs;
        const k4o_entry* e = k4oi_map_chunk_hit2entry(ix, (uint64_t)left);
        if (e == NULL || (uint64_t)left + (uint32_t)probe_len - 1 > e->end_ofs) continue;
        const uint32_t targ_id = (uint32_t)(1 + pos - (uint32_t)ofs);
        if (!k4oi_idset_insert(&ids, targ_id)) continue;
        n_nodes++;
        iter++;
        if (ctr) ctr->n_cand++;
        /* :6064-6189 (the call's return value is not looked at: a parameter error leaves the outputs zero) */
        uint32_t t_len = 0, t5 = 0, t3 = 0, t_mm = 0;
        k4o_adaptive_trim((uint32_t)probe_len, probe, ix->seq + left, min_probe_chim, (uint32_t)max_tot_mm, 3, &t_len, &t5, &t3, &t_mm);
        const int c_len = (int)t_len, c_mms = (int)t_mm;
        if (c_len < (int)min_probe_chim) continue;
        if (c_len > best_len || (c_len == best_len && c_mms < best_mms)) {
          if (best_len > 0 && c_len > best_len) low = c_mms + mm_delta + 1;
          best_len = c_len; best_mms = c_mms;
          cur_hit = 0;
          inst = 1;
          nxt = low;
          low = c_mms;
          e = k4oi_map_chunk_hit2entry(ix, (uint64_t)left + t5);
          store_chimeric(&hits[0], e, left, cur_strand, probe_len, c_mms, t5, t3);
        } else if (c_len == best_len && c_mms == best_mms) {
          inst += 1;
          if (cur_hit != -1 && inst <= max_hits) {
            cur_hit += 1;
            e = k4oi_map_chunk_hit2entry(ix, (uint64_t)left + t5);
            store_chimeric(&hits[cur_hit], e, left, cur_strand, probe_len, c_mms, t5, t3);
          }
        } else if (c_len == best_len && c_mms < nxt)
          nxt = c_mms;
        if (c_len == probe_len && inst > max_hits && low == 0) break; /* :6187 */
      }
      if (inst > max_hits && low == 0) { strand = 3; break; }
    }
    if (cur_strand == '+' && strand == K4O_STRAND_BOTH) {
      k4o_revcomp(probe, probe_len);
      cur_strand = '-';
      strand = K4O_STRAND_CRICK;
    } else
      strand = 3;
  } while (!(inst > max_hits && low == 0) && strand != 3);
  k4oi_idset_free(&ids);
  if (cur_strand == '-') k4o_revcomp(probe, probe_len);
  return lcm_result(p_inst, p_low, p_nxt, inst, low, nxt, mm_delta, max_hits);
}

/* ---- ExploreInDelMatchRight, SfxArray.cpp:9277-9495.  targ_ofs: concat offset the probe's first base is laid on -------- */
static int explore_indel_right(const k4o_index* ix, char cur_strand, int micro_indel_len, int max_tot_mm, int probe_len,
                               const uint8_t* probe, const k4o_entry* e, int64_t targ_ofs, xhit* hit) {
  memset(hit, 0, sizeof(*hit));
  if (targ_ofs < (int64_t)e->start_ofs || (targ_ofs + probe_len - 1) > (int64_t)e->end_ofs) return 0;
  const uint32_t targ_seq_len = (uint32_t)(e->seq_len - (targ_ofs - e->start_ofs));
  xhit ins, del;
  memset(&ins, 0, sizeof(ins));
  memset(&del, 0, sizeof(del));
  if (max_tot_mm > C_MAX_PUT_INDEL_OFSS) max_tot_mm = C_MAX_PUT_INDEL_OFSS;
  int mm_ofs[C_MAX_PUT_INDEL_OFSS + 8];
  int n_mm = 0;
  uint32_t idx;
  uint8_t pb = 0, tv = 0;
  for (idx = 0; idx < (uint32_t)probe_len && n_mm <= IMAX(max_tot_mm, C_MAX_MM_EXPLORE_INDEL); idx++) {
    pb = probe[idx] & 0x07; tv = tb(ix, targ_ofs + idx);
    if (tv > K4O_N || pb > K4O_N) return 0;
    if (pb == tv && pb <= K4O_T) continue;
    mm_ofs[n_mm++] = (int)idx;
  }
  if (n_mm < C_MAX_MM_EXPLORE_INDEL || C_MIN_INDEL_SEQ_LEN > (probe_len - mm_ofs[0])) { /* :9344-9357 */
    if (n_mm > max_tot_mm) return 0;
    hit->seg[0].match_len = (uint16_t)probe_len;
    hit->seg[0].match_loci = (uint64_t)targ_ofs;
    hit->seg[0].mismatches = (uint8_t)n_mm;
    hit->seg[0].strand = (uint8_t)cur_strand;
    hit->score = (uint16_t)(C_BASE_SCORE + probe_len * C_SCORE_MATCH - n_mm * C_SCORE_MISMATCH);
    return 1;
  }
  const int tot_mm = IMIN(max_tot_mm, n_mm);
  for (int pass = 0; pass < 2; pass++) { /* 0: insertion into the probe :9363-9423, 1: deletion from it :9426-9482 */
    xhit* best = pass == 0 ? &ins : &del;
    for (int m = 0; m <= tot_mm && C_MIN_INDEL_SEQ_LEN < (probe_len - mm_ofs[m]); m++) {
      for (int gl = 1; gl <= micro_indel_len; gl++) {
        int score = C_BASE_SCORE + probe_len * C_SCORE_MATCH - ((gl - 1) * C_SCORE_INDEL_EXTN + C_SCORE_INDEL_OPN);
        const int p0 = pass == 0 ? mm_ofs[m] + gl : mm_ofs[m];
        const int t0 = pass == 0 ? mm_ofs[m] : mm_ofs[m] + gl;
        const uint32_t tmp_probe_len = (uint32_t)(probe_len - p0);
        if (tmp_probe_len < C_MIN_INDEL_SEQ_LEN) break;
        const uint32_t tmp_targ_len = targ_seq_len - (uint32_t)t0;
        if (tmp_targ_len < tmp_probe_len) break;
        int mms = 0;
        for (idx = 0; idx < tmp_probe_len && (m + mms) <= max_tot_mm; idx++) {
          pb = probe[p0 + idx] & 0x07; tv = tb(ix, targ_ofs + t0 + idx);
          if (pb > K4O_N || tv > K4O_N) break;
          if (pb == tv && pb <= K4O_T) continue;
          mms += 1;
          score -= C_SCORE_MISMATCH;
          if (tmp_probe_len < (uint32_t)(C_MIN_INDEL_SEQ_LEN * mms)) break;
        }
        if (idx != tmp_probe_len) continue;
        if (score > (int)best->score) {
          memset(best->seg, 0, sizeof(best->seg));
          best->seg[0].match_len = (uint16_t)mm_ofs[m];
          best->seg[0].match_loci = (uint64_t)targ_ofs;
          best->seg[0].mismatches = (uint8_t)m;
          best->seg[0].strand = (uint8_t)cur_strand;
          best->seg[1].match_len = (uint16_t)tmp_probe_len;
          best->seg[1].match_loci = pass == 0 ? best->seg[0].match_loci + best->seg[0].match_len
                                              : (uint64_t)targ_ofs + best->seg[0].match_len + (uint64_t)gl;
          best->seg[1].mismatches = (uint8_t)mms;
          best->seg[1].read_ofs = (uint16_t)(pass == 0 ? best->seg[0].match_len + gl : best->seg[0].match_len);
          best->seg[1].strand = (uint8_t)cur_strand;
          best->score = (uint16_t)score;
          best->f_indel = 1;
          best->f_insert = pass == 0 ? 1 : 0;
          best->f_splice = 0;
        }
      }
    }
  }
  if (del.score == 0 && ins.score == 0) return 0;
  if (del.score > ins.score) { *hit = del; return 3; }
  *hit = ins;
  return 2;
}

/* ---- ExploreInDelMatchLeft, SfxArray.cpp:9506-9735 ------------------------------------------------------------------------ */
static int explore_indel_left(const k4o_index* ix, char cur_strand, int micro_indel_len, int max_tot_mm, int probe_len,
                              const uint8_t* probe, const k4o_entry* e, int64_t targ_ofs, xhit* hit) {
  memset(hit, 0, sizeof(*hit));
  if (targ_ofs < (int64_t)e->start_ofs || (targ_ofs + probe_len - 1) > (int64_t)e->end_ofs) return 0;
  xhit ins, del;
  memset(&ins, 0, sizeof(ins));
  memset(&del, 0, sizeof(del));
  if (max_tot_mm > C_MAX_PUT_INDEL_OFSS) max_tot_mm = C_MAX_PUT_INDEL_OFSS;
  int mm_ofs[C_MAX_PUT_INDEL_OFSS + 8];
  int n_mm = 0, idx;
  uint8_t pb = 0, tv = 0;
  for (idx = probe_len - 1; idx >= 0 && n_mm <= IMAX(max_tot_mm, C_MAX_MM_EXPLORE_INDEL); idx--) {
    pb = probe[idx] & 0x07; tv = tb(ix, targ_ofs + idx);
    if (tv > K4O_N || pb > K4O_N) return 0;
    if (pb == tv && pb <= K4O_T) continue;
    mm_ofs[n_mm++] = idx;
  }
  if (n_mm < C_MAX_MM_EXPLORE_INDEL || C_MIN_INDEL_SEQ_LEN > mm_ofs[0]) { /* :9574-9587 */
    if (n_mm > max_tot_mm) return 0;
    hit->seg[0].match_len = (uint16_t)probe_len;
    hit->seg[0].match_loci = (uint64_t)targ_ofs;
    hit->seg[0].mismatches = (uint8_t)n_mm;
    hit->seg[0].strand = (uint8_t)cur_strand;
    hit->score = (uint16_t)(C_BASE_SCORE + probe_len * C_SCORE_MATCH - n_mm * C_SCORE_MISMATCH);
    return 1;
  }
  const int tot_mm = IMIN(max_tot_mm, n_mm);
  for (int pass = 0; pass < 2; pass++) { /* 0: insertion :9592-9658, 1: deletion :9662-9722 */
    xhit* best = pass == 0 ? &ins : &del;
    for (int m = 0; m <= tot_mm && C_MIN_INDEL_SEQ_LEN < mm_ofs[m]; m++) {
      for (int gl = 1; gl <= micro_indel_len; gl++) {
        int score = C_BASE_SCORE + probe_len * C_SCORE_MATCH - ((gl - 1) * C_SCORE_INDEL_EXTN + C_SCORE_INDEL_OPN);
        score -= m * C_SCORE_MISMATCH;
        if (score < (int)best->score) break;
        const int p0 = pass == 0 ? mm_ofs[m] - gl : mm_ofs[m];  /* both walk leftwards from here */
        const int t0 = pass == 0 ? mm_ofs[m] : mm_ofs[m] - gl;
        const uint32_t tmp_probe_len = pass == 0 ? (uint32_t)(mm_ofs[m] - (gl - 1)) : (uint32_t)(mm_ofs[m] + 1);
        if (tmp_probe_len < C_MIN_INDEL_SEQ_LEN) break;
        if (pass == 1 && (uint32_t)gl > (uint64_t)targ_ofs) break; /* :9678 (the comparison is unsigned 64-bit there) */
        int mms = 0;
        for (idx = 0; idx < (int)tmp_probe_len && (m + mms) <= max_tot_mm; idx++) {
          pb = probe[p0 - idx] & 0x07; tv = tb(ix, targ_ofs + t0 - idx);
          if (pb > K4O_N || tv > K4O_N) break;
          if (pb == tv && pb <= K4O_T) continue;
          mms += 1;
          score -= C_SCORE_MISMATCH;
          if (tmp_probe_len < (uint32_t)(C_MIN_INDEL_SEQ_LEN * mms)) break;
        }
        if (idx != (int)tmp_probe_len) continue;
        if (score > (int)best->score) {
          memset(best->seg, 0, sizeof(best->seg));
          best->seg[0].match_len = (uint16_t)tmp_probe_len;
          /* :9640 / :9705 the concat offset goes through a 32-bit cast here */
          best->seg[0].match_loci = pass == 0 ? (uint64_t)(uint32_t)(targ_ofs + gl) : (uint64_t)(uint32_t)(targ_ofs - gl);
          best->seg[0].mismatches = (uint8_t)mms;
          best->seg[0].strand = (uint8_t)cur_strand;
          if (pass == 0) {
            best->seg[1].match_len = (uint16_t)(probe_len - (int)(tmp_probe_len + (uint32_t)gl));
            best->seg[1].match_loci = best->seg[0].match_loci + best->seg[0].match_len;
            best->seg[1].read_ofs = (uint16_t)(best->seg[0].match_len + gl);
          } else {
            best->seg[1].match_len = (uint16_t)(probe_len - (int)tmp_probe_len);
            best->seg[1].match_loci = best->seg[0].match_loci + tmp_probe_len + (uint64_t)gl;
            best->seg[1].read_ofs = (uint16_t)tmp_probe_len;
          }
          best->seg[1].mismatches = (uint8_t)m;
          best->seg[1].strand = (uint8_t)cur_strand;
          best->score = (uint16_t)score;
          best->f_indel = 1;
          best->f_insert = pass == 0 ? 1 : 0;
          best->f_splice = 0;
        }
      }
    }
  }
  if (del.score == 0 && ins.score == 0) return 0;
  if (del.score > ins.score) { *hit = del; return 3; }
  *hit = ins;
  return 2;
}

/* ---- ExploreSpliceRight, SfxArray.cpp:8771-9011 ---------------------------------------------------------------------------- */
static int explore_splice_right(const k4o_index* ix, char cur_strand, int max_splice_junct_len, int max_tot_mm, int core_len,
                                int probe_len, const uint8_t* probe, int64_t targ_ofs, int64_t targ_len, xhit* hit) {
  memset(hit, 0, sizeof(*hit));
  if ((targ_ofs + probe_len + C_MIN_JUNCT_ALIGN_SEP) > targ_len) return 0;
  xhit cur;
  memset(&cur, 0, sizeof(cur));
  if (max_tot_mm > C_MAX_JUNCT_ALIGN_MM) max_tot_mm = C_MAX_JUNCT_ALIGN_MM;
  int mm_ofs[C_MAX_PUT_INDEL_OFSS + 8];
  int n_mm = 0;
  uint32_t idx;
  uint8_t pb = 0, tv = 0;
  for (idx = (uint32_t)core_len; idx < (uint32_t)probe_len && n_mm <= IMAX(max_tot_mm, C_MAX_JUNCT_ALIGN_MM * 5); idx++) {
    pb = probe[idx] & 0x07; tv = tb(ix, targ_ofs + idx);
    if (tv > K4O_N || pb > K4O_N) return 0;
    if (pb == tv && pb <= K4O_T) continue;
    mm_ofs[n_mm++] = (int)idx;
  }
  if (n_mm < (C_MAX_JUNCT_ALIGN_MM * 4) || C_MIN_JUNCT_SEG_LEN > (probe_len - mm_ofs[0])) { /* :8838-8853 */
    if (n_mm > max_tot_mm) return 0;
    hit->seg[0].match_len = (uint16_t)probe_len;
    hit->seg[0].match_loci = (uint64_t)targ_ofs;
    hit->seg[0].mismatches = (uint8_t)n_mm;
    hit->seg[0].strand = (uint8_t)cur_strand;
    hit->score = (uint16_t)(C_BASE_SCORE + probe_len * C_SCORE_MATCH - n_mm * C_SCORE_MISMATCH);
    return 1;
  }
  const int tot_mm = IMIN(n_mm, max_tot_mm);
  /* :8868-8873 the first 35 bases right of the last usable mismatch must stay on the chromosome */
  /* (TotMM-1 indexes below the first element when MaxTotMM is 0: the reference then reads the word in front of its
   * array; ProbeMMOfss[-1] is whatever the stack holds -- restated as offset 0, see the GPU tests' -s0 case) */
  {
    const int64_t p0 = targ_ofs + (tot_mm >= 1 ? mm_ofs[tot_mm - 1] : 0);
    for (idx = 0; idx < (uint32_t)(C_MIN_JUNCT_ALIGN_SEP + C_MIN_JUNCT_SEG_LEN); idx++)
      if (tb(ix, p0 + idx) > K4O_N) return 0;
  }
  for (int m = 0; m <= tot_mm && C_MIN_JUNCT_SEG_LEN < (probe_len - mm_ofs[m]); m++) {
    if (cur.score >= C_MAX_SCORE) break;
    const uint32_t max_seg_len = (uint32_t)(probe_len - mm_ofs[m]);
    const uint8_t* cur_p = probe + mm_ofs[m];
    const int64_t donor = targ_ofs + mm_ofs[m]; /* pTDonor */
    int64_t t_start = donor + C_MIN_JUNCT_ALIGN_SEP;
    const int max_hash_diff = 4 * (max_tot_mm - m);
    int probe_hash = 100000;
    for (idx = 0; idx < max_seg_len; idx++) probe_hash += cur_p[idx] & 0x07;
    const int min_hash = probe_hash - max_hash_diff, max_hash = probe_hash + max_hash_diff;
    int targ_hash = 100000;
    int64_t t_end = t_start;
    for (idx = 0; idx < max_seg_len - 1; idx++) {
      if (tb(ix, t_end) > K4O_N) break;
      targ_hash += tb(ix, t_end++);
    }
    if (idx < (max_seg_len - 1)) break;
    for (int gap = C_MIN_JUNCT_ALIGN_SEP; gap < max_splice_junct_len - (int)max_seg_len; gap++, t_start++, t_end++) {
      if ((tv = tb(ix, t_end)) > K4O_N) break;
      targ_hash += tv;
      if (targ_hash < min_hash || targ_hash > max_hash) {
        targ_hash -= tb(ix, t_start);
        continue;
      }
      targ_hash -= tb(ix, t_start);
      const uint32_t tmp_targ_len = (uint32_t)(targ_len - (targ_ofs + mm_ofs[m] + gap + 1));
      if (tmp_targ_len < max_seg_len) break;
      int mms = 0;
      for (idx = 0; idx < max_seg_len && (m + mms) < max_tot_mm; idx++) {
        pb = cur_p[idx] & 0x07; tv = tb(ix, t_start + idx);
        if (pb > K4O_N || tv > K4O_N) break;
        if (pb == tv && pb <= K4O_T) continue;
        mms += 1;
      }
      if (idx != max_seg_len) {
        if (pb > K4O_N || tv > K4O_N) break;
        continue;
      }
      uint32_t score = (uint32_t)(C_BASE_SCORE + probe_len * C_SCORE_MATCH - (((m + mms) * C_SCORE_MISMATCH) + ((gap / 1000) * C_SPLICE_LEN)));
      const uint8_t d0 = tb(ix, donor), d1 = tb(ix, donor + 1), a0 = tb(ix, t_start - 1), a1 = tb(ix, t_start - 2);
      const int gt_ag = d0 == K4O_G && d1 == K4O_T && a0 == K4O_G && a1 == K4O_A;
      const int ct_ac = d0 == K4O_C && d1 == K4O_T && a0 == K4O_C && a1 == K4O_A;
      if (cur_strand == '+') {
        if (gt_ag) score += C_SPLICE_DONOR_ACCEPT; else if (ct_ac) score += C_SPLICE_DONOR_ACCEPT / 2;
      } else {
        if (ct_ac) score += C_SPLICE_DONOR_ACCEPT; else if (gt_ag) score += C_SPLICE_DONOR_ACCEPT / 2;
      }
      if (score > cur.score) {
        memset(cur.seg, 0, sizeof(cur.seg));
        cur.seg[0].match_len = (uint16_t)mm_ofs[m];
        cur.seg[0].match_loci = (uint64_t)targ_ofs;
        cur.seg[0].mismatches = (uint8_t)m;
        cur.seg[0].strand = (uint8_t)cur_strand;
        cur.seg[1].match_len = (uint16_t)(probe_len - mm_ofs[m]);
        cur.seg[1].match_loci = (uint64_t)(targ_ofs + mm_ofs[m] + gap);
        cur.seg[1].mismatches = (uint8_t)mms;
        cur.seg[1].read_ofs = (uint16_t)mm_ofs[m];
        cur.seg[1].strand = (uint8_t)cur_strand;
        cur.score = (uint16_t)score;
        cur.f_splice = 1;
      }
    }
  }
  if (cur.score == 0) return 0;
  *hit = cur;
  return 3;
}

/* ---- ExploreSpliceLeft, SfxArray.cpp:9022-9265 ------------------------------------------------------------------------------ */
static int explore_splice_left(const k4o_index* ix, char cur_strand, int max_splice_junct_len, int max_tot_mm, int core_len,
                               int probe_len, const uint8_t* probe, int64_t targ_ofs, int64_t targ_len, xhit* hit) {
  (void)targ_len;
  memset(hit, 0, sizeof(*hit));
  if ((uint64_t)targ_ofs < (uint32_t)(C_MIN_JUNCT_ALIGN_SEP + C_MIN_JUNCT_SEG_LEN)) return 0;
  xhit cur;
  memset(&cur, 0, sizeof(cur));
  if (max_tot_mm > C_MAX_JUNCT_ALIGN_MM) max_tot_mm = C_MAX_JUNCT_ALIGN_MM;
  /* the reference moves both pointers to the 3' ends and walks leftwards: x3(k) is k bases left of the last base */
  const int64_t t3 = targ_ofs + probe_len - 1;
  const int p3 = probe_len - 1;
  int mm_ofs[C_MAX_PUT_INDEL_OFSS + 8];
  int n_mm = 0;
  uint32_t idx;
  uint8_t pb = 0, tv = 0;
  for (idx = (uint32_t)core_len; idx < (uint32_t)probe_len && n_mm <= IMAX(max_tot_mm, C_MAX_JUNCT_ALIGN_MM * 5); idx++) {
    pb = probe[p3 - (int)idx] & 0x07; tv = tb(ix, t3 - idx);
    if (tv > K4O_N || pb > K4O_N) return 0;
    if (pb == tv && pb <= K4O_T) continue;
    mm_ofs[n_mm++] = (int)idx;
  }
  if (n_mm < (C_MAX_JUNCT_ALIGN_MM * 4) || C_MIN_JUNCT_SEG_LEN > (probe_len - mm_ofs[0])) { /* :9093-9107 */
    if (n_mm > max_tot_mm) return 0;
    hit->seg[0].match_len = (uint16_t)probe_len;
    hit->seg[0].match_loci = (uint64_t)targ_ofs;
    hit->seg[0].mismatches = (uint8_t)n_mm;
    hit->seg[0].strand = (uint8_t)cur_strand;
    hit->score = (uint16_t)(C_BASE_SCORE + probe_len * C_SCORE_MATCH - n_mm * C_SCORE_MISMATCH);
    return 1;
  }
  const int tot_mm = IMIN(n_mm, max_tot_mm);
  for (idx = 0; idx < (uint32_t)(C_MIN_JUNCT_ALIGN_SEP + C_MIN_JUNCT_SEG_LEN); idx++) /* :9122-9127 (index TotMM here) */
    if (tb(ix, t3 - mm_ofs[tot_mm] - idx) > K4O_N) return 0;
  for (int m = 0; m <= tot_mm && C_MIN_JUNCT_SEG_LEN < (probe_len - mm_ofs[m]); m++) {
    if (cur.score >= C_MAX_SCORE) break;
    const uint32_t max_seg_len = (uint32_t)(probe_len - mm_ofs[m]);
    const int cur_p = p3 - mm_ofs[m];             /* pCurP: walks leftwards */
    const int64_t donor = t3 - mm_ofs[m];         /* pTDonor */
    int64_t t_start = donor - C_MIN_JUNCT_ALIGN_SEP;
    const int max_hash_diff = 4 * (max_tot_mm - m);
    int probe_hash = 100000;
    for (idx = 0; idx < max_seg_len; idx++) probe_hash += probe[cur_p - (int)idx] & 0x07;
    const int min_hash = probe_hash - max_hash_diff, max_hash = probe_hash + max_hash_diff;
    int targ_hash = 100000;
    int64_t t_end = t_start;
    for (idx = 0; idx < max_seg_len - 1; idx++) {
      if (tb(ix, t_end) > K4O_N) break;
      targ_hash += tb(ix, t_end--);
    }
    if (idx < (max_seg_len - 1)) break;
    for (int gap = C_MIN_JUNCT_ALIGN_SEP; gap < max_splice_junct_len - (int)max_seg_len; gap++, t_start--, t_end--) {
      if ((tv = tb(ix, t_end)) > K4O_N) break;
      targ_hash += tv;
      if (targ_hash < min_hash || targ_hash > max_hash) {
        targ_hash -= tb(ix, t_start);
        continue;
      }
      targ_hash -= tb(ix, t_start);
      const uint32_t tmp_targ_len = (uint32_t)(targ_ofs - gap);
      if (tmp_targ_len < 1) break;
      int mms = 0;
      for (idx = 0; idx < max_seg_len && (m + mms) < max_tot_mm; idx++) {
        pb = probe[cur_p - (int)idx] & 0x07; tv = tb(ix, t_start - idx);
        if (pb > K4O_N || tv > K4O_N) break;
        if (pb == tv && pb <= K4O_T) continue;
        mms += 1;
      }
      if (idx != max_seg_len) {
        if (pb > K4O_N || tv > K4O_N) break;
        continue;
      }
      uint32_t score = (uint32_t)(C_BASE_SCORE + probe_len * C_SCORE_MATCH - (((m + mms) * C_SCORE_MISMATCH) + ((gap / 1000) * C_SPLICE_LEN)));
      const uint8_t a0 = tb(ix, donor), a1 = tb(ix, donor - 1), d0 = tb(ix, t_start + 1), d1 = tb(ix, t_start + 2);
      const int gt_ag = d0 == K4O_G && d1 == K4O_T && a0 == K4O_G && a1 == K4O_A;
      const int ct_ac = d0 == K4O_C && d1 == K4O_T && a0 == K4O_C && a1 == K4O_A;
      if (cur_strand == '+') {
        if (gt_ag) score += C_SPLICE_DONOR_ACCEPT; else if (ct_ac) score += C_SPLICE_DONOR_ACCEPT / 2;
      } else {
        if (ct_ac) score += C_SPLICE_DONOR_ACCEPT; else if (gt_ag) score += C_SPLICE_DONOR_ACCEPT / 2;
      }
      if (score > cur.score) {
        memset(cur.seg, 0, sizeof(cur.seg));
        cur.seg[0].match_len = (uint16_t)(probe_len - mm_ofs[m]);
        cur.seg[0].match_loci = (uint64_t)(targ_ofs - gap);
        cur.seg[0].mismatches = (uint8_t)mms;
        cur.seg[0].strand = (uint8_t)cur_strand;
        cur.seg[1].match_len = (uint16_t)mm_ofs[m];
        cur.seg[1].match_loci = cur.seg[0].match_loci + cur.seg[0].match_len + (uint64_t)gap;
        cur.seg[1].mismatches = (uint8_t)m;
        cur.seg[1].read_ofs = cur.seg[0].match_len;
        cur.seg[1].strand = (uint8_t)cur_strand;
        cur.score = (uint16_t)score;
        cur.f_splice = 1;
      }
    }
  }
  if (cur.score == 0) return 0;
  *hit = cur;
  return 3;
}

/* pHits[0] as the two locate functions leave it -> the flat records */
static void flatten(const xhit* x, k4o_hit* hit, k4o_seg2* seg2) {
  memset(hit, 0, sizeof(*hit));
  memset(seg2, 0, sizeof(*seg2));
  hit->chrom_id = x->seg[0].chrom_id;
  hit->match_loci = (uint32_t)x->seg[0].match_loci;
  hit->match_len = x->seg[0].match_len;
  hit->strand = x->seg[0].strand;
  hit->mismatches = x->seg[0].mismatches;
  hit->ext = (x->f_indel ? K4O_EXT_INDEL : 0) | (x->f_insert ? K4O_EXT_INSERT : 0) | (x->f_splice ? K4O_EXT_SPLICE : 0);
  if (!(x->f_indel || x->f_splice)) return; /* a one-segment result: Seg[1] is all zero, Score is of no consequence */
  seg2->chrom_id = x->seg[1].chrom_id;
  seg2->match_loci = (uint32_t)x->seg[1].match_loci;
  seg2->match_len = x->seg[1].match_len;
  seg2->read_ofs = x->seg[1].read_ofs;
  seg2->mismatches = x->seg[1].mismatches;
  seg2->score = x->score;
}

/* ---- LocateInDels (:7526-7832) and LocateSpliceJuncts (:7208-7523) share their frame: two cores per strand (the read's
 * first and last core_len bases), every suffix that starts with the core explored to the right / to the left ---------------- */
static int locate_two_seg(const k4o_index* ix, int splice, int limit_len, int max_tot_mm, int core_len, int strand, int* p_inst,
                          int* p_low, int* p_nxt, uint8_t* probe, int probe_len, int max_hits, k4o_hit* hit, k4o_seg2* seg2,
                          int* p_score, k4o_counters* ctr) {
  if (ix->n == 0) return -1;
  const int64_t n = (int64_t)ix->n;
  const int max_iter = ix->max_iter;
  if (splice && max_tot_mm > C_MAX_JUNCT_ALIGN_MM) max_tot_mm = C_MAX_JUNCT_ALIGN_MM; /* :7271 */
  *p_inst = 0; *p_low = 0; *p_nxt = 0;
  xhit best; /* pHits[0]: with MaxHits 1 no other slot is ever written (a tie only counts) */
  memset(&best, 0, sizeof(best));
  memset(hit, 0, sizeof(*hit));
  memset(seg2, 0, sizeof(*seg2));
  int best_inst = 0;
  char cur_strand = '+';
  if (strand == K4O_STRAND_CRICK) { k4o_revcomp(probe, probe_len); cur_strand = '-'; }
  do {
    for (int phase = 0; phase < 2; phase++) {
      const int ofs = phase == 0 ? 0 : probe_len - core_len;
      int64_t t = k4o_locate_first_exact(ix, probe + ofs, core_len, 0, n - 1, ctr);
      if (t == 0) continue;
      t -= 1;
      int iter = 0, first = 1;
      while (!max_iter || iter < max_iter) {
        if (!first) {
          if (splice) { /* :7323 */
            if (t + 1 >= n || (k4o_sa_at(ix, t + 1) + (phase == 0 ? probe_len : core_len)) >= n) break;
          } else if (t + 1 >= n || (k4o_sa_at(ix, t + 1) + core_len) > n) /* :7635 */
            break;
          if (k4oi_cmp_probe_targ(probe + ofs, ix->seq + k4o_sa_at(ix, t + 1), core_len) != 0) break;
          t += 1;
        }
        first = 0;
        const int64_t pos = k4o_sa_at(ix, t);
        if (pos < (int64_t)(uint32_t)ofs) continue;
        const int64_t left = pos - ofs;
        const k4o_entry* e;
        if (splice) { /* :7376-7383 */
          if ((left + probe_len) >= n) continue;
          e = k4oi_map_chunk_hit2entry(ix, (uint64_t)pos);
          if (e == NULL) continue;
          if (left < (int64_t)e->start_ofs || (left + probe_len) > (int64_t)e->end_ofs) continue;
        } else { /* :7685-7693 */
          e = k4oi_map_chunk_hit2entry(ix, (uint64_t)pos);
          if (e == NULL) continue;
          if (left < (int64_t)e->start_ofs || (left + probe_len - 1) > (int64_t)e->end_ofs) continue;
          if ((left + probe_len) > n) continue;
        }
        iter++;
        if (ctr) ctr->n_cand++;
        xhit x;
        int r = 0;
        if (!splice)
          r = phase == 0 ? explore_indel_right(ix, cur_strand, limit_len, max_tot_mm, probe_len, probe, e, left, &x)
                         : explore_indel_left(ix, cur_strand, limit_len, max_tot_mm, probe_len, probe, e, left, &x);
        else if (phase == 0) { /* :7392-7426 */
          int lim = (int)(n - left);
          if (lim > (C_MIN_JUNCT_ALIGN_SEP + C_MIN_JUNCT_SEG_LEN)) {
            lim -= (C_MIN_JUNCT_ALIGN_SEP + C_MIN_JUNCT_SEG_LEN);
            if (lim > limit_len) lim = limit_len;
            r = explore_splice_right(ix, cur_strand, lim, max_tot_mm, core_len, probe_len, probe, left, n, &x);
          }
        } else if ((uint64_t)left >= (uint32_t)(ofs + C_MIN_JUNCT_SEG_LEN)) { /* :7429-7461 */
          int lim = IMIN((int32_t)left, (int32_t)limit_len);
          if (lim >= (C_MIN_JUNCT_ALIGN_SEP + C_MIN_JUNCT_SEG_LEN)) {
            lim -= C_MIN_JUNCT_SEG_LEN;
            r = explore_splice_left(ix, cur_strand, lim, max_tot_mm, core_len, probe_len, probe, left, n, &x);
          }
        }
        if (r > 0 && x.score >= best.score) {
          if (x.score == best.score) {
            if (best.seg[0].match_loci == x.seg[0].match_loci) continue;
            if (++best_inst > max_hits) continue;
          } else
            best_inst = 0;
          /* pHits[BestScoreInstances++] = hit: with max_hits == 1 the index is always 0 here */
          best = x;
          best_inst++;
        }
      }
      if (best_inst >= 1 && best.score >= C_MAX_SCORE) { strand = 3; break; }
    }
    if (cur_strand == '+' && strand == K4O_STRAND_BOTH) {
      k4o_revcomp(probe, probe_len);
      cur_strand = '-';
      strand = K4O_STRAND_CRICK;
    } else
      strand = 3;
  } while (!(best_inst >= 1 && best.score >= C_MAX_SCORE) && strand != 3);
  if (cur_strand == '-') k4o_revcomp(probe, probe_len);
  if (best_inst == 0) return K4O_HR_NONE;
  if (best.score > C_MAX_SCORE) best.score = C_MAX_SCORE;
  /* concat offsets -> chromosome + locus, :7501-7516 / :7800-7825 (one hit: max_hits is 1) */
  if (!splice && best.seg[0].strand != '-' && best.seg[0].strand != '+') best.seg[0].strand = '?';
  const k4o_entry* e0 = k4oi_map_chunk_hit2entry(ix, best.seg[0].match_loci);
  if (e0 == NULL) { flatten(&best, hit, seg2); return K4O_HR_NONE; }
  if (!splice) {
    const k4o_entry* e1 = k4oi_map_chunk_hit2entry(ix, best.seg[1].match_loci);
    if (e1 == NULL || e0->entry_id != e1->entry_id) { flatten(&best, hit, seg2); return K4O_HR_NONE; } /* :7811-7814 */
    best.seg[0].chrom_id = e0->entry_id;
    best.seg[0].match_loci -= e0->start_ofs;
    if (best.seg[1].match_loci > 0) {
      best.seg[1].chrom_id = e1->entry_id;
      best.seg[1].match_loci -= e1->start_ofs;
    }
  } else {
    best.seg[0].chrom_id = e0->entry_id;
    best.seg[0].match_loci -= e0->start_ofs;
    if (best.seg[1].match_loci > 0) {
      const k4o_entry* e1 = k4oi_map_chunk_hit2entry(ix, best.seg[1].match_loci);
      if (e1 == NULL) { flatten(&best, hit, seg2); return K4O_HR_NONE; }
      best.seg[1].chrom_id = e1->entry_id;
      best.seg[1].match_loci -= e1->start_ofs;
    }
  }
  flatten(&best, hit, seg2);
  if (p_score) *p_score = best.score;
  *p_inst = splice ? best_inst : IMIN(max_hits, best_inst);
  *p_low = best.seg[0].mismatches + best.seg[1].mismatches;
  *p_nxt = *p_low + 2;
  return best_inst <= max_hits ? K4O_HR_HITS : K4O_HR_NONE;
}

int k4o_locate_indels(const k4o_index* ix, int micro_indel_len, int max_tot_mm, int core_len, int strand, int* inst, int* low,
                      int* nxt, uint8_t* probe, int probe_len, int max_hits, k4o_hit* hit, k4o_seg2* seg2, int* score,
                      k4o_counters* ctr) {
  return locate_two_seg(ix, 0, micro_indel_len, max_tot_mm, core_len, strand, inst, low, nxt, probe, probe_len, max_hits, hit, seg2,
                        score, ctr);
}
int k4o_locate_splice_juncts(const k4o_index* ix, int max_splice_junct_len, int max_tot_mm, int core_len, int strand, int* inst,
                             int* low, int* nxt, uint8_t* probe, int probe_len, int max_hits, k4o_hit* hit, k4o_seg2* seg2,
                             int* score, k4o_counters* ctr) {
  return locate_two_seg(ix, 1, max_splice_junct_len, max_tot_mm, core_len, strand, inst, low, nxt, probe, probe_len, max_hits, hit,
                        seg2, score, ctr);
}

/* ---- AlignReads with every argument, SfxArray.cpp:7838-7933 ------------------------------------------------------------------ */
int k4o_align_reads_ext(const k4o_index* ix, const k4o_ext_params* ext, int tot_mm, int core_len, int core_delta, int max_slides,
                        int min_core_len, int mm_delta, int strand, int* inst, int* low, int* nxt, uint8_t* probe,
                        int probe_len, int max_hits, k4o_hit* hits, k4o_seg2* seg2, k4o_counters* ctr) {
  k4o_seg2 dummy;
  if (!seg2) seg2 = &dummy;
  memset(seg2, 0, sizeof(*seg2));
  int rslt = k4o_align_reads(ix, tot_mm, core_len, core_delta, max_slides, min_core_len, mm_delta, strand, inst, low, nxt, probe,
                             probe_len, max_hits, hits, ctr);
  if (rslt != 0) return rslt;
  int score = 0;
  if (ext->micro_indel_len > 0) { /* :7894-7906 */
    rslt = k4o_locate_indels(ix, ext->micro_indel_len, tot_mm > C_MAX_MICRO_INDEL_MM ? C_MAX_MICRO_INDEL_MM : tot_mm, core_len,
                             strand, inst, low, nxt, probe, probe_len, 1, &hits[0], seg2, &score, ctr);
    if (rslt != 0) return rslt;
  }
  if (ext->max_splice_junct_len > 0) { /* :7908-7921 */
    rslt = k4o_locate_splice_juncts(ix, ext->max_splice_junct_len, tot_mm > C_MAX_JUNCT_ALIGN_MM ? C_MAX_JUNCT_ALIGN_MM : tot_mm,
                                    core_len, strand, inst, low, nxt, probe, probe_len, 1, &hits[0], seg2, &score, ctr);
    if (rslt != 0) return rslt;
  }
  if (ext->min_chimeric_len > 0) { /* :7923-7930 */
    if (max_slides <= 1) return ERR_PARAMS; /* the reference divides by MaxNumCoreSlides-1 */
    const int cl = IMAX(min_core_len, probe_len / (tot_mm + 4));
    const int cd = IMAX(probe_len / (max_slides - 1), cl);
    rslt = k4o_locate_core_multiples_chimeric(ix, ext->min_chimeric_len, tot_mm, cl, cd, max_slides, mm_delta, strand, inst, low,
                                              nxt, probe, probe_len, max_hits, hits, ctr);
    /* Slot 0 may still hold what the last two-segment phase left there when it gave up over too many equally good loci:
     * that phase's instance count is carried in, and with more than MaxHits exact instances this pass returns
     * eHRHitInsts at once (:5890) without touching the slot.  Whenever the pass does store a hit it clears both segments
     * first (:6129), so the second segment is gone unless slot 0 still is that two-segment record. */
    if (!(hits[0].ext & (K4O_EXT_INDEL | K4O_EXT_SPLICE))) memset(seg2, 0, sizeof(*seg2));
    return rslt;
  }
  return 0;
}

/* ---- batches ----------------------------------------------------------------------------------------------------------- */
typedef struct {
  const k4o_index* ix; const k4o_kalign_params* kp; k4o_ext_params ext;
  int raw, tot_mm, core_len, core_delta, slides, min_core_len, mm_delta, strand, max_hits, mcl, spm;
  int64_t n; const uint8_t* reads; const uint64_t* offs; const uint32_t* lens;
  int32_t *rslt, *inst, *low, *nxt; k4o_read_result* out; k4o_hit* hits; k4o_seg2* seg2;
  int64_t next; pthread_mutex_t mtx; k4o_counters ctr;
} xjob;

/* CKAligner::AlignRead (KAligner.cpp:9583-10105) around k4o_align_reads_ext: SE, MLMode default / PE classification */
static void align_read_ext(const xjob* j, const uint8_t* read, int read_len, uint8_t* scratch, k4o_read_result* out, k4o_hit* hits,
                           k4o_seg2* seg2, k4o_counters* ctr) {
  const k4o_kalign_params* kp = j->kp;
  memset(out, 0, sizeof(*out));
  memset(seg2, 0, sizeof(*seg2));
  out->nar = K4O_NAR_NOHIT;
  int max_ns_seq = 0, ns = 0, i;
  if (kp->max_ns) max_ns_seq = IMAX((read_len * kp->max_ns) / 100, kp->max_ns);
  for (i = 0; i < read_len; i++) {
    const uint8_t b = read[i] & 0x07;
    scratch[i] = b;
    if (b > K4O_N) break;
    if (b == K4O_N && ++ns > max_ns_seq) break;
  }
  const int max_ml = kp->max_ml < 1 ? 1 : kp->max_ml;
  memset(hits, 0, sizeof(k4o_hit) * (size_t)max_ml);
  if (i != read_len) { out->nar = K4O_NAR_NS; out->hit_rslt = K4O_HR_SEQERRS; return; }
  int tot_mm, core_len, core_delta, slides;
  k4o_read_params(kp, j->mcl, j->spm, read_len, &tot_mm, &core_len, &core_delta, &slides);
  int inst = 0, low = 0, nxt = 0;
  int r = k4o_align_reads_ext(j->ix, &j->ext, tot_mm, core_len, core_delta, slides, j->mcl, kp->min_edit_dist, kp->strand, &inst,
                              &low, &nxt, scratch, read_len, max_ml, hits, seg2, ctr);
  if (inst > max_ml) inst = max_ml + 1;
  if (kp->pe_mode >= 3 && r == K4O_HR_HITINSTS) { inst = max_ml; r = K4O_HR_HITS; }
  /* :9866-9887 trimming belongs to chimeric hits only; a two-segment hit is never chimeric */
  if (r == K4O_HR_HITINSTS || r == K4O_HR_HITS)
    for (int q = 0; q < IMIN(inst, max_ml); q++) {
      if (!(hits[q].ext & K4O_EXT_CHIMERIC)) hits[q].ext &= ~0xFFFFFFu;
      if (hits[q].ext & (K4O_EXT_INDEL | K4O_EXT_SPLICE)) hits[q].ext &= ~K4O_EXT_CHIMERIC;
    }
  out->hit_rslt = r; out->inst = inst; out->low_mm = low; out->nxt_mm = nxt;
  switch (r) {
    case K4O_HR_NONE: out->nar = K4O_NAR_NOHIT; out->low_mm = 0; out->inst = 0; out->nxt_mm = 0; break;
    case K4O_HR_HITS:
      if (kp->pe_mode >= 2) { out->nar = K4O_NAR_ACCEPTED; out->num_hits = IMIN(inst, max_ml); }
      else if (!kp->pe_mode || inst == 1) { out->nar = K4O_NAR_ACCEPTED; out->num_hits = 1; }
      else { out->nar = K4O_NAR_MULTIALIGN; out->num_hits = inst; }
      break;
    case K4O_HR_MMDELTA: out->nar = K4O_NAR_MMDELTA; break;
    case K4O_HR_HITINSTS: out->nar = K4O_NAR_MULTIALIGN; break;
    default: break;
  }
  /* what the caller never looks at is cleared so that batches can be compared byte for byte */
  if (!(r == K4O_HR_HITS || r == K4O_HR_MMDELTA || r == K4O_HR_HITINSTS)) { memset(hits, 0, sizeof(k4o_hit) * (size_t)max_ml); memset(seg2, 0, sizeof(*seg2)); }
  else for (int q = IMIN(inst, max_ml); q < max_ml; q++) memset(&hits[q], 0, sizeof(k4o_hit));
}

int k4oi_align_read_ext(const k4o_index* ix, const k4o_kalign_params* kp, int mcl, int spm, const uint8_t* read, int read_len,
                        uint8_t* scratch, k4o_read_result* out, k4o_hit* hits) {
  xjob j;
  memset(&j, 0, sizeof(j));
  j.ix = ix; j.kp = kp; j.mcl = mcl; j.spm = spm;
  j.ext.min_chimeric_len = kp->min_chimeric_len;  /* (two-segment phases are not part of the paired-end build: seg2 is dropped) */
  k4o_seg2 s2;
  align_read_ext(&j, read, read_len, scratch, out, hits, &s2, NULL);
  return out->hit_rslt;
}

/* ---- AlignPairedRead with chimeric trimming, SfxArray.cpp:8571-8767 --------------------------------------------------------- */
static void store_rescued(k4o_hit* out, uint32_t chrom_id, uint32_t loci, int read_len, int antisense, uint32_t t5, uint32_t t3,
                          uint32_t mms, int chimeric) {
  memset(out, 0, sizeof(*out));
  out->chrom_id = chrom_id;
  out->match_loci = loci;
  out->match_len = (uint16_t)read_len;
  out->strand = antisense ? '-' : '+';
  out->mismatches = (uint8_t)mms;
  const uint32_t tl = antisense ? t3 : t5, tr = antisense ? t5 : t3; /* :8702-8711 */
  out->ext = (tl & 0xFFF) | ((tr & 0xFFF) << 12) | (chimeric ? K4O_EXT_CHIMERIC : 0);
}

int k4o_align_paired_read_x(const k4o_index* ix, int b3prime, int antisense, uint32_t chrom_id, uint32_t start_loci,
                            uint32_t end_loci, int min_insert, int max_insert, int max_allowed_mm, int read_len,
                            int min_chimeric_len, int core_len, int core_delta, const uint8_t* read, k4o_hit* out) {
  memset(out, 0, sizeof(*out));
  if (chrom_id < 1 || chrom_id > ix->n_entries) return -1;
  const k4o_entry* e = &ix->entries[chrom_id - 1];
  const uint32_t chrom_len = e->seq_len;
  if (chrom_len == 0) return -1;
  if (start_loci >= end_loci || end_loci >= chrom_len) return -1;
  int min_put_len;
  if (core_len > 0 && min_chimeric_len >= 15 && min_chimeric_len <= 99) { /* :8610-8621 */
    min_put_len = ((read_len * min_chimeric_len) + 50) / 100;
    if (core_len > min_put_len) core_len = min_put_len;
  } else
    min_put_len = read_len;
  if (min_put_len == read_len) { min_chimeric_len = 0; core_len = 0; }
  if (min_insert > max_insert) return 0;
  if (min_insert < read_len) { max_insert += read_len - min_insert; min_insert = read_len; }
  uint32_t sp, ep;
  if (b3prime) {
    if ((uint32_t)(start_loci + min_insert) >= chrom_len) return 0;
    sp = start_loci + min_insert - read_len;
    const uint32_t a = chrom_len - read_len, b = (uint32_t)(start_loci + max_insert - read_len);
    ep = a < b ? a : b;
  } else {
    if (end_loci < (uint32_t)min_insert) return 0;
    sp = end_loci <= (uint32_t)max_insert ? 0 : end_loci - max_insert;
    ep = end_loci - min_insert;
  }
  uint8_t* rs = (uint8_t*)malloc((size_t)read_len + 1);
  memcpy(rs, read, (size_t)read_len);
  rs[read_len] = K4O_EOS;
  if (antisense) k4o_revcomp(rs, read_len);
  const uint8_t* chrom = ix->seq + e->start_ofs;
  uint32_t prev_best = (uint32_t)max_allowed_mm + 1;
  if ((ep - sp) >= 1000) { /* :8685-8726; with core_len 0 the reference never returns from here (see k4o_align_paired_read) */
    if (core_len <= 0 || core_delta <= 0) { free(rs); return -3; }
    const int64_t n = (int64_t)ix->n;
    for (uint32_t ofs = 0; (int)ofs + core_len <= read_len; ofs += (uint32_t)core_delta) {
      int64_t t = k4o_locate_first_exact(ix, rs + ofs, core_len, 0, n - 1, NULL); /* IterateExactsRange, :3461-3553 */
      if (t == 0) continue;
      for (t -= 1; t < n; t++) { /* every suffix that starts with the core, in suffix-array order */
        const int64_t pos = k4o_sa_at(ix, t);
        if (k4oi_cmp_probe_targ(rs + ofs, ix->seq + pos, core_len) != 0) break;
        const k4o_entry* he = k4oi_map_chunk_hit2entry(ix, (uint64_t)pos);
        if (!he || he->entry_id != chrom_id) continue;
        const uint32_t hit_loci = (uint32_t)((uint64_t)pos - he->start_ofs);
        if (hit_loci < sp || hit_loci > ep) continue;
        if (ofs > hit_loci || (hit_loci + (uint32_t)read_len - ofs) >= chrom_len) continue; /* :8693 */
        uint32_t tl, t5, t3, mms;
        const int r = k4o_adaptive_trim((uint32_t)read_len, rs, chrom + (hit_loci - ofs), (uint32_t)min_put_len, (uint32_t)max_allowed_mm, 3,
                                        &tl, &t5, &t3, &mms);
        if (r > min_put_len || (r == min_put_len && mms < prev_best)) {
          prev_best = mms;
          min_put_len = r;
          store_rescued(out, chrom_id, hit_loci - ofs, read_len, antisense, t5, t3, mms, min_put_len != read_len);
        }
      }
    }
  } else
    for (uint32_t loci = sp; loci <= ep; loci++) { /* :8731-8766 */
      uint32_t tl, t5, t3, mms;
      const int r = k4o_adaptive_trim((uint32_t)read_len, rs, chrom + loci, (uint32_t)min_put_len, (uint32_t)max_allowed_mm, 3, &tl, &t5, &t3, &mms);
      if (r > min_put_len || (r == min_put_len && mms < prev_best)) {
        prev_best = mms;
        min_put_len = r;
        store_rescued(out, chrom_id, loci, read_len, antisense, t5, t3, mms, min_put_len != read_len);
        if (min_put_len == read_len && mms == 0) break;
      }
      if (loci == 0xFFFFFFFFu) break;
    }
  free(rs);
  return prev_best <= (uint32_t)max_allowed_mm ? 1 : 0;
}

static void* xworker(void* arg) {
  xjob* j = (xjob*)arg;
  k4o_counters c = { 0, 0, 0 };
  uint8_t* scratch = (uint8_t*)malloc(1 << 16);
  for (;;) {
    pthread_mutex_lock(&j->mtx);
    const int64_t b = j->next;
    j->next += 64;
    pthread_mutex_unlock(&j->mtx);
    if (b >= j->n) break;
    const int64_t e = b + 64 < j->n ? b + 64 : j->n;
    for (int64_t i = b; i < e; i++) {
      const uint8_t* rd = j->reads + j->offs[i];
      const int len = (int)j->lens[i];
      k4o_seg2 s2;
      if (j->raw) {
        const int mh = j->max_hits;
        k4o_hit* h = j->hits + (size_t)i * mh;
        memset(h, 0, sizeof(k4o_hit) * (size_t)mh);
        memcpy(scratch, rd, (size_t)len);
        int in = 0, lo = 0, nx = 0;
        const int r = k4o_align_reads_ext(j->ix, &j->ext, j->tot_mm, j->core_len, j->core_delta, j->slides, j->min_core_len,
                                          j->mm_delta, j->strand, &in, &lo, &nx, scratch, len, mh, h, &s2, &c);
        j->rslt[i] = r; j->inst[i] = in; j->low[i] = lo; j->nxt[i] = nx;
        if (!(r == K4O_HR_HITS || r == K4O_HR_MMDELTA || r == K4O_HR_HITINSTS)) { memset(h, 0, sizeof(k4o_hit) * (size_t)mh); memset(&s2, 0, sizeof(s2)); }
        else for (int q = IMIN(in, mh); q < mh; q++) memset(&h[q], 0, sizeof(k4o_hit));
      } else {
        const int mh = j->kp->max_ml < 1 ? 1 : j->kp->max_ml;
        align_read_ext(j, rd, len, scratch, &j->out[i], j->hits + (size_t)i * mh, &s2, &c);
      }
      if (j->seg2) j->seg2[i] = s2;
    }
  }
  free(scratch);
  pthread_mutex_lock(&j->mtx);
  j->ctr.n_lookup += c.n_lookup; j->ctr.n_probe += c.n_probe; j->ctr.n_cand += c.n_cand;
  pthread_mutex_unlock(&j->mtx);
  return NULL;
}

static void xrun(xjob* j, int nthreads, k4o_counters* ctr) {
  pthread_mutex_init(&j->mtx, NULL);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256];
  for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, xworker, j);
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  pthread_mutex_destroy(&j->mtx);
  if (ctr) *ctr = j->ctr;
}

int k4o_align_reads_ext_batch(const k4o_index* ix, const k4o_ext_params* ext, int tot_mm, int core_len, int core_delta,
                              int max_slides, int min_core_len, int mm_delta, int strand, int max_hits, int64_t n_reads,
                              const uint8_t* reads, const uint64_t* offs, const uint32_t* lens, int32_t* rslt, int32_t* inst,
                              int32_t* low, int32_t* nxt, k4o_hit* hits, k4o_seg2* seg2, int nthreads, k4o_counters* ctr) {
  xjob j;
  memset(&j, 0, sizeof(j));
  j.ix = ix; j.ext = *ext; j.raw = 1; j.tot_mm = tot_mm; j.core_len = core_len; j.core_delta = core_delta; j.slides = max_slides;
  j.min_core_len = min_core_len; j.mm_delta = mm_delta; j.strand = strand; j.max_hits = max_hits; j.n = n_reads; j.reads = reads;
  j.offs = offs; j.lens = lens; j.rslt = rslt; j.inst = inst; j.low = low; j.nxt = nxt; j.hits = hits; j.seg2 = seg2;
  xrun(&j, nthreads, ctr);
  return 0;
}

int k4o_align_ext_batch(const k4o_index* ix, const k4o_kalign_params* kp, int64_t n_reads, const uint8_t* reads,
                        const uint64_t* offs, const uint32_t* lens, k4o_read_result* out, k4o_hit* hits, k4o_seg2* seg2,
                        int nthreads, k4o_counters* ctr) {
  xjob j;
  memset(&j, 0, sizeof(j));
  j.ix = ix; j.kp = kp; j.n = n_reads; j.reads = reads; j.offs = offs; j.lens = lens; j.out = out; j.hits = hits; j.seg2 = seg2;
  j.ext.min_chimeric_len = kp->min_chimeric_len; j.ext.micro_indel_len = kp->micro_indel_len;
  j.ext.max_splice_junct_len = kp->max_splice_junct_len;
  j.mcl = kp->min_core_len; j.spm = kp->max_num_slides;
  if (j.mcl <= 0 || j.spm <= 0) {
    int s2, m2 = k4o_min_core_len(ix, kp->pmode, &s2);
    if (j.mcl <= 0) j.mcl = m2;
    if (j.spm <= 0) j.spm = s2;
  }
  xrun(&j, nthreads, ctr);
  return 0;
}

/* ---- the Adj* helpers of CKAligner, KAligner.cpp:1633-1655 ---------------------------------------------------------------- */
static uint32_t trim_l(const k4o_hit* h) { return h->ext & 0xFFF; }
static uint32_t trim_r(const k4o_hit* h) { return (h->ext >> 12) & 0xFFF; }
static uint32_t adj_start(uint32_t loci, uint8_t strand, uint32_t tl, uint32_t tr) { return loci + (strand == '+' ? tl : tr); }
static uint32_t adj_end(uint32_t loci, uint32_t len, uint8_t strand, uint32_t tl, uint32_t tr) {
  return loci + (len - (strand == '+' ? tr : tl) - 1);
}

/* ---- AutoTrimFlanks, KAligner.cpp:1714-1917 (base space).  A read it eliminates gets NumHits 0 and NAR eNARTrim. ------------ */
int64_t k4o_auto_trim_flanks(const k4o_index* ix, int min_flank_exacts, int pe, int64_t n_reads, const uint8_t* reads,
                             const uint64_t* offs, const uint32_t* lens, int max_ml, k4o_read_result* rr, k4o_hit* hits,
                             const k4o_seg2* seg2) {
  (void)seg2;
  if (min_flank_exacts <= 0) return 0;
  int64_t elim = 0;
  uint8_t* rs = (uint8_t*)malloc(1 << 16);
  uint8_t* ts = (uint8_t*)malloc(1 << 16);
  for (int64_t i = 0; i < n_reads; i++) {
    k4o_hit* h = &hits[(size_t)i * max_ml];
    if (rr[i].nar != K4O_NAR_ACCEPTED || (h->ext & (K4O_EXT_INDEL | K4O_EXT_SPLICE | K4O_EXT_CHIMERIC))) continue; /* :1748 */
    uint32_t match_len = h->match_len;
    const uint32_t read_len = lens[i];
    if (match_len != read_len) { rr[i].num_hits = 0; elim++; continue; } /* :1751-1759 (NAR is left as it is) */
    int min_trimmed = (int)(match_len + 1) / 2;
    if (min_trimmed < 15) min_trimmed = 15;
    for (uint32_t q = 0; q < match_len; q++) rs[q] = reads[offs[i] + q] & 0x07;
    const k4o_entry* e = &ix->entries[h->chrom_id - 1];
    for (uint32_t q = 0; q < match_len; q++) /* GetSeq, SfxArray.cpp:2396-2424 */
      ts[q] = (h->match_loci + q) < e->seq_len ? (uint8_t)(ix->seq[e->start_ofs + h->match_loci + q] & 0x0f) : 0;
    if (h->strand == '-') k4o_revcomp(ts, (int)match_len);
    int exact = 0, trim_mm = 0;
    uint32_t idx;
    const int pemin5 = !pe ? (int)match_len : (int)match_len / 3;
    for (idx = 0; idx <= (match_len - (uint32_t)min_trimmed) && idx < (uint32_t)pemin5; idx++) { /* :1817-1828 */
      if (rs[idx] != ts[idx]) { exact = 0; trim_mm += 1; continue; }
      exact += 1;
      if (exact == min_flank_exacts) break;
    }
    if (!pe && ((idx + (uint32_t)min_trimmed) > match_len || exact < min_flank_exacts)) { /* :1830-1843 */
      rr[i].num_hits = 0; rr[i].nar = K4O_NAR_TRIM; elim++;
      continue;
    }
    const int left_ofs = (int)idx - (min_flank_exacts - 1);
    exact = 0;
    const int pemin3 = !pe ? 0 : (int)(match_len * 2) / 3;
    for (idx = match_len - 1; idx >= (uint32_t)(left_ofs + min_trimmed) && idx > (uint32_t)pemin3; idx--) { /* :1855-1866 */
      if (rs[idx] != ts[idx]) { exact = 0; trim_mm += 1; continue; }
      exact += 1;
      if (exact == min_flank_exacts) break;
    }
    if (!pe && (exact != min_flank_exacts || idx < (uint32_t)(left_ofs + min_trimmed))) { /* :1868-1881 */
      rr[i].num_hits = 0; rr[i].nar = K4O_NAR_TRIM; elim++;
      continue;
    }
    const int right_ofs = (int)idx + min_flank_exacts;
    const uint32_t tl = (uint32_t)left_ofs, tr = match_len - (uint32_t)right_ofs;
    h->ext = (h->ext & ~0xFFFFFFu) | (tl & 0xFFF) | ((tr & 0xFFF) << 12);
    (void)trim_mm; /* TrimMismatches (:1895) is not part of the flat record: nothing downstream of SAM reads it */
  }
  free(rs);
  free(ts);
  return elim;
}

/* ---- RemoveOrphanSpliceJuncts / RemoveOrphanMicroInDels, KAligner.cpp:2406-2594 ----------------------------------------------- */
typedef struct { int64_t read; uint32_t chrom, starts, ends; } junct;
static int junct_cmp(const void* a, const void* b) { /* SortSegJuncts, KAligner.cpp:11104-11124 */
  const junct* x = (const junct*)a; const junct* y = (const junct*)b;
  if (x->chrom != y->chrom) return x->chrom < y->chrom ? -1 : 1;
  if (x->starts != y->starts) return x->starts < y->starts ? -1 : 1;
  if (x->ends != y->ends) return x->ends < y->ends ? -1 : 1;
  return 0;
}

int64_t k4o_remove_orphan_juncts(uint32_t which, int64_t n_reads, int max_ml, k4o_read_result* rr, k4o_hit* hits,
                                 const k4o_seg2* seg2) {
  int64_t nj = 0;
  for (int64_t i = 0; i < n_reads; i++)
    if (rr[i].nar == K4O_NAR_ACCEPTED && (hits[(size_t)i * max_ml].ext & which)) nj++;
  if (nj == 0) return 0;
  junct* js = (junct*)malloc(sizeof(junct) * (size_t)nj);
  int64_t k = 0;
  for (int64_t i = 0; i < n_reads; i++) {
    k4o_hit* h = &hits[(size_t)i * max_ml];
    if (rr[i].nar != K4O_NAR_ACCEPTED || !(h->ext & which)) continue;
    js[k].read = i;
    js[k].chrom = h->chrom_id;
    js[k].starts = adj_end(h->match_loci, h->match_len, h->strand, trim_l(h), trim_r(h));
    js[k].ends = adj_start(seg2[i].match_loci, h->strand, 0, 0); /* Seg[1] carries no trimming and the hit's strand */
    k++;
  }
  int64_t removed = 0;
  if (nj > 1) {
    qsort(js, (size_t)nj, sizeof(junct), junct_cmp);
    for (k = 0; k + 1 < nj; k++) { /* :2456-2465 neighbours in the sorted order only; 32-bit unsigned arithmetic */
      const junct *a = &js[k], *b = &js[k + 1];
      if (a->chrom == b->chrom && (a->starts <= (uint32_t)(b->starts + 3) && a->starts >= (uint32_t)(b->starts - 3)) &&
          (a->ends <= (uint32_t)(b->ends + 3) && a->ends >= (uint32_t)(b->ends - 3))) {
        hits[(size_t)a->read * max_ml].ext |= K4O_EXT_NONORPHAN;
        hits[(size_t)b->read * max_ml].ext |= K4O_EXT_NONORPHAN;
      }
    }
  }
  for (k = 0; k < nj; k++) { /* a lone junction (nj == 1) is an orphan too, :2482-2489 */
    const int64_t i = js[k].read;
    if (!(hits[(size_t)i * max_ml].ext & K4O_EXT_NONORPHAN)) {
      rr[i].nar = which == K4O_EXT_SPLICE ? K4O_NAR_SPLICEJCTN : K4O_NAR_MICROINDEL;
      rr[i].num_hits = 0;
      rr[i].inst = 0;
      removed++;
    }
  }
  free(js);
  return removed;
}
