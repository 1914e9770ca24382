// kit4b_amd/csrc/k4_ext.h -- the OPTIONAL phases of CSfxArray::AlignReads on gfx950; included by k4_align.hip
// in front of the general kernel, whose wave-per-read frame (K4Slow: probe in LDS, 64-way seed search, run walk with one
// suffix per lane) they run in.  A read gets here only when the standard phases found nothing (SfxArray.cpp:7894-7930):
//   microInDels       CSfxArray::LocateInDels        SfxArray.cpp:7526-7832, ExploreInDelMatchRight/Left :9277-9735
//   splice junctions  CSfxArray::LocateSpliceJuncts  SfxArray.cpp:7208-7523, ExploreSpliceRight/Left :8771-9265
//   chimeric trimming CSfxArray::LocateCoreMultiples SfxArray.cpp:6064-6189 with CSfxArray::AdaptiveTrim :5561-5795
// Division of labour: what is bound by memory latency -- the seed search, the suffix elements and core comparisons of a run,
// the entry lookups -- is spread over the 64 lanes; each candidate locus is then explored by ONE lane with the reference's
// own sequential rules (they are full of order-dependent early-outs), 64 candidates at a time; what the reference makes
// depend on the order of the candidates (best score / tie counting, the chimeric fold) is replayed in suffix order.
#pragma once

#include "k4_trim.h"

// the mismatch vector of the probe laid on [left, left+len) into this lane's column of mk (N == N is a match, :5618,5651)
K4_DEV void k4d_build_mm_vector(const K4DevIndex& ix, const K4Slow& sc, int len, uint64_t left, uint32_t* mk, int s = 0) {
  if (sc.packed && !k4d_any_exc_sup(ix, sc.sup, (int64_t)left, (int64_t)left + len)) {
    for (int c0 = 0; 32 * c0 < len; c0 += 4) {
      const int rem = len - 32 * c0;
      uint64_t rc[4];
      k4d_ref_chunks4(ix, (int64_t)left, c0, rem + (int)(left & 15) <= 128, rc);
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (32 * c < rem) mk[(c0 + c) * 64] = k4d_mm_bits((rc[c] ^ k4d_probe_chunk(sc, 32 * (c0 + c), s)) & k4d_range_mask(0, rem - 32 * c));
    }
    return;
  }
  K4Tb t;
  t.init(ix);
  for (int c = 0; 32 * c < len; c++) {
    uint32_t m = 0;
    for (int q = 0; q < 32 && 32 * c + q < len; q++)
      if ((sc.probe[(s ? sc.pstride : 0u) + 32 * c + q] & 0x0f) != t.get((int64_t)left + 32 * c + q)) m |= 1u << q;
    mk[c * 64] = m;
  }
}

// ---- the two-segment explorations: one lane, the reference's sequential rules ------------------------------------------------
#define K4X_MAX_OFSS 12  // mismatch offsets a lane records: max(MaxTotMM, 10) + 1 with MaxTotMM clamped to 2 by AlignReads
struct K4XHit {
  uint64_t l0, l1;          // Seg[0] / Seg[1].MatchLoci: concat offsets until the caller converts them
  uint32_t len0, len1, ofs1, mm0, mm1, score, fl;  // fl: 1 FlgInDel, 2 FlgInsert, 4 FlgSplice, 8 found on the '-' strand
};
K4_DEV void k4x_zero(K4XHit& h) { h.l0 = h.l1 = 0; h.len0 = h.len1 = h.ofs1 = h.mm0 = h.mm1 = h.score = h.fl = 0; }
K4_DEV void k4x_one_seg(K4XHit& h, int len, int64_t targ_ofs, int n_mm) {  // the "mismatches only" result, e.g. :9349-9356
  k4x_zero(h);
  h.len0 = (uint32_t)len; h.l0 = (uint64_t)targ_ofs; h.mm0 = (uint32_t)n_mm;
  h.score = (uint32_t)(500 + len * 3 - n_mm * 5) & 0xFFFFu;
}


// The explorations begin by listing the first mismatches of the read laid on its locus.  With the locus' packed window in
// registers that list comes from the mismatch bit vector (bit j of word j >> 5: base j differs) by find-first-set, instead of
// a walk over every base through LDS bytes and cached target words: a read of 100 bases took ~100 dependent LDS reads there.
struct K4XMask {
  uint32_t w[4];
  bool have;  // the window was clean and the probe packed (no N anywhere): w is valid and no symbol above T can turn up
};
K4_DEV int k4x_next_up(const K4XMask& m, int from, int end) {  // lowest set bit in [from, end), or end
  for (int j = from; j < end;) {
    const uint32_t v = m.w[j >> 5] >> (j & 31);
    if (v) { const int r = j + (__ffs((int)v) - 1); return r < end ? r : end; }
    j = (j | 31) + 1;
  }
  return end;
}
K4_DEV int k4x_next_down(const K4XMask& m, int from, int end) {  // highest set bit in (end, from], or end
  for (int j = from; j > end;) {
    const uint32_t v = m.w[j >> 5] << (31 - (j & 31));
    if (v) { const int r = j - __clz((int)v); return r > end ? r : end; }
    j = (j & ~31) - 1;
  }
  return end;
}

// ExploreInDelMatchRight, SfxArray.cpp:9277-9495
K4_DEV int k4x_indel_right(K4Tb& t, const uint8_t* probe, int micro_indel_len, int max_tot_mm, int len, uint64_t e_start,
                           uint64_t e_end, int64_t targ_ofs, K4XHit& hit, const K4XMask& mk) {
  k4x_zero(hit);
  if (targ_ofs < (int64_t)e_start || (targ_ofs + len - 1) > (int64_t)e_end) return 0;
  const uint32_t targ_seq_len = (uint32_t)((e_end - e_start + 1) - ((uint64_t)targ_ofs - e_start));
  int mm_ofs[K4X_MAX_OFSS];
  int n_mm = 0;
  uint32_t pb = 0, tv = 0;
  const int lim = max(max_tot_mm, 7);
  if (mk.have)
    for (int idx = k4x_next_up(mk, 0, len); idx < len && n_mm <= lim; idx = k4x_next_up(mk, idx + 1, len)) mm_ofs[n_mm++] = idx;
  else
    for (int idx = 0; idx < len && n_mm <= lim; idx++) {
      pb = probe[idx] & 7; tv = t.get(targ_ofs + idx);
      if (tv > 4 || pb > 4) return 0;
      if (pb == tv && pb <= 3) continue;
      mm_ofs[n_mm++] = idx;
    }
  if (n_mm < 7 || 7 > (len - mm_ofs[0])) {
    if (n_mm > max_tot_mm) return 0;
    k4x_one_seg(hit, len, targ_ofs, n_mm);
    return 1;
  }
  const int tot_mm = min(max_tot_mm, n_mm);
  K4XHit ins, del;
  k4x_zero(ins); k4x_zero(del);
  for (int pass = 0; pass < 2; pass++) {  // insertion into the probe :9363-9423, deletion from it :9426-9482
    K4XHit& best = pass == 0 ? ins : del;
    for (int m = 0; m <= tot_mm && 7 < (len - mm_ofs[m]); m++) {
      for (int gl = 1; gl <= micro_indel_len; gl++) {
        int score = 500 + len * 3 - ((gl - 1) + 20);
        const int p0 = pass == 0 ? mm_ofs[m] + gl : mm_ofs[m];
        const int t0 = pass == 0 ? mm_ofs[m] : mm_ofs[m] + gl;
        const uint32_t tmp_probe_len = (uint32_t)(len - p0);
        if (tmp_probe_len < 7) break;
        if ((targ_seq_len - (uint32_t)t0) < tmp_probe_len) break;
        int mms = 0;
        uint32_t idx;
        for (idx = 0; idx < tmp_probe_len && (m + mms) <= max_tot_mm; idx++) {
          pb = probe[p0 + (int)idx] & 7; tv = t.get(targ_ofs + t0 + (int64_t)idx);
          if (pb > 4 || tv > 4) break;
          if (pb == tv && pb <= 3) continue;
          mms += 1;
          score -= 5;
          if (tmp_probe_len < (uint32_t)(7 * mms)) break;
        }
        if (idx != tmp_probe_len) continue;
        if (score > (int)best.score) {
          best.len0 = (uint32_t)mm_ofs[m]; best.l0 = (uint64_t)targ_ofs; best.mm0 = (uint32_t)m;
          best.len1 = tmp_probe_len & 0xFFFFu;
          best.l1 = pass == 0 ? best.l0 + best.len0 : (uint64_t)targ_ofs + best.len0 + (uint64_t)gl;
          best.mm1 = (uint32_t)mms;
          best.ofs1 = pass == 0 ? best.len0 + (uint32_t)gl : best.len0;
          best.score = (uint32_t)score & 0xFFFFu;
          best.fl = pass == 0 ? 3u : 1u;
        }
      }
    }
  }
  if (del.score == 0 && ins.score == 0) return 0;
  if (del.score > ins.score) { hit = del; return 3; }
  hit = ins;
  return 2;
}

// ExploreInDelMatchLeft, SfxArray.cpp:9506-9735
K4_DEV int k4x_indel_left(K4Tb& t, const uint8_t* probe, int micro_indel_len, int max_tot_mm, int len, uint64_t e_start,
                          uint64_t e_end, int64_t targ_ofs, K4XHit& hit, const K4XMask& mk) {
  k4x_zero(hit);
  if (targ_ofs < (int64_t)e_start || (targ_ofs + len - 1) > (int64_t)e_end) return 0;
  int mm_ofs[K4X_MAX_OFSS];
  int n_mm = 0, idx;
  uint32_t pb = 0, tv = 0;
  const int lim = max(max_tot_mm, 7);
  if (mk.have)
    for (idx = k4x_next_down(mk, len - 1, -1); idx >= 0 && n_mm <= lim; idx = k4x_next_down(mk, idx - 1, -1)) mm_ofs[n_mm++] = idx;
  else
    for (idx = len - 1; idx >= 0 && n_mm <= lim; idx--) {
      pb = probe[idx] & 7; tv = t.get(targ_ofs + idx);
      if (tv > 4 || pb > 4) return 0;
      if (pb == tv && pb <= 3) continue;
      mm_ofs[n_mm++] = idx;
    }
  if (n_mm < 7 || 7 > mm_ofs[0]) {
    if (n_mm > max_tot_mm) return 0;
    k4x_one_seg(hit, len, targ_ofs, n_mm);
    return 1;
  }
  const int tot_mm = min(max_tot_mm, n_mm);
  K4XHit ins, del;
  k4x_zero(ins); k4x_zero(del);
  for (int pass = 0; pass < 2; pass++) {  // insertion :9592-9658, deletion :9662-9722
    K4XHit& best = pass == 0 ? ins : del;
    for (int m = 0; m <= tot_mm && 7 < mm_ofs[m]; m++) {
      for (int gl = 1; gl <= micro_indel_len; gl++) {
        int score = 500 + len * 3 - ((gl - 1) + 20) - m * 5;
        if (score < (int)best.score) break;
        const int p0 = pass == 0 ? mm_ofs[m] - gl : mm_ofs[m];
        const int t0 = pass == 0 ? mm_ofs[m] : mm_ofs[m] - gl;
        const uint32_t tmp_probe_len = pass == 0 ? (uint32_t)(mm_ofs[m] - (gl - 1)) : (uint32_t)(mm_ofs[m] + 1);
        if (tmp_probe_len < 7) break;
        if (pass == 1 && (int64_t)gl > targ_ofs) break;
        int mms = 0;
        for (idx = 0; idx < (int)tmp_probe_len && (m + mms) <= max_tot_mm; idx++) {
          pb = probe[p0 - idx] & 7; tv = t.get(targ_ofs + t0 - idx);
          if (pb > 4 || tv > 4) break;
          if (pb == tv && pb <= 3) continue;
          mms += 1;
          score -= 5;
          if (tmp_probe_len < (uint32_t)(7 * mms)) break;
        }
        if (idx != (int)tmp_probe_len) continue;
        if (score > (int)best.score) {
          best.len0 = tmp_probe_len & 0xFFFFu;
          best.l0 = pass == 0 ? (uint64_t)(uint32_t)(targ_ofs + gl) : (uint64_t)(uint32_t)(targ_ofs - gl);  // :9640 / :9705 32-bit casts
          best.mm0 = (uint32_t)mms;
          if (pass == 0) {
            best.len1 = (uint32_t)(len - (int)(tmp_probe_len + (uint32_t)gl)) & 0xFFFFu;
            best.l1 = best.l0 + best.len0;
            best.ofs1 = best.len0 + (uint32_t)gl;
          } else {
            best.len1 = (uint32_t)(len - (int)tmp_probe_len) & 0xFFFFu;
            best.l1 = best.l0 + tmp_probe_len + (uint64_t)gl;
            best.ofs1 = tmp_probe_len & 0xFFFFu;
          }
          best.mm1 = (uint32_t)m;
          best.score = (uint32_t)score & 0xFFFFu;
          best.fl = pass == 0 ? 3u : 1u;
        }
      }
    }
  }
  if (del.score == 0 && ins.score == 0) return 0;
  if (del.score > ins.score) { hit = del; return 3; }
  hit = ins;
  return 2;
}

K4_DEV uint32_t k4x_splice_bonus(char cur_strand, uint32_t d0, uint32_t d1, uint32_t a0, uint32_t a1) {  // :8969-8984
  const bool gt_ag = d0 == 2 && d1 == 3 && a0 == 2 && a1 == 0;
  const bool ct_ac = d0 == 1 && d1 == 3 && a0 == 1 && a1 == 0;
  if (cur_strand == '+') return gt_ag ? 50u : ct_ac ? 25u : 0u;
  return ct_ac ? 50u : gt_ag ? 25u : 0u;
}

// ExploreSpliceRight, SfxArray.cpp:8771-9011
K4_DEV int k4x_splice_right(K4Tb& t, const uint8_t* probe, char cur_strand, int max_junct_len, int max_tot_mm, int core_len,
                            int len, int64_t targ_ofs, int64_t targ_len, K4XHit& hit, const K4XMask& mk) {
  k4x_zero(hit);
  if ((targ_ofs + len + 25) > targ_len) return 0;
  if (max_tot_mm > 2) max_tot_mm = 2;
  int mm_ofs[K4X_MAX_OFSS];
  int n_mm = 0;
  uint32_t pb = 0, tv = 0, idx;
  const int lim = max(max_tot_mm, 10);
  if (mk.have)
    for (int j = k4x_next_up(mk, core_len, len); j < len && n_mm <= lim; j = k4x_next_up(mk, j + 1, len)) mm_ofs[n_mm++] = j;
  else
    for (idx = (uint32_t)core_len; idx < (uint32_t)len && n_mm <= lim; idx++) {
      pb = probe[idx] & 7; tv = t.get(targ_ofs + idx);
      if (tv > 4 || pb > 4) return 0;
      if (pb == tv && pb <= 3) continue;
      mm_ofs[n_mm++] = (int)idx;
    }
  if (n_mm < 8 || 10 > (len - mm_ofs[0])) {
    if (n_mm > max_tot_mm) return 0;
    k4x_one_seg(hit, len, targ_ofs, n_mm);
    return 1;
  }
  const int tot_mm = min(n_mm, max_tot_mm);
  {  // :8868-8873 (with MaxTotMM 0 the reference indexes in front of its array; offset 0 stands in for that word)
    const int64_t p0 = targ_ofs + (tot_mm >= 1 ? mm_ofs[tot_mm - 1] : 0);
    for (idx = 0; idx < 35u; idx++)
      if (t.get(p0 + idx) > 4) return 0;
  }
  K4XHit cur;
  k4x_zero(cur);
  for (int m = 0; m <= tot_mm && 10 < (len - mm_ofs[m]); m++) {
    if (cur.score >= 1000) break;
    const uint32_t max_seg_len = (uint32_t)(len - mm_ofs[m]);
    const uint8_t* cur_p = probe + mm_ofs[m];
    const int64_t donor = targ_ofs + mm_ofs[m];
    int64_t t_start = donor + 25;
    const int max_hash_diff = 4 * (max_tot_mm - m);
    int probe_hash = 100000;
    for (idx = 0; idx < max_seg_len; idx++) probe_hash += cur_p[idx] & 7;
    const int min_hash = probe_hash - max_hash_diff, max_hash = probe_hash + max_hash_diff;
    int targ_hash = 100000;
    int64_t t_end = t_start;
    for (idx = 0; idx < max_seg_len - 1; idx++) {
      const uint32_t v = t.get(t_end);
      if (v > 4) break;
      targ_hash += (int)v;
      t_end++;
    }
    if (idx < (max_seg_len - 1)) break;
    // two readers: the window's leading and trailing ends move in step (each keeps its own cached word)
    K4Tb ts = t;
    for (int gap = 25; gap < max_junct_len - (int)max_seg_len; gap++, t_start++, t_end++) {
      if ((tv = t.get(t_end)) > 4) break;
      targ_hash += (int)tv;
      const uint32_t out = ts.get(t_start);
      if (targ_hash < min_hash || targ_hash > max_hash) { targ_hash -= (int)out; continue; }
      targ_hash -= (int)out;
      const uint32_t tmp_targ_len = (uint32_t)(targ_len - (targ_ofs + mm_ofs[m] + gap + 1));
      if (tmp_targ_len < max_seg_len) break;
      int mms = 0;
      K4Tb tc = ts;
      for (idx = 0; idx < max_seg_len && (m + mms) < max_tot_mm; idx++) {
        pb = cur_p[idx] & 7; tv = tc.get(t_start + idx);
        if (pb > 4 || tv > 4) break;
        if (pb == tv && pb <= 3) continue;
        mms += 1;
      }
      if (idx != max_seg_len) {
        if (pb > 4 || tv > 4) break;
        continue;
      }
      uint32_t score = (uint32_t)(500 + len * 3 - (((m + mms) * 5) + ((gap / 1000) * 10)));
      score += k4x_splice_bonus(cur_strand, tc.get(donor), tc.get(donor + 1), tc.get(t_start - 1), tc.get(t_start - 2));
      if (score > cur.score) {
        cur.len0 = (uint32_t)mm_ofs[m]; cur.l0 = (uint64_t)targ_ofs; cur.mm0 = (uint32_t)m;
        cur.len1 = (uint32_t)(len - mm_ofs[m]); cur.l1 = (uint64_t)(targ_ofs + mm_ofs[m] + gap); cur.mm1 = (uint32_t)mms;
        cur.ofs1 = (uint32_t)mm_ofs[m];
        cur.score = score & 0xFFFFu;
        cur.fl = 4u;
      }
    }
  }
  if (cur.score == 0) return 0;
  hit = cur;
  return 3;
}

// ExploreSpliceLeft, SfxArray.cpp:9022-9265
K4_DEV int k4x_splice_left(K4Tb& t, const uint8_t* probe, char cur_strand, int max_junct_len, int max_tot_mm, int core_len,
                           int len, int64_t targ_ofs, K4XHit& hit, const K4XMask& mk) {
  k4x_zero(hit);
  if ((uint64_t)targ_ofs < 35u) return 0;
  if (max_tot_mm > 2) max_tot_mm = 2;
  const int64_t t3 = targ_ofs + len - 1;
  const int p3 = len - 1;
  int mm_ofs[K4X_MAX_OFSS];
  int n_mm = 0;
  uint32_t pb = 0, tv = 0, idx;
  const int lim = max(max_tot_mm, 10);
  if (mk.have)  // (offset idx counts from the read's end: base p3 - idx)
    for (int j = k4x_next_down(mk, p3 - core_len, -1); j >= 0 && n_mm <= lim; j = k4x_next_down(mk, j - 1, -1)) mm_ofs[n_mm++] = p3 - j;
  else
    for (idx = (uint32_t)core_len; idx < (uint32_t)len && n_mm <= lim; idx++) {
      pb = probe[p3 - (int)idx] & 7; tv = t.get(t3 - idx);
      if (tv > 4 || pb > 4) return 0;
      if (pb == tv && pb <= 3) continue;
      mm_ofs[n_mm++] = (int)idx;
    }
  if (n_mm < 8 || 10 > (len - mm_ofs[0])) {
    if (n_mm > max_tot_mm) return 0;
    k4x_one_seg(hit, len, targ_ofs, n_mm);
    return 1;
  }
  const int tot_mm = min(n_mm, max_tot_mm);
  for (idx = 0; idx < 35u; idx++)
    if (t.get(t3 - mm_ofs[tot_mm] - idx) > 4) return 0;
  K4XHit cur;
  k4x_zero(cur);
  for (int m = 0; m <= tot_mm && 10 < (len - mm_ofs[m]); m++) {
    if (cur.score >= 1000) break;
    const uint32_t max_seg_len = (uint32_t)(len - mm_ofs[m]);
    const int cur_p = p3 - mm_ofs[m];
    const int64_t donor = t3 - mm_ofs[m];
    int64_t t_start = donor - 25;
    const int max_hash_diff = 4 * (max_tot_mm - m);
    int probe_hash = 100000;
    for (idx = 0; idx < max_seg_len; idx++) probe_hash += probe[cur_p - (int)idx] & 7;
    const int min_hash = probe_hash - max_hash_diff, max_hash = probe_hash + max_hash_diff;
    int targ_hash = 100000;
    int64_t t_end = t_start;
    for (idx = 0; idx < max_seg_len - 1; idx++) {
      const uint32_t v = t.get(t_end);
      if (v > 4) break;
      targ_hash += (int)v;
      t_end--;
    }
    if (idx < (max_seg_len - 1)) break;
    K4Tb ts = t;
    for (int gap = 25; gap < max_junct_len - (int)max_seg_len; gap++, t_start--, t_end--) {
      if ((tv = t.get(t_end)) > 4) break;
      targ_hash += (int)tv;
      const uint32_t out = ts.get(t_start);
      if (targ_hash < min_hash || targ_hash > max_hash) { targ_hash -= (int)out; continue; }
      targ_hash -= (int)out;
      if ((uint32_t)(targ_ofs - gap) < 1u) break;
      int mms = 0;
      K4Tb tc = ts;
      for (idx = 0; idx < max_seg_len && (m + mms) < max_tot_mm; idx++) {
        pb = probe[cur_p - (int)idx] & 7; tv = tc.get(t_start - idx);
        if (pb > 4 || tv > 4) break;
        if (pb == tv && pb <= 3) continue;
        mms += 1;
      }
      if (idx != max_seg_len) {
        if (pb > 4 || tv > 4) break;
        continue;
      }
      uint32_t score = (uint32_t)(500 + len * 3 - (((m + mms) * 5) + ((gap / 1000) * 10)));
      score += k4x_splice_bonus(cur_strand, tc.get(t_start + 1), tc.get(t_start + 2), tc.get(donor), tc.get(donor - 1));
      if (score > cur.score) {
        cur.len0 = (uint32_t)(len - mm_ofs[m]); cur.l0 = (uint64_t)(targ_ofs - gap); cur.mm0 = (uint32_t)mms;
        cur.len1 = (uint32_t)mm_ofs[m]; cur.l1 = cur.l0 + cur.len0 + (uint64_t)gap; cur.mm1 = (uint32_t)m;
        cur.ofs1 = cur.len0;
        cur.score = score & 0xFFFFu;
        cur.fl = 4u;
      }
    }
  }
  if (cur.score == 0) return 0;
  hit = cur;
  return 3;
}

K4_DEV uint64_t k4x_shfl64(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
  return ((uint64_t)hi << 32) | lo;
}
K4_DEV K4XHit k4x_from_lane(const K4XHit& x, int src) {  // (every lane gets lane src's record: wave-uniform)
  K4XHit r;
  r.l0 = k4d_uni(k4x_shfl64(x.l0, src)); r.l1 = k4d_uni(k4x_shfl64(x.l1, src));
  r.len0 = k4d_uni((uint32_t)__shfl((int)x.len0, src, 64)); r.len1 = k4d_uni((uint32_t)__shfl((int)x.len1, src, 64));
  r.ofs1 = k4d_uni((uint32_t)__shfl((int)x.ofs1, src, 64)); r.mm0 = k4d_uni((uint32_t)__shfl((int)x.mm0, src, 64));
  r.mm1 = k4d_uni((uint32_t)__shfl((int)x.mm1, src, 64)); r.score = k4d_uni((uint32_t)__shfl((int)x.score, src, 64));
  r.fl = k4d_uni((uint32_t)__shfl((int)x.fl, src, 64));
  return r;
}

// ---- LocateInDels (SfxArray.cpp:7526-7832) / LocateSpliceJuncts (:7208-7523): two cores per strand -- the read's first and
// last core_len bases -- every suffix of the core's run explored to the right / to the left.  MaxHits is 1 in both calls
// (:7903,7918): only pHits[0] is ever written, a tie in score only counts.
// The four (strand, core) pairs are looked up TOGETHER (k4d_group_lookup: one round of k-mer table loads, the buckets as one
// sequence of slots) and walked 64 slots a step, pair by pair in the reference's order; the microInDel and the splice call of
// one read use the same cores, so the second call walks the slots the first one laid out (reuse_lookup). --------------------
template <int EL>
K4_DEV int k4d_two_seg(const K4AlignArgs& a, K4Slow& sc, bool splice, int limit_len, int max_tot_mm, int core_len, int strand,
                       int len, int* p_inst, int* p_low, int* p_nxt, k4_hit* hit0, k4_seg2* seg2, uint32_t& n_lookup,
                       uint32_t& n_probe, uint32_t& n_cand, bool reuse_lookup) {
  const K4DevIndex& ix = a.ix;
  const int64_t n = (int64_t)ix.n;
  const int max_iter = ix.max_iter;
  const int lane = sc.lane;
  if (splice && max_tot_mm > 2) max_tot_mm = 2;
  *p_inst = 0; *p_low = 0; *p_nxt = 0;
  if (lane == 0) {  // memset(pHits, 0, sizeof(tsHitLoci)), :7280 / :7592
    *reinterpret_cast<uint4*>(hit0) = make_uint4(0, 0, 0, 0);
    if (seg2) *reinterpret_cast<uint4*>(seg2) = make_uint4(0, 0, 0, 0);
  }
  K4XHit best;
  k4x_zero(best);
  int best_inst = 0;
  if (core_len > len) return K4_HR_NONE;
  // pair j: strand s_first + (j >> 1), core at the read's start (j even) or end (j odd)
  const int s_first = strand == K4_STRAND_CRICK ? 1 : 0, s_last = strand == K4_STRAND_WATSON ? 0 : 1;
  const int np = 2 * (s_last - s_first + 1);
  unsigned long long smask = 0;
  for (int j = 0; j < np; j++)
    if (s_first + (j >> 1)) smask |= 1ull << j;
  const int my_o = (lane & 1) ? len - core_len : 0;
  K4_PROF_T(px0);
  const uint64_t total = reuse_lookup ? k4d_uni(sc.g_pre[np]) : k4d_group_lookup<EL>(ix, sc, np, smask, my_o, core_len, n_probe);
  K4_PROF_T(px1);
  K4_PROF_ADD(16, px1 - px0);
  K4_PROF_ADD(21, 1);
  int opened = 0, iter = 0;
  bool pair_done = false, seen_first = false, stop_all = false;
  bool v_n = false, reload = true;
  int pj_n = 0;
  uint64_t pos_n = 0, base = 0;
  while (base < total) {
    if (reload) {
      const uint64_t q = base + lane;
      v_n = q < total;
      if (v_n) {
        while (q >= sc.g_pre[pj_n + 1]) pj_n++;
        pos_n = k4d_sa_at<EL>(ix, sc.g_lb[pj_n] + (q - sc.g_pre[pj_n]));
      }
      reload = false;
    }
    const bool valid = v_n;
    const int pj = pj_n;
    const uint64_t pos = pos_n;
    {  // the next step's suffix elements are fetched during this one
      const uint64_t q = base + 64 + lane;
      v_n = q < total;
      if (v_n) {
        while (q >= sc.g_pre[pj_n + 1]) pj_n++;
        pos_n = k4d_sa_at<EL>(ix, sc.g_lb[pj_n] + (q - sc.g_pre[pj_n]));
      }
    }
    K4_PROF_T(py0);
    bool core_eq = false;
    if (valid) core_eq = k4d_lane_cmp(ix, sc, (pj & 1) ? len - core_len : 0, core_len, pos, (int)((smask >> pj) & 1ull)) == 0;
    K4_PROF_T(py1);
    K4_PROF_ADD(17, py1 - py0);
    K4_PROF_ADD(22, 1);
    const unsigned long long validm = __ballot(valid);
    n_probe += (uint32_t)__popcll(validm);
    const int j_lo = k4d_uni(__shfl(pj, 0, 64)), j_hi = k4d_uni(__shfl(pj, 63 - __clzll(validm), 64));
    uint64_t next_base = base + 64;
    for (int jj = j_lo; jj <= j_hi; jj++) {
      const unsigned long long seg = __ballot(valid && pj == jj);
      if (!seg) continue;
      for (; opened <= jj; opened++) {  // behind a core's walk: a full score cannot be beaten (:7469 / :7776), nothing further is looked at
        if (best_inst >= 1 && best.score >= 1000) { stop_all = true; break; }
        n_lookup++;
        iter = 0;
        pair_done = false;
        seen_first = false;
      }
      if (stop_all) break;
      if (!pair_done) {
        const int cs = (int)((smask >> jj) & 1ull), phase = jj & 1;
        const char cur_strand = cs ? '-' : '+';
        const int ofs = phase ? len - core_len : 0;
        const bool in_seg = (seg >> lane) & 1ull;
        unsigned long long memberm = __ballot(in_seg && core_eq);
        // the run's first suffix is taken as LocateFirstExact gives it; each further one must leave room behind its start
        // (:7355-7362 / :7666-7672), else the walk over this core ends there
        bool fits = true;
        if (in_seg && core_eq) fits = splice ? !(((int64_t)pos + (phase == 0 ? len : core_len)) >= n) : !(((int64_t)pos + core_len) > n);
        unsigned long long unfit = memberm & __ballot(!fits);
        if (!seen_first && memberm) unfit &= ~(memberm & (0ull - memberm));
        if (memberm) seen_first = true;
        bool walk_ends = false;
        if (unfit) {
          memberm &= (unfit & (0ull - unfit)) - 1ull;
          walk_ends = true;
        }
        // filters in front of the exploration, :7368-7383 / :7679-7693 (the entry is looked up at the CORE's position)
        bool inb = ((memberm >> lane) & 1ull) && pos >= (uint64_t)ofs;
        const int64_t left = (int64_t)pos - ofs;
        uint64_t e_start = 0, e_end = 0;
        if (inb) {
          const int e = k4d_map_entry_slow(ix, sc.ent, pos, e_start, e_end);
          if (splice) inb = (left + len) < n && e >= 0 && left >= (int64_t)e_start && (left + len) <= (int64_t)e_end;
          else inb = e >= 0 && left >= (int64_t)e_start && (left + len - 1) <= (int64_t)e_end && (left + len) <= n;
        }
        unsigned long long inm = __ballot(inb);
        bool hit_limit = false;
        if (max_iter) {  // IterCnt counts the candidates that got this far; the walk ends with the MaxIter-th
          const int rem = max_iter - iter;
          if ((int)__popcll(inm) >= rem) {
            unsigned long long keep = inm;
            for (int q = 1; q < rem; q++) keep &= keep - 1;
            const int last = __ffsll((long long)keep) - 1;
            inm &= last >= 63 ? ~0ull : ((1ull << (last + 1)) - 1ull);
            hit_limit = true;
          }
        }
        const int took = (int)__popcll(inm);
        iter += took;
        n_cand += (uint32_t)took;
        K4_PROF_T(py2);
        K4XHit x;
        k4x_zero(x);
        int xr = 0;
        if ((inm >> lane) & 1ull) {
          K4Tb tb;
          tb.init(ix);
          const uint8_t* probe = sc.probe + (cs ? sc.pstride : 0u);
          K4XMask xm;
          xm.w[0] = xm.w[1] = xm.w[2] = xm.w[3] = 0;
          xm.have = sc.packed && len <= 128 && (uint64_t)left + (uint64_t)len <= ix.n && !k4d_any_exc_sup(ix, sc.sup, left, left + len);
          if (xm.have) {  // the locus' window in one round of loads; read and window as mismatch bits
            uint64_t rc[4];
            k4d_ref_chunks4(ix, left, 0, len + (int)(left & 15) <= 128, rc);
#pragma unroll
            for (int c = 0; c < 4; c++)
              if (32 * c < len) xm.w[c] = k4d_mm_bits((rc[c] ^ k4d_probe_chunk(sc, 32 * c, cs)) & k4d_range_mask(0, len - 32 * c));
          }
          if (!splice)
            xr = phase == 0 ? k4x_indel_right(tb, probe, limit_len, max_tot_mm, len, e_start, e_end, left, x, xm)
                            : k4x_indel_left(tb, probe, limit_len, max_tot_mm, len, e_start, e_end, left, x, xm);
          else if (phase == 0) {  // :7392-7426
            int lim = (int)(n - left);
            if (lim > 35) {
              lim -= 35;
              if (lim > limit_len) lim = limit_len;
              xr = k4x_splice_right(tb, probe, cur_strand, lim, max_tot_mm, core_len, len, left, n, x, xm);
            }
          } else if ((uint64_t)left >= (uint32_t)(ofs + 10)) {  // :7429-7461
            int lim = min((int32_t)left, (int32_t)limit_len);
            if (lim >= 35) {
              lim -= 10;
              xr = k4x_splice_left(tb, probe, cur_strand, lim, max_tot_mm, core_len, len, left, x, xm);
            }
          }
        }
        K4_PROF_T(py3);
        K4_PROF_ADD(18, py3 - py2);
        K4_PROF_ADD(20, took);
        if (xr > 0 && cur_strand == '-') x.fl |= 8u;
        // replay in suffix order: `>=` the best score so far; an equal score at another locus only counts
        unsigned long long todo = __ballot(xr > 0);
        while (todo) {
          const int c = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          const uint32_t xs = k4d_uni((uint32_t)__shfl((int)x.score, c, 64));
          if (xs < best.score) continue;
          if (xs == best.score) {
            if (best.l0 == k4d_uni(k4x_shfl64(x.l0, c))) continue;
            if (++best_inst > 1) continue;
          } else
            best_inst = 0;
          best = k4x_from_lane(x, c);
          best_inst++;
        }
        if (hit_limit || walk_ends) pair_done = true;
      }
      if (pair_done && jj == j_hi && k4d_uni(sc.g_pre[jj + 1]) > base + 64) {  // on to the next pair's slots
        next_base = k4d_uni(sc.g_pre[jj + 1]);
        reload = true;
      }
    }
    if (stop_all) break;
    base = next_base;
  }
  if (!stop_all)
    for (; opened < np; opened++) {  // the pairs behind the last slot (no suffix starts with their core) are reached as well
      if (best_inst >= 1 && best.score >= 1000) break;
      n_lookup++;
    }
  if (best_inst == 0) return K4_HR_NONE;
  if (best.score > 1000) best.score = 1000;
  // concat offsets -> chromosome + locus, :7501-7516 / :7800-7825
  uint64_t s0 = 0, e0e = 0, s1 = 0, e1e = 0;
  const int e0 = k4d_map_entry_slow(ix, sc.ent, best.l0, s0, e0e);
  if (e0 < 0) return K4_HR_NONE;
  uint32_t chrom1 = 0;
  uint64_t l1 = best.l1;
  if (!splice) {
    const int e1 = k4d_map_entry_slow(ix, sc.ent, best.l1, s1, e1e);
    if (e1 < 0 || e1 != e0) return K4_HR_NONE;  // (an InDel across two chromosomes, :7811-7814)
    if (best.l1 > 0) { chrom1 = ix.ent_id[e1]; l1 -= s1; }
  } else if (best.l1 > 0) {
    const int e1 = k4d_map_entry_slow(ix, sc.ent, best.l1, s1, e1e);
    if (e1 < 0) return K4_HR_NONE;
    chrom1 = ix.ent_id[e1];
    l1 -= s1;
  }
  // (with more than one instance the call returns eHRnone, but the slot and the counts stay as they are: a chimeric pass
  // that follows sees them carried in, SfxArray.cpp:5890)
  if (lane == 0) {
    const uint32_t ext = ((best.fl & 1) ? K4_EXT_INDEL : 0u) | ((best.fl & 2) ? K4_EXT_INSERT : 0u) | ((best.fl & 4) ? K4_EXT_SPLICE : 0u);
    uint4 v;
    v.x = ix.ent_id[e0];
    v.y = (uint32_t)(best.l0 - s0);
    v.z = (best.len0 & 0xFFFFu) | ((uint32_t)(uint8_t)((best.fl & 8) ? '-' : '+') << 16) | ((best.mm0 & 0xFFu) << 24);
    v.w = ext;
    *reinterpret_cast<uint4*>(hit0) = v;
    if (seg2 && (best.fl & 5)) {  // (a one-segment result leaves the record zero)
      uint4 w;
      w.x = chrom1;
      w.y = (uint32_t)l1;
      w.z = (best.len1 & 0xFFFFu) | ((best.ofs1 & 0xFFFFu) << 16);
      w.w = (best.mm1 & 0xFFu) | ((best.score & 0xFFFFu) << 16);
      *reinterpret_cast<uint4*>(seg2) = w;
    }
  }
  *p_inst = splice ? best_inst : min(1, best_inst);
  *p_low = (int)(best.mm0 + best.mm1);
  *p_nxt = *p_low + 2;
  return best_inst <= 1 ? K4_HR_HITS : K4_HR_NONE;
}
