// kit4b_amd/csrc/k4_align_common.h -- what the two translation units of the alignment kernels share: k4_align.hip (the step
// kernels, one lane per read, and the host entry points) and k4_general.hip (the general kernel, one wave per read).
#pragma once
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include "k4_device.h"

#define K4_NEED_SLOW (-100)
#define K4_DEFER (-101)  // first launch only: the read met a deep k-mer bucket; it is taken again in the launch of its like
#define K4_RF_HAS_N 1u
#define K4_RF_INVALID 2u
#define K4_RF_TOOLONG 4u
#ifndef K4_STEP_WAVES
#define K4_STEP_WAVES 4  // waves per SIMD the step kernel is register-budgeted for
#endif
#ifndef K4_STEP_WAVES_LONG
#define K4_STEP_WAVES_LONG 2  // ... for reads over 256 bp (16 packed words per strand; LDS allows one block per CU anyway)
#endif
#ifndef K4_STEP_WAVES_5
#define K4_STEP_WAVES_5 4     // ... for 129..160 bp (5 words per strand); 3 (no spills) measured 8 % slower on C3
#endif
#ifndef K4_STEP_WAVES_MID
#define K4_STEP_WAVES_MID 3   // ... for 161..256 bp (8 words per strand; LDS allows three blocks per CU)
#endif
#define K4_CHUNK 512         // survivor slots a wave reserves per atomic
#define K4_NO_READ 0xFFFFFFFFu  // hole in a survivor list
#define K4_ROW_WORDS(nch) (2 * (nch) + 2)  // survivor row: forward + reverse-complement words, then the two offset-0 memos
#ifndef K4_PF
#define K4_PF 4  // k-mer table entries fetched ahead per strand pass
#endif
#ifndef K4_PF5
#define K4_PF5 4 // ... in the 5-word (129..160 bp) instantiation (3 measured +1 % on C3: noise)
#endif
// general kernel: waves per SIMD the compiler must fit its registers to.  Standard phases: 4 (128 VGPRs, ~60 spilled) -- the
// kernel waits on memory, a fourth wave is worth more than the spills cost (measured: 3 waves 36.1 ms, 4 waves 33.8 ms per
// 20 M reads of the repeat-rich workload).  With the optional phases compiled in (EXT): 2, without spills.
#ifndef K4_SLOW_WAVES_PER_EU
#define K4_SLOW_WAVES_PER_EU 4
#endif
#ifndef K4_SLOW_WAVES_PER_EU_EXT
#define K4_SLOW_WAVES_PER_EU_EXT 3
#endif
// A read whose first phase meets a k-mer bucket deeper than this (a repeat family: the lower-bound search alone costs
// log2(depth) dependent probes where its 63 wave mates need one or two) is set aside by the first launch and taken in a
// launch of its own together with the others of its kind, so that the many waves without such a read do not wait for it;
// its survivors stay together in their own chunks through the later phases.
#ifndef K4_DEFER_BUCKET
#define K4_DEFER_BUCKET K4_DEEP_BUCKET
#endif
// (Sending the reads with the deepest buckets straight to the general kernel instead was measured and lost: half of them are
// settled by the fast path -- 2.08 M instead of 1.06 M reads per 50 M in the general kernel, 91 ms instead of 74.)
#define K4_DEFER_MIN_FRAC 0.002  // of the index's suffixes in buckets that deep: below it the first launch is not split
#ifndef K4_SLOW_KB
#define K4_SLOW_KB 1  // general kernel: suffixes per lane per walk step (measured: 2 and 4 cost occupancy and lose 25 %)
#endif
#ifndef K4_SLOW_WAVES
#define K4_SLOW_WAVES 8192  // pass 0 of the general kernel: 32 waves per CU
#endif
#define K4_SMALL_HASH 4096  // entries of a pass-0 dedupe table (2047 candidates per strand pass)
#define K4_HUGE_WAVES 256
#define K4_CTL_HUGE 68  // ctl[68] huge count, ctl[69] huge head
#define K4_CTL_DEFER 70  // ctl[70] slots handed out in the deferred list of the first launch
#define K4_CTL_WORDS 72  // [0] slow count, [1] slow head, [2+t] survivors of step t

struct K4AlignArgs {
  K4DevIndex ix;
  const uint8_t* reads;
  const uint64_t* offs;
  const uint32_t* lens;
  int64_t n_reads;
  int32_t mode;       // 0: AlignReads with uniform parameters, 1: CKAligner::AlignRead
  int32_t sparse_hits; // hit slots that hold no reported instance are left as they are (internal callers that never read them)
  int32_t best;       // mode 0 only: LocateBestMatches instead of AlignReads (every read runs in the general kernel)
  int32_t deep_general; // a read that meets a k-mer bucket deeper than K4_DEFER_BUCKET leaves the step kernels for the general one (repeat-rich indexes)
  k4_align_params ap;
  k4_kalign_params kp;  // min_core_len / max_num_slides already resolved
  int32_t* rslt;
  int32_t* inst;
  int32_t* low;
  int32_t* nxt;
  k4_read_result* rr;
  k4_hit* hits;
  k4_seg2* seg2;      // second segments of microInDel / splice hits, one per read (null: those phases are off)
  int32_t ext_on;     // any optional AlignReads phase requested (SfxArray.cpp:7894-7930): reads the standard phases leave
                      // without a result go on to the general kernel instead of being finalised
  int32_t max_hits;
  uint32_t* slow_list;
  uint8_t* slow_step;  // phase ordinal at which the read left the fast path (its earlier phases are already tallied)
  uint32_t* huge_list; // reads whose strand pass outgrew the small dedupe tables: second general pass with big tables
  uint8_t* huge_step;
  uint32_t* defer_ids;  // first launch: list of the reads set aside (chunked like the survivor lists, ctl[K4_CTL_DEFER])
  uint32_t* ctl;
  unsigned long long* counters;
  uint8_t* slow_probe;
  uint64_t* slow_hash;
  uint32_t* slow_gen;
  uint32_t slow_hash_cap;
  int32_t nw;
};

struct K4ReadParams {
  int tot_mm, core_len, core_delta, max_slides, mm_delta, strand, max_hits;
  int min_core_len, min_chimeric_len, micro_indel_len, max_splice_junct_len;  // the optional phases (k4_ext.h)
};

struct K4State {
  int inst, low, nxt, cur_hit;
};

// CKAligner::AlignRead parameter derivation, ngskit4b/KAligner.cpp:9662-9672
K4_DEV K4ReadParams k4d_read_params(const K4AlignArgs& a, int len) {
  K4ReadParams p;
  if (a.mode == 0) {
    p.tot_mm = a.ap.tot_mm; p.core_len = a.ap.core_len; p.core_delta = a.ap.core_delta;
    p.max_slides = a.ap.max_core_slides; p.mm_delta = a.ap.mm_delta; p.strand = a.ap.strand;
    p.max_hits = a.ap.max_hits;
    p.min_core_len = a.ap.min_core_len; p.min_chimeric_len = a.ap.min_chimeric_len;
    p.micro_indel_len = a.ap.micro_indel_len; p.max_splice_junct_len = a.ap.max_splice_junct_len;
    return p;
  }
  int mm = a.kp.max_subs == 0 ? 0 : (int)(0.5 + (len * a.kp.max_subs) / 100.0);
  if (a.kp.max_subs != 0 && mm < 1) mm = 1;
  if (mm > 63) mm = 63;  // cMaxTotAllowedSubs, KAligner.h:38
  int cl = len / (a.kp.min_edit_dist == 1 ? mm + 1 : mm + 2);
  if (cl < a.kp.min_core_len) cl = a.kp.min_core_len;
  int sl = (a.kp.max_num_slides * len + 99) / 100;
  if (sl < 1) sl = 1;
  int cd = len / sl - 1;
  if (cd < cl) cd = cl;
  p.tot_mm = mm; p.core_len = cl; p.core_delta = cd; p.max_slides = sl;
  p.mm_delta = a.kp.min_edit_dist; p.strand = a.kp.strand; p.max_hits = a.kp.max_ml < 1 ? 1 : a.kp.max_ml;
  p.min_core_len = a.kp.min_core_len; p.min_chimeric_len = a.kp.min_chimeric_len;
  p.micro_indel_len = a.kp.micro_indel_len; p.max_splice_junct_len = a.kp.max_splice_junct_len;
  return p;
}

K4_DEV void k4d_store_hit(k4_hit* h, uint32_t chrom_id, uint32_t loci, int len, char strand, int mm, uint32_t ext = 0) {
  uint4 v;
  v.x = chrom_id;
  v.y = loci;
  v.z = (uint32_t)(len & 0xFFFF) | ((uint32_t)(uint8_t)strand << 16) | ((uint32_t)(mm & 0xFF) << 24);
  v.w = ext;
  *reinterpret_cast<uint4*>(h) = v;
}

// fold of one accepted candidate into (LowMMCnt, NxtLowMMCnt, LowHitInstances, pHits), SfxArray.cpp:6264-6312
K4_DEV void k4d_fold(K4State& st, int mm, k4_hit* hits, int max_hits, uint32_t chrom_id, uint32_t loci, int len,
                     char strand) {
  if (mm < st.low) {
    st.cur_hit = 0;
    st.inst = 1;
    st.nxt = st.low;
    st.low = mm;
    if (hits) k4d_store_hit(&hits[0], chrom_id, loci, len, strand, mm);
  } else if (mm == st.low) {
    st.inst += 1;
    if (st.cur_hit != -1 && st.inst <= max_hits) {
      st.cur_hit += 1;
      if (hits && st.cur_hit < max_hits) k4d_store_hit(&hits[st.cur_hit], chrom_id, loci, len, strand, mm);
    }
  } else if (mm < st.nxt)
    st.nxt = mm;
}

// result code of one LocateCoreMultiples call, SfxArray.cpp:6345-6368 (p_* = values on entry, after initialisation)
K4_DEV int k4d_lcm_result(int p_inst, int p_low, int* p_nxt, const K4State& st, int mm_delta, int max_hits,
                          int* o_inst, int* o_low) {
  if (p_low == st.low && p_inst == st.inst) {
    if (*p_nxt > st.nxt) {
      *p_nxt = st.nxt;
      if (st.nxt - p_low < mm_delta) return K4_HR_MMDELTA;
      return K4_HR_RMMDELTA;
    }
    return K4_HR_NONE;
  }
  *o_low = st.low; *o_inst = st.inst; *p_nxt = st.nxt;
  if (st.inst >= 1 && (st.nxt - st.low) < mm_delta) return K4_HR_MMDELTA;
  if (st.inst > max_hits) return K4_HR_HITINSTS;
  return K4_HR_HITS;
}

// writes the per-read outputs; zeroes hit slots that do not hold a reported instance
K4_DEV void k4d_finalize(const K4AlignArgs& a, int64_t i, int len, const K4ReadParams& rp, int rslt, int inst,
                         int low, int nxt) {
  k4_hit* hits = a.hits + i * a.max_hits;
  int nvalid = (rslt == K4_HR_HITS || rslt == K4_HR_MMDELTA || rslt == K4_HR_HITINSTS) ? min(inst, rp.max_hits) : 0;
  if (!a.sparse_hits)
    for (int q = nvalid; q < a.max_hits; q++) *reinterpret_cast<uint4*>(&hits[q]) = make_uint4(0, 0, 0, 0);
  if (a.mode == 0) {
    a.rslt[i] = rslt; a.inst[i] = inst;
    if (a.low) a.low[i] = low;  // (the LocateBestMatches entry points have no low / nxt outputs)
    if (a.nxt) a.nxt[i] = nxt;
    return;
  }
  // CKAligner::AlignRead classification, KAligner.cpp:9854,9890-10079 (SE default MLMode / PE / eMLall)
  k4_read_result r;
  if (inst > rp.max_hits) inst = rp.max_hits + 1;
  if (a.kp.pe_mode >= 3 && rslt == K4_HR_HITINSTS) { inst = rp.max_hits; rslt = K4_HR_HITS; }  // -X / -N clamp, :9856-9861
  r.hit_rslt = rslt; r.inst = inst; r.low_mm = low; r.nxt_mm = nxt; r.nar = K4_NAR_NOHIT; r.num_hits = 0;
  switch (rslt) {
    case K4_HR_NONE: r.inst = 0; r.low_mm = 0; r.nxt_mm = 0; break;
    case K4_HR_HITS:
      if (a.kp.pe_mode >= 2) { r.nar = K4_NAR_ACCEPTED; r.num_hits = min(inst, rp.max_hits); }  // eMLall: every instance is reported (:9913-9931)
      else if (!a.kp.pe_mode || inst == 1) { r.nar = K4_NAR_ACCEPTED; r.num_hits = 1; }
      else { r.nar = K4_NAR_MULTIALIGN; r.num_hits = inst; }
      break;
    case K4_HR_MMDELTA: r.nar = K4_NAR_MMDELTA; break;
    case K4_HR_HITINSTS: r.nar = K4_NAR_MULTIALIGN; break;
    default: break;
  }
  (void)len;
  a.rr[i] = r;
}

K4_DEV void k4d_push_slow(const K4AlignArgs& a, int64_t i, int step) {
  uint32_t slot = atomicAdd(&a.ctl[0], 1u);
  a.slow_list[slot] = (uint32_t)i;
  a.slow_step[slot] = (uint8_t)step;
}

// k4_general.hip: the two passes of the general kernel behind the step kernels of a batch, on the same stream
int k4i_launch_general(k4_index* ix, K4AlignArgs& a, int max_len, hipStream_t st);
