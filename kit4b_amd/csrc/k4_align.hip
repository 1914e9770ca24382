// kit4b_amd/csrc/k4_align.hip -- the hot path: batched CSfxArray::AlignReads / CKAligner::AlignRead on gfx950.
//
// Reference semantics reproduced (bit-identical results are the contract):
//   CSfxArray::AlignReads           libkit4b/SfxArray.cpp:7838-7933   phase escalation
//   CSfxArray::LocateCoreMultiples  libkit4b/SfxArray.cpp:5806-6369   cores -> SA run -> dedupe -> Hamming extension -> fold
//   CSfxArray::LocateFirstExact     libkit4b/SfxArray.cpp:7938-8058   lowest SA index whose suffix starts with the core
//   CKAligner::AlignRead            ngskit4b/KAligner.cpp:9583-10105  per-read parameters + NAR classification
//
// Kernels:
//   k4k_align_step   one launch per AlignReads phase ("step"), one lane per surviving read.  Step 0 packs the read
//                    bytes to 2-bit words (forward + reverse complement) in LDS; a read leaves at the first phase
//                    whose result is non-zero, the others are compacted (wave ballot + prefix popcount + one atomic
//                    per wave) together with their packed rows into the next step's input, so every step runs with
//                    full waves.  Seed lookup = one k-mer table fetch + a short lower-bound search whose every
//                    probe fetches the whole read-aligned reference window, so the same registers give the core
//                    comparison (ordering) and the Hamming distance (XOR + popcount).  Reads that meet anything the
//                    2-bit form cannot express (N, a separator in a window) or more than K4_DEDUP_CAP candidates in
//                    one strand pass are handed to ...
//   k4k_align_slow   ... the general kernel: a literal lane-per-read restatement over exact 4-bit symbols with a
//                    hash-set dedupe in HBM scratch.
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
#define K4_SLOW_WAVES_PER_EU_EXT 2
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

// ==== fast kernel ==================================================================================================
// threads per block: the per-lane LDS columns of a 16-word read (257..512 bp) would leave room for one 256-thread block per CU
#define K4_BS(nch) ((nch) >= 16 ? 128 : 256)
template <int NCH>
struct K4Lane {
  const uint64_t* rd;  // LDS: word (s*NW + c) of this lane's read at rd[(s*NW + c) * K4_BS(NCH)]
  uint32_t* ded;       // LDS: dedupe slot q at ded[q * K4_BS(NCH)]
  const uint32_t* sup; // LDS (block-shared): coarse exception bitmap, K4_SUP_WORDS words
  const uint64_t* ent; // LDS (block-shared): chromosome starts [0..K4_LDS_ENTRIES) then ends, when they fit
  uint64_t* memo;      // LDS: what the offset-0 lookup of strand s found in the first phase, at memo[s * K4_BS(NCH)]
  static constexpr int NW = NCH + 1;
  K4_DEV uint64_t word(int s, int c) const { return rd[(s * NW + c) * K4_BS(NCH)]; }
  K4_DEV uint64_t chunk_at(int s, int o) const {  // 32 bases of strand s starting at base o
    int w = o >> 5, sh = 2 * (o & 31);
    uint64_t hi = word(s, w);
    if (!sh) return hi;
    return (hi << sh) | (word(s, w + 1) >> (64 - sh));
  }
};

struct K4Probe {
  int cmp;   // core vs suffix: -1 probe<target, 0 match, 1 probe>target
  int mm;    // Hamming distance of the whole read against the read-aligned window
  int fm;    // offset of the first mismatching base of the read (len when there is none)
  bool exc;  // window touches a non-ACGT symbol
};

// Offset-0 memo.  Every phase of AlignReads starts each strand pass with the core at offset 0, i.e. with the same first
// k bases, so the same k-mer bucket; on a large genome that bucket is empty or a single suffix nearly always.  The first
// phase records what it found: kind 1 = empty bucket; kind 2 = one suffix at pos whose window differs from the read
// first at base fm (mm = Hamming distance of the whole read, 0xFF when the window was never fetched because the 16-base
// signature already disagreed).  A later phase with core length cl then knows without any memory access that nothing
// starts with its offset-0 core (empty, or fm < cl) or that exactly the suffix at pos does, with distance mm.
#define K4_MEMO_NONE 0ull
K4_DEV uint64_t k4d_memo_pack(int kind, uint64_t pos, int mm, int fm) {
  return ((uint64_t)kind << 62) | ((uint64_t)(fm & 0x3FFF) << 48) | ((uint64_t)(mm & 0xFF) << 40) | (pos & 0xFFFFFFFFFFull);
}

// One suffix-array probe: the suffix at p is where core (offset o, length cl) of strand s would sit, so the read would
// sit at left = p - o.  Fetch the window [left, left+len) once; derive both the core ordering and the distance.
template <int NCH>
K4_DEV K4Probe k4d_probe(const K4DevIndex& ix, const K4Lane<NCH>& ln, int s, int o, int cl, int len, uint64_t p) {
  K4Probe r;
  int64_t left = (int64_t)p - o;
  {  // coarse test in LDS first; the fine bitmap in L2 is only consulted near a separator / N run
    const int64_t st = left < 0 ? 0 : left;
    const uint64_t b0 = (uint64_t)st >> ix.sup_shift, b1 = (uint64_t)(left + len - 1) >> ix.sup_shift;
    const uint64_t v = (((uint64_t)ln.sup[(b0 >> 5) + 1] << 32) | ln.sup[b0 >> 5]) >> (b0 & 31);
    r.exc = (v & ((2ull << (b1 - b0)) - 1ull)) != 0;
    if (r.exc && ix.sup_shift != K4_EXC_SHIFT) r.exc = k4d_any_exc(ix, left, left + len);
  }
  r.cmp = 0;
  r.mm = 0;
  r.fm = len;
  const uint32_t* wp = ix.ref2 + (left >> 4);
  uint32_t sh = (uint32_t)(left & 15) * 2;
  uint32_t wv[2 * NCH + 1];
  if (NCH <= 5 && len + 15 <= 32 * NCH) {  // the window spans at most 2*NCH words: two 16-byte loads for <= 113 bp
    uint32_t even[2 * NCH];
    k4d_load_words<2 * NCH>(wp, even);
#pragma unroll
    for (int j = 0; j < 2 * NCH + 1; j++) wv[j] = j < 2 * NCH ? even[j] : 0u;
  } else if (NCH <= 5 || len > 32 * (NCH / 2))
    k4d_load_words<2 * NCH + 1>(wp, wv);
  else {
    uint32_t half[NCH + 1];               // a short read in a long-read batch: only the words it covers
    k4d_load_words<NCH + 1>(wp, half);
#pragma unroll
    for (int j = 0; j < 2 * NCH + 1; j++) wv[j] = j < NCH + 1 ? half[j] : 0u;
  }
  bool decided = false;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    if (32 * c < len) {
      uint64_t hi = ((uint64_t)wv[2 * c] << 32) | wv[2 * c + 1];
      uint64_t refc = sh ? (hi << sh) | (wv[2 * c + 2] >> (32 - sh)) : hi;
      uint64_t rdc = ln.word(s, c);
      uint64_t x = rdc ^ refc;
      const uint64_t xr = x & k4d_range_mask(0, len - 32 * c);
      r.mm += (int)k4d_mm_count(xr);
      if (xr && r.fm == len) r.fm = 32 * c + (__clzll(xr) >> 1);
      uint64_t cm = k4d_range_mask(o - 32 * c, o + cl - 32 * c);
      if (!decided && (x & cm)) {
        r.cmp = (rdc & cm) < (refc & cm) ? -1 : 1;
        decided = true;
      }
    }
  }
  return r;
}

// One LocateCoreMultiples call for one read (SfxArray.cpp:5806-6369), fast form.  Returns tHRslt or K4_NEED_SLOW.
template <int EL, int NCH, typename KT, bool CAPTURE>
K4_DEV int k4d_lcm_fast(const K4AlignArgs& a, const K4Lane<NCH>& ln, int len, int allow_mm, int cl, int core_delta,
                        const K4ReadParams& rp, int* p_inst, int* p_low, int* p_nxt, k4_hit* hits,
                        uint32_t& n_lookup, uint32_t& n_probe, uint32_t& n_cand, bool defer_deep = false) {
  const K4DevIndex& ix = a.ix;
  if (*p_inst > rp.max_hits && *p_low == 0) return K4_HR_HITINSTS;  // :5889-5895 (unreachable for fresh reads)
  if (*p_inst >= 1 && *p_low == 0 && (*p_nxt - *p_low) < rp.mm_delta) return K4_HR_MMDELTA;
  K4State st;
  if (*p_inst <= 0 || *p_low < 0 || *p_nxt < 0) {
    st.inst = *p_inst = 0;
    st.low = *p_low = allow_mm + rp.mm_delta + 1;
    st.nxt = *p_nxt = st.low;
  } else {
    st.inst = *p_inst; st.low = *p_low; st.nxt = *p_nxt;
  }
  st.cur_hit = st.inst < rp.max_hits ? st.inst : -1;
  const int max_iter = ix.max_iter;
  const int64_t n = (int64_t)ix.n;
  const int kk = min((int)ix.k, cl);
  const int tshift = 2 * ((int)ix.k - kk);
  int s = rp.strand == K4_STRAND_CRICK ? 1 : 0;
  const int s_end = rp.strand == K4_STRAND_WATSON ? 0 : 1;
  bool stop = false;
  for (; s <= s_end && !stop; s++) {
    const char strand_c = s ? '-' : '+';
    int n_ded = 0;
    int cur_delta = core_delta;
    int slides = 0;
    int o_next = 0;
    bool more = true;
    bool first_group = true;  // its core 0 sits at offset 0: the memoised lookup
    while (more && !stop) {
      // The core offsets of a strand pass depend only on (len, cl, delta) (:5948-5959), so the k-mer table entries of
      // the next K4_PF cores are fetched together before any of them is searched: one memory round trip, not K4_PF.
      constexpr int PFN = NCH == 5 ? K4_PF5 : K4_PF;
      int oo[PFN];
      KT lb0[PFN], ps0[PFN], lb1[PFN];
      uint32_t sig[PFN];
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < PFN; j++) {
        oo[j] = 0; lb0[j] = 0; ps0[j] = 0; lb1[j] = 0; sig[j] = 0;
        if (cnt == j && slides < rp.max_slides && o_next <= len - cl && cur_delta > cl / 3) {
          if (o_next + cl + cur_delta > len) cur_delta = len - (o_next + cl);
          oo[j] = o_next;
          cnt++;
          slides++;
          o_next += cur_delta;
        }
      }
      more = cnt == PFN;
      bool memo_hit = false, memo_pending = false;
      int memo_mm = 0;
#pragma unroll
      for (int j = 0; j < PFN; j++) {
        if (j < cnt) {
          // (64-bit table: the first phase may have looked at a sub-bucket only, k + 2 bases deep -- what it noted holds for
          // cores at least that long)
          if (!CAPTURE && j == 0 && first_group && tshift == 0 && (sizeof(KT) == 4 || cl >= kk + 2)) {
            const uint64_t mv = ln.memo[s * K4_BS(NCH)];
            const int kind = (int)(mv >> 62), fm = (int)((mv >> 48) & 0x3FFF), mmv = (int)((mv >> 40) & 0xFF);
            if (kind == 1 || (kind == 2 && fm < cl)) continue;                   // nothing starts with this core
            if (kind == 2 && mmv != 0xFF) {                                      // exactly the suffix at pos does
              lb0[0] = 0; lb1[0] = 1; ps0[0] = (KT)(mv & 0xFFFFFFFFFFull);
              memo_hit = true; memo_mm = mmv;
              continue;
            }
          }
          const uint64_t code = ln.chunk_at(s, oo[j]) >> (64 - 2 * kk);
          uint64_t sub;
          k4d_ktab_fetch<KT>(ix, code << tshift, (code + 1) << tshift, lb0[j], ps0[j], sig[j], lb1[j], sub);
          if (CAPTURE && defer_deep && tshift == 0 && (uint64_t)(lb1[j] - lb0[j]) > K4_DEFER_BUCKET) return K4_DEFER;
          if (sizeof(KT) == 8 && tshift == 0 && cl >= kk + 2 && sub != K4_KTAB64_IRREGULAR) {
            // straight to the suffixes that continue with the core's next two bases
            uint32_t before, count;
            k4d_ktab_sub(sub, (uint32_t)(ln.chunk_at(s, oo[j] + kk) >> 60), before, count);
            if (before) ps0[j] = (KT)K4_KTAB64_MASK;  // pos0 is the whole bucket's first suffix, not this one's
            lb0[j] += (KT)before;
            lb1[j] = lb0[j] + (KT)count;
          }
          if (CAPTURE && j == 0 && first_group && tshift == 0) {
            if (lb1[0] == lb0[0]) ln.memo[s * K4_BS(NCH)] = k4d_memo_pack(1, 0, 0, 0);
            else memo_pending = lb1[0] == lb0[0] + 1;
          }
          // a bucket of one suffix whose next bases already disagree with the core cannot hold a match: drop it here
          if (sizeof(KT) == 4 && tshift == 0 && lb1[j] == lb0[j] + 1 && cl > kk) {
            const int nb = min(K4_SIG_BASES32, cl - kk);
            const uint32_t cb = (uint32_t)(ln.chunk_at(s, oo[j] + kk) >> 32);
            const uint32_t df = (cb ^ sig[j]) & (0xFFFFFFFFu << (32 - 2 * nb));
            if (df) {
              lb1[j] = lb0[j];
              if (CAPTURE && j == 0 && first_group) {
                ln.memo[s * K4_BS(NCH)] = k4d_memo_pack(2, (uint64_t)ps0[0], 0xFF, kk + (__clz(df) >> 1));
                memo_pending = false;
              }
            }
          }
        }
      }
      // touch the first word of every non-empty bucket's first window now: the lines are on their way (and land in
      // L2) while the cores are searched one after the other below
      uint32_t touch = 0;
#pragma unroll
      for (int j = 0; j < PFN; j++)
        if (j < cnt && tshift == 0 && lb1[j] > lb0[j] && !(j == 0 && memo_hit) && (sizeof(KT) == 4 || (uint64_t)ps0[j] != K4_KTAB64_MASK))
          touch |= ix.ref2[((int64_t)ps0[j] - oo[j]) >> 4];

#pragma unroll
      for (int j = 0; j < PFN; j++) {
        if (j >= cnt || stop) continue;
        const int o = oo[j];
        n_lookup++;
        // ---- seed lookup: lower bound inside the k-mer bucket [lb0, lb1) -------------------------------------------
        int64_t lo = (int64_t)lb0[j];
        int64_t hi = (int64_t)lb1[j] - 1;
        const int64_t bucket_hi = hi;
        int64_t found = -1;
        uint64_t fpos = 0;
        int fmm = 0;
        while (lo <= hi) {
          const int64_t mid = (lo + hi) >> 1;
          // pos0 belongs to the bucket of the exact k-mer: with a core shorter than k the interval spans several
          // buckets and the first of them may be empty (pos0 unset), so the suffix array is read instead
          const uint64_t p = (tshift == 0 && mid == (int64_t)lb0[j] && (sizeof(KT) == 4 || (uint64_t)ps0[j] != K4_KTAB64_MASK))
                                 ? (uint64_t)ps0[j] : k4d_sa_at<EL>(ix, (uint64_t)mid);
          K4Probe pr;
          if (!CAPTURE && j == 0 && memo_hit) { pr.cmp = 0; pr.mm = memo_mm; pr.fm = len; pr.exc = false; }
          else {
            pr = k4d_probe<NCH>(ix, ln, s, o, cl, len, p);
            n_probe++;
            if (CAPTURE && j == 0 && memo_pending) ln.memo[s * K4_BS(NCH)] = k4d_memo_pack(2, p, pr.mm > 254 ? 254 : pr.mm, pr.fm);
          }
          if (pr.exc) return K4_NEED_SLOW;  // (handling it here instead costs the hot path 8 % in registers: measured)
          if (pr.cmp > 0) lo = mid + 1;
          else {
            if (pr.cmp == 0) { found = mid; fpos = p; fmm = pr.mm; }
            hi = mid - 1;
          }
        }
        if (found != lo) continue;  // no suffix starts with this core
        // ---- walk the run of equal cores in SA order, :5971-6321 -------------------------------------------------
        int64_t idx = found;
        uint64_t p = fpos;
        int mm = fmm;
        int iter = 0;
        bool first = true;
        while (!max_iter || iter < max_iter) {
          if (!first) {
            // a suffix beyond the bucket does not share the core's first kk bases: the reference's compare (:5986-6016)
            // would end the run there, so the probe is skipped
            if (idx >= bucket_hi || idx + 1 >= n) break;
            const uint64_t p2 = k4d_sa_at<EL>(ix, (uint64_t)idx + 1);
            if ((int64_t)p2 + cl > n) break;
            const K4Probe pr = k4d_probe<NCH>(ix, ln, s, o, cl, len, p2);
            n_probe++;
            if (pr.exc) return K4_NEED_SLOW;
            if (pr.cmp != 0) break;
            idx += 1; p = p2; mm = pr.mm;
          }
          first = false;
          if (p < (uint64_t)o) continue;
          const uint64_t left = p - (uint64_t)o;
          int e;
          uint64_t e_start, e_end;
          if (ix.n_entries <= K4_LDS_ENTRIES) {  // MapChunkHit2Entry (SfxArray.cpp:2609-2654) over the LDS copy
            int lo_e = 0, hi_e = (int)ix.n_entries - 1;
            e = -1; e_start = 0; e_end = 0;
            while (hi_e >= lo_e) {
              const int mid_e = (hi_e + lo_e) >> 1;
              const uint64_t sv = ln.ent[mid_e];
              if (sv > left) { hi_e = mid_e - 1; continue; }
              const uint64_t ev = ln.ent[K4_LDS_ENTRIES + mid_e];
              if (ev >= left) { e = mid_e; e_start = sv; e_end = ev; break; }
              lo_e = mid_e + 1;
            }
          } else {
            e = k4d_map_entry(ix, left);
            e_start = e >= 0 ? ix.ent_start[e] : 0;
            e_end = e >= 0 ? ix.ent_end[e] : 0;
          }
          if (e < 0 || left + (uint64_t)len - 1 > e_end) continue;
          const uint32_t targ_id = (uint32_t)(1 + p - (uint32_t)o);  // :6037 (truncation is the reference's)
          bool dup = false;
          for (int q = 0; q < n_ded; q++) dup |= (ln.ded[q * K4_BS(NCH)] == targ_id);
          if (dup) continue;
          if (n_ded >= K4_DEDUP_CAP) return K4_NEED_SLOW;
          ln.ded[n_ded * K4_BS(NCH)] = targ_id;
          n_ded++;
          iter++;
          n_cand++;
          if (mm > allow_mm || mm >= st.nxt) continue;  // the two early-outs of :6200-6261
          k4d_fold(st, mm, hits, rp.max_hits, ix.ent_id[e], (uint32_t)(left - e_start), len, strand_c);
          if (st.inst > rp.max_hits && st.low == 0) break;
        }
        if (st.inst > rp.max_hits && st.low == 0) stop = true;
      }
      if (touch == 0x5A5A5A5Au && len < 0) n_probe++;  // never true: only keeps the touch loads alive until here
      first_group = false;
    }
  }
  return k4d_lcm_result(*p_inst, *p_low, p_nxt, st, rp.mm_delta, rp.max_hits, p_inst, p_low);
}

// reverse the order of the 32 two-bit groups of a word
K4_DEV uint64_t k4d_rev2(uint64_t x) {
  uint64_t y = __brevll(x);
  return ((y & 0x5555555555555555ull) << 1) | ((y >> 1) & 0x5555555555555555ull);
}

// etSeqBase bytes -> this lane's LDS column: forward words [0][c], reverse-complement words [1][c], pad words zero.
// Returns K4_RF_* flags; n_ns = number of N (the fast path cannot express N: such reads go to the general kernel).
template <int NCH>
K4_DEV uint32_t k4d_pack_read(const uint8_t* __restrict__ src, int len, uint64_t* col, uint32_t& n_ns) {
  constexpr int NW = NCH + 1;
  uint32_t fl = 0;
  n_ns = 0;
  // reads sit at any byte offset (150-byte reads back to back, parsed FASTQ): always load aligned words -- the word that
  // holds the read's first byte may start up to 3 bytes before it, the one that holds its last byte may end up to 3 bytes
  // after it, neither leaves the 4-byte cells the read itself occupies -- and shift the pair into place
  const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3);
  const uint32_t* __restrict__ src32 = reinterpret_cast<const uint32_t*>(src - sh);
  const int span = len + (int)sh;  // bytes from src32 to the end of the read
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    uint64_t acc = 0;
    if (32 * c < len) {
      uint32_t dq[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (c * 32 + 32 <= span) {  // two 16-byte loads
        uint32_t d8[8];
        k4d_load_words<8>(src32 + c * 8, d8);
#pragma unroll
        for (int q = 0; q < 8; q++) dq[q] = d8[q];
      } else {
#pragma unroll
        for (int q = 0; q < 8; q++)
          if (c * 32 + q * 4 < span) dq[q] = src32[c * 8 + q];
      }
      if (c * 32 + 32 < span) dq[8] = src32[c * 8 + 8];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int base = c * 32 + q * 4;
        uint32_t d = __builtin_amdgcn_alignbyte(dq[q + 1], dq[q], sh);
        d &= 0x07070707u;
        if (base + 4 > len) d &= len > base ? (0xFFFFFFFFu >> (8 * (4 - (len - base)))) : 0u;  // bytes past the end
        // N / invalid detection on the four bytes at once: bit 2 set => symbol >= 4
        const uint32_t hi4 = d & 0x04040404u;
        if (hi4) {
          const uint32_t inval = hi4 & ((d << 1) | (d << 2)) & 0x04040404u;  // 5,6,7: bit2 and (bit1 or bit0)
          if (inval) fl |= K4_RF_INVALID;
          n_ns += __popc(hi4 & ~inval);
          d &= ~(hi4 | (hi4 >> 1) | (hi4 >> 2));  // such symbols pack as 0
        }
        acc = (acc << 8) | ((d & 3) << 6) | (((d >> 8) & 3) << 4) | (((d >> 16) & 3) << 2) | ((d >> 24) & 3);
      }
    }
    col[c * K4_BS(NCH)] = acc;
  }
  col[NCH * K4_BS(NCH)] = 0;
  if (n_ns) fl |= K4_RF_HAS_N;
  // reverse complement from the stored forward words: rc bases [32c, 32c+32) = complement of forward bases
  // [len-32(c+1), len-32c) in reverse order
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    uint64_t r = 0;
    if (32 * c < len) {
      const int o = len - 32 * (c + 1);
      uint64_t f;
      if (o >= 0) {
        const int w = o >> 5, sh = 2 * (o & 31);
        const uint64_t hi = col[w * K4_BS(NCH)];
        f = sh ? (hi << sh) | (col[(w + 1) * K4_BS(NCH)] >> (64 - sh)) : hi;
      } else
        f = col[0] >> (2 * (-o));  // fewer than 32 bases left: they sit at the low end, zeros above
      r = k4d_rev2(~f) & k4d_range_mask(0, len - 32 * c);
    }
    col[(NW + c) * K4_BS(NCH)] = r;
  }
  col[(NW + NCH) * K4_BS(NCH)] = 0;
  return fl;
}

// One AlignReads phase per launch.  FIRST: lanes take reads j = 0..n_reads-1 and pack them; later steps take the
// compacted survivors (ids + packed rows) of the previous step.
template <int EL, int NCH, bool FIRST, typename KT, bool DEFER = false>
__global__ void __launch_bounds__(K4_BS(NCH), (NCH >= 16 ? K4_STEP_WAVES_LONG : NCH >= 8 ? K4_STEP_WAVES_MID : NCH == 5 ? K4_STEP_WAVES_5 : K4_STEP_WAVES)) k4k_align_step(K4AlignArgs a, int step, const uint32_t* __restrict__ in_ids,
                                                      const uint64_t* __restrict__ in_rows,
                                                      const uint32_t* __restrict__ in_count, uint32_t* __restrict__ out_ids,
                                                      uint64_t* __restrict__ out_rows, uint32_t* __restrict__ out_count) {
  extern __shared__ uint64_t lds[];
  constexpr int NW = NCH + 1;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  K4Lane<NCH> ln;
  ln.rd = lds + tid;
  ln.ded = reinterpret_cast<uint32_t*>(lds + 2 * NW * K4_BS(NCH)) + tid;
  ln.memo = lds + 2 * NW * K4_BS(NCH) + (K4_DEDUP_CAP * K4_BS(NCH)) / 2 + tid;
  uint64_t* ent_l = lds + 2 * NW * K4_BS(NCH) + (K4_DEDUP_CAP * K4_BS(NCH)) / 2 + 2 * K4_BS(NCH);
  uint32_t* sup_l = reinterpret_cast<uint32_t*>(ent_l + 2 * K4_LDS_ENTRIES);
  ln.ent = ent_l;
  ln.sup = sup_l;
  for (int q = tid; q < K4_SUP_WORDS; q += K4_BS(NCH)) sup_l[q] = a.ix.excsup[q];
  if (a.ix.n_entries <= K4_LDS_ENTRIES)
    for (int q = tid; q < (int)a.ix.n_entries; q += K4_BS(NCH)) {
      ent_l[q] = a.ix.ent_start[q];
      ent_l[K4_LDS_ENTRIES + q] = a.ix.ent_end[q];
    }
  __syncthreads();
  uint64_t* col = lds + tid;
  uint32_t n_lookup = 0, n_probe = 0, n_cand = 0, n_slow = 0, n_bases = 0, n_done = 0;
  const int64_t count = (FIRST && !(DEFER && in_ids != nullptr)) ? a.n_reads : (int64_t)*in_count;
  const int64_t stride = (int64_t)gridDim.x * K4_BS(NCH);
  // Survivor slots are reserved K4_CHUNK at a time (one atomic per chunk, not per wave iteration: a single counter
  // word serialises at ~88 atomics/us).  ch_cur / ch_left are wave-uniform: every lane of the wave runs every
  // iteration of this loop, inactive lanes masked, so the copies never diverge.  Unused slots hold K4_NO_READ.
  uint32_t ch_cur = 0, ch_left = 0;
  // FIRST comes in two launches: every read (in_ids null), then the reads the first one set aside (in_ids = that list)
  const bool from_list = !FIRST || (DEFER && in_ids != nullptr);
  const bool may_defer = DEFER && FIRST && in_ids == nullptr && a.defer_ids != nullptr;
  uint32_t dch_cur = 0, dch_left = 0;  // the deferred list's chunk, as ch_cur / ch_left
  for (int64_t jb = (int64_t)blockIdx.x * K4_BS(NCH) + (tid & ~63); jb < count; jb += stride) {
    const int64_t j = jb + lane;
    bool active = j < count;
    int64_t i = 0;
    if (active) {
      i = from_list ? (int64_t)in_ids[j] : j;
      if (from_list && (uint32_t)i == K4_NO_READ) active = false;
    }
    bool slow = false, survive = false, deferred = false;
    int ext_from = -1;  // >= 0: every standard phase ran here without a result; the general kernel starts at the optional ones
    if (active) {
      const int len = (int)a.lens[i];
      const K4ReadParams rp = k4d_read_params(a, len);
      bool skip = false;
      if (FIRST) {
        if (!from_list) {  // (a read taken from the deferred list was counted when it was set aside)
          n_done++;
          n_bases += (uint32_t)len;
        }
        uint32_t fl = 0, n_ns = 0;
        if (len < 1 || len > 32 * NCH || len > K4_MAX_FAST_READ_LEN) {
          fl = K4_RF_TOOLONG;
          if (a.mode == 1) {  // the N rule still applies
            const uint8_t* src = a.reads + a.offs[i];
            for (int q = 0; q < len; q++) {
              uint32_t b = src[q] & 7;
              if (b == 4) n_ns++; else if (b > 4) fl |= K4_RF_INVALID;
            }
          }
        } else
          fl = k4d_pack_read<NCH>(a.reads + a.offs[i], len, col, n_ns);
        if (a.mode == 1) {  // AlignRead: too many Ns / a symbol above N -> NAR Ns, KAligner.cpp:9618-9640
          int max_ns = 0;
          if (a.kp.max_ns) max_ns = max((len * a.kp.max_ns) / 100, a.kp.max_ns);
          if ((fl & K4_RF_INVALID) || (int)n_ns > max_ns) {
            k4_read_result r = {K4_HR_SEQERRS, 0, 0, 0, K4_NAR_NS, 0};
            a.rr[i] = r;
            if (!a.sparse_hits)
              for (int q = 0; q < a.max_hits; q++) *reinterpret_cast<uint4*>(&a.hits[i * a.max_hits + q]) = make_uint4(0, 0, 0, 0);
            skip = true;
          }
        }
        slow = fl != 0 && !skip;
        if (((a.mode == 1 && a.kp.pe_mode == 4) || a.best) && !skip) slow = true;  // LocateBestMatches lives in the general kernel
      } else {
        const uint64_t* row = in_rows + (int64_t)j * K4_ROW_WORDS(NCH);
        {
          const k4_u64x2_a8 mv = *reinterpret_cast<const k4_u64x2_a8*>(row + 2 * NCH);
          ln.memo[0] = mv.x; ln.memo[K4_BS(NCH)] = mv.y;
        }
        uint64_t rw[2 * NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) {  // rows are 16-byte aligned: NCH 16-byte loads
          const k4_u64x2_a8 v = *reinterpret_cast<const k4_u64x2_a8*>(row + 2 * c);
          rw[2 * c] = v.x; rw[2 * c + 1] = v.y;
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          col[c * K4_BS(NCH)] = rw[c];
          col[(NW + c) * K4_BS(NCH)] = rw[NCH + c];
        }
        col[NCH * K4_BS(NCH)] = 0;
        col[(NW + NCH) * K4_BS(NCH)] = 0;
      }
      if (!skip && (rp.core_len < 1 || rp.max_hits < 1 || rp.max_hits > a.max_hits)) slow = true;  // the general kernel reports it
      if (!skip && !slow) {
        // which AlignReads phase is this read's step-th?  (SfxArray.cpp:7867-7891)
        int n_esc = 0;
        if (rp.tot_mm > 0)
          for (; n_esc <= rp.tot_mm; n_esc++)
            if (len / (n_esc + rp.mm_delta) <= rp.core_len) break;
        const bool has_final = rp.tot_mm > 0 ? n_esc <= rp.tot_mm : true;
        const int n_phases = n_esc + (has_final ? 1 : 0);
        int allow, cl, delta;
        if (step < n_esc) { allow = step; cl = len / (step + rp.mm_delta); delta = cl; }
        else { allow = rp.tot_mm; cl = rp.core_len; delta = rp.core_delta; }
        k4_hit* hits = a.hits + i * a.max_hits;
        int inst = 0, low = 0, nxt = 0;
        const uint32_t c0 = n_lookup, c1 = n_probe, c2 = n_cand;
        if (FIRST) { ln.memo[0] = K4_MEMO_NONE; ln.memo[K4_BS(NCH)] = K4_MEMO_NONE; }
        int rslt = k4d_lcm_fast<EL, NCH, KT, FIRST>(a, ln, len, allow, cl, delta, rp, &inst, &low, &nxt, hits, n_lookup, n_probe, n_cand, DEFER && may_defer);
        if (rslt == K4_NEED_SLOW) {  // the general kernel redoes (and tallies) this phase
          slow = true;
          n_lookup = c0; n_probe = c1; n_cand = c2;
        }
        else if (DEFER && rslt == K4_DEFER) {  // the second launch of this step runs the phase from its start (what was stored so far is stored again)
          deferred = true;
          n_lookup = c0; n_probe = c1; n_cand = c2;
        }
        else if (rslt == 0 && step + 1 >= n_phases && a.ext_on) { slow = true; ext_from = n_phases; }  // :7894-7930
        else if (rslt != 0 || step + 1 >= n_phases) k4d_finalize(a, i, len, rp, rslt, inst, low, nxt);
        else survive = true;
      }
      if (slow) {
        k4d_push_slow(a, i, ext_from >= 0 ? ext_from : step);
        n_slow++;
      }
    }
    if (FIRST && DEFER) {  // the reads set aside: ids only, in chunks like the survivors
      const unsigned long long dm = __ballot(deferred);
      const uint32_t dcnt = (uint32_t)__popcll(dm);
      if (dcnt) {
        if (dcnt > dch_left) {
          for (uint32_t q = lane; q < dch_left; q += 64) a.defer_ids[dch_cur + q] = K4_NO_READ;
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&a.ctl[K4_CTL_DEFER], (uint32_t)K4_CHUNK);
          dch_cur = __shfl(base, 0, 64);
          dch_left = K4_CHUNK;
        }
        if (deferred) a.defer_ids[dch_cur + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull))] = (uint32_t)i;
        dch_cur += dcnt;
        dch_left -= dcnt;
      }
    }
    // compaction of the survivors: slots by prefix popcount of the wave's ballot inside the wave's current chunk
    const unsigned long long m = __ballot(survive);
    const uint32_t cnt = (uint32_t)__popcll(m);
    if (cnt) {
      if (cnt > ch_left) {
        for (uint32_t q = lane; q < ch_left; q += 64) out_ids[ch_cur + q] = K4_NO_READ;  // retire the old chunk's tail
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(out_count, (uint32_t)K4_CHUNK);
        ch_cur = __shfl(base, 0, 64);
        ch_left = K4_CHUNK;
      }
      if (survive) {
        const uint32_t slot = ch_cur + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        out_ids[slot] = (uint32_t)i;
        uint64_t* row = out_rows + (int64_t)slot * K4_ROW_WORDS(NCH);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
          k4_u64x2_a8 v;
          const int w0 = 2 * c, w1 = 2 * c + 1;
          v.x = w0 < NCH ? col[w0 * K4_BS(NCH)] : col[(NW + w0 - NCH) * K4_BS(NCH)];
          v.y = w1 < NCH ? col[w1 * K4_BS(NCH)] : col[(NW + w1 - NCH) * K4_BS(NCH)];
          *reinterpret_cast<k4_u64x2_a8*>(row + 2 * c) = v;
        }
        {
          k4_u64x2_a8 mv;
          mv.x = ln.memo[0]; mv.y = ln.memo[K4_BS(NCH)];
          *reinterpret_cast<k4_u64x2_a8*>(row + 2 * NCH) = mv;
        }
      }
      ch_cur += cnt;
      ch_left -= cnt;
    }
  }
  for (uint32_t q = lane; q < ch_left; q += 64) out_ids[ch_cur + q] = K4_NO_READ;
  if (FIRST && DEFER)
    for (uint32_t q = lane; q < dch_left; q += 64) a.defer_ids[dch_cur + q] = K4_NO_READ;
  // per-wave tallies -> one atomic per counter per wave (lookups of reads that went slow are recounted there)
  {
    unsigned long long v[6] = {n_done, n_lookup, n_probe, n_cand, n_slow, n_bases};
#pragma unroll
    for (int q = 0; q < 6; q++) {
      unsigned long long x = v[q];
      for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
      if (lane == 0 && x) atomicAdd(&a.counters[q], x);
    }
  }
}

// ==== general kernel =================================================================================================
// One WAVE per read: the literal LocateCoreMultiples / AlignReads control flow (wave-uniform), with the two inner loops
// of the reference -- the core comparison and the Hamming extension -- spread over the 64 lanes on exact 4-bit symbols.
// It takes whatever the 2-bit fast path cannot decide: N in the read, windows touching N runs / separators, deep repeats
// (more candidates than the fast path's dedupe list), reads longer than 512 bp.
struct K4Slow {
  uint8_t* probe;   // LDS: the probe, reverse-complemented in place like the reference does
  uint64_t* hash;   // HBM scratch of this wave: (generation << 32 | TargSeqID), open addressing
  uint32_t cap;     // power of two
  uint32_t gen;
  int lane;
  const uint64_t* ent;  // LDS copy of the entry table (starts, then ends at +K4_LDS_ENTRIES) or null
  uint64_t* pk;         // LDS: the probe in its current orientation as 2-bit words, MSB first, zero word behind the end
  bool packed;          // pk is usable: the probe holds only A/C/G/T
  bool small;       // first general pass: small tables, overflow defers the read to the pass with big tables
  const uint32_t* sup;  // LDS copy of the coarse exception bitmap (K4_SUP_WORDS words)
  const uint32_t* ent_id;  // entry ids: LDS copy when the entry table is in LDS, the index's array otherwise
  uint32_t* lhash;   // first general pass over 4-byte suffix elements: the dedupe table in LDS (ids only, cleared per strand pass)
  uint32_t lcap;     // its slots (power of two); lused: slots taken so far in this strand pass, retracted inserts included
  uint32_t lused;
  // the batched LocateCoreMultiples (k4d_lcm_batched) looks at both strands in one go: the reverse complement of the probe sits
  // behind the forward one -- bytes at probe + pstride, packed words at pk + pkstride -- and never changes while a read is worked on
  uint32_t pstride, pkstride;
  uint64_t* g_lb;    // LDS [K4_GROUP]: first suffix-array index of (strand, core) pair j's bucket / run
  uint64_t* g_pre;   // LDS [K4_GROUP + 1]: slots in front of pair j; [pairs] = slots of the group
  uint16_t* g_o;     // LDS [K4_GROUP]: core offset of pair j
#ifdef K4_SLOW_PROF
  unsigned long long prof[16];
#endif
};

// CmpProbeTarg (SfxArray.cpp:2508-2525): lanes compare 64 symbols at a time, the first differing / EOS position decides
K4_DEV int k4d_cmp_wave(const K4DevIndex& ix, const K4Slow& sc, int o, uint64_t pos, int len) {
  for (int j0 = 0; j0 < len; j0 += 64) {
    const int j = j0 + sc.lane;
    uint32_t t = 7, p = 0;
    bool diff = false;
    if (j < len) {
      t = pos + j < ix.n ? k4d_ref_base(ix, pos + j) : 7u;
      p = sc.probe[o + j] & 0x0f;
      diff = t == 7 || p != t;
    }
    const unsigned long long m = __ballot(diff);
    if (m) {
      const int f = __ffsll((long long)m) - 1;
      const uint32_t tf = __shfl(t, f, 64), pf = __shfl(p, f, 64);
      if (tf == 7) return -1;
      return pf > tf ? 1 : -1;
    }
  }
  return 0;
}

K4_DEV void k4d_pack_probe_wave(K4Slow& sc, int len);
K4_DEV void k4d_revcomp_wave(K4Slow& sc, int len) {  // CSeqTrans::ReverseComplement, SeqTrans.cpp:497-545
  // complement stops at the first symbol that is not a base / N / InDel / Undef (values > 6): reads hold 0..7 here
  int stop = len;
  for (int j0 = 0; j0 < len; j0 += 64) {
    const int j = j0 + sc.lane;
    const bool bad = j < len && (sc.probe[j] & 0x0f) > 6;
    const unsigned long long m = __ballot(bad);
    if (m) { stop = j0 + __ffsll((long long)m) - 1; break; }
  }
  for (int j = sc.lane; j < stop; j += 64) {
    const uint8_t b = sc.probe[j];
    if (b <= 3) sc.probe[j] = 3 - b;
  }
  __syncthreads();
  for (int x = sc.lane; x < len / 2; x += 64) {
    const uint8_t t = sc.probe[x];
    sc.probe[x] = sc.probe[len - 1 - x];
    sc.probe[len - 1 - x] = t;
  }
  __syncthreads();
  k4d_pack_probe_wave(sc, len);
}

// Dedupe table of a strand pass (tsIdentNode, SfxArray.cpp:5946,6037-6058): open addressing on (generation, TargSeqID).
// One insert per lane, concurrently: the ids of one batch are distinct (one SA run, one core offset), so the only
// interaction between lanes is the race for a free slot, which the compare-and-swap settles.  Returns whether the id was
// new in this strand pass and the slot it occupies (for k4d_hash_retract).
// The LDS form (K4Slow::lhash): 32-bit slots holding the id itself; TargSeqID = 1 + offset stays below both markers while
// suffix elements are 4 bytes (offsets < 4 000 000 000).  No generation: k4d_hash_new_pass clears it.  An LDS compare-and-swap
// costs a few hundred cycles where the HBM table's load + compare-and-swap cost two round trips to L2.
#define K4_LH_EMPTY 0xFFFFFFFFu
#define K4_LH_TOMB 0xFFFFFFFEu
#define K4_LDS_HASH 1024
K4_DEV void k4d_hash_new_pass(K4Slow& sc) {
  sc.gen++;
  if (sc.lhash) {
    __syncthreads();  // (blocks are one wave)
    for (uint32_t q = sc.lane; q < sc.lcap; q += 64) sc.lhash[q] = K4_LH_EMPTY;
    sc.lused = 0;
    __syncthreads();
  }
}
K4_DEV bool k4d_hash_insert_lane(const K4Slow& sc, uint32_t id, uint32_t& slot) {
  if (sc.lhash) {
    uint32_t h = (id * 2654435761u) >> 22 & (sc.lcap - 1);
    for (;;) {
      uint32_t v = sc.lhash[h];
      if (v == K4_LH_EMPTY) {
        v = atomicCAS(&sc.lhash[h], K4_LH_EMPTY, id);
        if (v == K4_LH_EMPTY) { slot = h; return true; }
      }
      if (v == id) { slot = h; return false; }
      h = (h + 1) & (sc.lcap - 1);
    }
  }
  const unsigned long long key = ((unsigned long long)sc.gen << 32) | id;
  unsigned long long* tab = reinterpret_cast<unsigned long long*>(sc.hash);
  uint32_t h = (id * 2654435761u) & (sc.cap - 1);
  for (;;) {
    // a plain (possibly stale) read is enough: a slot only ever moves from an older generation to the current one, so
    // a stale "free" is caught by the compare-and-swap failing, and what it returns is then examined like a fresh read
    unsigned long long v = tab[h];
    if ((uint32_t)(v >> 32) != sc.gen) {
      const unsigned long long old = atomicCAS(&tab[h], v, key);
      if (old == v) { slot = h; return true; }
      v = old;
      if ((uint32_t)(v >> 32) != sc.gen) continue;  // (cannot happen: slots only move to the current generation)
    }
    if (v == key) { slot = h; return false; }
    h = (h + 1) & (sc.cap - 1);
  }
}
// an insert that the reference would not have made (its walk had already stopped): the slot stays occupied for this
// generation so that probe chains through it stay intact, but holds the impossible id 0 (TargSeqID is 1 + offset)
K4_DEV void k4d_hash_retract(const K4Slow& sc, uint32_t slot) {
  if (sc.lhash) { sc.lhash[slot] = K4_LH_TOMB; return; }
  atomicExch(reinterpret_cast<unsigned long long*>(sc.hash) + slot, (unsigned long long)sc.gen << 32);
}

// A divergent wave pays per lane-request, not per byte (k4_device.h): the general kernel's lanes therefore fetch a
// candidate's window with 16-byte loads (nine words hold 128 bases at any alignment) and ask the LDS copy of the coarse
// exception bitmap before the fine one in L2, as the fast kernel's k4d_probe does.
K4_DEV bool k4d_any_exc_sup(const K4DevIndex& ix, const uint32_t* sup, int64_t start, int64_t end) {
  if (start < 0) start = 0;
  if (end <= start) return false;
  const uint64_t b0 = (uint64_t)start >> ix.sup_shift, b1 = (uint64_t)(end - 1) >> ix.sup_shift;
  const uint64_t v = (((uint64_t)sup[(b0 >> 5) + 1] << 32) | sup[b0 >> 5]) >> (b0 & 31);
  bool f = (v & ((2ull << (b1 - b0)) - 1ull)) != 0;
  if (f && ix.sup_shift != K4_EXC_SHIFT) f = k4d_any_exc(ix, start, end);
  return f;
}
// One lane: compare probe[j] with the target symbol at left + j for j in [jlo, jhi).  all_eq: every symbol equal and no
// target EOS (CmpProbeTarg == 0 when the range is a core); mm: number of unequal symbols (N == N is equal, :6202-6234);
// eos: the range holds a target EOS.  Exact symbols: when no 256-base block of the range is flagged the packed words are
// fetched eight at a time (independent loads, one memory latency per 128 bases) -- the 2 Kbase pads make the over-read
// safe; otherwise symbol by symbol through the nibble store.  stop_early: return at the first difference.
K4_DEV void k4d_lane_range(const K4DevIndex& ix, const uint8_t* probe, int jlo, int jhi, uint64_t left, bool stop_early,
                           bool& all_eq, bool& eos, int& mm) {
  all_eq = true;
  eos = false;
  mm = 0;
  if (jhi <= jlo) return;
  const uint64_t g0 = left + (uint64_t)jlo, g1 = left + (uint64_t)jhi;
  bool flagged = g1 > ix.n;
  for (uint64_t bb = g0 >> K4_EXC_SHIFT; !flagged && bb <= ((g1 - 1) >> K4_EXC_SHIFT); bb++)
    flagged = (ix.excbm[bb >> 5] >> (bb & 31)) & 1;
  if (!flagged) {
    const uint64_t w0 = g0 >> 4, w1 = (g1 - 1) >> 4;
    int j = jlo;
    for (uint64_t wb = w0; wb <= w1; wb += 8) {
      uint32_t wv[8];
      k4d_load_words<8>(ix.ref2 + wb, wv);
      const uint64_t gend = min(g1, (wb + 8) << 4);
      for (uint64_t g = left + (uint64_t)j; g < gend; g++, j++) {
        const uint32_t t = (wv[(g >> 4) - wb] >> (30 - 2 * (uint32_t)(g & 15))) & 3;
        if ((probe[j] & 0x0f) != t) {
          all_eq = false;
          mm++;
          if (stop_early) return;
        }
      }
    }
    return;
  }
  for (int j = jlo; j < jhi; j++) {
    const uint64_t g = left + (uint64_t)j;
    const uint32_t t = g < ix.n ? k4d_ref_base(ix, g) : 7u;
    if (t == 7) eos = true;
    if ((probe[j] & 0x0f) != t) {  // (a target EOS never equals a probe symbol)
      all_eq = false;
      mm++;
      if (stop_early) return;
    }
  }
}

// MapChunkHit2Entry (libkit4b/SfxArray.cpp:2609-2654) over the LDS copy of the entry table when there is one
K4_DEV int k4d_map_entry_slow(const K4DevIndex& ix, const uint64_t* ent_lds, uint64_t ofs, uint64_t& e_start, uint64_t& e_end) {
  if (!ent_lds) {
    const int e = k4d_map_entry(ix, ofs);
    e_start = e >= 0 ? ix.ent_start[e] : 0;
    e_end = e >= 0 ? ix.ent_end[e] : 0;
    return e;
  }
  int lo = 0, hi = (int)ix.n_entries - 1;
  while (hi >= lo) {
    const int mid = (hi + lo) >> 1;
    const uint64_t s = ent_lds[mid];
    if (s > ofs) { hi = mid - 1; continue; }
    const uint64_t ev = ent_lds[K4_LDS_ENTRIES + mid];
    if (ev >= ofs) { e_start = s; e_end = ev; return mid; }
    lo = mid + 1;
  }
  e_start = e_end = 0;
  return -1;
}

// probe bytes -> sc.pk (call after every change of sc.probe); sc.packed = no symbol above T
K4_DEV void k4d_pack_probe_wave(K4Slow& sc, int len) {
  const int nw = (len + 31) >> 5;
  bool bad = false;
  for (int w = sc.lane; w <= nw; w += 64) {
    uint64_t acc = 0;
    if (w < nw)
      for (int q = 0; q < 32; q++) {
        const int j = 32 * w + q;
        uint32_t b = j < len ? (sc.probe[j] & 0x0f) : 0u;
        if (b > 3) { bad = true; b = 0; }
        acc = (acc << 2) | b;
      }
    sc.pk[w] = acc;
  }
  sc.packed = __ballot(bad) == 0;
  __syncthreads();
}
K4_DEV uint64_t k4d_probe_chunk(const K4Slow& sc, int j, int s = 0) {  // 32 probe bases from base j (s = 1: of the reverse complement)
  const uint64_t* pk = sc.pk + (s ? sc.pkstride : 0u);
  const int w = j >> 5, sh = 2 * (j & 31);
  const uint64_t hi = pk[w];
  return sh ? (hi << sh) | (pk[w + 1] >> (64 - sh)) : hi;
}

// Hamming distance of the packed probe against the window [left, left + len) (no exception in it): two 16-byte loads per
// 113 bases
// first: the words of the first 128 bases when the caller fetched them already (k4d_ref_words9 with c0 = 0)
K4_DEV int k4d_lane_hamming(const K4DevIndex& ix, const K4Slow& sc, int len, int64_t left, const uint32_t (*first)[9] = nullptr) {
  int mm = 0;
  const int a = (int)(left & 15);
  for (int c0 = 0; 32 * c0 < len; c0 += 4) {
    const int rem = len - 32 * c0;
    uint64_t rc[4];
    if (first && c0 == 0) k4d_words_to_chunks4(*first, left, rc);
    else k4d_ref_chunks4(ix, left, c0, rem + a <= 128, rc);
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (32 * c < rem) mm += (int)k4d_mm_count((rc[c] ^ k4d_probe_chunk(sc, 32 * (c0 + c))) & k4d_range_mask(0, rem - 32 * c));
  }
  return mm;
}

// CmpProbeTarg (SfxArray.cpp:2508-2525) by one lane: core [o, o+cl) of the probe against the suffix at pos; 0 equal,
// 1 probe greater, -1 probe smaller (a target EOS, or the end of the block, sorts above every probe symbol)
K4_DEV int k4d_lane_cmp(const K4DevIndex& ix, const K4Slow& sc, int o, int cl, uint64_t pos, int s = 0) {
  const uint8_t* probe = sc.probe + (s ? sc.pstride : 0u);
  if (sc.packed && pos + (uint64_t)cl <= ix.n && !k4d_any_exc_sup(ix, sc.sup, (int64_t)pos, (int64_t)pos + cl)) {
    // XOR of packed chunks, MSB-first order == symbol order.  One 16-byte load holds the first 49 bases or more: most
    // comparisons end there.
    const int al = (int)(pos & 15);
    {
      uint32_t w[4];
      k4d_load_words<4>(ix.ref2 + (pos >> 4), w);
      const uint32_t sh = 2 * (uint32_t)al;
      const uint64_t hi0 = ((uint64_t)w[0] << 32) | w[1], hi1 = ((uint64_t)w[2] << 32) | w[3];
      uint64_t m = k4d_range_mask(0, cl);
      uint64_t rc = (sh ? (hi0 << sh) | (w[2] >> (32 - sh)) : hi0) & m, pc = k4d_probe_chunk(sc, o, s) & m;
      if (rc != pc) return pc > rc ? 1 : -1;
      if (cl <= 32) return 0;
      if (cl <= 64 - al) {  // (what the fifth word would add lies behind the core)
        m = k4d_range_mask(0, cl - 32);
        rc = (hi1 << sh) & m; pc = k4d_probe_chunk(sc, o + 32, s) & m;
        return rc == pc ? 0 : pc > rc ? 1 : -1;
      }
    }
    for (int c0 = 1; 32 * c0 < cl; c0 += 4) {
      const int rem = cl - 32 * c0;
      uint64_t rc4[4];
      k4d_ref_chunks4(ix, (int64_t)pos, c0, rem + al <= 128, rc4);
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (32 * c < rem) {
          const uint64_t m = k4d_range_mask(0, rem - 32 * c);
          const uint64_t rc = rc4[c] & m, pc = k4d_probe_chunk(sc, o + 32 * (c0 + c), s) & m;
          if (rc != pc) return pc > rc ? 1 : -1;
        }
    }
    return 0;
  }
  bool flagged = pos + (uint64_t)cl > ix.n;
  for (uint64_t bb = pos >> K4_EXC_SHIFT; !flagged && bb <= ((pos + cl - 1) >> K4_EXC_SHIFT); bb++)
    flagged = (ix.excbm[bb >> 5] >> (bb & 31)) & 1;
  if (!flagged) {
    const uint64_t w1 = (pos + cl - 1) >> 4;
    int j = 0;
    for (uint64_t wb = pos >> 4; wb <= w1; wb += 8) {
      uint32_t wv[8];
      k4d_load_words<8>(ix.ref2 + wb, wv);
      const uint64_t gend = min(pos + (uint64_t)cl, (wb + 8) << 4);
      for (uint64_t g = pos + (uint64_t)j; g < gend; g++, j++) {
        const uint32_t t = (wv[(g >> 4) - wb] >> (30 - 2 * (uint32_t)(g & 15))) & 3;
        const uint32_t pb = probe[o + j] & 0x0f;
        if (pb != t) return pb > t ? 1 : -1;
      }
    }
    return 0;
  }
  for (int j = 0; j < cl; j++) {
    const uint64_t g = pos + (uint64_t)j;
    const uint32_t t = g < ix.n ? k4d_ref_base(ix, g) : 7u;
    const uint32_t pb = probe[o + j] & 0x0f;
    if (t == 7) return -1;
    if (pb != t) return pb > t ? 1 : -1;
  }
  return 0;
}

// LocateFirstExact (SfxArray.cpp:7938-8058): index+1 of the lowest suffix that starts with the core, or 0.  The k-mer
// table narrows the range to the core's bucket; inside it the 64 lanes compare 64 evenly spaced suffixes at once, so a
// bucket of up to 64 suffixes is settled in one round of memory accesses and one of 4096 in two (the reference's binary
// search takes one dependent round per halving).
template <int EL>
K4_DEV int64_t k4d_first_exact_wave(const K4DevIndex& ix, const K4Slow& sc, int o, int cl, uint32_t& n_probe) {
  int64_t lo = 0, hi = (int64_t)ix.n - 1;
  const int kk = min((int)ix.k, cl);
  bool acgt = true;
  uint64_t code = 0;
  for (int j = 0; j < kk; j++) {  // uniform: every lane reads the same LDS bytes
    const uint32_t b = sc.probe[o + j] & 0x0f;
    if (b > 3) { acgt = false; break; }
    code = (code << 2) | b;
  }
  if (acgt) {
    const int sh = 2 * ((int)ix.k - kk);
    lo = (int64_t)k4d_ktab_lb(ix, code << sh);
    hi = (int64_t)k4d_ktab_lb(ix, (code + 1) << sh) - 1;
  }
  int64_t found = -1;
  while (lo <= hi) {
    const int64_t size = hi - lo + 1;
    const int64_t step = (size + 63) / 64;
    const int64_t pv = lo + (int64_t)sc.lane * step;  // this lane's pivot (ascending with the lane)
    const bool have = pv <= hi;
    int c = 1;
    if (have) c = k4d_lane_cmp(ix, sc, o, cl, k4d_sa_at<EL>(ix, (uint64_t)pv));
    const unsigned long long hm = __ballot(have);
    n_probe += (uint32_t)__popcll(hm);
    const unsigned long long le = __ballot(have && c <= 0);  // pivots whose suffix is not below the core
    if (!le) {  // every pivot is below the core: what is left lies behind the last one
      lo = lo + (int64_t)(__popcll(hm) - 1) * step + 1;
      continue;
    }
    const int f = __ffsll((long long)le) - 1;
    const int64_t pvf = lo + (int64_t)f * step;
    if (step == 1) {  // every suffix of the range was a pivot: f is the lowest that is not below the core
      if (__shfl(c, f, 64) == 0) found = pvf;
      break;
    }
    if (f > 0) lo = lo + (int64_t)(f - 1) * step + 1;
    hi = pvf;
  }
  return found >= 0 ? found + 1 : 0;
}

// Profiling build (-DK4_SLOW_PROF, tools/slow_prof.py): where the general kernel's cycles go, summed over waves into the
// slots behind k4_counters.  0 run search, 1 walk (suffix elements, entries, dedupe), 2 Hamming extension, 3 replay,
// 4 whole reads, 5 read set-up; 6 lookups, 7 pivots of the run searches, 10 in-bounds run members, 8 runs, 9 walk steps, 11 reads, 12 run members, 13..15 reads that arrive with
// 0, 1, 2 or more phases already done by the fast kernel.
#ifdef K4_SLOW_PROF
#define K4_PROF_T(v) const long long v = clock64()
#define K4_PROF_ADD(slot, x) do { sc.prof[slot] += (unsigned long long)(x); } while (0)  // flushed once per wave
#else
#define K4_PROF_T(v)
#define K4_PROF_ADD(slot, x)
#endif

// The whole run of suffixes that start with the core, [first, last] (first > last: none), for the walk of
// LocateCoreMultiples: the reference finds the first by LocateFirstExact and then compares suffix after suffix until one
// differs (:5971-6016) -- one random window per suffix visited, which is what a read from a 400-copy repeat family spends
// its time on.  The suffix array is sorted by the very comparison that loop uses, so the run is the interval between two
// lower bounds (first suffix not below the core, first suffix above it); lanes 0..31 search the one and lanes 32..63 the
// other in the same rounds, 32 evenly spaced pivots each: a bucket of 32 suffixes is settled in one round of memory
// accesses, one of 1024 in two.  end_cmp: would the reference have compared the suffix behind the run (it does not when
// there is none or when it is closer than the core length to the end of the block, :5981-5985).
template <int EL>
K4_DEV void k4d_exact_run_wave(const K4DevIndex& ix, K4Slow& sc, int o, int cl, uint32_t& n_probe, int64_t& first,
                               int64_t& last, bool& end_cmp, int s = 0) {
  int64_t lo = 0, hi = (int64_t)ix.n - 1;
  const int kk = min((int)ix.k, cl);
  bool acgt = true;
  uint64_t code = 0;
  if (sc.packed)  // the k-mer straight from the packed probe (two LDS words instead of kk byte reads)
    code = k4d_probe_chunk(sc, o, s) >> (64 - 2 * kk);
  else
    for (int j = 0; j < kk; j++) {  // uniform: every lane reads the same LDS bytes
      const uint32_t b = sc.probe[(s ? sc.pstride : 0u) + o + j] & 0x0f;
      if (b > 3) { acgt = false; break; }
      code = (code << 2) | b;
    }
  if (acgt) {
    const int sh = 2 * ((int)ix.k - kk);
    lo = (int64_t)k4d_ktab_lb(ix, code << sh);
    hi = (int64_t)k4d_ktab_lb(ix, (code + 1) << sh) - 1;
  }
  // search h (0: lowest index whose suffix is not below the core, 1: lowest whose suffix is above it): the answer lies in
  // [slo[h], shi[h] + 1]; everything below slo[h] fails the predicate, shi[h] + 1 passes it or is the end of the bucket
  int64_t slo[2] = {lo, lo}, shi[2] = {hi, hi}, ans[2] = {hi + 1, hi + 1};
  bool open[2] = {lo <= hi, lo <= hi};
  const int half = sc.lane >> 5, hl = sc.lane & 31;
  while (open[0] || open[1]) {
    const int64_t my_lo = half ? slo[1] : slo[0], my_hi = half ? shi[1] : shi[0];
    const int64_t my_step = (my_hi - my_lo + 1 + 31) / 32;
    const int64_t pv = my_lo + (int64_t)hl * my_step;
    const bool have = (half ? open[1] : open[0]) && pv <= my_hi;
    int c = 1;
    if (have) c = k4d_lane_cmp(ix, sc, o, cl, k4d_sa_at<EL>(ix, (uint64_t)pv), s);
    const unsigned long long hm = __ballot(have);
    const unsigned long long pm = __ballot(have && (half ? c < 0 : c <= 0));
    n_probe += (uint32_t)__popcll(hm);
    K4_PROF_ADD(7, __popcll(hm));
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (!open[h]) continue;
      const uint32_t hm_h = (uint32_t)(hm >> (32 * h)), pm_h = (uint32_t)(pm >> (32 * h));
      const int64_t step = (shi[h] - slo[h] + 1 + 31) / 32;
      if (!pm_h) {  // every pivot fails: the answer lies behind the last one
        slo[h] += (int64_t)(__popc(hm_h) - 1) * step + 1;
        if (slo[h] > shi[h]) { ans[h] = shi[h] + 1; open[h] = false; }
        continue;
      }
      const int f = __ffs((int)pm_h) - 1;
      const int64_t pvf = slo[h] + (int64_t)f * step;
      if (f == 0 || step == 1) { ans[h] = pvf; open[h] = false; continue; }
      slo[h] = pvf - step + 1;  // behind the last failing pivot
      shi[h] = pvf - 1;         // (pvf itself passes)
      // (slo <= shi here: step > 1)
    }
  }
  first = ans[0];
  last = ans[1] - 1;
  end_cmp = false;
  if (last >= first && last + 1 < (int64_t)ix.n) end_cmp = (int64_t)k4d_sa_at<EL>(ix, (uint64_t)last + 1) + cl <= (int64_t)ix.n;
}

#include "k4_ext.h"

// CHIM: the chimeric branch (:6064-6189) -- every new in-bounds candidate is flank-trimmed by AdaptiveTrim (one lane each,
// its mismatch vector in the lane's column of mk) instead of being counted out by the Hamming extension, and the fold ranks
// by trimmed length first.  min_probe_chim = MinProbeChimericLen (:5880).
template <int EL, bool CHIM>
K4_DEV int k4d_lcm_slow(const K4AlignArgs& a, K4Slow& sc, int len, int allow_mm, int cl, int core_delta,
                        const K4ReadParams& rp, int* p_inst, int* p_low, int* p_nxt, k4_hit* hits,
                        uint32_t& n_lookup, uint32_t& n_probe, uint32_t& n_cand, int min_probe_chim = 0,
                        uint32_t* mk = nullptr) {
  const K4DevIndex& ix = a.ix;
  int best_len = 0, best_mms = 0;  // BestChimericLen / BestMaxChimericMMs: one per call, not per strand (:5936-5940)
  if (*p_inst > rp.max_hits && *p_low == 0) return K4_HR_HITINSTS;
  if (*p_inst >= 1 && *p_low == 0 && (*p_nxt - *p_low) < rp.mm_delta) return K4_HR_MMDELTA;
  K4State st;
  if (*p_inst <= 0 || *p_low < 0 || *p_nxt < 0) {
    st.inst = *p_inst = 0;
    st.low = *p_low = allow_mm + rp.mm_delta + 1;
    st.nxt = *p_nxt = st.low;
  } else {
    st.inst = *p_inst; st.low = *p_low; st.nxt = *p_nxt;
  }
  st.cur_hit = st.inst < rp.max_hits ? st.inst : -1;
  const int max_iter = ix.max_iter;
  const int lane = sc.lane;
  int strand = rp.strand;
  char cur_strand = '+';
  // hits are stored by lane 0 only (every lane folds the same wave-uniform state)
  k4_hit* hits_w = lane == 0 ? hits : nullptr;
  if (strand == K4_STRAND_CRICK) { k4d_revcomp_wave(sc, len); cur_strand = '-'; }
  do {
    int cur_delta = core_delta;
    int slides = 0;
    uint32_t n_nodes = 0;
    k4d_hash_new_pass(sc);
    // cMaxNumIdentNodes (SfxArray.h:15); additionally bounded by the scratch table so an insert always terminates.
    // A pass that fills a small table before the reference's own limit is redone with a big one (K4_NEED_SLOW).
    const uint32_t node_cap = min((uint32_t)K4_MAX_IDENT_NODES, sc.lhash ? sc.lcap * 3 / 4 : sc.cap / 2 - 1);
    for (int o = 0; slides < rp.max_slides && o <= len - cl && cur_delta > cl / 3 && n_nodes < node_cap;
         slides++, o += cur_delta) {
      if (o + cl + cur_delta > len) cur_delta = len - (o + cl);
      n_lookup++;
      int64_t t, t_last;
      bool end_cmp;
      K4_PROF_T(pt0);
      K4_PROF_ADD(6, 1);
      k4d_exact_run_wave<EL>(ix, sc, o, cl, n_probe, t, t_last, end_cmp);
      K4_PROF_T(pt1);
      K4_PROF_ADD(0, pt1 - pt0);
      if (t > t_last) continue;
      K4_PROF_ADD(8, 1);
      K4_PROF_ADD(12, t_last - t + 1);
      // The walk over the run of suffixes that start with the core (:5971-6321), 64 suffixes per step, one per lane.  Where
      // the run ends is known (k4d_exact_run_wave), so no suffix is compared with the core again; the memory-bound part
      // (suffix element, entry lookup, dedupe insert, Hamming distance of the new candidates) runs in parallel; what the
      // reference's sequential loop makes order-dependent -- MaxIter and the node limit counting only new in-bounds
      // candidates, the fold into (LowMMCnt, NxtLowMMCnt, instances, hits) and its early exit -- is then replayed in suffix
      // order from the lanes' results.
      int iter = 0;
      bool run_over = false, done_all = false, limit_seen = false;
      constexpr int KB = K4_SLOW_KB;  // suffixes per lane per step: KB * 64 per step, their memory accesses in flight together
      // (the suffix elements of a step are fetched during the step before it)
      uint64_t pos_ahead[KB];
#pragma unroll
      for (int k = 0; k < KB; k++) pos_ahead[k] = t + 64 * k + lane <= t_last ? k4d_sa_at<EL>(ix, (uint64_t)(t + 64 * k + lane)) : 0;
      for (int64_t base = t; !run_over; base += 64 * KB) {
        K4_PROF_T(ps0);
        K4_PROF_ADD(9, 1);
        uint64_t pos[KB];
        bool isnew[KB];
        int r[KB], e[KB], mm[KB];
        uint32_t loci[KB], slot[KB];
        bool eos[KB], clean[KB];
        uint32_t win[KB][9];
        unsigned long long newm[KB];
        // 1. suffix elements; members of sub-batch k are its lanes [0, r[k])
#pragma unroll
        for (int k = 0; k < KB; k++) {
          const int64_t idx_ahead = base + 64 * KB + 64 * k + lane;
          pos[k] = pos_ahead[k];
          pos_ahead[k] = idx_ahead <= t_last ? k4d_sa_at<EL>(ix, (uint64_t)idx_ahead) : 0;
          const int64_t left_in_run = t_last - (base + 64 * k) + 1;
          r[k] = left_in_run >= 64 ? 64 : left_in_run > 0 ? (int)left_in_run : 0;
        }
        const bool ended = base + 64 * KB > t_last;
        // the LDS table must keep room for this step's inserts (slots of retracted inserts count): else the pass with the big tables
        if (sc.lhash && sc.lused + 64 * KB + 1 > sc.lcap) {
          if (cur_strand == '-') k4d_revcomp_wave(sc, len);
          return K4_NEED_SLOW;
        }
        // 4. filters that precede the dedupe (:6019-6036: before the core offset, on a separator, over the entry end),
        //    then the dedupe insert, all lanes at once
#pragma unroll
        for (int k = 0; k < KB; k++) {
          const uint64_t left = pos[k] - (uint64_t)o;
          uint64_t e_start = 0, e_end = 0;
          e[k] = -1;
          const bool member = lane < r[k] && pos[k] >= (uint64_t)o;
          if (member) e[k] = k4d_map_entry_slow(ix, sc.ent, left, e_start, e_end);
          const bool in_bounds = member && e[k] >= 0 && left + (uint64_t)len - 1 <= e_end;
          loci[k] = (uint32_t)(left - e_start);
          isnew[k] = false;
          slot[k] = 0;
          // the window's first words are on their way while the dedupe table is probed (nearly every in-bounds member of
          // a run turns out to be new: the same trip, not one more)
          clean[k] = !CHIM && in_bounds && sc.packed && !k4d_any_exc_sup(ix, sc.sup, (int64_t)left, (int64_t)left + len);
          if (clean[k]) k4d_ref_words9(ix, (int64_t)left, 0, len + (int)(left & 15) <= 128, win[k]);
          if (in_bounds) isnew[k] = k4d_hash_insert_lane(sc, (uint32_t)(1 + pos[k] - (uint32_t)o), slot[k]);
          K4_PROF_ADD(10, __popcll(__ballot(in_bounds)));
          if (sc.lhash) sc.lused += (uint32_t)__popcll(__ballot(isnew[k]));  // (a retracted insert keeps its slot)
        }
        // 5. MaxIter / node limit: both count new in-bounds candidates only; the walk stops before the suffix after the
        //    last one it may take.  Inserts behind that point are retracted.
        const uint32_t rem_iter = max_iter ? (uint32_t)(max_iter - iter) : 0xFFFFFFFFu;
        uint32_t remaining = min(rem_iter, node_cap - n_nodes);
        bool hit_limit = false;
        int last[KB];  // last member of sub-batch k the reference's loop reaches (-1: none)
#pragma unroll
        for (int k = 0; k < KB; k++) {
          newm[k] = __ballot(isnew[k]);
          last[k] = r[k] - 1;
          if (hit_limit) {  // the walk ended in an earlier sub-batch
            if (isnew[k]) k4d_hash_retract(sc, slot[k]);
            newm[k] = 0; last[k] = -1; r[k] = 0;
            continue;
          }
          const uint32_t c = (uint32_t)__popcll(newm[k]);
          if (c >= remaining) {
            unsigned long long mrem = newm[k];
            for (uint32_t q = 1; q < remaining; q++) mrem &= mrem - 1;  // drop the lowest remaining-1 bits
            last[k] = __ffsll((long long)mrem) - 1;
            hit_limit = true;
            const unsigned long long beyond = last[k] >= 63 ? 0ull : (~0ull << (last[k] + 1));
            if (isnew[k] && ((beyond >> lane) & 1)) k4d_hash_retract(sc, slot[k]);
            newm[k] &= ~beyond;
          } else
            remaining -= c;
        }
        run_over = hit_limit || ended;
        limit_seen = hit_limit;
        // 6. the Hamming extension (:6200-6261) for the candidates that are new in this strand pass
        K4_PROF_T(ps1);
        K4_PROF_ADD(1, ps1 - ps0);
        K4Trim trim[KB];
#pragma unroll
        for (int k = 0; k < KB; k++) {
          mm[k] = 0;
          eos[k] = false;
          trim[k].len = trim[k].t5 = trim[k].t3 = trim[k].mms = 0;
          if (CHIM) {  // :6097 AdaptiveTrim(ProbeLen, probe, target, MinProbeChimericLen, MaxTotMM, 3 flank matches)
            if ((newm[k] >> lane) & 1) {
              k4d_build_mm_vector(ix, sc, len, pos[k] - (uint64_t)o, mk);
              trim[k] = k4d_adaptive_trim(mk, len, min_probe_chim, allow_mm, 3);
            }
            continue;
          }
          if ((newm[k] >> lane) & 1) {
            const uint64_t left = pos[k] - (uint64_t)o;
            if (clean[k]) {
              mm[k] = k4d_lane_hamming(ix, sc, len, (int64_t)left, &win[k]);
            } else {
              bool all_eq;
              k4d_lane_range(ix, sc.probe, 0, len, left, false, all_eq, eos[k], mm[k]);
            }
          }
        }
        // 7. replay in suffix order: only candidates that pass the order-independent part of the acceptance test can
        //    change the state; the rest just count
        K4_PROF_T(ps2);
        K4_PROF_ADD(2, ps2 - ps1);
#pragma unroll
        for (int k = 0; k < KB; k++) {
          if (done_all) break;
          const bool cand = CHIM ? ((newm[k] >> lane) & 1) && trim[k].len >= min_probe_chim && trim[k].len > 0
                                 : ((newm[k] >> lane) & 1) && !eos[k] && mm[k] <= allow_mm;
          unsigned long long todo = __ballot(cand);
          const uint32_t ent_id_l = cand ? sc.ent_id[e[k]] : 0u;  // (looked up by all lanes at once, not per candidate in the loop below)
          int stop_lane = -1;
          while (todo) {
            const int c = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            if (CHIM) {  // the fold of :6106-6188
              const int c_len = __shfl(trim[k].len, c, 64), c_mms = __shfl(trim[k].mms, c, 64);
              const int t5 = __shfl(trim[k].t5, c, 64), t3 = __shfl(trim[k].t3, c, 64);
              const uint32_t ent_c = (uint32_t)__shfl((int)ent_id_l, c, 64);
              const uint32_t loci_c = (uint32_t)__shfl((int)loci[k], c, 64);
              const uint32_t tl = cur_strand == '+' ? (uint32_t)t5 : (uint32_t)t3, tr = cur_strand == '+' ? (uint32_t)t3 : (uint32_t)t5;
              const uint32_t ext = K4_EXT_CHIMERIC | (tl & 0xFFFu) | ((tr & 0xFFFu) << 12);
              if (c_len > best_len || (c_len == best_len && c_mms < best_mms)) {
                if (best_len > 0 && c_len > best_len) st.low = c_mms + rp.mm_delta + 1;
                best_len = c_len; best_mms = c_mms;
                st.cur_hit = 0;
                st.inst = 1;
                st.nxt = st.low;
                st.low = c_mms;
                if (hits_w) k4d_store_hit(&hits_w[0], ent_c, loci_c, len, cur_strand, c_mms, ext);
              } else if (c_len == best_len && c_mms == best_mms) {
                st.inst += 1;
                if (st.cur_hit != -1 && st.inst <= rp.max_hits) {
                  st.cur_hit += 1;
                  if (hits_w && st.cur_hit < rp.max_hits) k4d_store_hit(&hits_w[st.cur_hit], ent_c, loci_c, len, cur_strand, c_mms, ext);
                }
              } else if (c_len == best_len && c_mms < st.nxt)
                st.nxt = c_mms;
              if (c_len == len && st.inst > rp.max_hits && st.low == 0) { stop_lane = c; break; }  // :6187
              continue;
            }
            const int mm_c = __shfl(mm[k], c, 64);
            if (mm_c >= st.nxt) continue;
            const uint32_t ent_c = (uint32_t)__shfl((int)ent_id_l, c, 64);
            const uint32_t loci_c = (uint32_t)__shfl((int)loci[k], c, 64);
            k4d_fold(st, mm_c, hits_w, rp.max_hits, ent_c, loci_c, len, cur_strand);
            if (st.inst > rp.max_hits && st.low == 0) { stop_lane = c; break; }
            // what is left of a repeat family's batch mostly cannot change the state any more: candidates at or above
            // NxtLowMMCnt are no-ops (it only ever drops), and once the hit slots are full a candidate that ties with
            // LowMMCnt only counts -- those in front of the next better one are counted in one go
            todo &= __ballot(mm[k] < st.nxt);
            if (st.inst >= rp.max_hits && st.low > 0) {
              const unsigned long long better = todo & __ballot(mm[k] < st.low);
              const unsigned long long ties = todo & __ballot(mm[k] == st.low) & (better ? (better & (0ull - better)) - 1ull : ~0ull);
              st.inst += (int)__popcll(ties);
              todo &= ~ties;
            }
          }
          const uint32_t first_adj = (base == t && k == 0) ? 1u : 0u;  // the first suffix of the run is not a probe
          if (stop_lane >= 0) {  // early exit of :6313-6321: candidates behind it were never examined
            const unsigned long long upto = stop_lane >= 63 ? ~0ull : ((1ull << (stop_lane + 1)) - 1ull);
            const uint32_t took = (uint32_t)__popcll(newm[k] & upto);
            iter += (int)took; n_cand += took; n_nodes += took;
            n_probe += (uint32_t)(stop_lane + 1) - first_adj;
            done_all = true;
          } else if (last[k] >= 0) {
            const uint32_t took = (uint32_t)__popcll(newm[k]);
            iter += (int)took; n_cand += took; n_nodes += took;
            n_probe += (uint32_t)(last[k] + 1) - first_adj;  // probes the reference counted: every member reached after the first suffix
          }
        }
        K4_PROF_T(ps3);
        K4_PROF_ADD(3, ps3 - ps2);
        if (done_all) break;
        if (run_over && !limit_seen && end_cmp) n_probe++;  // ... plus the comparison that ended the run
        if (n_nodes >= node_cap && node_cap < (uint32_t)K4_MAX_IDENT_NODES && sc.small) {
          if (cur_strand == '-') k4d_revcomp_wave(sc, len);
          return K4_NEED_SLOW;
        }
      }
      if (done_all || (st.inst > rp.max_hits && st.low == 0)) { strand = 3; break; }
    }
    if (cur_strand == '+' && strand == K4_STRAND_BOTH) {
      k4d_revcomp_wave(sc, len);
      cur_strand = '-';
      strand = K4_STRAND_CRICK;
    } else
      strand = 3;
  } while (!(st.inst > rp.max_hits && st.low == 0) && strand != 3);
  if (cur_strand == '-') k4d_revcomp_wave(sc, len);
  return k4d_lcm_result(*p_inst, *p_low, p_nxt, st, rp.mm_delta, rp.max_hits, p_inst, p_low);
}

// LocateBestMatches (SfxArray.cpp:6836-7205; CKAligner's -N): at most max_hits alignments with no more than max_tot_mm
// mismatches, kept sorted by mismatches.  One wave per read, every lane runs the same control flow; lane 0 keeps the hit
// list.  Returns 0, 1..max_hits, or max_hits + 1 when further matches were sloughed (K4_NEED_SLOW: small table outgrown).
template <int EL>
K4_DEV int k4d_best_slow(const K4AlignArgs& a, K4Slow& sc, int len, int max_tot_mm, int cl, int core_delta,
                         const K4ReadParams& rp, int* p_inst, k4_hit* hits, uint32_t& n_lookup, uint32_t& n_probe,
                         uint32_t& n_cand) {
  const K4DevIndex& ix = a.ix;
  const int max_hits = rp.max_hits, max_iter = ix.max_iter;
  const int64_t n = (int64_t)ix.n;
  int inst = 0;
  bool sloughed = false;
  int strand = rp.strand;
  char cur_strand = '+';
  if (strand == K4_STRAND_CRICK) { k4d_revcomp_wave(sc, len); cur_strand = '-'; }
  do {
    int cur_delta = core_delta, slides = 0;
    uint32_t n_nodes = 0;
    k4d_hash_new_pass(sc);
    const uint32_t node_cap = min((uint32_t)K4_MAX_IDENT_NODES, sc.lhash ? sc.lcap * 3 / 4 : sc.cap / 2 - 1);
    for (int o = 0; slides < rp.max_slides && o <= len - cl && cur_delta > cl / 3 && n_nodes < node_cap; slides++, o += cur_delta) {
      if (o + cl + cur_delta > len) cur_delta = len - (o + cl);
      n_lookup++;
      int64_t t = k4d_first_exact_wave<EL>(ix, sc, o, cl, n_probe);
      if (t == 0) continue;
      t -= 1;
      int iter = 0;
      bool first = true;
      uint32_t num_copies = 0;
      while (!max_iter || iter < max_iter) {
        if (n_nodes >= node_cap) break;
        if (!first) {
          if (t + 1 >= n) break;
          const uint64_t p2 = k4d_sa_at<EL>(ix, (uint64_t)t + 1);
          if ((int64_t)p2 + cl > n) break;
          if (iter == 100 && !num_copies) {  // :6969-6976 too many copies of this core: give it up
            int64_t lo = t, hi = n - 1;       // LocateLastExact: the last suffix that still starts with the core
            while (lo < hi) {
              const int64_t mid = lo + (hi - lo + 1) / 2;
              n_probe++;
              if (k4d_lane_cmp(ix, sc, o, cl, k4d_sa_at<EL>(ix, (uint64_t)mid)) == 0) lo = mid; else hi = mid - 1;
            }
            num_copies = (uint32_t)(1 + (lo + 1) - t);
            if (max_iter && num_copies > (uint32_t)max_iter) break;
          }
          n_probe++;
          if (k4d_lane_cmp(ix, sc, o, cl, p2) != 0) break;
          t += 1;
        }
        first = false;
        const uint64_t pos = k4d_sa_at<EL>(ix, (uint64_t)t);
        if (pos < (uint64_t)o) continue;
        const uint64_t left = pos - (uint64_t)o;
        if (left + (uint64_t)len > ix.n) continue;  // :7034 (no entry test here: a separator shows up as EOS below)
        int isnew = 0;
        if (sc.lane == 0) {
          uint32_t slot;
          isnew = k4d_hash_insert_lane(sc, (uint32_t)(1 + pos - (uint32_t)o), slot) ? 1 : 0;
        }
        if (!__shfl(isnew, 0, 64)) continue;
        n_nodes++;
        if (n_nodes >= node_cap && node_cap < (uint32_t)K4_MAX_IDENT_NODES && sc.small) {
          if (cur_strand == '-') k4d_revcomp_wave(sc, len);
          return K4_NEED_SLOW;
        }
        iter++;
        n_cand++;
        int mm = 0;
        bool eos = false, all_eq;
        if (sc.packed && !k4d_any_exc_sup(ix, sc.sup, (int64_t)left, (int64_t)left + len))
          mm = k4d_lane_hamming(ix, sc, len, (int64_t)left);
        else
          k4d_lane_range(ix, sc.probe, 0, len, left, false, all_eq, eos, mm);
        if (eos || mm > max_tot_mm) continue;  // :7060-7127
        // :7129-7176 sorted insert (lane 0 owns the list), then every lane learns the new state
        int st_inst = inst, st_mm = max_tot_mm, st_sl = sloughed ? 1 : 0;
        if (sc.lane == 0) {
          int cur = -1;
          if (inst) {
            if (inst == max_hits) st_sl = 1;
            int b;
            for (b = 0; b < inst; b++)
              if ((int)hits[b].mismatches > mm) {
                cur = b;
                if (b + 1 < max_hits)
                  for (int q = min(inst, max_hits - 1); q > b; q--) hits[q] = hits[q - 1];
                break;
              }
            if (b == inst && inst < max_hits) cur = inst;
          } else
            cur = 0;
          if (cur >= 0) {
            uint64_t e_start = 0, e_end = 0;
            const int e = k4d_map_entry_slow(ix, sc.ent, left, e_start, e_end);
            if (e >= 0) {
              k4d_store_hit(&hits[cur], ix.ent_id[e], (uint32_t)(left - e_start), len, cur_strand, mm);
              if (inst < max_hits) st_inst = inst + 1;
              else st_mm = (int)hits[inst - 1].mismatches;  // :7171-7175 only better ones from now on
            }
          }
        }
        inst = __shfl(st_inst, 0, 64);
        max_tot_mm = __shfl(st_mm, 0, 64);
        sloughed = __shfl(st_sl, 0, 64) != 0;
      }
      if (inst == max_hits && max_tot_mm == 0 && !sloughed) { strand = 3; break; }
    }
    if (cur_strand == '+' && strand == K4_STRAND_BOTH) {
      k4d_revcomp_wave(sc, len);
      cur_strand = '-';
      strand = K4_STRAND_CRICK;
    } else
      strand = 3;
  } while (!(inst == max_hits && max_tot_mm == 0 && !sloughed) && strand != 3);
  if (cur_strand == '-') k4d_revcomp_wave(sc, len);
  *p_inst = inst;
  if (inst == 0) return 0;
  return sloughed ? inst + 1 : inst;
}

// The optional phases of AlignReads for a read the standard ones left without a result (SfxArray.cpp:7894-7930), in the
// reference's order: microInDels, splice junctions (both with MaxHits 1, into hit slot 0 + the read's k4_seg2), then the
// chimeric LocateCoreMultiples pass with its own core length.  Returns tHRslt, K4_NEED_SLOW or a negative error code.
template <int EL>
K4_DEV int k4d_ext_phases(const K4AlignArgs& a, K4Slow& sc, int len, const K4ReadParams& rp, int* inst, int* low, int* nxt,
                          k4_hit* hits, k4_seg2* seg2, uint32_t* mk, uint32_t& n_lookup, uint32_t& n_probe, uint32_t& n_cand) {
  int rslt = 0;
  // no hit has been stored for this read so far; its slots start out zero (what the reference's caller would find in slots a
  // phase counts but never writes is its own stale memory)
  if (sc.lane == 0)
    for (int q = 0; q < rp.max_hits; q++) *reinterpret_cast<uint4*>(&hits[q]) = make_uint4(0, 0, 0, 0);
  if (rp.micro_indel_len > 0) {
    rslt = k4d_two_seg<EL>(a, sc, false, rp.micro_indel_len, min(rp.tot_mm, 2), rp.core_len, rp.strand, len, inst, low, nxt, &hits[0],
                           seg2, n_lookup, n_probe, n_cand);
    if (rslt != 0) return rslt;
  }
  if (rp.max_splice_junct_len > 0) {
    rslt = k4d_two_seg<EL>(a, sc, true, rp.max_splice_junct_len, min(rp.tot_mm, 2), rp.core_len, rp.strand, len, inst, low, nxt,
                           &hits[0], seg2, n_lookup, n_probe, n_cand);
    if (rslt != 0) return rslt;
  }
  if (rp.min_chimeric_len > 0) {
    if (rp.max_slides <= 1) return K4_ERR_PARAMS;  // (the reference divides by MaxNumCoreSlides - 1, :7926)
    const int cl = max(rp.min_core_len, len / (rp.tot_mm + 4));
    const int cd = max(len / (rp.max_slides - 1), cl);
    if (cl < 1) return K4_ERR_PARAMS;
    if (rp.min_chimeric_len >= 15 && rp.min_chimeric_len <= 99)  // :5878-5883 any other value: the default branch
      rslt = k4d_lcm_slow<EL, true>(a, sc, len, rp.tot_mm, cl, cd, rp, inst, low, nxt, hits, n_lookup, n_probe, n_cand,
                                    max(cl, (rp.min_chimeric_len * len) / 100), mk);
    else
      rslt = k4d_lcm_slow<EL, false>(a, sc, len, rp.tot_mm, cl, cd, rp, inst, low, nxt, hits, n_lookup, n_probe, n_cand);
    // a hit this pass stored cleared both segments of slot 0 (:6129); the second segment survives only with the two-segment
    // record a microInDel / splice phase left there (it gave up over several equally good loci; its count was carried in)
    if (seg2 && sc.lane == 0 && !(hits[0].ext & (K4_EXT_INDEL | K4_EXT_SPLICE))) *reinterpret_cast<uint4*>(seg2) = make_uint4(0, 0, 0, 0);
    return rslt;
  }
  return 0;
}

// persistent waves pull read ids from their list until it is drained (every wave reaches the exit test).
// pass 0: many waves with small dedupe tables (list = slow_list, ctl[0]/[1]); pass 1: few waves with tables sized for
// the reference's own limits (list = huge_list, ctl[K4_CTL_HUGE]/[+1]).
// EXT: the instantiation that also holds the optional phases (k4_ext.h); launched only when one of them is requested, so
// that the standard path keeps the register budget (and with it the occupancy) of the lean one.
template <int EL, bool EXT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(EXT ? K4_SLOW_WAVES_PER_EU_EXT : K4_SLOW_WAVES_PER_EU))) k4k_align_slow(K4AlignArgs a, uint32_t n_waves, int pass, uint64_t* hash_base,
                                                     uint32_t hash_cap, uint32_t* gen_base, int max_len) {
  // dynamic LDS, sized by the batch: entry table copy (starts, ends, ids) | coarse exception bitmap | packed probe | probe bytes
  extern __shared__ uint64_t slow_lds[];
  uint64_t* ent_s = slow_lds;
  uint32_t* entid_s = reinterpret_cast<uint32_t*>(slow_lds + (a.ix.n_entries <= K4_LDS_ENTRIES ? 2 * K4_LDS_ENTRIES : 0));
  uint32_t* sup_s = entid_s + (a.ix.n_entries <= K4_LDS_ENTRIES ? K4_LDS_ENTRIES : 0);
  uint64_t* pk_s = reinterpret_cast<uint64_t*>(sup_s + K4_SUP_WORDS);
  uint8_t* probe_s = reinterpret_cast<uint8_t*>(pk_s + (max_len / 32 + 2));
  // (chimeric phase only) one mismatch bit vector per lane behind the probe bytes, word w of lane l at mk_s[w * 64 + l]
  uint32_t* mk_s = reinterpret_cast<uint32_t*>(probe_s + ((max_len + 64 + 7) & ~7)) + threadIdx.x;
  // the pass-0 dedupe table (EL == 4, lean instantiation): behind the probe bytes, where the chimeric masks of the other one go
  uint32_t* lhash_s = reinterpret_cast<uint32_t*>(probe_s + ((max_len + 64 + 7) & ~7));
  const uint32_t wave = blockIdx.x;
  const int lane = threadIdx.x;
  uint32_t n_lookup = 0, n_probe = 0, n_cand = 0;
  const bool ent_in_lds = a.ix.n_entries <= K4_LDS_ENTRIES;
  if (ent_in_lds)
    for (int q = lane; q < (int)a.ix.n_entries; q += 64) {
      ent_s[q] = a.ix.ent_start[q];
      ent_s[K4_LDS_ENTRIES + q] = a.ix.ent_end[q];
      entid_s[q] = a.ix.ent_id[q];
    }
  for (int q = lane; q < K4_SUP_WORDS; q += 64) sup_s[q] = a.ix.excsup[q];
  __syncthreads();
  if (wave < n_waves) {
    K4Slow sc;
    sc.sup = sup_s;
    sc.ent_id = ent_in_lds ? entid_s : a.ix.ent_id;
    sc.lhash = (EL == 4 && !EXT && pass == 0) ? lhash_s : nullptr;
    sc.lcap = K4_LDS_HASH;
    sc.lused = 0;
#ifdef K4_SLOW_PROF
    for (int q = 0; q < 16; q++) sc.prof[q] = 0;
#endif
    sc.probe = probe_s;
    sc.ent = ent_in_lds ? ent_s : nullptr;
    sc.pk = pk_s;
    sc.packed = false;
    sc.hash = hash_base + (size_t)wave * hash_cap;
    sc.cap = hash_cap;
    sc.gen = gen_base[wave];
    sc.lane = lane;
    sc.small = pass == 0;
    const uint32_t* list = pass == 0 ? a.slow_list : a.huge_list;
    const uint8_t* steps = pass == 0 ? a.slow_step : a.huge_step;
    uint32_t* cnt = a.ctl + (pass == 0 ? 0 : K4_CTL_HUGE);
    const uint32_t total = cnt[0];
    for (;;) {
      uint32_t q = 0;
      if (lane == 0) q = atomicAdd(&cnt[1], 1u);
      q = __shfl(q, 0, 64);
      if (q >= total) break;
      const int64_t i = list[q];
      const int from_phase = steps[q];
      int phase = 0;
      K4_PROF_T(pr0);
      K4_PROF_ADD(from_phase < 3 ? 13 + from_phase : 15, 1);
      const uint32_t r0 = n_lookup, r1 = n_probe, r2 = n_cand;
      const int len = (int)a.lens[i];
      const K4ReadParams rp = k4d_read_params(a, len);
      k4_hit* hits = a.hits + i * a.max_hits;
      int inst = 0, low = 0, nxt = 0, rslt = 0, allow = 0;
      if (len < 1 || len > K4_MAX_READ_LEN || len > max_len || rp.core_len < 1 || rp.max_hits < 1 || rp.max_hits > a.max_hits) {
        if (lane == 0) k4d_finalize(a, i, len, rp, a.mode == 0 ? K4_ERR_PARAMS : K4_HR_FATAL, 0, 0, 0);
        continue;
      }
      const uint8_t* src = a.reads + a.offs[i];
      __syncthreads();
      for (int j = lane; j < len; j += 64) probe_s[j] = src[j] & 7;
      __syncthreads();
      k4d_pack_probe_wave(sc, len);
      K4_PROF_T(pr1);
      K4_PROF_ADD(5, pr1 - pr0);
      if ((a.mode == 1 && a.kp.pe_mode == 4) || a.best) {  // -N (KAligner.cpp:9776-9796): LocateBestMatches instead of AlignReads
        const int r = k4d_best_slow<EL>(a, sc, len, rp.tot_mm, rp.core_len, rp.core_delta, rp, &inst, hits, n_lookup, n_probe, n_cand);
        if (r == K4_NEED_SLOW) {
          n_lookup = r0; n_probe = r1; n_cand = r2;
          if (lane == 0) {
            const uint32_t slot = atomicAdd(&a.ctl[K4_CTL_HUGE], 1u);
            a.huge_list[slot] = (uint32_t)i;
            a.huge_step[slot] = (uint8_t)from_phase;
          }
          continue;
        }
        if (lane == 0) {
          if (a.mode == 0) {  // the raw call: its own return value and instance count; unused slots zeroed
            for (int q = inst; q < a.max_hits; q++) *reinterpret_cast<uint4*>(&hits[q]) = make_uint4(0, 0, 0, 0);
            a.rslt[i] = r; a.inst[i] = inst;
          } else
            k4d_finalize(a, i, len, rp, r == 0 ? K4_HR_NONE : K4_HR_HITS, inst, 0, 0);
        }
        continue;
      }
      // the standard phases -- unless the fast path ran all of them without a result and only the optional ones are left:
      // then the In/Out state is what the last LocateCoreMultiples initialised it to (:5902-5907), no instance seen
      int n_std = 0;
      if (rp.tot_mm > 0)
        for (; n_std <= rp.tot_mm; n_std++)
          if (len / (n_std + rp.mm_delta) <= rp.core_len) break;
      n_std += (rp.tot_mm > 0 ? n_std <= rp.tot_mm : true) ? 1 : 0;
      const bool std_done = EXT && a.ext_on && from_phase >= n_std;
      if (std_done) {
        inst = 0; low = nxt = rp.tot_mm + rp.mm_delta + 1;
      } else {
        if (rp.tot_mm > 0) {
          // A phase the fast kernel completed (and tallied) is not run again: the read is still unaligned, so that phase
          // returned eHRnone, i.e. it folded no candidate -- every candidate it accepts has fewer mismatches than the
          // LowMMCnt it starts from -- stored no hit and left (instances, LowMMCnt, NxtLowMMCnt) as LocateCoreMultiples
          // initialises them, which the next call does again (:5902-5907).
          for (allow = 0; allow <= rp.tot_mm; allow++) {
            int cl = len / (allow + rp.mm_delta);
            if (cl <= rp.core_len) break;
            if (phase++ < from_phase) continue;
            rslt = k4d_lcm_slow<EL, false>(a, sc, len, allow, cl, cl, rp, &inst, &low, &nxt, hits, n_lookup, n_probe, n_cand);
            if (rslt != 0) break;
          }
        }
        if (rslt == 0 && allow <= rp.tot_mm && phase++ >= from_phase)
          rslt = k4d_lcm_slow<EL, false>(a, sc, len, rp.tot_mm, rp.core_len, rp.core_delta, rp, &inst, &low, &nxt, hits,
                                         n_lookup, n_probe, n_cand);
      }
      if (EXT && rslt == 0 && a.ext_on)  // SfxArray.cpp:7894-7930
        rslt = k4d_ext_phases<EL>(a, sc, len, rp, &inst, &low, &nxt, hits, a.seg2 ? a.seg2 + i : nullptr, mk_s, n_lookup, n_probe, n_cand);
      if (rslt == K4_NEED_SLOW) {  // outgrew the small table: the big-table pass redoes the read (and tallies it)
        n_lookup = r0; n_probe = r1; n_cand = r2;
        if (lane == 0) {
          const uint32_t slot = atomicAdd(&a.ctl[K4_CTL_HUGE], 1u);
          a.huge_list[slot] = (uint32_t)i;
          a.huge_step[slot] = (uint8_t)from_phase;
        }
        continue;
      }
      if (lane == 0) {
        // a read without a reported hit has no second segment either (a two-segment phase may have left one behind)
        if (EXT && a.seg2 && !(rslt == K4_HR_HITS || rslt == K4_HR_MMDELTA || rslt == K4_HR_HITINSTS))
          *reinterpret_cast<uint4*>(a.seg2 + i) = make_uint4(0, 0, 0, 0);
        if (rslt < 0) k4d_finalize(a, i, len, rp, a.mode == 0 ? rslt : K4_HR_FATAL, 0, 0, 0);
        else k4d_finalize(a, i, len, rp, rslt, inst, low, nxt);
      }
      K4_PROF_T(pr2);
      K4_PROF_ADD(4, pr2 - pr0);
      K4_PROF_ADD(11, 1);
    }
    if (lane == 0) gen_base[wave] = sc.gen;
#ifdef K4_SLOW_PROF
    if (lane == 0)
      for (int q = 0; q < 16; q++)
        if (sc.prof[q]) atomicAdd(&a.counters[6 + q], sc.prof[q]);
#endif
  }
  if (lane == 0) {  // the tallies are wave-uniform
    if (n_lookup) atomicAdd(&a.counters[1], (unsigned long long)n_lookup);
    if (n_probe) atomicAdd(&a.counters[2], (unsigned long long)n_probe);
    if (n_cand) atomicAdd(&a.counters[3], (unsigned long long)n_cand);
  }
}

// ==== host side ====================================================================================================
static uint32_t next_pow2(uint64_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// slots of one survivor buffer for batches of up to cap reads (see k4_reserve)
static size_t k4_survivor_slots(size_t cap) {
  static_assert(K4_CHUNK * 7 >= 8 * 63, "cap / 7 must cover the abandoned chunk tails");
  return cap + cap / 7 + (size_t)2048 * 4 * K4_CHUNK + K4_CHUNK;
}

static int nch_for(int max_len) {
  if (max_len <= 128) return 4;
  if (max_len <= 160) return 5;
  if (max_len <= 256) return 8;
  return 16;
}

extern "C" int k4_reserve(k4_index* ix, int64_t max_reads, int32_t max_read_len, int32_t max_hits) {
  if (!ix || max_reads < 0 || max_read_len < 1 || max_hits < 1) return K4_ERR_PARAMS;
  if (max_read_len > K4_MAX_READ_LEN) return k4_fail(ix, K4_ERR_PARAMS, "read length %d exceeds %d", max_read_len, K4_MAX_READ_LEN);
  if (max_reads >= 0xFFFFFFF0ll) return k4_fail(ix, K4_ERR_PARAMS, "at most 2^32-16 reads per batch");
  K4_HIP(ix, hipSetDevice(ix->device));
  K4Workspace& w = ix->ws;
  bool touched = false;  // something was (re)allocated or filled on the null stream
  int fast_len = std::min<int>(max_read_len, K4_MAX_FAST_READ_LEN);
  if (max_reads > w.cap_reads || fast_len > w.cap_len) {
    int64_t cap = std::max<int64_t>(max_reads, w.cap_reads);
    cap = (cap + 255) / 256 * 256;
    int len = std::max(fast_len, w.cap_len);
    int nch = nch_for(len);
    touched = true;
    // the capacities say what the pointers below can hold: zero while they are being replaced, so that a failed
    // allocation leaves a workspace every *_dev call refuses (K4_ERR_PARAMS) instead of one with null buffers
    w.cap_reads = 0;
    w.cap_len = 0;
    for (void* q : {(void*)w.ids[0], (void*)w.ids[1], (void*)w.rows[0], (void*)w.rows[1], (void*)w.slow_list, (void*)w.slow_step,
                    (void*)w.huge_list, (void*)w.huge_step})
      if (q) hipFree(q);
    w.ids[0] = w.ids[1] = nullptr; w.rows[0] = w.rows[1] = nullptr; w.slow_list = nullptr; w.slow_step = nullptr;
    w.huge_list = nullptr; w.huge_step = nullptr;
    // Survivors of step t (ids + packed rows) ping-pong between two buffers.  Slots are handed out K4_CHUNK at a time
    // per wave; a wave abandons the tail of its chunk (at most 63 slots) when the next ballot does not fit, and leaves
    // one partly used chunk behind when it ends.  Step t+1 walks over step t's slots, holes included, but places only
    // real survivors again.  So every chunk a wave has moved on from holds at least K4_CHUNK - 63 real entries: a step
    // uses at most n * K4_CHUNK / (K4_CHUNK - 63) slots plus one chunk per wave of the (at most 2048-block) grid.
    const size_t slots = k4_survivor_slots((size_t)cap);
    for (int b = 0; b < 2; b++) {
      K4_HIP(ix, hipMalloc(&w.ids[b], slots * 4));
      K4_HIP(ix, hipMalloc(&w.rows[b], slots * K4_ROW_WORDS(nch) * 8));
    }
    K4_HIP(ix, hipMalloc(&w.slow_list, (size_t)cap * 4));
    K4_HIP(ix, hipMalloc(&w.slow_step, (size_t)cap));
    K4_HIP(ix, hipMalloc(&w.huge_list, (size_t)cap * 4));
    K4_HIP(ix, hipMalloc(&w.huge_step, (size_t)cap));
    w.cap_reads = cap;
    w.cap_len = len;
  }
  if (!w.ctl) {
    touched = true;
    K4_HIP(ix, hipMalloc(&w.ctl, K4_CTL_WORDS * 4));
    K4_HIP(ix, hipMemset(w.ctl, 0, K4_CTL_WORDS * 4));
  }
  // general-kernel scratch.  Pass 0: K4_SLOW_WAVES waves with small dedupe tables; pass 1: K4_HUGE_WAVES waves whose
  // tables hold one strand pass at the reference's own limits (<= MaxIter per core, <= 1,024,000 nodes, SfxArray.h:15)
  uint64_t nodes = ix->d.max_iter ? std::min<uint64_t>((uint64_t)ix->d.max_iter * 48, K4_MAX_IDENT_NODES) : K4_MAX_IDENT_NODES;
  uint32_t hcap = next_pow2(std::max<uint64_t>(2 * nodes + 2, 1024));
  if (!w.slow_hash || hcap > w.slow_hash_cap) {
    touched = true;
    if (w.slow_hash) hipFree(w.slow_hash);
    w.slow_hash = nullptr;
    const size_t words = (size_t)K4_SLOW_WAVES * K4_SMALL_HASH + (size_t)K4_HUGE_WAVES * hcap;
    const size_t bytes = words * 8 + (size_t)(K4_SLOW_WAVES + K4_HUGE_WAVES) * 4;
    K4_HIP(ix, hipMalloc(&w.slow_hash, bytes));
    K4_HIP(ix, hipMemset(w.slow_hash, 0, bytes));
    w.slow_hash_cap = hcap;
    w.slow_lanes = K4_SLOW_WAVES;
  }
  w.cap_hits = std::max(w.cap_hits, max_hits);
  // (the fills above ran on the null stream: callers launch on streams of their own, possibly non-blocking ones)
  if (touched) K4_HIP(ix, hipDeviceSynchronize());
  return K4_OK;
}

template <int EL, int NCH, typename KT>
static int launch_steps(k4_index* ix, K4AlignArgs& a, int n_steps, hipStream_t st) {
  K4Workspace& w = ix->ws;
  const size_t lds = (size_t)2 * (NCH + 1) * K4_BS(NCH) * 8 + (size_t)K4_DEDUP_CAP * K4_BS(NCH) * 4 + (size_t)2 * K4_BS(NCH) * 8 + (size_t)2 * K4_LDS_ENTRIES * 8 +
                     (size_t)K4_SUP_WORDS * 4 + 16;
  if (lds > 48 * 1024) {
    K4_HIP(ix, hipFuncSetAttribute((const void*)k4k_align_step<EL, NCH, true, KT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    K4_HIP(ix, hipFuncSetAttribute((const void*)k4k_align_step<EL, NCH, true, KT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    K4_HIP(ix, hipFuncSetAttribute((const void*)k4k_align_step<EL, NCH, false, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  // grid-stride kernels: enough blocks to fill the chip at the kernel's occupancy, never more than the work
  const unsigned full = 256 * 8 * (256 / K4_BS(NCH));
  const bool timed = ix->timing && ix->ev_used < 4096;
  if (timed) {
    if (ix->ev_used == ix->ev0.size()) {
      hipEvent_t e0, e1, e2;
      K4_HIP(ix, hipEventCreate(&e0));
      K4_HIP(ix, hipEventCreate(&e1));
      K4_HIP(ix, hipEventCreate(&e2));
      ix->ev0.push_back(e0);
      ix->ev1.push_back(e1);
      ix->ev2.push_back(e2);
    }
    K4_HIP(ix, hipEventRecord(ix->ev0[ix->ev_used], st));
  }
  unsigned grid0 = (unsigned)std::min<int64_t>((a.n_reads + K4_BS(NCH) - 1) / K4_BS(NCH), full);
  // step 0 in two launches: every read, then those it set aside (deep k-mer buckets), listed in the id buffer step 1 will
  // overwrite; both append their survivors to the same list
  // -- on an index with repeat families (share of the suffixes in deep k-mer buckets, measured when the table was built);
  // elsewhere one launch without the code for it, which costs the first phase 4 % in registers (C2: 2335 -> 2254 M reads/s)
  if (ix->deep_bucket_frac >= K4_DEFER_MIN_FRAC) {
    a.defer_ids = w.ids[1];
    hipLaunchKernelGGL((k4k_align_step<EL, NCH, true, KT, true>), dim3(grid0), dim3(K4_BS(NCH)), lds, st, a, 0, (const uint32_t*)nullptr,
                       (const uint64_t*)nullptr, (const uint32_t*)nullptr, w.ids[0], w.rows[0], w.ctl + 2);
    hipLaunchKernelGGL((k4k_align_step<EL, NCH, true, KT, true>), dim3(grid0), dim3(K4_BS(NCH)), lds, st, a, 0, (const uint32_t*)w.ids[1],
                       (const uint64_t*)nullptr, (const uint32_t*)(w.ctl + K4_CTL_DEFER), w.ids[0], w.rows[0], w.ctl + 2);
  } else {
    a.defer_ids = nullptr;
    hipLaunchKernelGGL((k4k_align_step<EL, NCH, true, KT, false>), dim3(grid0), dim3(K4_BS(NCH)), lds, st, a, 0, (const uint32_t*)nullptr,
                       (const uint64_t*)nullptr, (const uint32_t*)nullptr, w.ids[0], w.rows[0], w.ctl + 2);
  }
  for (int t = 1; t < n_steps; t++) {
    const int in = (t - 1) & 1, out = t & 1;
    hipLaunchKernelGGL((k4k_align_step<EL, NCH, false, KT>), dim3(grid0), dim3(K4_BS(NCH)), lds, st, a, t, w.ids[in], w.rows[in],
                       w.ctl + 2 + (t - 1), w.ids[out], w.rows[out], w.ctl + 2 + t);
  }
  if (timed) K4_HIP(ix, hipEventRecord(ix->ev1[ix->ev_used], st));  // (launch_all records ev2 behind the general kernel and counts the set)
  return K4_OK;
}

template <int EL, typename KT>
static int launch_all(k4_index* ix, K4AlignArgs& a, int max_len, int n_steps, hipStream_t st) {
  K4Workspace& w = ix->ws;
  const int nch = nch_for(std::min(max_len, K4_MAX_FAST_READ_LEN));
  if (nch > nch_for(w.cap_len)) return k4_fail(ix, K4_ERR_INTERNAL, "workspace not reserved for read length %d", max_len);
  if (n_steps < 1 || n_steps > K4_CTL_WORDS - 4) return k4_fail(ix, K4_ERR_INTERNAL, "bad phase count %d", n_steps);
  a.slow_list = w.slow_list;
  a.slow_step = w.slow_step;
  a.ctl = w.ctl;
  a.counters = (unsigned long long*)ix->counters;
  a.huge_list = w.huge_list;
  a.huge_step = w.huge_step;
  a.slow_probe = nullptr;
  a.slow_hash = w.slow_hash;
  a.slow_hash_cap = w.slow_hash_cap;
  uint64_t* small_base = w.slow_hash;
  uint64_t* big_base = w.slow_hash + (size_t)K4_SLOW_WAVES * K4_SMALL_HASH;
  uint32_t* gen_small = reinterpret_cast<uint32_t*>(big_base + (size_t)K4_HUGE_WAVES * w.slow_hash_cap);
  uint32_t* gen_big = gen_small + K4_SLOW_WAVES;
  a.slow_gen = gen_small;
  a.nw = nch + 1;
  if (a.n_reads == 0) return K4_OK;
  K4_HIP(ix, hipMemsetAsync(w.ctl, 0, K4_CTL_WORDS * 4, st));
  int rc;
  switch (nch) {
    case 4: rc = launch_steps<EL, 4, KT>(ix, a, n_steps, st); break;
    case 5: rc = launch_steps<EL, 5, KT>(ix, a, n_steps, st); break;
    case 8: rc = launch_steps<EL, 8, KT>(ix, a, n_steps, st); break;
    default: rc = launch_steps<EL, 16, KT>(ix, a, n_steps, st); break;
  }
  if (rc != K4_OK) return rc;
  const int slow_len = std::min(std::max(max_len, 1), K4_MAX_READ_LEN);
  const bool chim = a.ext_on && (a.mode == 0 ? a.ap.min_chimeric_len : a.kp.min_chimeric_len) > 0;
  const size_t slow_lds = (size_t)(a.ix.n_entries <= K4_LDS_ENTRIES ? 2 * K4_LDS_ENTRIES * 8 + K4_LDS_ENTRIES * 4 : 0) + (size_t)K4_SUP_WORDS * 4 + (size_t)(slow_len / 32 + 2) * 8 +
                          (size_t)((slow_len + 64 + 7) & ~7) + std::max<size_t>(chim ? (size_t)64 * 4 * ((slow_len + 31) / 32 + 1) : 0, (size_t)K4_LDS_HASH * 4) + 16;
  if (slow_lds > 48 * 1024) K4_HIP(ix, hipFuncSetAttribute((const void*)k4k_align_slow<EL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slow_lds));
  // (no more waves than reads: a batch of one -- the facade's AlignReads -- should not pay for 8192 idle blocks)
  const uint32_t sw = (uint32_t)std::min<int64_t>(K4_SLOW_WAVES, std::max<int64_t>(a.n_reads, 1));
  const uint32_t hw = (uint32_t)std::min<int64_t>(K4_HUGE_WAVES, std::max<int64_t>(a.n_reads, 1));
  if (a.ext_on) {
    hipLaunchKernelGGL((k4k_align_slow<EL, true>), dim3(sw), dim3(64), slow_lds, st, a, sw, 0, small_base, (uint32_t)K4_SMALL_HASH,
                       gen_small, slow_len);
    hipLaunchKernelGGL((k4k_align_slow<EL, true>), dim3(hw), dim3(64), slow_lds, st, a, hw, 1, big_base, w.slow_hash_cap, gen_big, slow_len);
  } else {
    hipLaunchKernelGGL((k4k_align_slow<EL, false>), dim3(sw), dim3(64), slow_lds, st, a, sw, 0, small_base, (uint32_t)K4_SMALL_HASH,
                       gen_small, slow_len);
    hipLaunchKernelGGL((k4k_align_slow<EL, false>), dim3(hw), dim3(64), slow_lds, st, a, hw, 1, big_base, w.slow_hash_cap, gen_big, slow_len);
  }
  if (ix->timing && ix->ev_used < ix->ev0.size() && ix->ev_used < 4096) {
    K4_HIP(ix, hipEventRecord(ix->ev2[ix->ev_used], st));
    ix->ev_used++;
  }
  K4_HIP(ix, hipGetLastError());
  return K4_OK;
}

static int run_dev(k4_index* ix, K4AlignArgs& a, int max_len, void* stream) {
  K4Workspace& w = ix->ws;
  if (a.n_reads > w.cap_reads || std::min(max_len, K4_MAX_FAST_READ_LEN) > w.cap_len || a.max_hits > w.cap_hits ||
      !w.slow_hash || !w.ids[0] || !w.ids[1] || !w.rows[0] || !w.rows[1] || !w.slow_list || !w.huge_list || !w.ctl)
    return k4_fail(ix, K4_ERR_PARAMS, "k4_reserve(%lld, %d, %d) must precede the *_dev call", (long long)a.n_reads,
                   max_len, a.max_hits);
  a.ix = ix->d;
  hipStream_t st = (hipStream_t)stream;
  {  // the optional phases (SfxArray.cpp:7894-7930): argument ranges as kalign enforces them (KAlignerCL.cpp:667-761)
    const int mc = a.mode == 0 ? a.ap.min_chimeric_len : a.kp.min_chimeric_len;
    const int mi = a.mode == 0 ? a.ap.micro_indel_len : a.kp.micro_indel_len;
    const int ms = a.mode == 0 ? a.ap.max_splice_junct_len : a.kp.max_splice_junct_len;
    if (mc < 0 || mi < 0 || mi > 20 || ms < 0 || (ms != 0 && (ms < 25 || ms > 100000)))
      return k4_fail(ix, K4_ERR_PARAMS, "MinChimericLen / microInDelLen (0..20) / MaxSpliceJunctLen (0, 25..100000) out of range");
    a.ext_on = (mc > 0 || mi > 0 || ms > 0) ? 1 : 0;
    if ((mi > 0 || ms > 0) && !a.seg2)
      return k4_fail(ix, K4_ERR_PARAMS, "microInDel / splice phases report two segments: use the *_ext entry points (k4_seg2 output)");
    if (a.ext_on && (a.best || (a.mode == 1 && a.kp.pe_mode == 4)))
      return k4_fail(ix, K4_ERR_PARAMS, "the optional AlignReads phases cannot be combined with LocateBestMatches (-N)");
    if (a.seg2 && a.n_reads > 0) K4_HIP(ix, hipMemsetAsync(a.seg2, 0, (size_t)a.n_reads * sizeof(k4_seg2), st));
  }
  // number of AlignReads phases any read can have: escalation 0..TotMM, or fewer plus the final phase (SfxArray.cpp:7867-7891)
  int tot_mm = a.ap.tot_mm;
  if (a.mode == 1) {
    tot_mm = a.kp.max_subs == 0 ? 0 : std::max(1, (int)(0.5 + (max_len * a.kp.max_subs) / 100.0));
    tot_mm = std::min(tot_mm, 63);
  }
  const int n_steps = tot_mm + 1;
  if (ix->d.el == 4) return launch_all<4, uint32_t>(ix, a, max_len, n_steps, st);
  return ix->d.ktab64 ? launch_all<5, uint64_t>(ix, a, max_len, n_steps, st) : launch_all<5, uint32_t>(ix, a, max_len, n_steps, st);
}

static int check_align_params(k4_index* ix, const k4_align_params* p) {
  if (!p) return K4_ERR_PARAMS;
  if (p->tot_mm < 0 || p->tot_mm > 63 || p->core_len < 1 || p->core_delta < 0 || p->max_core_slides < 1 ||
      p->mm_delta < 1 || p->strand < 0 || p->strand > 2 || p->max_hits < 1 || p->max_hits > 4096)
    return k4_fail(ix, K4_ERR_PARAMS, "AlignReads parameters out of range");
  return K4_OK;
}

extern "C" int k4_align_reads_batch_dev(k4_index* ix, const k4_align_params* p, int64_t n, int32_t max_len,
                                        const void* d_reads, const void* d_offs, const void* d_lens, void* d_rslt,
                                        void* d_inst, void* d_low, void* d_nxt, void* d_hits, void* stream) {
  return k4_align_reads_ext_batch_dev(ix, p, n, max_len, d_reads, d_offs, d_lens, d_rslt, d_inst, d_low, d_nxt, d_hits, nullptr, stream);
}
extern "C" int k4_align_reads_ext_batch_dev(k4_index* ix, const k4_align_params* p, int64_t n, int32_t max_len,
                                            const void* d_reads, const void* d_offs, const void* d_lens, void* d_rslt,
                                            void* d_inst, void* d_low, void* d_nxt, void* d_hits, void* d_seg2, void* stream) {
  if (!ix) return K4_ERR_PARAMS;
  int rc = check_align_params(ix, p);
  if (rc != K4_OK) return rc;
  if (n < 0 || (n > 0 && (!d_reads || !d_offs || !d_lens || !d_rslt || !d_inst || !d_low || !d_nxt || !d_hits)))
    return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  K4AlignArgs a;
  memset(&a, 0, sizeof(a));
  a.reads = (const uint8_t*)d_reads; a.offs = (const uint64_t*)d_offs; a.lens = (const uint32_t*)d_lens;
  a.n_reads = n; a.mode = 0; a.ap = *p;
  a.rslt = (int32_t*)d_rslt; a.inst = (int32_t*)d_inst; a.low = (int32_t*)d_low; a.nxt = (int32_t*)d_nxt;
  a.hits = (k4_hit*)d_hits; a.max_hits = p->max_hits;
  a.seg2 = (k4_seg2*)d_seg2;
  return run_dev(ix, a, max_len, stream);
}

// CSfxArray::LocateBestMatches for n reads (SfxArray.h:793): rslt = its return value (0, 1..max_hits, max_hits+1 when
// matches were sloughed), inst = alignments in the read's max_hits hit slots, sorted by mismatches
extern "C" int k4_best_matches_batch_dev(k4_index* ix, const k4_align_params* p, int64_t n, int32_t max_len, const void* d_reads,
                                         const void* d_offs, const void* d_lens, void* d_rslt, void* d_inst, void* d_hits,
                                         void* stream) {
  if (!ix) return K4_ERR_PARAMS;
  int rc = check_align_params(ix, p);
  if (rc != K4_OK) return rc;
  if (n < 0 || (n > 0 && (!d_reads || !d_offs || !d_lens || !d_rslt || !d_inst || !d_hits))) return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  K4AlignArgs a;
  memset(&a, 0, sizeof(a));
  a.reads = (const uint8_t*)d_reads; a.offs = (const uint64_t*)d_offs; a.lens = (const uint32_t*)d_lens;
  a.n_reads = n; a.mode = 0; a.best = 1; a.ap = *p;
  a.rslt = (int32_t*)d_rslt; a.inst = (int32_t*)d_inst;  // (low / nxt are not outputs of this call and stay null)
  a.hits = (k4_hit*)d_hits; a.max_hits = p->max_hits;
  return run_dev(ix, a, max_len, stream);
}

static int resolve_kalign(k4_index* ix, const k4_kalign_params* p, k4_kalign_params* out) {
  if (!p) return K4_ERR_PARAMS;
  *out = *p;
  if (p->max_subs < 0 || p->max_subs > 15 || (p->min_edit_dist != 1 && p->min_edit_dist != 2) || p->max_ns < 0 ||
      p->strand < 0 || p->strand > 2 || p->max_ml < 1 || p->max_ml > 4096 || p->pe_mode < 0 || p->pe_mode > 4)
    return k4_fail(ix, K4_ERR_PARAMS, "kalign parameters out of range");
  int slides = 0;
  int mcl = k4_min_core_len(ix, p->pmode, &slides);
  if (out->min_core_len <= 0) out->min_core_len = mcl;
  if (out->max_num_slides <= 0) out->max_num_slides = slides;
  return K4_OK;
}

// sparse_hits: the paired-end pass reads only the slots of reported instances, so the 10 slots per end need no zero fill
int k4i_kalign_batch_dev(k4_index* ix, const k4_kalign_params* p, int64_t n, int32_t max_len, const void* d_reads,
                         const void* d_offs, const void* d_lens, void* d_out, void* d_hits, void* stream, int sparse_hits);
extern "C" int k4_kalign_batch_dev(k4_index* ix, const k4_kalign_params* p, int64_t n, int32_t max_len,
                                   const void* d_reads, const void* d_offs, const void* d_lens, void* d_out,
                                   void* d_hits, void* stream) {
  return k4i_kalign_batch_dev(ix, p, n, max_len, d_reads, d_offs, d_lens, d_out, d_hits, stream, 0);
}
static int kalign_dev(k4_index* ix, const k4_kalign_params* p, int64_t n, int32_t max_len, const void* d_reads, const void* d_offs,
                      const void* d_lens, void* d_out, void* d_hits, void* d_seg2, void* stream, int sparse_hits);
extern "C" int k4_kalign_ext_batch_dev(k4_index* ix, const k4_kalign_params* p, int64_t n, int32_t max_len, const void* d_reads,
                                       const void* d_offs, const void* d_lens, void* d_out, void* d_hits, void* d_seg2,
                                       void* stream) {
  return kalign_dev(ix, p, n, max_len, d_reads, d_offs, d_lens, d_out, d_hits, d_seg2, stream, 0);
}
int k4i_kalign_batch_dev(k4_index* ix, const k4_kalign_params* p, int64_t n, int32_t max_len, const void* d_reads,
                         const void* d_offs, const void* d_lens, void* d_out, void* d_hits, void* stream, int sparse_hits) {
  return kalign_dev(ix, p, n, max_len, d_reads, d_offs, d_lens, d_out, d_hits, nullptr, stream, sparse_hits);
}
static int kalign_dev(k4_index* ix, const k4_kalign_params* p, int64_t n, int32_t max_len, const void* d_reads, const void* d_offs,
                      const void* d_lens, void* d_out, void* d_hits, void* d_seg2, void* stream, int sparse_hits) {
  if (!ix) return K4_ERR_PARAMS;
  K4AlignArgs a;
  memset(&a, 0, sizeof(a));
  int rc = resolve_kalign(ix, p, &a.kp);
  if (rc != K4_OK) return rc;
  if (n < 0 || (n > 0 && (!d_reads || !d_offs || !d_lens || !d_out || !d_hits))) return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  a.reads = (const uint8_t*)d_reads; a.offs = (const uint64_t*)d_offs; a.lens = (const uint32_t*)d_lens;
  a.n_reads = n; a.mode = 1;
  a.rr = (k4_read_result*)d_out; a.hits = (k4_hit*)d_hits; a.max_hits = a.kp.max_ml;
  a.seg2 = (k4_seg2*)d_seg2;
  a.sparse_hits = sparse_hits;
  return run_dev(ix, a, max_len, stream);
}

// ---- host-pointer entry points: stage through the index's own buffers and stream -------------------------------
static int stage_in(k4_index* ix, int64_t n, const uint8_t* reads, const uint64_t* offs, const uint32_t* lens,
                    int max_hits, int* max_len_out, size_t out_bytes_per_read) {
  K4Workspace& w = ix->ws;
  uint64_t tot = 0;
  int max_len = 1;
  for (int64_t i = 0; i < n; i++) {
    tot = std::max<uint64_t>(tot, offs[i] + lens[i]);
    max_len = std::max<int>(max_len, (int)lens[i]);
  }
  if (max_len > K4_MAX_READ_LEN) return k4_fail(ix, K4_ERR_PARAMS, "read longer than %d bases", K4_MAX_READ_LEN);
  *max_len_out = max_len;
  int rc = k4_reserve(ix, n, max_len, max_hits);
  if (rc != K4_OK) return rc;
  // small batch: everything through one pinned block (one copy up, one down instead of three and two from pageable memory)
  const size_t in_bytes = (((size_t)n * 12 + 15) & ~(size_t)15) + tot + 64;
  const size_t hits_off = ((size_t)n * 24 + 15) & ~(size_t)15;
  const size_t out_bytes = hits_off + (size_t)n * max_hits * sizeof(k4_hit);
  w.c_small = n <= K4_SMALL_READS && in_bytes <= K4_SMALL_STAGE / 2 && out_bytes <= K4_SMALL_STAGE / 2;
  if (w.c_small) {
    if (!w.h_small) {
      K4_HIP(ix, hipHostMalloc((void**)&w.h_small, K4_SMALL_STAGE, hipHostMallocDefault));
      K4_HIP(ix, hipMalloc((void**)&w.d_small, K4_SMALL_STAGE));
    }
    const size_t reads_off = ((size_t)n * 12 + 15) & ~(size_t)15;
    memcpy(w.h_small, offs, (size_t)n * 8);
    memcpy(w.h_small + (size_t)n * 8, lens, (size_t)n * 4);
    memcpy(w.h_small + reads_off, reads, tot);
    K4_HIP(ix, hipMemcpyAsync(w.d_small, w.h_small, reads_off + tot, hipMemcpyHostToDevice, ix->stream));
    w.c_offs = (const uint64_t*)w.d_small;
    w.c_lens = (const uint32_t*)(w.d_small + (size_t)n * 8);
    w.c_reads = w.d_small + reads_off;
    w.c_out = (int32_t*)(w.d_small + K4_SMALL_STAGE / 2);
    w.c_hits = (k4_hit*)(w.d_small + K4_SMALL_STAGE / 2 + hits_off);
    return K4_OK;
  }
  if (tot + 64 > w.d_reads_cap) {
    if (w.d_reads) hipFree(w.d_reads);
    w.d_reads = nullptr;
    K4_HIP(ix, hipMalloc(&w.d_reads, tot + 64));
    w.d_reads_cap = tot + 64;
  }
  if (n > w.stage_reads || max_hits > w.stage_hits) {
    int64_t cap = std::max(n, w.stage_reads);
    int mh = std::max(max_hits, w.stage_hits);
    for (void* p : {(void*)w.d_offs, (void*)w.d_lens, (void*)w.d_out4, (void*)w.d_hits})
      if (p) hipFree(p);
    w.d_offs = nullptr; w.d_lens = nullptr; w.d_out4 = nullptr; w.d_hits = nullptr;
    K4_HIP(ix, hipMalloc(&w.d_offs, (size_t)(cap + 1) * 8));
    K4_HIP(ix, hipMalloc(&w.d_lens, (size_t)(cap + 1) * 4));
    K4_HIP(ix, hipMalloc(&w.d_out4, (size_t)(cap + 1) * 24));
    K4_HIP(ix, hipMalloc(&w.d_hits, (size_t)(cap + 1) * mh * sizeof(k4_hit)));
    w.stage_reads = cap;
    w.stage_hits = mh;
  }
  (void)out_bytes_per_read;
  w.c_reads = w.d_reads; w.c_offs = w.d_offs; w.c_lens = w.d_lens; w.c_out = w.d_out4; w.c_hits = w.d_hits;
  if (n) {
    K4_HIP(ix, hipMemcpyAsync(w.d_reads, reads, tot, hipMemcpyHostToDevice, ix->stream));
    K4_HIP(ix, hipMemcpyAsync(w.d_offs, offs, (size_t)n * 8, hipMemcpyHostToDevice, ix->stream));
    K4_HIP(ix, hipMemcpyAsync(w.d_lens, lens, (size_t)n * 4, hipMemcpyHostToDevice, ix->stream));
  }
  return K4_OK;
}

// small batch: results and hits come down in one copy into the pinned block; returns where they are
static int fetch_small(k4_index* ix, int64_t n, int max_hits, const int32_t** out, const k4_hit** hits) {
  K4Workspace& w = ix->ws;
  const size_t hits_off = ((size_t)n * 24 + 15) & ~(size_t)15;
  const size_t bytes = hits_off + (size_t)n * max_hits * sizeof(k4_hit);
  K4_HIP(ix, hipMemcpyAsync(w.h_small + K4_SMALL_STAGE / 2, w.d_small + K4_SMALL_STAGE / 2, bytes, hipMemcpyDeviceToHost, ix->stream));
  K4_HIP(ix, hipStreamSynchronize(ix->stream));
  *out = (const int32_t*)(w.h_small + K4_SMALL_STAGE / 2);
  *hits = (const k4_hit*)(w.h_small + K4_SMALL_STAGE / 2 + hits_off);
  return K4_OK;
}

// second segments of a host-pointer batch: a temporary device array, copied down after the stream has drained
struct Seg2Stage {
  void* d = nullptr;
  ~Seg2Stage() { if (d) hipFree(d); }
  int alloc(k4_index* ix, int64_t n) { K4_HIP(ix, hipMalloc(&d, (size_t)std::max<int64_t>(n, 1) * sizeof(k4_seg2))); return K4_OK; }
  int fetch(k4_index* ix, k4_seg2* out, int64_t n) {
    K4_HIP(ix, hipMemcpyAsync(out, d, (size_t)n * sizeof(k4_seg2), hipMemcpyDeviceToHost, ix->stream));
    K4_HIP(ix, hipStreamSynchronize(ix->stream));
    return K4_OK;
  }
};

extern "C" int k4_align_reads_batch(k4_index* ix, const k4_align_params* p, int64_t n, const uint8_t* reads,
                                    const uint64_t* offs, const uint32_t* lens, int32_t* rslt, int32_t* inst,
                                    int32_t* low, int32_t* nxt, k4_hit* hits) {
  return k4_align_reads_ext_batch(ix, p, n, reads, offs, lens, rslt, inst, low, nxt, hits, nullptr);
}
extern "C" int k4_align_reads_ext_batch(k4_index* ix, const k4_align_params* p, int64_t n, const uint8_t* reads,
                                        const uint64_t* offs, const uint32_t* lens, int32_t* rslt, int32_t* inst,
                                        int32_t* low, int32_t* nxt, k4_hit* hits, k4_seg2* seg2) {
  if (!ix) return K4_ERR_PARAMS;
  int rc = check_align_params(ix, p);
  if (rc != K4_OK) return rc;
  if (n < 0 || (n > 0 && (!reads || !offs || !lens || !rslt || !inst || !low || !nxt || !hits)))
    return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  if (n == 0) return K4_OK;
  K4_HIP(ix, hipSetDevice(ix->device));
  int max_len = 1;
  rc = stage_in(ix, n, reads, offs, lens, p->max_hits, &max_len, 16);
  if (rc != K4_OK) return rc;
  K4Workspace& w = ix->ws;
  int32_t* o = w.c_out;
  Seg2Stage s2;
  if (seg2 && (rc = s2.alloc(ix, n)) != K4_OK) return rc;
  rc = k4_align_reads_ext_batch_dev(ix, p, n, max_len, w.c_reads, w.c_offs, w.c_lens, o, o + n, o + 2 * n, o + 3 * n,
                                    w.c_hits, s2.d, ix->stream);
  if (rc != K4_OK) return rc;
  if (seg2 && (rc = s2.fetch(ix, seg2, n)) != K4_OK) return rc;
  if (w.c_small) {
    const int32_t* r;
    const k4_hit* h;
    rc = fetch_small(ix, n, p->max_hits, &r, &h);
    if (rc != K4_OK) return rc;
    memcpy(rslt, r, (size_t)n * 4); memcpy(inst, r + n, (size_t)n * 4);
    memcpy(low, r + 2 * n, (size_t)n * 4); memcpy(nxt, r + 3 * n, (size_t)n * 4);
    memcpy(hits, h, (size_t)n * p->max_hits * sizeof(k4_hit));
    return K4_OK;
  }
  K4_HIP(ix, hipMemcpyAsync(rslt, o, (size_t)n * 4, hipMemcpyDeviceToHost, ix->stream));
  K4_HIP(ix, hipMemcpyAsync(inst, o + n, (size_t)n * 4, hipMemcpyDeviceToHost, ix->stream));
  K4_HIP(ix, hipMemcpyAsync(low, o + 2 * n, (size_t)n * 4, hipMemcpyDeviceToHost, ix->stream));
  K4_HIP(ix, hipMemcpyAsync(nxt, o + 3 * n, (size_t)n * 4, hipMemcpyDeviceToHost, ix->stream));
  K4_HIP(ix, hipMemcpyAsync(hits, w.c_hits, (size_t)n * p->max_hits * sizeof(k4_hit), hipMemcpyDeviceToHost, ix->stream));
  K4_HIP(ix, hipStreamSynchronize(ix->stream));
  return K4_OK;
}

extern "C" int k4_best_matches_batch(k4_index* ix, const k4_align_params* p, int64_t n, const uint8_t* reads, const uint64_t* offs,
                                     const uint32_t* lens, int32_t* rslt, int32_t* inst, k4_hit* hits) {
  if (!ix) return K4_ERR_PARAMS;
  int rc = check_align_params(ix, p);
  if (rc != K4_OK) return rc;
  if (n < 0 || (n > 0 && (!reads || !offs || !lens || !rslt || !inst || !hits))) return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  if (n == 0) return K4_OK;
  K4_HIP(ix, hipSetDevice(ix->device));
  int max_len = 1;
  rc = stage_in(ix, n, reads, offs, lens, p->max_hits, &max_len, 16);
  if (rc != K4_OK) return rc;
  K4Workspace& w = ix->ws;
  int32_t* o = w.c_out;
  rc = k4_best_matches_batch_dev(ix, p, n, max_len, w.c_reads, w.c_offs, w.c_lens, o, o + n, w.c_hits, ix->stream);
  if (rc != K4_OK) return rc;
  if (w.c_small) {
    const int32_t* r;
    const k4_hit* h;
    rc = fetch_small(ix, n, p->max_hits, &r, &h);
    if (rc != K4_OK) return rc;
    memcpy(rslt, r, (size_t)n * 4); memcpy(inst, r + n, (size_t)n * 4);
    memcpy(hits, h, (size_t)n * p->max_hits * sizeof(k4_hit));
    return K4_OK;
  }
  K4_HIP(ix, hipMemcpyAsync(rslt, o, (size_t)n * 4, hipMemcpyDeviceToHost, ix->stream));
  K4_HIP(ix, hipMemcpyAsync(inst, o + n, (size_t)n * 4, hipMemcpyDeviceToHost, ix->stream));
  K4_HIP(ix, hipMemcpyAsync(hits, w.c_hits, (size_t)n * p->max_hits * sizeof(k4_hit), hipMemcpyDeviceToHost, ix->stream));
  K4_HIP(ix, hipStreamSynchronize(ix->stream));
  return K4_OK;
}

extern "C" int k4_kalign_batch(k4_index* ix, const k4_kalign_params* p, int64_t n, const uint8_t* reads,
                               const uint64_t* offs, const uint32_t* lens, k4_read_result* out, k4_hit* hits) {
  return k4_kalign_ext_batch(ix, p, n, reads, offs, lens, out, hits, nullptr);
}
extern "C" int k4_kalign_ext_batch(k4_index* ix, const k4_kalign_params* p, int64_t n, const uint8_t* reads,
                                   const uint64_t* offs, const uint32_t* lens, k4_read_result* out, k4_hit* hits, k4_seg2* seg2) {
  if (!ix || !p) return K4_ERR_PARAMS;
  if (n < 0 || (n > 0 && (!reads || !offs || !lens || !out || !hits))) return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  if (p->max_ml < 1) return k4_fail(ix, K4_ERR_PARAMS, "max_ml must be >= 1");
  if (n == 0) return K4_OK;
  K4_HIP(ix, hipSetDevice(ix->device));
  int max_len = 1;
  int rc = stage_in(ix, n, reads, offs, lens, p->max_ml, &max_len, 24);
  if (rc != K4_OK) return rc;
  K4Workspace& w = ix->ws;
  Seg2Stage s2;
  if (seg2 && (rc = s2.alloc(ix, n)) != K4_OK) return rc;
  rc = k4_kalign_ext_batch_dev(ix, p, n, max_len, w.c_reads, w.c_offs, w.c_lens, w.c_out, w.c_hits, s2.d, ix->stream);
  if (rc != K4_OK) return rc;
  if (seg2 && (rc = s2.fetch(ix, seg2, n)) != K4_OK) return rc;
  if (w.c_small) {
    const int32_t* r;
    const k4_hit* h;
    rc = fetch_small(ix, n, p->max_ml, &r, &h);
    if (rc != K4_OK) return rc;
    memcpy(out, r, (size_t)n * sizeof(k4_read_result));
    memcpy(hits, h, (size_t)n * p->max_ml * sizeof(k4_hit));
    return K4_OK;
  }
  K4_HIP(ix, hipMemcpyAsync(out, w.c_out, (size_t)n * sizeof(k4_read_result), hipMemcpyDeviceToHost, ix->stream));
  K4_HIP(ix, hipMemcpyAsync(hits, w.c_hits, (size_t)n * p->max_ml * sizeof(k4_hit), hipMemcpyDeviceToHost, ix->stream));
  K4_HIP(ix, hipStreamSynchronize(ix->stream));
  return K4_OK;
}

extern "C" int k4_enable_kernel_timing(k4_index* ix, int on) {
  if (!ix) return K4_ERR_PARAMS;
  ix->timing = on != 0;
  return K4_OK;
}

extern "C" int k4_get_kernel_times_split(k4_index* ix, double* step_ms, double* general_ms, int32_t* launches) {
  if (!ix || !step_ms || !general_ms || !launches) return K4_ERR_PARAMS;
  K4_HIP(ix, hipSetDevice(ix->device));
  K4_HIP(ix, hipDeviceSynchronize());
  double tot = 0, tot_g = 0;
  for (size_t j = 0; j < ix->ev_used; j++) {
    float ms = 0;
    K4_HIP(ix, hipEventElapsedTime(&ms, ix->ev0[j], ix->ev1[j]));
    tot += ms;
    K4_HIP(ix, hipEventElapsedTime(&ms, ix->ev1[j], ix->ev2[j]));
    tot_g += ms;
  }
  *step_ms = tot;
  *general_ms = tot_g;
  *launches = (int32_t)ix->ev_used;
  ix->ev_used = 0;
  return K4_OK;
}
extern "C" int k4_get_kernel_times(k4_index* ix, double* fast_ms, int32_t* launches) {
  double g = 0;
  return k4_get_kernel_times_split(ix, fast_ms, &g, launches);
}

extern "C" int k4_get_counters(k4_index* ix, k4_counters* out) {
  if (!ix || !out) return K4_ERR_PARAMS;
  K4_HIP(ix, hipSetDevice(ix->device));
  K4_HIP(ix, hipDeviceSynchronize());
  K4_HIP(ix, hipMemcpy(out, ix->counters, sizeof(k4_counters), hipMemcpyDeviceToHost));
  return K4_OK;
}

// test/profiling hook (not part of the ABI header): the K4_PROF_SLOTS slots behind the counters; zeroes them
extern "C" int k4i_debug_prof(k4_index* ix, uint64_t* out) {
  if (!ix || !out) return K4_ERR_PARAMS;
  K4_HIP(ix, hipSetDevice(ix->device));
  K4_HIP(ix, hipDeviceSynchronize());
  K4_HIP(ix, hipMemcpy(out, (char*)ix->counters + sizeof(k4_counters), K4_PROF_SLOTS * 8, hipMemcpyDeviceToHost));
  K4_HIP(ix, hipMemset((char*)ix->counters + sizeof(k4_counters), 0, K4_PROF_SLOTS * 8));
  return K4_OK;
}

extern "C" int k4_reset_counters(k4_index* ix) {
  if (!ix) return K4_ERR_PARAMS;
  K4_HIP(ix, hipSetDevice(ix->device));
  K4_HIP(ix, hipDeviceSynchronize());
  K4_HIP(ix, hipMemset(ix->counters, 0, sizeof(k4_counters)));
  return K4_OK;
}
