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
#include <stdlib.h>
#include <type_traits>
#include "k4_align_common.h"

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
  // what a lane keeps of a table entry: with 4-byte suffix elements (fewer than 2^32 - 1 symbols) 32 bits hold lb and pos0 whatever
  // the entry's own width -- the 16-byte entries in 64-bit registers cost the 100 bp kernel 40 spilled registers more
  typedef typename std::conditional<EL == 4, uint32_t, KT>::type RT;
  constexpr uint64_t NOPOS = sizeof(RT) == 4 ? 0xFFFFFFFFull : K4_KTAB64_MASK;  // pos0 not usable (it is the whole bucket's first suffix)
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
      RT lb0[PFN], ps0[PFN], lb1[PFN];
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
              lb0[0] = 0; lb1[0] = 1; ps0[0] = (RT)(mv & 0xFFFFFFFFFFull);
              memo_hit = true; memo_mm = mmv;
              continue;
            }
          }
          const uint64_t code = ln.chunk_at(s, oo[j]) >> (64 - 2 * kk);
          uint64_t sub;
          {
            KT e_lb0, e_ps0, e_lb1;
            k4d_ktab_fetch<KT>(ix, code << tshift, (code + 1) << tshift, e_lb0, e_ps0, sig[j], e_lb1, sub,
                               sizeof(KT) == 8 && tshift == 0 && cl >= kk + 2);  // (measured on the repeat-rich index too: 24.9 -> 23.8 ms)
            lb0[j] = (RT)e_lb0; ps0[j] = (RT)e_ps0; lb1[j] = (RT)e_lb1;
          }
          if (CAPTURE && defer_deep && tshift == 0 && (uint64_t)(lb1[j] - lb0[j]) > K4_DEFER_BUCKET) return K4_DEFER;
          if (a.deep_general && (!CAPTURE || a.deep_general == 1) && tshift == 0 && (uint64_t)(lb1[j] - lb0[j]) > K4_DEFER_BUCKET) return K4_NEED_SLOW;
          if (sizeof(KT) == 8 && tshift == 0 && cl >= kk + 2 && sub != K4_KTAB64_IRREGULAR) {
            // straight to the suffixes that continue with the core's next two bases
            uint32_t before, count;
            k4d_ktab_sub(sub, (uint32_t)(ln.chunk_at(s, oo[j] + kk) >> 60), before, count);
            if (before) ps0[j] = (RT)NOPOS;  // pos0 is the whole bucket's first suffix, not this one's
            lb0[j] += (RT)before;
            lb1[j] = lb0[j] + (RT)count;
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
        if (j < cnt && tshift == 0 && lb1[j] > lb0[j] && !(j == 0 && memo_hit) && (sizeof(KT) == 4 || (uint64_t)ps0[j] != NOPOS))
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
          const uint64_t p = (tshift == 0 && mid == (int64_t)lb0[j] && (sizeof(KT) == 4 || (uint64_t)ps0[j] != NOPOS))
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
  static const int deep_mode = getenv("K4_DEEP_TO_GENERAL") ? atoi(getenv("K4_DEEP_TO_GENERAL")) : 0;  // (experiment)
  a.deep_general = (deep_mode && ix->deep_bucket_frac >= K4_DEFER_MIN_FRAC) ? deep_mode : 0;  // 1: from the first phase on, 2: from the second
  if (ix->deep_bucket_frac >= K4_DEFER_MIN_FRAC && a.deep_general != 1) {
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
  a.slow_gen = reinterpret_cast<uint32_t*>(w.slow_hash + (size_t)K4_SLOW_WAVES * K4_SMALL_HASH + (size_t)K4_HUGE_WAVES * w.slow_hash_cap);
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
  rc = k4i_launch_general(ix, a, max_len, st);
  if (rc != K4_OK) return rc;
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
  if (ix->d.el == 4) return ix->d.ktab64 ? launch_all<4, uint64_t>(ix, a, max_len, n_steps, st) : launch_all<4, uint32_t>(ix, a, max_len, n_steps, st);
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
