// kit4b_amd/csrc/k4_general.hip -- the general kernel of the kalign hot path on gfx950: one WAVE per read, for whatever the
// step kernels' 2-bit fast path (k4_align.hip) cannot decide -- N in the read, windows touching N runs / separators, deep
// repeats (more candidates than the fast path's dedupe list), reads longer than 512 bp -- and for the optional phases of
// AlignReads (k4_ext.h) and LocateBestMatches.  Reference semantics as listed at the top of k4_align.hip.
#include "k4_align_common.h"

// rarely taken or large paths stay out of line: inlined, their registers and hoisted values weigh on the hot loop's allocation
#define K4_DEV_OUT __device__ __attribute__((noinline))

// ==== general kernel =================================================================================================
// A value every lane of the wave holds alike, said so to the compiler: it then lives in a scalar register and branches on it are
// scalar branches -- without this the results of cross-lane reads (and everything computed from them: loop bounds, the replayed
// fold's state) count as divergent and the "sequential" replay runs as masked vector code.
K4_DEV int k4d_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
K4_DEV uint32_t k4d_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
K4_DEV uint64_t k4d_uni(uint64_t v) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
// The four waves of a block work on independent reads: a hand-off through LDS between the lanes of ONE wave needs no barrier
// (a wave's LDS instructions execute in order); only the compiler must not move accesses across the point.
#define K4_WSYNC()                                          \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)
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
  uint64_t* g_cm;    // LDS [K4_GROUP][4]: pair j's core as masks over the four 32-base chunks of a read of up to 128 bases
  uint64_t* g_lm;    // LDS [4]: the read's own bases, likewise
#ifdef K4_SLOW_PROF
  unsigned long long prof[24];
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
  K4_WSYNC();
  for (int x = sc.lane; x < len / 2; x += 64) {
    const uint8_t t = sc.probe[x];
    sc.probe[x] = sc.probe[len - 1 - x];
    sc.probe[len - 1 - x] = t;
  }
  K4_WSYNC();
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
    K4_WSYNC();
    for (uint32_t q = sc.lane; q < sc.lcap; q += 64) sc.lhash[q] = K4_LH_EMPTY;
    sc.lused = 0;
    K4_WSYNC();
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
  K4Tb tb;
  tb.init(ix);
  for (int j = jlo; j < jhi; j++) {
    const uint32_t t = tb.get((int64_t)(left + (uint64_t)j));
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
  K4_WSYNC();
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
  K4Tb tb;
  tb.init(ix);
  for (int j = 0; j < cl; j++) {
    const uint32_t t = tb.get((int64_t)(pos + (uint64_t)j));
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
    lo = (int64_t)k4d_uni(k4d_ktab_lb(ix, code << sh));
    hi = (int64_t)k4d_uni(k4d_ktab_lb(ix, (code + 1) << sh)) - 1;
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
template <int EL, bool END_CMP = true>
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
    lo = (int64_t)k4d_uni(k4d_ktab_lb(ix, code << sh));
    hi = (int64_t)k4d_uni(k4d_ktab_lb(ix, (code + 1) << sh)) - 1;
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
  if (END_CMP && last >= first && last + 1 < (int64_t)ix.n) end_cmp = (int64_t)k4d_sa_at<EL>(ix, (uint64_t)last + 1) + cl <= (int64_t)ix.n;
}

#ifndef K4_SCAN_MAX
#define K4_SCAN_MAX 128  // k-mer buckets up to this many suffixes are laid on the read whole; deeper ones are searched for the run's bounds first
#endif
#define K4_GROUP 64      // (strand, core) pairs looked up together
#define K4_TRIM_MAX_LEN 2048  // AdaptiveTrim turns longer reads down (:5601-5605): no mismatch vector is kept for them
K4_DEV uint64_t k4d_wave_excl_scan(uint64_t v, int lane, uint64_t& total) {
  unsigned long long x = v;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long y = __shfl_up(x, (unsigned)d, 64);
    if (lane >= d) x += y;
  }
  total = k4d_uni((uint64_t)__shfl(x, 63, 64));
  return x - v;
}

// The buckets / runs of up to K4_GROUP (strand, core) pairs as ONE sequence of slots in LDS (g_lb: first suffix-array index of
// pair j, g_pre: slots in front of it, g_pre[np]: all of them): lane j holds pair j's strand and core offset.  One round of
// k-mer table loads for all pairs; a bucket of up to K4_SCAN_MAX suffixes is taken whole (the walk's own window fetch tells the
// members), a deeper one has the two bounds of its run searched (k4d_exact_run_wave), one pair after the other.
template <int EL>
K4_DEV uint64_t k4d_group_lookup(const K4DevIndex& ix, K4Slow& sc, int np, unsigned long long smask, int my_o, int cl, uint32_t& n_probe) {
  const int lane = sc.lane;
  const int kk = min((int)ix.k, cl);
  const int ksh = 2 * ((int)ix.k - kk);
  uint64_t lb0 = 0, size = 0;
  bool big = false;
  if (lane < np) {
    const int my_s = (int)((smask >> lane) & 1ull);
    bool acgt = true;
    uint64_t code = 0;
    if (sc.packed)
      code = k4d_probe_chunk(sc, my_o, my_s) >> (64 - 2 * kk);
    else {
      const uint8_t* pb = sc.probe + (my_s ? sc.pstride : 0u) + my_o;
      for (int j = 0; j < kk; j++) {
        const uint32_t b = pb[j] & 0x0f;
        if (b > 3) { acgt = false; break; }
        code = (code << 2) | b;
      }
    }
    if (acgt) {
      lb0 = k4d_ktab_lb(ix, code << ksh);
      size = k4d_ktab_lb(ix, (code + 1) << ksh) - lb0;
    } else
      size = ix.n;  // (a core that holds N: the whole array is searched, as LocateFirstExact would)
    big = size > K4_SCAN_MAX;
  }
  for (unsigned long long bigm = __ballot(big); bigm; bigm &= bigm - 1) {
    const int j = __ffsll((long long)bigm) - 1;
    const int o_j = k4d_uni(__shfl(my_o, j, 64));
    int64_t first, last;
    bool end_cmp;
    k4d_exact_run_wave<EL, false>(ix, sc, o_j, cl, n_probe, first, last, end_cmp, (int)((smask >> j) & 1ull));
    if (lane == j) { lb0 = (uint64_t)first; size = last >= first ? (uint64_t)(last - first + 1) : 0ull; }
  }
  uint64_t total;
  const uint64_t pre = k4d_wave_excl_scan(lane < np ? size : 0ull, lane, total);
  if (lane < np) { sc.g_lb[lane] = lb0; sc.g_pre[lane] = pre; }
  if (lane == 63) sc.g_pre[np] = total;
  K4_WSYNC();
  return total;
}

#include "k4_ext.h"

// ---- the batched LocateCoreMultiples of the general kernel ------------------------------------------------------------

// the reverse complement of the probe behind the forward one (bytes as k4d_revcomp_wave would leave them, then packed)
K4_DEV void k4d_make_rc_wave(K4Slow& sc, int len) {
  uint8_t* prc = sc.probe + sc.pstride;
  int stop = len;  // CSeqTrans::ReverseComplement (SeqTrans.cpp:497-545) complements up to the first symbol above 6
  for (int j0 = 0; j0 < len; j0 += 64) {
    const int j = j0 + sc.lane;
    const bool bad = j < len && (sc.probe[j] & 0x0f) > 6;
    const unsigned long long m = __ballot(bad);
    if (m) { stop = j0 + __ffsll((long long)m) - 1; break; }
  }
  for (int j = sc.lane; j < len; j += 64) {
    uint8_t b = sc.probe[j];
    if (j < stop && b <= 3) b = 3 - b;
    prc[len - 1 - j] = b;
  }
  K4_WSYNC();
  const int nw = (len + 31) >> 5;
  for (int w = sc.lane; w <= nw; w += 64) {
    uint64_t acc = 0;
    if (w < nw)
      for (int q = 0; q < 32; q++) {
        const int j = 32 * w + q;
        uint32_t b = j < len ? (prc[j] & 0x0f) : 0u;
        if (b > 3) b = 0;
        acc = (acc << 2) | b;
      }
    sc.pk[sc.pkstride + w] = acc;
  }
  K4_WSYNC();
}

// bit k of x -> bit 2k (the even bits of the result)
K4_DEV uint64_t k4d_spread32(uint32_t v) {
  uint64_t x = v;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}
// A read into the wave's LDS: its bytes and those of its reverse complement (for the exact-symbol paths), and both as packed
// words.  64 bases a round, a lane per base: the two bit planes of the symbols are wave ballots, a packed word is their
// interleave (scalar arithmetic, no per-base loop); the reverse complement's words come from the forward ones (complement,
// order of the 2-bit groups reversed) as the step kernels make theirs.  Same results as k4d_pack_probe_wave + k4d_make_rc_wave.
K4_DEV void k4d_load_read_wave(K4Slow& sc, const uint8_t* __restrict__ src, int len) {
  uint8_t* prc = sc.probe + sc.pstride;
  const int lane = sc.lane;
  int stop = len;  // CSeqTrans::ReverseComplement (SeqTrans.cpp:497-545) complements up to the first symbol above 6
  bool bad = false;
  const int nw = (len + 31) >> 5;
  for (int j0 = 0; j0 < len; j0 += 64) {
    const int j = j0 + lane;
    const uint32_t b = j < len ? (uint32_t)(src[j] & 7) : 0u;
    const unsigned long long m6 = __ballot(b > 6);
    if (m6 && stop == len) stop = j0 + __ffsll((long long)m6) - 1;
    if (j < len) {
      sc.probe[j] = (uint8_t)b;
      prc[len - 1 - j] = (uint8_t)((j < stop && b <= 3) ? 3 - b : b);
    }
    if (__ballot(b > 3)) bad = true;
    const unsigned long long m0 = __ballot((b & 1u) && b <= 3), m1 = __ballot((b & 2u) && b <= 3);  // (symbols above T pack as 0)
    // base i of the round (bit i of the planes) goes to bits (63 - 2i, 62 - 2i) of its word: reverse, spread, interleave
    const uint64_t w_lo = (k4d_spread32(__brev((uint32_t)m1)) << 1) | k4d_spread32(__brev((uint32_t)m0));
    const uint64_t w_hi = (k4d_spread32(__brev((uint32_t)(m1 >> 32))) << 1) | k4d_spread32(__brev((uint32_t)(m0 >> 32)));
    if (lane == 0) {
      sc.pk[j0 >> 5] = w_lo;
      sc.pk[(j0 >> 5) + 1] = w_hi;  // (the word behind the read's last one is zero: pk has nw + 1 words at least)
    }
  }
  if (lane == 0 && (nw & 1) == 0) sc.pk[nw] = 0;  // (an odd nw had its pad word written as the round's upper half)
  sc.packed = !bad;
  K4_WSYNC();
  for (int c = lane; c <= nw; c += 64) {  // reverse complement words from the forward ones
    uint64_t r = 0;
    if (c < nw) {
      const int o = len - 32 * (c + 1);
      uint64_t f;
      if (o >= 0) f = k4d_probe_chunk(sc, o);
      else f = sc.pk[0] >> (2 * (-o));  // fewer than 32 bases left: they sit at the low end, zeros above
      const uint64_t y = __brevll(~f);
      r = (((y & 0x5555555555555555ull) << 1) | ((y >> 1) & 0x5555555555555555ull)) & k4d_range_mask(0, len - 32 * c);
    }
    sc.pk[sc.pkstride + c] = r;
  }
  K4_WSYNC();
}

// One lane: the read (strand s) laid on the clean window [left, left + len): does its core [o, o + cl) equal the reference
// there, and the Hamming distance of the whole read -- both from the same 16-byte loads (two per 113 bases).
K4_DEV void k4d_lane_window(const K4DevIndex& ix, const K4Slow& sc, int s, int o, int cl, int len, int64_t left, bool& core_eq, int& mm) {
  uint64_t diff = 0;
  mm = 0;
  const int al = (int)(left & 15);
  for (int c0 = 0; 32 * c0 < len; c0 += 4) {
    const int rem = len - 32 * c0;
    uint64_t rc[4];
    k4d_ref_chunks4(ix, left, c0, rem + al <= 128, rc);
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (32 * c < rem) {
        const int b = 32 * (c0 + c);
        const uint64_t x = (rc[c] ^ k4d_probe_chunk(sc, b, s)) & k4d_range_mask(0, rem - 32 * c);
        mm += (int)k4d_mm_count(x);
        diff |= x & k4d_range_mask(o - b, o + cl - b);
      }
  }
  core_eq = diff == 0;
}

// ... for a read of up to 128 bases from its nine words (fetched before the exception test, so that the test's own memory
// access runs beside them): chunk masks of the core (cm) and of the read (lm) come from LDS, the 64-bit funnel shifts from two
// 32-bit v_alignbit each
template <bool MK>
K4_DEV void k4d_lane_window128(const K4Slow& sc, const uint32_t (&wv)[9], int64_t left, int s, const uint64_t* cm, bool& core_eq, int& mm,
                               uint32_t* mk = nullptr) {
  const uint32_t sh = (uint32_t)(left & 15) * 2;
  const uint64_t* pk = sc.pk + (s ? sc.pkstride : 0u);
  uint64_t diff = 0;
  mm = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    // 64 bits of the window from bit offset sh of the words 2c, 2c+1, 2c+2 (sh < 32): alignbit(hi, lo, 32 - sh) = (hi:lo) >> (32 - sh)
    const uint32_t hi = sh ? __builtin_amdgcn_alignbit(wv[2 * c], wv[2 * c + 1], 32 - sh) : wv[2 * c];
    const uint32_t lo = sh ? __builtin_amdgcn_alignbit(wv[2 * c + 1], wv[2 * c + 2], 32 - sh) : wv[2 * c + 1];
    const uint64_t x = ((((uint64_t)hi << 32) | lo) ^ pk[c]) & sc.g_lm[c];
    mm += (int)k4d_mm_count(x);
    diff |= x & cm[c];
    if (MK) mk[c * 64] = k4d_mm_bits(x);  // the chimeric pass trims from the mismatch vector (k4d_build_mm_vector's form)
  }
  core_eq = diff == 0;
}


// One LocateCoreMultiples call (SfxArray.cpp:5806-6369) by one wave, every memory-bound part of it batched.  The cores of a
// call depend only on (ProbeLen, CoreLen, CoreDelta, MaxNumCoreSlides) (:5948-5959) and both strands use the same offsets,
// so the (strand, core) pairs are known before anything is looked up:
//   1. k-mer table entries of up to K4_GROUP pairs in ONE round of loads (a lane per pair) -> each pair's bucket;
//   2. a bucket of up to K4_SCAN_MAX suffixes is not searched at all: the suffix array is sorted by the comparison the
//      reference's walk uses, so the run of suffixes that start with the core is a contiguous part of the bucket, and a
//      suffix is a member iff its window equals the core -- which the Hamming extension's own window fetch decides for free.
//      Only deeper buckets get their run's two bounds searched first (k4d_exact_run_wave);
//   3. all buckets of the group become one sequence of slots (prefix sums over the sizes), 64 slots per step, a lane per
//      slot: suffix element (fetched a step ahead) -> window -> (member?, distance).  A call is three dependent rounds of
//      memory accesses plus one per further 64 slots, where the walk core by core took eight and more per core;
//   4. what the reference's sequential loop makes order-dependent -- the dedupe table per strand pass, MaxIter and the node
//      limit counting only new in-bounds candidates, the fold and its early exit, the tallies -- is replayed pair by pair, in
//      suffix order, from the lanes' results (LDS and registers only).
// CHIM: the chimeric branch (:6064-6189) -- every new in-bounds candidate is flank-trimmed by AdaptiveTrim (one lane each,
// its mismatch vector in the lane's column of mk) instead of being counted out by the Hamming extension, and the fold ranks
// by trimmed length first.  min_probe_chim = MinProbeChimericLen (:5880).
template <int EL, bool CHIM>
K4_DEV int k4d_lcm_batched(const K4AlignArgs& a, K4Slow& sc, int len, int allow_mm, int cl, int core_delta,
                           const K4ReadParams& rp, int* p_inst, int* p_low, int* p_nxt, k4_hit* hits, uint32_t& n_lookup,
                           uint32_t& n_probe, uint32_t& n_cand, int min_probe_chim = 0, uint32_t* mk = nullptr) {
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
  k4_hit* hits_w = lane == 0 ? hits : nullptr;  // hits are stored by lane 0 only (every lane folds the same wave-uniform state)
  // cMaxNumIdentNodes (SfxArray.h:15); additionally bounded by the scratch table so an insert always terminates.
  // A pass that fills a small table before the reference's own limit is redone with a big one (K4_NEED_SLOW).
  const uint32_t node_cap = min((uint32_t)K4_MAX_IDENT_NODES, sc.lhash ? sc.lcap * 3 / 4 : sc.cap / 2 - 1);
  // generator of the (strand, core) pairs in the reference's order: '+' cores, then '-' cores (:5925-5934,5948-5959,6323-6336)
  int gs = rp.strand == K4_STRAND_CRICK ? 1 : 0;
  const int gs_end = rp.strand == K4_STRAND_WATSON ? 0 : 1;
  int go_next = 0, g_delta = core_delta, g_slides = 0;
  // state of the replayed walk
  int cur_s = -1, iter = 0;
  bool strand_dead = false, core_done = false;
  uint32_t n_nodes = 0;
  for (;;) {
    // ---- the next group of pairs ------------------------------------------------------------------------------------
    int np = 0, my_o = 0;
    unsigned long long smask = 0;  // bit j: pair j is on the '-' strand
    while (np < K4_GROUP && gs <= gs_end) {
      if (g_slides < rp.max_slides && go_next <= len - cl && g_delta > cl / 3) {
        if (go_next + cl + g_delta > len) g_delta = len - (go_next + cl);
        if (lane == np) { my_o = go_next; sc.g_o[np] = (uint16_t)go_next; }
        if (gs) smask |= 1ull << np;
        np++; g_slides++; go_next += g_delta;
      } else {
        gs++; go_next = 0; g_delta = core_delta; g_slides = 0;
      }
    }
    if (np == 0) break;
    if (len <= 128 && lane < np) {  // (lane j made pair j)
#pragma unroll
      for (int c = 0; c < 4; c++) sc.g_cm[4 * lane + c] = k4d_range_mask(my_o - 32 * c, my_o + cl - 32 * c);
    }
    K4_PROF_T(pg0);
    K4_PROF_ADD(8, 1);
    K4_PROF_ADD(6, np);
    // ---- 1.-3. the k-mer table (a lane per pair), deep buckets searched, the slots of the group --------------------------
    const uint64_t total = k4d_group_lookup<EL>(ix, sc, np, smask, my_o, cl, n_probe);
    K4_PROF_T(pg1);
    K4_PROF_ADD(0, pg1 - pg0);
    int opened = 0;  // pairs of this group the walk has reached so far
    bool v_n = false;
    int pj_n = 0;
    uint64_t pos_n = 0;
    uint64_t base = 0;
    bool reload = true;  // the suffix elements of the step at `base` are not on their way yet
    bool stop_all = false;
    while (base < total) {
      K4_PROF_T(ps0);
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
      K4_PROF_ADD(9, 1);
      // the slot's window: member of its pair's run?  distance of the whole read?
      bool core_eq = false, clean = false;
      int mm = 0, o = 0, s = 0;
      int64_t left = 0;
      if (valid) {
        o = (int)sc.g_o[pj];
        s = (int)((smask >> pj) & 1ull);
        left = (int64_t)pos - o;
        const bool inside = sc.packed && left >= 0 && (uint64_t)left + (uint64_t)len <= ix.n;
        uint32_t wv[9];
        if (inside && len <= 128) k4d_ref_words9(ix, left, 0, len + (int)(left & 15) <= 128, wv);  // on their way during the test below
        clean = inside && !k4d_any_exc_sup(ix, sc.sup, left, left + len);
        if (clean) {
          if (len <= 128) k4d_lane_window128<CHIM>(sc, wv, left, s, sc.g_cm + 4 * pj, core_eq, mm, mk);
          else k4d_lane_window(ix, sc, s, o, cl, len, left, core_eq, mm);
        } else
          core_eq = k4d_lane_cmp(ix, sc, o, cl, pos, s) == 0;
      }
      // CHIM: AdaptiveTrim (:6097: ProbeLen, probe, target, MinProbeChimericLen, MaxTotMM, 3 flank matches) depends on the locus
      // alone, not on the walk's state: every member's trim is worked out here, all lanes side by side, instead of pair after
      // pair in the replay (a short chimeric core gives a dozen pairs with one or two members each)
      K4Trim trim;
      trim.len = trim.t5 = trim.t3 = trim.mms = 0;
      if (CHIM && valid && core_eq && left >= 0 && len <= K4_TRIM_MAX_LEN) {
        if (!(clean && len <= 128)) k4d_build_mm_vector(ix, sc, len, (uint64_t)left, mk, s);
        if (k4d_trim_possible(mk, len, min_probe_chim, allow_mm)) trim = k4d_adaptive_trim(mk, len, min_probe_chim, allow_mm, 3);
      }
      const unsigned long long validm = __ballot(valid);
      n_probe += (uint32_t)__popcll(validm);
      K4_PROF_ADD(7, __popcll(validm));
      K4_PROF_ADD(12, __popcll(__ballot(core_eq)));
      K4_PROF_T(ps1);
      K4_PROF_ADD(1, ps1 - ps0);
      // filters that precede the dedupe (:6019-6036: before the core offset, on a separator, over the entry end): for every
      // slot of the step at once, not pair by pair
      uint32_t loci = 0, ent_id_l = 0;
      bool in_bounds_all = false;
      {
        uint64_t e_start = 0, e_end = 0;
        int e = -1;
        const bool member = valid && core_eq && pos >= (uint64_t)o;
        if (member) e = k4d_map_entry_slow(ix, sc.ent, (uint64_t)left, e_start, e_end);
        in_bounds_all = member && e >= 0 && (uint64_t)left + (uint64_t)len - 1 <= e_end;
        loci = (uint32_t)((uint64_t)left - e_start);
        if (in_bounds_all) ent_id_l = sc.ent_id[e];
      }
      // ---- 4. replay, pair by pair in the reference's order ------------------------------------------------------------
      const int j_lo = k4d_uni(__shfl(pj, 0, 64)), j_hi = k4d_uni(__shfl(pj, 63 - __clzll(validm), 64));
      uint64_t next_base = base + 64;
      for (int jj = j_lo; jj <= j_hi; jj++) {
        const unsigned long long seg = __ballot(valid && pj == jj);
        if (!seg) continue;
        K4_PROF_T(pq0);
        K4_PROF_ADD(21, 1);
        for (; opened <= jj; opened++) {  // the for-loop head of :5948-5959 for every pair up to this one
          // behind a core's walk (:6313-6321): nothing can improve on more than MaxHits exact instances.  The fold below leaves
          // at once when that happens -- except the chimeric one, which does so only for untrimmed candidates (:6187)
          if (CHIM && st.inst > rp.max_hits && st.low == 0) { stop_all = true; break; }
          const int s_p = (int)((smask >> opened) & 1ull);
          if (s_p != cur_s) {  // a new strand pass: fresh dedupe table (:5946-5947)
            cur_s = s_p;
            k4d_hash_new_pass(sc);
            n_nodes = 0;
            strand_dead = false;
          }
          if (n_nodes >= node_cap) strand_dead = true;
          if (!strand_dead) n_lookup++;
          iter = 0;
          core_done = false;
        }
        if (stop_all) break;
        K4_PROF_T(pq1);
        K4_PROF_ADD(16, pq1 - pq0);
        const bool pair_over = strand_dead || core_done;
        if (!pair_over) {
          const int cs = (int)((smask >> jj) & 1ull);
          const char cur_strand = cs ? '-' : '+';
          const bool in_seg = (seg >> lane) & 1ull;
          const bool in_bounds = in_seg && in_bounds_all;
          K4_PROF_ADD(10, __popcll(__ballot(in_bounds)));
          // the LDS table must keep room for this step's inserts (slots of retracted inserts count): else the pass with the big tables
          if (sc.lhash && sc.lused + 64 + 1 > sc.lcap) return K4_NEED_SLOW;
          bool isnew = false;
          uint32_t slot = 0;
          if (in_bounds) isnew = k4d_hash_insert_lane(sc, (uint32_t)(1 + pos - (uint32_t)o), slot);
          unsigned long long newm = __ballot(isnew);
          if (sc.lhash) sc.lused += (uint32_t)__popcll(newm);  // (a retracted insert keeps its slot)
          K4_PROF_T(pq2);
          K4_PROF_ADD(17, pq2 - pq1);
          // MaxIter / node limit: both count new in-bounds candidates only; the walk stops before the suffix after the last
          // one it may take.  Inserts behind that point are retracted.
          const uint32_t rem_iter = max_iter ? (uint32_t)(max_iter - iter) : 0xFFFFFFFFu;
          const uint32_t remaining = min(rem_iter, node_cap - n_nodes);
          bool hit_limit = false;
          if ((uint32_t)__popcll(newm) >= remaining) {
            unsigned long long mrem = newm;
            for (uint32_t q = 1; q < remaining; q++) mrem &= mrem - 1;  // drop the lowest remaining-1 bits
            const int lastl = __ffsll((long long)mrem) - 1;
            hit_limit = true;
            const unsigned long long beyond = lastl >= 63 ? 0ull : (~0ull << (lastl + 1));
            if (isnew && ((beyond >> lane) & 1ull)) k4d_hash_retract(sc, slot);
            newm &= ~beyond;
          }
          const bool is_cand = (newm >> lane) & 1ull;
          bool eos = false;
          if (!CHIM && is_cand && !clean) {  // the Hamming extension (:6200-6261) over exact symbols
            bool all_eq;
            k4d_lane_range(ix, sc.probe + (cs ? sc.pstride : 0u), 0, len, (uint64_t)left, false, all_eq, eos, mm);
          }
          K4_PROF_T(pq3);
          K4_PROF_ADD(18, pq3 - pq2);
          K4_PROF_ADD(20, __popcll(__ballot(is_cand && !clean)));
          // only candidates that pass the order-independent part of the acceptance test can change the state; the rest just count
          const bool cand = CHIM ? is_cand && trim.len >= min_probe_chim && trim.len > 0 : is_cand && !eos && mm <= allow_mm;
          unsigned long long todo = __ballot(cand);
          int stop_lane = -1;
          while (todo) {
            K4_PROF_ADD(22, 1);
            const int c = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            if (CHIM) {  // the fold of :6106-6188
              const int c_len = k4d_uni(__shfl(trim.len, c, 64)), c_mms = k4d_uni(__shfl(trim.mms, c, 64));
              const int t5 = k4d_uni(__shfl(trim.t5, c, 64)), t3 = k4d_uni(__shfl(trim.t3, c, 64));
              const uint32_t ent_c = (uint32_t)k4d_uni(__shfl((int)ent_id_l, c, 64));
              const uint32_t loci_c = (uint32_t)k4d_uni(__shfl((int)loci, c, 64));
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
            const int mm_c = k4d_uni(__shfl(mm, c, 64));
            if (mm_c >= st.nxt) continue;
            const uint32_t ent_c = (uint32_t)k4d_uni(__shfl((int)ent_id_l, c, 64));
            const uint32_t loci_c = (uint32_t)k4d_uni(__shfl((int)loci, c, 64));
            k4d_fold(st, mm_c, hits_w, rp.max_hits, ent_c, loci_c, len, cur_strand);
            if (st.inst > rp.max_hits && st.low == 0) { stop_lane = c; break; }
            // what is left of a repeat family's batch mostly cannot change the state any more: candidates at or above
            // NxtLowMMCnt are no-ops (it only ever drops), and once the hit slots are full a candidate that ties with
            // LowMMCnt only counts -- those in front of the next better one are counted in one go
            todo &= __ballot(mm < st.nxt);
            if (st.inst >= rp.max_hits && st.low > 0) {
              const unsigned long long better = todo & __ballot(mm < st.low);
              const unsigned long long ties = todo & __ballot(mm == st.low) & (better ? (better & (0ull - better)) - 1ull : ~0ull);
              st.inst += (int)__popcll(ties);
              todo &= ~ties;
            }
          }
          K4_PROF_T(pq4);
          K4_PROF_ADD(19, pq4 - pq3);
          if (stop_lane >= 0) {  // early exit of :6313-6321: candidates behind it were never examined
            const unsigned long long upto = stop_lane >= 63 ? ~0ull : ((1ull << (stop_lane + 1)) - 1ull);
            const uint32_t took = (uint32_t)__popcll(newm & upto);
            iter += (int)took; n_cand += took; n_nodes += took;
            stop_all = true;
            break;
          }
          {
            const uint32_t took = (uint32_t)__popcll(newm);
            iter += (int)took; n_cand += took; n_nodes += took;
          }
          if (hit_limit) core_done = true;
          if (n_nodes >= node_cap && node_cap < (uint32_t)K4_MAX_IDENT_NODES && sc.small) return K4_NEED_SLOW;
        }
        // a pair whose walk is over (MaxIter, the node limit) is not looked at any further: on to the next pair's slots
        if ((strand_dead || core_done) && jj == j_hi && k4d_uni(sc.g_pre[jj + 1]) > base + 64) {
          next_base = k4d_uni(sc.g_pre[jj + 1]);
          reload = true;
        }
      }
      K4_PROF_T(ps2);
      K4_PROF_ADD(2, ps2 - ps1);
      if (stop_all) break;
      base = next_base;
    }
    if (stop_all) break;
    for (; opened < np; opened++) {  // the pairs behind the last slot (empty buckets) are reached as well
      if (CHIM && st.inst > rp.max_hits && st.low == 0) { stop_all = true; break; }
      const int s_p = (int)((smask >> opened) & 1ull);
      if (s_p != cur_s) {
        cur_s = s_p;
        k4d_hash_new_pass(sc);
        n_nodes = 0;
        strand_dead = false;
      }
      if (n_nodes >= node_cap) strand_dead = true;
      if (!strand_dead) n_lookup++;
      iter = 0;
      core_done = false;
    }
    if (stop_all) break;
    K4_WSYNC();  // (the group's LDS arrays are rewritten by the next group)
  }
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
  K4_PROF_T(pe0);
  if (rp.micro_indel_len > 0) {
    rslt = k4d_two_seg<EL>(a, sc, false, rp.micro_indel_len, min(rp.tot_mm, 2), rp.core_len, rp.strand, len, inst, low, nxt, &hits[0],
                           seg2, n_lookup, n_probe, n_cand, false);
    if (rslt != 0) return rslt;
  }
  if (rp.max_splice_junct_len > 0) {
    rslt = k4d_two_seg<EL>(a, sc, true, rp.max_splice_junct_len, min(rp.tot_mm, 2), rp.core_len, rp.strand, len, inst, low, nxt,
                           &hits[0], seg2, n_lookup, n_probe, n_cand, rp.micro_indel_len > 0 && rp.core_len <= len);  // (same cores: the slots are laid out)
    if (rslt != 0) return rslt;
  }
  K4_PROF_T(pe1);
  K4_PROF_ADD(19, pe1 - pe0);
  if (rp.min_chimeric_len > 0) {
    if (rp.max_slides <= 1) return K4_ERR_PARAMS;  // (the reference divides by MaxNumCoreSlides - 1, :7926)
    const int cl = max(rp.min_core_len, len / (rp.tot_mm + 4));
    const int cd = max(len / (rp.max_slides - 1), cl);
    if (cl < 1) return K4_ERR_PARAMS;
    if (rp.min_chimeric_len >= 15 && rp.min_chimeric_len <= 99)  // :5878-5883 any other value: the default branch
      rslt = k4d_lcm_batched<EL, true>(a, sc, len, rp.tot_mm, cl, cd, rp, inst, low, nxt, hits, n_lookup, n_probe, n_cand,
                                    max(cl, (rp.min_chimeric_len * len) / 100), mk);
    else
      rslt = k4d_lcm_batched<EL, false>(a, sc, len, rp.tot_mm, cl, cd, rp, inst, low, nxt, hits, n_lookup, n_probe, n_cand);
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
// LDS of a block (dynamic, sized by the batch): [shared by its K4_SLOW_WPB waves: entry table copy (starts, ends, ids) | coarse
// exception bitmap] then one private region per wave, k4_slow_wave_lds() bytes: packed probe (forward, reverse complement) | probe
// bytes (likewise) | the pair tables of k4d_lcm_batched | the pass-0 dedupe table or the chimeric phase's mismatch vectors
#define K4_SLOW_WPB 4
__host__ __device__ static inline size_t k4_slow_pk_words(int max_len) { return (size_t)(max_len / 32 + 2); }
__host__ __device__ static inline size_t k4_slow_probe_bytes(int max_len) { return (size_t)((max_len + 64 + 7) & ~7); }
__host__ __device__ static inline size_t k4_slow_shared_lds(uint32_t n_entries) {
  return (size_t)(n_entries <= K4_LDS_ENTRIES ? 2 * K4_LDS_ENTRIES * 8 + K4_LDS_ENTRIES * 4 : 0) + (size_t)K4_SUP_WORDS * 4;
}
__host__ __device__ static inline size_t k4_slow_wave_lds(int max_len, bool chim) {
  const int mk_len = max_len < K4_TRIM_MAX_LEN ? max_len : K4_TRIM_MAX_LEN;
  return 2 * k4_slow_pk_words(max_len) * 8 + 2 * k4_slow_probe_bytes(max_len) + (size_t)K4_GROUP * 8 + (size_t)(K4_GROUP + 2) * 8 + (size_t)K4_GROUP * 2 +
         (size_t)K4_GROUP * 4 * 8 + 4 * 8 + (size_t)K4_LDS_HASH * 4 +
         (chim ? (size_t)64 * 4 * ((mk_len + 31) / 32 + 1 < 4 ? 4 : (mk_len + 31) / 32 + 1) : 0);  // (k4d_lane_window128 writes four words whatever the length)
}
// BEST: the instantiation for LocateBestMatches (-N) only -- a different walk with its own registers, kept out of the other two
template <int EL, bool EXT, bool BEST = false>
__global__ void __launch_bounds__(64 * K4_SLOW_WPB) __attribute__((amdgpu_waves_per_eu(EXT ? K4_SLOW_WAVES_PER_EU_EXT : K4_SLOW_WAVES_PER_EU))) k4k_align_slow(K4AlignArgs a, uint32_t n_waves, int pass, uint64_t* hash_base,
                                                     uint32_t hash_cap, uint32_t* gen_base, int max_len, int chim) {
  extern __shared__ uint64_t slow_lds[];
  const bool ent_in_lds = a.ix.n_entries <= K4_LDS_ENTRIES;
  uint64_t* ent_s = slow_lds;
  uint32_t* entid_s = reinterpret_cast<uint32_t*>(slow_lds + (ent_in_lds ? 2 * K4_LDS_ENTRIES : 0));
  uint32_t* sup_s = entid_s + (ent_in_lds ? K4_LDS_ENTRIES : 0);
  const int wib = k4d_uni((int)(threadIdx.x >> 6));  // wave in block
  const int lane = threadIdx.x & 63;
  uint8_t* wbase = reinterpret_cast<uint8_t*>(slow_lds) + k4_slow_shared_lds(a.ix.n_entries) + (size_t)wib * k4_slow_wave_lds(max_len, chim != 0);
  uint64_t* pk_s = reinterpret_cast<uint64_t*>(wbase);
  uint8_t* probe_s = reinterpret_cast<uint8_t*>(pk_s + 2 * k4_slow_pk_words(max_len));
  uint64_t* glb_s = reinterpret_cast<uint64_t*>(probe_s + 2 * k4_slow_probe_bytes(max_len));
  uint64_t* gpre_s = glb_s + K4_GROUP;
  uint64_t* gcm_s = gpre_s + K4_GROUP + 2;
  uint64_t* glm_s = gcm_s + 4 * K4_GROUP;
  uint16_t* go_s = reinterpret_cast<uint16_t*>(glm_s + 4);
  // the pass-0 dedupe table (4-byte suffix elements), then -- chimeric phase only -- one mismatch bit vector per lane, word w of
  // lane l at mk_s[w * 64 + l]
  uint32_t* lhash_s = reinterpret_cast<uint32_t*>(go_s + K4_GROUP);
  uint32_t* mk_s = lhash_s + K4_LDS_HASH + lane;
  const uint32_t wave = blockIdx.x * K4_SLOW_WPB + (uint32_t)wib;
  uint32_t n_lookup = 0, n_probe = 0, n_cand = 0;
  if (ent_in_lds)
    for (int q = threadIdx.x; q < (int)a.ix.n_entries; q += 64 * K4_SLOW_WPB) {
      ent_s[q] = a.ix.ent_start[q];
      ent_s[K4_LDS_ENTRIES + q] = a.ix.ent_end[q];
      entid_s[q] = a.ix.ent_id[q];
    }
  for (int q = threadIdx.x; q < K4_SUP_WORDS; q += 64 * K4_SLOW_WPB) sup_s[q] = a.ix.excsup[q];
  __syncthreads();  // the only block-wide barrier: from here on the waves go their own ways
  if (wave < n_waves) {
    K4Slow sc;
    sc.sup = sup_s;
    sc.ent_id = ent_in_lds ? entid_s : a.ix.ent_id;
    sc.lhash = (EL == 4 && !BEST && pass == 0) ? lhash_s : nullptr;
    sc.lcap = K4_LDS_HASH;
    sc.lused = 0;
#ifdef K4_SLOW_PROF
    for (int q = 0; q < 24; q++) sc.prof[q] = 0;
#endif
    sc.probe = probe_s;
    sc.ent = ent_in_lds ? ent_s : nullptr;
    sc.pk = pk_s;
    sc.pstride = (uint32_t)k4_slow_probe_bytes(max_len);
    sc.pkstride = (uint32_t)k4_slow_pk_words(max_len);
    sc.g_lb = glb_s;
    sc.g_pre = gpre_s;
    sc.g_o = go_s;
    sc.g_cm = gcm_s;
    sc.g_lm = glm_s;
    sc.packed = false;
    sc.hash = hash_base + (size_t)wave * hash_cap;
    sc.cap = hash_cap;
    sc.gen = k4d_uni(gen_base[wave]);
    sc.lane = lane;
    sc.small = pass == 0;
    const uint32_t* list = pass == 0 ? a.slow_list : a.huge_list;
    const uint8_t* steps = pass == 0 ? a.slow_step : a.huge_step;
    uint32_t* cnt = a.ctl + (pass == 0 ? 0 : K4_CTL_HUGE);
    const uint32_t total = k4d_uni(cnt[0]);
    for (;;) {
      uint32_t q = 0;
      if (lane == 0) q = atomicAdd(&cnt[1], 1u);
      q = k4d_uni((uint32_t)__shfl(q, 0, 64));
      if (q >= total) break;
      const int64_t i = (int64_t)k4d_uni(list[q]);
      const int from_phase = k4d_uni((int)steps[q]);
      int phase = 0;
      K4_PROF_T(pr0);
      K4_PROF_ADD(from_phase < 3 ? 13 + from_phase : 15, 1);
      const uint32_t r0 = n_lookup, r1 = n_probe, r2 = n_cand;
      const int len = k4d_uni((int)a.lens[i]);
      const K4ReadParams rp = k4d_read_params(a, len);
      k4_hit* hits = a.hits + i * a.max_hits;
      int inst = 0, low = 0, nxt = 0, rslt = 0, allow = 0;
      if (len < 1 || len > K4_MAX_READ_LEN || len > max_len || rp.core_len < 1 || rp.max_hits < 1 || rp.max_hits > a.max_hits) {
        if (lane == 0) k4d_finalize(a, i, len, rp, a.mode == 0 ? K4_ERR_PARAMS : K4_HR_FATAL, 0, 0, 0);
        continue;
      }
      const uint8_t* src = a.reads + k4d_uni(a.offs[i]);
      K4_WSYNC();
      k4d_load_read_wave(sc, src, len);
      if (lane < 4) sc.g_lm[lane] = k4d_range_mask(0, len - 32 * lane);  // (k4d_lcm_batched's first K4_WSYNC comes before any use)
      K4_PROF_T(pr1);
      K4_PROF_ADD(5, pr1 - pr0);
      if (BEST) {  // -N (KAligner.cpp:9776-9796): LocateBestMatches instead of AlignReads (launch_general picks the instantiation)
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
        // A phase the fast kernel completed (and tallied) is not run again: the read is still unaligned, so that phase
        // returned eHRnone, i.e. it folded no candidate -- every candidate it accepts has fewer mismatches than the
        // LowMMCnt it starts from -- stored no hit and left (instances, LowMMCnt, NxtLowMMCnt) as LocateCoreMultiples
        // initialises them, which the next call does again (:5902-5907).
        // The phases of :7867-7891 from ONE call site (the function is large; every inlined copy weighs on the registers):
        // escalation with AllowMM = 0, 1, .. while the core stays longer than CoreLen, then the call with the caller's cores
        for (;;) {
          int p_allow, p_cl, p_cd;
          bool last = false;
          if (rp.tot_mm > 0 && allow <= rp.tot_mm && len / (allow + rp.mm_delta) > rp.core_len) {
            p_allow = allow; p_cl = len / (allow + rp.mm_delta); p_cd = p_cl;
            allow++;
          } else if (allow <= rp.tot_mm) {
            p_allow = rp.tot_mm; p_cl = rp.core_len; p_cd = rp.core_delta;
            last = true;
          } else
            break;
          if (phase++ >= from_phase) {
            rslt = k4d_lcm_batched<EL, false>(a, sc, len, p_allow, p_cl, p_cd, rp, &inst, &low, &nxt, hits, n_lookup, n_probe, n_cand);
            if (rslt != 0) break;
          }
          if (last) break;
        }
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
      for (int q = 0; q < 24; q++)
        if (sc.prof[q]) atomicAdd(&a.counters[6 + q], sc.prof[q]);
#endif
  }
  if (lane == 0) {  // the tallies are wave-uniform
    if (n_lookup) atomicAdd(&a.counters[1], (unsigned long long)n_lookup);
    if (n_probe) atomicAdd(&a.counters[2], (unsigned long long)n_probe);
    if (n_cand) atomicAdd(&a.counters[3], (unsigned long long)n_cand);
  }
}

// ==== host side: the two passes behind the step kernels of a batch ======================================================
template <int EL>
static int launch_general(k4_index* ix, K4AlignArgs& a, int max_len, hipStream_t st) {
  K4Workspace& w = ix->ws;
  uint64_t* small_base = w.slow_hash;
  uint64_t* big_base = w.slow_hash + (size_t)K4_SLOW_WAVES * K4_SMALL_HASH;
  uint32_t* gen_small = reinterpret_cast<uint32_t*>(big_base + (size_t)K4_HUGE_WAVES * w.slow_hash_cap);
  uint32_t* gen_big = gen_small + K4_SLOW_WAVES;
  const int slow_len = std::min(std::max(max_len, 1), K4_MAX_READ_LEN);
  const bool chim = a.ext_on && (a.mode == 0 ? a.ap.min_chimeric_len : a.kp.min_chimeric_len) > 0;
  const size_t slow_lds = k4_slow_shared_lds(a.ix.n_entries) + (size_t)K4_SLOW_WPB * k4_slow_wave_lds(slow_len, chim) + 16;
  if (slow_lds > 48 * 1024) {
    K4_HIP(ix, hipFuncSetAttribute((const void*)k4k_align_slow<EL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slow_lds));
    K4_HIP(ix, hipFuncSetAttribute((const void*)k4k_align_slow<EL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slow_lds));
    K4_HIP(ix, hipFuncSetAttribute((const void*)k4k_align_slow<EL, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slow_lds));
  }
  // (no more waves than reads: a batch of one -- the facade's AlignReads -- should not pay for 8192 idle waves)
  const uint32_t sw = (uint32_t)std::min<int64_t>(K4_SLOW_WAVES, std::max<int64_t>(a.n_reads, 1));
  const uint32_t hw = (uint32_t)std::min<int64_t>(K4_HUGE_WAVES, std::max<int64_t>(a.n_reads, 1));
  const dim3 sblk(64 * K4_SLOW_WPB), sgrid((sw + K4_SLOW_WPB - 1) / K4_SLOW_WPB), hgrid((hw + K4_SLOW_WPB - 1) / K4_SLOW_WPB);
  if ((a.mode == 1 && a.kp.pe_mode == 4) || a.best) {
    hipLaunchKernelGGL((k4k_align_slow<EL, false, true>), sgrid, sblk, slow_lds, st, a, sw, 0, small_base, (uint32_t)K4_SMALL_HASH, gen_small, slow_len, 0);
    hipLaunchKernelGGL((k4k_align_slow<EL, false, true>), hgrid, sblk, slow_lds, st, a, hw, 1, big_base, w.slow_hash_cap, gen_big, slow_len, 0);
  } else if (a.ext_on) {
    hipLaunchKernelGGL((k4k_align_slow<EL, true>), sgrid, sblk, slow_lds, st, a, sw, 0, small_base, (uint32_t)K4_SMALL_HASH, gen_small, slow_len, chim ? 1 : 0);
    hipLaunchKernelGGL((k4k_align_slow<EL, true>), hgrid, sblk, slow_lds, st, a, hw, 1, big_base, w.slow_hash_cap, gen_big, slow_len, chim ? 1 : 0);
  } else {
    hipLaunchKernelGGL((k4k_align_slow<EL, false>), sgrid, sblk, slow_lds, st, a, sw, 0, small_base, (uint32_t)K4_SMALL_HASH, gen_small, slow_len, 0);
    hipLaunchKernelGGL((k4k_align_slow<EL, false>), hgrid, sblk, slow_lds, st, a, hw, 1, big_base, w.slow_hash_cap, gen_big, slow_len, 0);
  }
  K4_HIP(ix, hipGetLastError());
  return K4_OK;
}

int k4i_launch_general(k4_index* ix, K4AlignArgs& a, int max_len, hipStream_t st) {
  return ix->d.el == 4 ? launch_general<4>(ix, a, max_len, st) : launch_general<5>(ix, a, max_len, st);
}
