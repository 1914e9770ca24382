// kit4b_amd/csrc/k4_device.h -- device-side accessors of the HBM index (gfx950 only).
#pragma once
#include "k4_internal.h"

#define K4_DEV __device__ __forceinline__

// gfx950 global loads of 2/3/4 dwords only need dword alignment: these types make hipcc emit one wide load where the
// address is merely 4-byte aligned (a divergent wave pays per lane-request in the vector L1, not per byte).
typedef uint32_t k4_u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t k4_u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t k4_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint64_t k4_u64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));

// n consecutive dwords from a 4-byte aligned address, in as few load instructions as possible
template <int N>
K4_DEV void k4d_load_words(const uint32_t* __restrict__ p, uint32_t (&out)[N]) {
  constexpr int Q = N / 4, R = N % 4;
#pragma unroll
  for (int q = 0; q < Q; q++) {
    const k4_u32x4_a4 v = *reinterpret_cast<const k4_u32x4_a4*>(p + 4 * q);
    out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
  }
  if (R == 1) out[4 * Q] = p[4 * Q];
  if (R == 2) {
    const k4_u32x2_a4 v = *reinterpret_cast<const k4_u32x2_a4*>(p + 4 * Q);
    out[4 * Q] = v.x; out[4 * Q + 1] = v.y;
  }
  if (R == 3) {
    const k4_u32x3_a4 v = *reinterpret_cast<const k4_u32x3_a4*>(p + 4 * Q);
    out[4 * Q] = v.x; out[4 * Q + 1] = v.y; out[4 * Q + 2] = v.z;
  }
}

// SfxOfsToLoci (libkit4b/SfxArray.cpp:49-60): element i of the 4- or 5-byte little-endian suffix array.
template <int EL>
K4_DEV uint64_t k4d_sa_at(const K4DevIndex& ix, uint64_t i) {
  if (EL == 4) {
    return reinterpret_cast<const uint32_t*>(ix.sa)[i];
  } else {
    uint64_t q = i * 5;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(ix.sa) + (q >> 2);
    uint32_t sh = (uint32_t)(q & 3) * 8;
    uint64_t v = ((uint64_t)w[1] << 32) | w[0];  // the allocation is padded so w[1] is always readable
    return (v >> sh) & 0xFFFFFFFFFFull;
  }
}

// k-mer table entry c = {lb, pos0, sig}: lb = number of suffixes sorting before k-mer c (its bucket is
// [lb(c), lb(c+1))), pos0 = SA[lb] (offset of the bucket's first suffix, valid when the bucket is not empty) and, in the
// 32-bit form, sig = the 16 bases that follow the k-mer in that first suffix (MSB-first), a filter that settles most
// single-suffix buckets without touching the suffix array or the reference.
// 64-bit form (blocks of 2^32 symbols or more: k stays at 16 while the block holds 4^17 suffixes and more, so a bucket
// averages several suffixes and a 12-base signature of its first one says little): two 64-bit words, lb and pos0 in their
// low 40 bits, and in the 2 x 24 bits above them SIXTEEN 3-bit counts -- how many of the bucket's suffixes continue with each
// of the 16 two-base extensions of the k-mer.  A core of k + 2 bases or more goes straight to its sub-bucket
// [lb + sum of the counts below it, + its own count), i.e. the table answers like a k = 18 table of 4^18 entries would.  All
// ones = the bucket is irregular (a count of 7 or more, or a suffix that meets N / a separator within k + 2 symbols): the
// whole bucket is searched as before.
#define K4_SIG_BASES32 16
#define K4_KTAB_STRIDE32 3
#define K4_KTAB_STRIDE64 2
#define K4_KTAB64_MASK 0xFFFFFFFFFFull
#define K4_KTAB64_IRREGULAR 0xFFFFFFFFFFFFull
K4_DEV uint64_t k4d_ktab_lb(const K4DevIndex& ix, uint64_t c) {
  return ix.ktab64 ? reinterpret_cast<const uint64_t*>(ix.ktab)[K4_KTAB_STRIDE64 * c] & K4_KTAB64_MASK
                   : reinterpret_cast<const uint32_t*>(ix.ktab)[K4_KTAB_STRIDE32 * c];
}
// bucket of the k-mer prefix range [c0, c1): lb0 = lb(c0), pos0 = pos0(c0), lb1 = lb(c1); sig = sig(c0) (32-bit form) /
// sub = the 48 bits of sub-bucket counts of c0 (64-bit form).  KT = field type.
template <typename KT>
K4_DEV void k4d_ktab_fetch(const K4DevIndex& ix, uint64_t c0, uint64_t c1, KT& lb0, KT& pos0, uint32_t& sig, KT& lb1, uint64_t& sub, bool lazy_lb1 = false) {
  sub = K4_KTAB64_IRREGULAR;
  if (sizeof(KT) == 4) {
    const uint32_t* t = reinterpret_cast<const uint32_t*>(ix.ktab);
    const uint32_t* e = t + K4_KTAB_STRIDE32 * c0;
    if (c1 == c0 + 1) {  // the common case (core at least k long): four adjacent fields, one 16-byte fetch
      const k4_u32x4_a4 v = *reinterpret_cast<const k4_u32x4_a4*>(e);
      lb0 = (KT)v.x; pos0 = (KT)v.y; sig = v.z; lb1 = (KT)v.w;
    } else {
      const k4_u32x3_a4 v = *reinterpret_cast<const k4_u32x3_a4*>(e);
      lb0 = (KT)v.x; pos0 = (KT)v.y; sig = v.z; lb1 = (KT)t[K4_KTAB_STRIDE32 * c1];
    }
  } else {
    const uint64_t* t = reinterpret_cast<const uint64_t*>(ix.ktab);
    const k4_u64x2_a8 v = *reinterpret_cast<const k4_u64x2_a8*>(t + K4_KTAB_STRIDE64 * c0);
    lb0 = (KT)(v.x & K4_KTAB64_MASK); pos0 = (KT)(v.y & K4_KTAB64_MASK); sig = 0;
    sub = (v.x >> 40) | ((v.y >> 40) << 24);
    // lazy_lb1: the caller goes on to the sub-bucket of a regular bucket and has no use for the bucket's end -- one load and,
    // for one entry in eight, one more 128-byte line less per lookup; an irregular bucket fetches it behind the entry
    if (!lazy_lb1 || sub == K4_KTAB64_IRREGULAR) lb1 = (KT)(t[K4_KTAB_STRIDE64 * c1] & K4_KTAB64_MASK);
    else lb1 = lb0;
  }
}
// the sub-bucket of two-base extension e (0..15) inside a regular bucket: (suffixes in front of it, its own count)
K4_DEV void k4d_ktab_sub(uint64_t sub, uint32_t e, uint32_t& before, uint32_t& count) {
  count = (uint32_t)(sub >> (3 * e)) & 7u;
  uint64_t below = sub & ((1ull << (3 * e)) - 1ull);  // the fields of the extensions that sort in front
  // sum of 3-bit fields: fold to 6-bit, then 12-bit lanes, then multiply-add
  below = (below & 0x1C71C71C71C7ull) + ((below >> 3) & 0x1C71C71C71C7ull);   // 8 sums of two fields, 6 bits apart
  below = (below & 0x03F03F03F03Full) + ((below >> 6) & 0x03F03F03F03Full);   // 4 sums of four fields, 12 bits apart
  before = (uint32_t)((below & 0xFFF) + ((below >> 12) & 0xFFF) + ((below >> 24) & 0xFFF) + ((below >> 36) & 0xFFF));
}

// 32 bases [pos, pos+32) as one MSB-first 64-bit chunk.  pos may be as low as -K4_PAD_BASES.
K4_DEV uint64_t k4d_ref_chunk(const K4DevIndex& ix, int64_t pos) {
  int64_t w = pos >> 4;  // arithmetic shift: floor for negative pos
  uint32_t s = (uint32_t)(pos & 15) * 2;
  const uint32_t* p = ix.ref2 + w;
  uint64_t hi = ((uint64_t)p[0] << 32) | p[1];
  return s ? (hi << s) | (p[2] >> (32 - s)) : hi;
}

// the nine packed words that hold the bases [pos + 32 c0, + 128); eight: the caller needs no base behind the eighth word
K4_DEV void k4d_ref_words9(const K4DevIndex& ix, int64_t pos, int c0, bool eight, uint32_t (&wv)[9]) {
  const uint32_t* wp = ix.ref2 + ((pos >> 4) + 2 * c0);
  if (eight) {
    uint32_t w8[8];
    k4d_load_words<8>(wp, w8);
#pragma unroll
    for (int j = 0; j < 9; j++) wv[j] = j < 8 ? w8[j] : 0u;
  } else
    k4d_load_words<9>(wp, wv);
}
// ... as four MSB-first chunks
K4_DEV void k4d_words_to_chunks4(const uint32_t (&wv)[9], int64_t pos, uint64_t (&out)[4]) {
  const uint32_t sh = (uint32_t)(pos & 15) * 2;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const uint64_t hi = ((uint64_t)wv[2 * c] << 32) | wv[2 * c + 1];
    out[c] = sh ? (hi << sh) | (wv[2 * c + 2] >> (32 - sh)) : hi;
  }
}
K4_DEV void k4d_ref_chunks4(const K4DevIndex& ix, int64_t pos, int c0, bool eight, uint64_t (&out)[4]) {
  uint32_t wv[9];
  k4d_ref_words9(ix, pos, c0, eight, wv);
  k4d_words_to_chunks4(wv, pos, out);
}

// any non-ACGT symbol in [start, end) ?  (end - start) must stay below 32 blocks; one 8-byte fetch
K4_DEV bool k4d_any_exc(const K4DevIndex& ix, int64_t start, int64_t end) {
  if (start < 0) start = 0;
  if (end <= start) return false;
  const uint64_t b0 = (uint64_t)start >> K4_EXC_SHIFT, b1 = (uint64_t)(end - 1) >> K4_EXC_SHIFT;
  const k4_u32x2_a4 w = *reinterpret_cast<const k4_u32x2_a4*>(ix.excbm + (b0 >> 5));
  const uint64_t v = (((uint64_t)w.y << 32) | w.x) >> (b0 & 31);  // bit 0 = block b0 (bitmap is padded)
  const uint32_t nb = (uint32_t)(b1 - b0) + 1;                      // 1..32 blocks
  return (v & ((1ull << nb) - 1ull)) != 0;
}

// exact symbol (etSeqBase low nibble: 0..4, 7) at pos
K4_DEV uint32_t k4d_ref_base(const K4DevIndex& ix, uint64_t pos) {
  uint64_t blk = pos >> K4_EXC_SHIFT;
  if ((ix.excbm[blk >> 5] >> (blk & 31)) & 1) {
    uint32_t lo = 0, hi = ix.n_exc;  // lower_bound over the sorted flagged-block list
    while (lo < hi) {
      uint32_t mid = (lo + hi) >> 1;
      if (ix.excblk[mid] < (uint32_t)blk) lo = mid + 1; else hi = mid;
    }
    uint32_t j = (uint32_t)(pos & (K4_EXC_BLOCK - 1));
    return (ix.excnib[(uint64_t)lo * (K4_EXC_BLOCK / 8) + (j >> 3)] >> (4 * (j & 7))) & 0xF;
  }
  uint32_t w = ix.ref2[pos >> 4];
  return (w >> (30 - 2 * (uint32_t)(pos & 15))) & 3;
}

// Exact target symbols for a lane that walks along the reference: one packed word per 16 bases; the exception bitmap is
// consulted once per 256-base block, and inside a flagged block the block's place in the nibble store is looked up ONCE
// (k4d_ref_base searches the block list for every symbol: eleven dependent loads at 2000 flagged blocks) and its nibbles come
// eight per load.  Beyond the block: a separator.
struct K4Tb {
  const K4DevIndex* ix;
  int64_t cw, cblk;   // the cached unit (64 bases of packed words / one nibble word of the cached block), the cached block
  uint32_t w4[4];     // four packed words = 64 bases, one 16-byte load (a lane that walks 100 bases waits for two loads, not seven)
  uint32_t exr;       // rank of the cached block among the flagged ones
  bool flagged;
  K4_DEV void init(const K4DevIndex& x) { ix = &x; cw = -1; cblk = -1; w4[0] = w4[1] = w4[2] = w4[3] = 0; exr = 0; flagged = false; }
  K4_DEV uint32_t get(int64_t pos) {
    if (pos < 0 || (uint64_t)pos >= ix->n) return 7u;
    const int64_t blk = pos >> K4_EXC_SHIFT;
    if (blk != cblk) {
      cblk = blk;
      cw = -1;
      flagged = (ix->excbm[blk >> 5] >> (blk & 31)) & 1;
      if (flagged) {
        uint32_t lo = 0, hi = ix->n_exc;  // lower_bound over the sorted flagged-block list
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (ix->excblk[mid] < (uint32_t)blk) lo = mid + 1; else hi = mid;
        }
        exr = lo;
      }
    }
    if (flagged) {
      const int64_t w = pos >> 3;
      if (w != cw) { cw = w; w4[0] = ix->excnib[(uint64_t)exr * (K4_EXC_BLOCK / 8) + (uint32_t)((pos & (K4_EXC_BLOCK - 1)) >> 3)]; }
      return (w4[0] >> (4 * (uint32_t)(pos & 7))) & 0xF;
    }
    const int64_t w = pos >> 6;
    if (w != cw) {
      cw = w;
      const k4_u32x4_a4 v = *reinterpret_cast<const k4_u32x4_a4*>(ix->ref2 + 4 * w);  // (the pads cover the last unit's tail)
      w4[0] = v.x; w4[1] = v.y; w4[2] = v.z; w4[3] = v.w;
    }
    const uint32_t q = (uint32_t)(pos >> 4) & 3u;
    const uint32_t word = q == 0 ? w4[0] : q == 1 ? w4[1] : q == 2 ? w4[2] : w4[3];
    return (word >> (30 - 2 * (uint32_t)(pos & 15))) & 3;
  }
};

// MapChunkHit2Entry (libkit4b/SfxArray.cpp:2609-2654): index of the entry holding concat offset ofs, -1 on a separator
K4_DEV int k4d_map_entry(const K4DevIndex& ix, uint64_t ofs) {
  int lo = 0, hi = (int)ix.n_entries - 1;
  while (hi >= lo) {
    int mid = (hi + lo) >> 1;
    uint64_t s = ix.ent_start[mid];
    if (s > ofs) { hi = mid - 1; continue; }
    if (ix.ent_end[mid] >= ofs) return mid;
    lo = mid + 1;
  }
  return -1;
}

// mask of the bases [lo, hi) of a 32-base MSB-first chunk (0 <= lo, hi <= 32)
K4_DEV uint64_t k4d_range_mask(int lo, int hi) {
  if (lo < 0) lo = 0;
  if (hi > 32) hi = 32;
  if (hi <= lo) return 0;
  uint64_t a = ~0ull >> (2 * lo);                       // lo < 32 here
  uint64_t b = hi >= 32 ? 0ull : (~0ull >> (2 * hi));
  return a & ~b;
}

K4_DEV uint32_t k4d_mm_count(uint64_t x) {  // mismatching bases in a XOR of two chunks
  uint64_t y = (x | (x >> 1)) & 0x5555555555555555ull;
  return (uint32_t)__popcll(y);
}
