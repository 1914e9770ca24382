// MLMode eMLuniq / eMLmulti (`-r3` / `-r4`): CKAligner::AssignMultiMatches (ngskit4b/KAligner.cpp:5092-5258) with the
// scoring of ProcAssignMultiMatches (:4944-5085), on the device, over the results k4_kalign_batch_dev leaves in HBM when
// it is run with pe_mode 1 (a read within the instance limit keeps its loci; uniquely aligned ones are accepted).
//
// The reference copies every locus of every such read into one array, qsorts it by locus (SortMultiHits, :11019), lets its
// threads score each multi-aligned locus against the loci that overlap it, re-sorts by read and score (SortMultiHitReadIDs,
// :11058) to pick each read's winner, sorts back by locus and walks the winners in order to drop the orphans.  Here:
//   entries   one per locus, keyed (chrom, start, len, mismatches, strand, read) -- two stable radix sorts
//   score     one thread per multi-aligned locus; a score depends only on static fields of its neighbours, so no order
//   winner    one thread per multi-aligned read over its <= max_ml loci (no second sort)
//   orphans   the reference's walk is sequential: an upstream neighbour counts with the state the walk left it in, a
//             downstream one with the state it had before the walk.  Relaxation rounds over the (few) loci that were won
//             by clustering with other multi-aligned reads: a locus is decided once every upstream neighbour in its
//             window is; the lowest undecided one always is, so the rounds terminate with the walk's result.
// The reference's shortcut of copying the previous locus' score when start, length, strand and chromosome repeat
// (:4967-4975, bounded by the blocks its threads happen to take) is not taken: both ways give the same score unless the
// score of uniquely aligned neighbours saturates (0x1fff upstream), which takes > 160-fold coverage.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <cstring>
#include <algorithm>
#include <rocprim/rocprim.hpp>
#include "k4_device.h"
#include "k4_internal.h"

namespace {

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  template <typename T> T* as() { return (T*)p; }
};

constexpr uint32_t kUniq = 0x8000u;  // cUniqueClustFlg and friends, KAligner.h:96-101
constexpr uint32_t kOverlap = 10, kUScore = 5, kMScore = 1, kScale = 10, kMinScore = 50;
// state of a locus in `ext` of its k4_hit (these modes are not combined with the optional AlignReads phases) while this runs (cleared before returning)
constexpr uint32_t kWon = 1u << 16, kWonAny = 1u << 17;

struct Ent {  // a locus in SortMultiHits order
  uint32_t chrom, loci, read;
  uint16_t len;
  uint8_t strand, mh;
};

__global__ void k4k_mm_count(int64_t n, const k4_read_result* __restrict__ rr, uint32_t* __restrict__ cnt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (int64_t)gridDim.x * blockDim.x)
    cnt[i] = (i < n && rr[i].hit_rslt == K4_HR_HITS) ? (uint32_t)rr[i].inst : 0u;  // AddMHitReads, :10002-10022
}

// entry k of read r, slot q: value = r*max_ml + q, minor key = len | mismatches | strand | read
__global__ void k4k_mm_fill(int64_t n, int32_t max_ml, const k4_read_result* __restrict__ rr, const k4_hit* __restrict__ hits,
                            const uint64_t* __restrict__ off, uint64_t* __restrict__ val, uint64_t* __restrict__ key) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (rr[i].hit_rslt != K4_HR_HITS) continue;
    const int inst = rr[i].inst;
    for (int q = 0; q < inst; q++) {
      const k4_hit h = hits[i * max_ml + q];
      const uint64_t k = off[i] + q;
      val[k] = (uint64_t)i * max_ml + q;
      key[k] = ((uint64_t)h.match_len << 48) | ((uint64_t)h.mismatches << 40) | ((uint64_t)h.strand << 32) | (uint32_t)i;
    }
  }
}

__global__ void k4k_mm_key_major(uint64_t m, const k4_hit* __restrict__ hits, const uint64_t* __restrict__ val,
                                 uint64_t* __restrict__ key) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
    const k4_hit h = hits[val[i]];
    key[i] = ((uint64_t)h.chrom_id << 32) | h.match_loci;
  }
}

__global__ void k4k_mm_gather(uint64_t m, int32_t max_ml, const k4_read_result* __restrict__ rr, const k4_hit* __restrict__ hits,
                              const uint64_t* __restrict__ val, Ent* __restrict__ ent) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t v = val[i];
    const k4_hit h = hits[v];
    const uint32_t r = (uint32_t)(v / (uint64_t)max_ml);
    Ent e;
    e.chrom = h.chrom_id; e.loci = h.match_loci; e.read = r; e.len = h.match_len; e.strand = h.strand;
    e.mh = rr[r].inst > 1;
    ent[i] = e;
  }
}

// ProcAssignMultiMatches, :4975-5081, for one multi-aligned locus
__global__ void k4k_mm_score(uint64_t m, int ml_mode, uint32_t max_reads_len, const Ent* __restrict__ ent,
                             const uint64_t* __restrict__ val, k4_hit* __restrict__ hits) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
    const Ent cur = ent[i];
    if (!cur.mh) continue;
    const uint32_t cs = cur.loci, clen = cur.len, cend = cs + clen - 1;
    uint32_t sc = 0;
    for (uint64_t j = i; j-- > 0;) {  // upstream
      const Ent c = ent[j];
      if (c.chrom != cur.chrom) break;
      if (cs - c.loci >= max_reads_len) break;
      const uint32_t ce = c.loci + c.len - 1;
      if (ce < cs + kOverlap) continue;
      const uint32_t ov = min(clen, ce - cs);
      if ((ml_mode == 3 && c.mh) || ((sc & kUniq) && (sc & ~kUniq) >= 0x1fffu)) continue;
      if (c.strand != cur.strand || c.read == cur.read) continue;
      if (!c.mh) {
        uint32_t s = 1 + (ov * kUScore) / kScale;
        if (sc & kUniq) s += sc & ~kUniq;
        s = min(s, 0x1fffu);
        sc = s | kUniq;
        if (s == 0x1fffu) break;
      } else if (!(sc & kUniq)) {
        sc = min(1 + (ov * kMScore) / kScale + sc, 0x1fffu);
      }
    }
    for (uint64_t j = i + 1; j < m; j++) {  // downstream
      const Ent c = ent[j];
      if (c.chrom != cur.chrom) break;
      if (c.loci > cend - kOverlap) break;
      const uint32_t ov = min((uint32_t)c.len, cend - c.loci);
      if ((ml_mode == 3 && c.mh) || ((sc & kUniq) && (sc & ~kUniq) >= 0x3fffu)) continue;
      if (c.strand != cur.strand || c.read == cur.read) continue;
      if (!c.mh) {
        uint32_t s = 1 + (ov * kUScore) / kScale;
        if (sc & kUniq) s += sc & ~kUniq;
        s = min(s, 0x3fffu);
        sc = s | kUniq;
        if (s == 0x3fffu) break;
      } else if (!(sc & kUniq)) {
        sc = min(1 + (ov * kMScore) / kScale + (sc & ~kUniq), 0x3fffu);
      }
    }
    hits[val[i]].ext = sc;
  }
}

// SortMultiHitReadIDs order among the loci of one read (:11058-11098): score descending, then chrom, len, mismatches,
// start, strand
K4_DEV bool k4d_mm_before(const k4_hit& a, const k4_hit& b) {
  if (a.ext != b.ext) return a.ext > b.ext;
  if (a.chrom_id != b.chrom_id) return a.chrom_id < b.chrom_id;
  if (a.match_len != b.match_len) return a.match_len < b.match_len;
  if (a.mismatches != b.mismatches) return a.mismatches < b.mismatches;
  if (a.match_loci != b.match_loci) return a.match_loci < b.match_loci;
  return a.strand < b.strand;
}

// the winner of each multi-aligned read (:5119-5163)
__global__ void k4k_mm_winner(int64_t n, int32_t max_ml, const k4_read_result* __restrict__ rr, k4_hit* __restrict__ hits) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (rr[i].hit_rslt != K4_HR_HITS || rr[i].inst < 2) continue;
    k4_hit* h = hits + i * max_ml;
    const int inst = rr[i].inst;
    int b0 = 0, b1 = -1;
    for (int q = 1; q < inst; q++) {
      if (k4d_mm_before(h[q], h[b0])) { b1 = b0; b0 = q; }
      else if (b1 < 0 || k4d_mm_before(h[q], h[b1])) b1 = q;
    }
    const uint32_t s0 = h[b0].ext, s1 = h[b1].ext;
    const uint32_t best = s0 & ~kUniq;
    if (best < kMinScore) continue;
    if ((s0 & kUniq) == (s1 & kUniq) && best < 2u * (s1 & ~kUniq)) continue;
    h[b0].ext = s0 | kWon | ((s0 & kUniq) ? 0u : kWonAny);
  }
}

// per sorted entry: 1 = in play for good (uniquely aligned read, or a locus won next to uniquely aligned reads),
// 2 = won next to other multi-aligned reads (undecided), 0 = not in play
__global__ void k4k_mm_state(uint64_t m, const Ent* __restrict__ ent, const uint64_t* __restrict__ val,
                             const k4_hit* __restrict__ hits, uint8_t* __restrict__ st0, uint8_t* __restrict__ st) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
    uint8_t s = 1;
    if (ent[i].mh) {
      const uint32_t r = hits[val[i]].ext;
      s = !(r & kWon) ? 0 : (r & kWonAny) ? 2 : 1;
    }
    st0[i] = s;
    st[i] = s;
  }
}

struct IsPending {
  const uint8_t* st;
  __device__ bool operator()(uint32_t i) const { return st[i] == 2; }
};

// one relaxation round of the orphan walk (:5176-5233) over the undecided loci
__global__ void k4k_mm_orphans(uint64_t n_pend, const uint32_t* __restrict__ pend, uint64_t m, const Ent* __restrict__ ent,
                               const uint8_t* __restrict__ st0, volatile uint8_t* st, uint32_t* __restrict__ left) {
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_pend; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = pend[t];
    if (st[i] != 2) continue;
    const Ent cur = ent[i];
    bool accept = false, wait = false;
    for (uint64_t j = i; j-- > 0;) {
      const Ent c = ent[j];
      if (cur.loci - c.loci > kOverlap + c.len) break;
      if (c.chrom != cur.chrom) break;
      const uint8_t s = st[j];
      if (s == 1) { accept = true; break; }
      if (s == 2) wait = true;  // its fate is not known yet; anything in play further up still decides
    }
    if (!accept && wait) {
      atomicAdd(left, 1u);
      continue;
    }
    if (!accept)
      for (uint64_t j = i + 1; j < m; j++) {
        const Ent c = ent[j];
        if (c.loci - cur.loci > kOverlap + cur.len) break;
        if (c.chrom != cur.chrom) break;
        if (st0[j] != 0) { accept = true; break; }  // downstream loci have not been walked yet
      }
    st[i] = accept ? 1 : 3;  // 3: dropped
  }
}

// the winners that stayed: the read is accepted with that locus (:5236-5246); every locus gets its reserved word back
__global__ void k4k_mm_assign(uint64_t m, int32_t max_ml, const Ent* __restrict__ ent, const uint64_t* __restrict__ val,
                              const uint8_t* __restrict__ st, k4_read_result* __restrict__ rr, k4_hit* __restrict__ hits,
                              unsigned long long* __restrict__ n_assigned) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x) {
    if (!ent[i].mh) continue;
    const uint64_t v = val[i];
    const bool won = (hits[v].ext & kWon) && st[i] == 1;
    if (!won) continue;
    const uint32_t r = ent[i].read;
    k4_hit h = hits[v];
    h.ext = (uint32_t)(v - (uint64_t)r * max_ml) | 0x80000000u;  // parked: slot 0 may still be read by its own thread
    hits[v] = h;
    rr[r].nar = K4_NAR_ACCEPTED;
    rr[r].num_hits = 1;
    rr[r].inst = 1;
    atomicAdd(n_assigned, 1ull);
  }
}

__global__ void k4k_mm_finish(int64_t n, int32_t max_ml, const uint32_t* __restrict__ cnt, k4_hit* __restrict__ hits) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int inst = (int)cnt[i];
    if (inst < 2) continue;
    k4_hit* h = hits + i * max_ml;
    int win = -1;
    for (int q = 0; q < inst; q++) {
      if (h[q].ext & 0x80000000u) win = q;
      h[q].ext = 0;
    }
    if (win > 0) h[0] = h[win];
  }
}

unsigned grid_for(uint64_t n) { return (unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 16); }

}  // namespace

extern "C" int k4_assign_multi_dev(k4_index* ix, int ml_mode, int32_t max_reads_len, int64_t n_reads, int32_t max_ml,
                                   void* d_rr, void* d_hits, int64_t* n_assigned, void* stream) {
  if (!ix || (ml_mode != 3 && ml_mode != 4) || n_reads < 0 || max_ml < 2 || max_reads_len < 1 ||
      (n_reads && (!d_rr || !d_hits)))
    return K4_ERR_PARAMS;
  if (n_assigned) *n_assigned = 0;
  if (!n_reads) return K4_OK;
  K4_HIP(ix, hipSetDevice(ix->device));
  hipStream_t st = (hipStream_t)stream;
  k4_read_result* rr = (k4_read_result*)d_rr;
  k4_hit* hits = (k4_hit*)d_hits;
  Buf cnt, off, tmp;
  K4_HIP(ix, cnt.alloc((size_t)(n_reads + 1) * 4));
  K4_HIP(ix, off.alloc((size_t)(n_reads + 1) * 8));
  hipLaunchKernelGGL(k4k_mm_count, dim3(grid_for(n_reads + 1)), dim3(256), 0, st, n_reads, rr, cnt.as<uint32_t>());
  {
    size_t tb = 0;
    K4_HIP(ix, rocprim::exclusive_scan(nullptr, tb, cnt.as<uint32_t>(), off.as<uint64_t>(), (uint64_t)0, (size_t)(n_reads + 1),
                                       rocprim::plus<uint64_t>(), st));
    K4_HIP(ix, tmp.alloc(tb));
    K4_HIP(ix, rocprim::exclusive_scan(tmp.p, tb, cnt.as<uint32_t>(), off.as<uint64_t>(), (uint64_t)0, (size_t)(n_reads + 1),
                                       rocprim::plus<uint64_t>(), st));
  }
  uint64_t m = 0;
  K4_HIP(ix, hipMemcpyAsync(&m, off.as<uint64_t>() + n_reads, 8, hipMemcpyDeviceToHost, st));
  K4_HIP(ix, hipStreamSynchronize(st));
  if (m == 0) return K4_OK;
  if (m > 0xFFFFFFFFull || n_reads > 0xFFFFFFFFll) return k4_fail(ix, K4_ERR_UNSUPPORTED, "more than 2^32 loci to cluster");
  // SortMultiHits order: stable sort on (len, mismatches, strand, read), then on (chrom, start)
  Buf va, vb, ka, kb, ent, st0, stc, pend, npend, left, nas;
  K4_HIP(ix, va.alloc(m * 8));
  K4_HIP(ix, vb.alloc(m * 8));
  K4_HIP(ix, ka.alloc(m * 8));
  K4_HIP(ix, kb.alloc(m * 8));
  hipLaunchKernelGGL(k4k_mm_fill, dim3(grid_for(n_reads)), dim3(256), 0, st, n_reads, max_ml, rr, hits, off.as<uint64_t>(),
                     va.as<uint64_t>(), ka.as<uint64_t>());
  rocprim::double_buffer<uint64_t> keys(ka.as<uint64_t>(), kb.as<uint64_t>());
  rocprim::double_buffer<uint64_t> vals(va.as<uint64_t>(), vb.as<uint64_t>());
  {
    size_t tb = 0;
    K4_HIP(ix, rocprim::radix_sort_pairs(nullptr, tb, keys, vals, (size_t)m, 0u, 64u, st));
    Buf t2;
    K4_HIP(ix, t2.alloc(tb));
    K4_HIP(ix, rocprim::radix_sort_pairs(t2.p, tb, keys, vals, (size_t)m, 0u, 64u, st));
    hipLaunchKernelGGL(k4k_mm_key_major, dim3(grid_for(m)), dim3(256), 0, st, m, hits, vals.current(), keys.current());
    K4_HIP(ix, rocprim::radix_sort_pairs(t2.p, tb, keys, vals, (size_t)m, 0u, 64u, st));
    K4_HIP(ix, hipStreamSynchronize(st));
  }
  const uint64_t* val = vals.current();
  K4_HIP(ix, ent.alloc(m * sizeof(Ent)));
  K4_HIP(ix, st0.alloc(m));
  K4_HIP(ix, stc.alloc(m));
  hipLaunchKernelGGL(k4k_mm_gather, dim3(grid_for(m)), dim3(256), 0, st, m, max_ml, rr, hits, val, ent.as<Ent>());
  hipLaunchKernelGGL(k4k_mm_score, dim3(grid_for(m)), dim3(256), 0, st, m, ml_mode, (uint32_t)max_reads_len, ent.as<Ent>(), val,
                     hits);
  hipLaunchKernelGGL(k4k_mm_winner, dim3(grid_for(n_reads)), dim3(256), 0, st, n_reads, max_ml, rr, hits);
  hipLaunchKernelGGL(k4k_mm_state, dim3(grid_for(m)), dim3(256), 0, st, m, ent.as<Ent>(), val, hits, st0.as<uint8_t>(),
                     stc.as<uint8_t>());
  K4_HIP(ix, hipGetLastError());
  // the undecided loci
  K4_HIP(ix, pend.alloc(m * 4));
  K4_HIP(ix, npend.alloc(8));
  K4_HIP(ix, left.alloc(4));
  K4_HIP(ix, nas.alloc(8));
  K4_HIP(ix, hipMemsetAsync(nas.p, 0, 8, st));
  uint64_t n_pend = 0;
  {
    rocprim::counting_iterator<uint32_t> all(0);
    IsPending pred{st0.as<uint8_t>()};
    size_t tb = 0;
    K4_HIP(ix, rocprim::select(nullptr, tb, all, pend.as<uint32_t>(), npend.as<uint64_t>(), (size_t)m, pred, st));
    Buf t3;
    K4_HIP(ix, t3.alloc(tb));
    K4_HIP(ix, rocprim::select(t3.p, tb, all, pend.as<uint32_t>(), npend.as<uint64_t>(), (size_t)m, pred, st));
    K4_HIP(ix, hipMemcpyAsync(&n_pend, npend.p, 8, hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipStreamSynchronize(st));
  }
  uint32_t n_left = n_pend ? 1 : 0;
  for (uint64_t round = 0; n_left && round <= n_pend; round++) {  // every round decides at least the lowest undecided locus
    K4_HIP(ix, hipMemsetAsync(left.p, 0, 4, st));
    hipLaunchKernelGGL(k4k_mm_orphans, dim3(grid_for(n_pend)), dim3(256), 0, st, n_pend, pend.as<uint32_t>(), m, ent.as<Ent>(),
                       st0.as<uint8_t>(), stc.as<uint8_t>(), left.as<uint32_t>());
    K4_HIP(ix, hipMemcpyAsync(&n_left, left.p, 4, hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipStreamSynchronize(st));
  }
  if (n_left) return k4_fail(ix, K4_ERR_INTERNAL, "orphan walk did not settle");
  hipLaunchKernelGGL(k4k_mm_assign, dim3(grid_for(m)), dim3(256), 0, st, m, max_ml, ent.as<Ent>(), val, stc.as<uint8_t>(), rr, hits,
                     nas.as<unsigned long long>());
  hipLaunchKernelGGL(k4k_mm_finish, dim3(grid_for(n_reads)), dim3(256), 0, st, n_reads, max_ml, cnt.as<uint32_t>(), hits);
  K4_HIP(ix, hipGetLastError());
  unsigned long long na = 0;
  K4_HIP(ix, hipMemcpyAsync(&na, nas.p, 8, hipMemcpyDeviceToHost, st));
  K4_HIP(ix, hipStreamSynchronize(st));
  if (n_assigned) *n_assigned = (int64_t)na;
  return K4_OK;
}
