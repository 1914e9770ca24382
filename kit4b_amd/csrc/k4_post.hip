// kit4b_amd/csrc/k4_post.hip -- the post-alignment stages that `kalign -x / -A / -a` run on the SE (or PE) results, on the
// device, over the records k4_kalign_*_batch_dev left in HBM:
//   k4_auto_trim_flanks_dev      <- CKAligner::AutoTrimFlanks            ngskit4b/KAligner.cpp:1714-1917
//   k4_remove_orphan_juncts_dev  <- CKAligner::RemoveOrphanSpliceJuncts  ngskit4b/KAligner.cpp:2406-2498
//                                   CKAligner::RemoveOrphanMicroInDels   ngskit4b/KAligner.cpp:2501-2594
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <rocprim/rocprim.hpp>
#include "k4_device.h"
#include "k4_internal.h"
#include "k4_pool.h"

namespace {

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) hipFree(p); }
  hipError_t alloc(size_t bytes) { return k4_malloc_retry(&p, bytes ? bytes : 1); }
  template <typename T> T* as() { return (T*)p; }
};

// read symbol j of the alignment in READ orientation: the target is reverse-complemented for a '-' hit (:1802-1805)
K4_DEV uint32_t k4d_targ_in_read_sense(const K4DevIndex& ix, uint64_t base, uint32_t len, bool minus, uint32_t j) {
  if (!minus) return k4d_ref_base(ix, base + j);
  const uint32_t t = k4d_ref_base(ix, base + (len - 1 - j));
  return t <= 3 ? 3 - t : t;  // CSeqTrans::ReverseComplement leaves N (and anything else) as it is
}

// AutoTrimFlanks, one thread per read: walk in from the 5' end until min_flank_exacts consecutive bases match, the same
// from the 3' end; what is outside becomes TrimLeft / TrimRight.  SE: a read that cannot keep (len+1)/2 (>= 15) bases
// between two such flanks is eliminated (eNARTrim).  One-segment, non-chimeric hits only (:1748).
__global__ void __launch_bounds__(256) k4k_auto_trim(K4DevIndex ix, int mfe, int pe, int64_t n, int max_ml, k4_read_result* __restrict__ rr,
                                                     k4_pe_read* __restrict__ pr, k4_hit* __restrict__ hits,
                                                     const uint8_t* __restrict__ reads, const uint64_t* __restrict__ offs,
                                                     const uint32_t* __restrict__ lens, unsigned long long* __restrict__ n_elim) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  k4_hit* hp = pr ? &pr[i].hit : &hits[i * max_ml];
  const int nar = pr ? pr[i].nar : rr[i].nar;
  k4_hit h = *hp;
  if (nar != K4_NAR_ACCEPTED || (h.ext & (K4_EXT_INDEL | K4_EXT_SPLICE | K4_EXT_CHIMERIC))) return;
  const uint32_t match_len = h.match_len, read_len = lens[i];
  if (match_len != read_len) {  // :1751-1759 (cannot happen for a one-segment hit; NumHits 0, NAR stays)
    if (pr) pr[i].num_hits = 0; else rr[i].num_hits = 0;
    atomicAdd(n_elim, 1ull);
    return;
  }
  int min_trimmed = (int)(match_len + 1) / 2;
  if (min_trimmed < 15) min_trimmed = 15;
  const uint8_t* rd = reads + offs[i];
  const uint64_t base = ix.ent_start[h.chrom_id - 1] + h.match_loci;
  const bool minus = h.strand == '-';
  int exact = 0;
  uint32_t idx;
  const int pemin5 = !pe ? (int)match_len : (int)match_len / 3;
  for (idx = 0; idx <= (match_len - (uint32_t)min_trimmed) && idx < (uint32_t)pemin5; idx++) {  // :1817-1828
    if ((uint32_t)(rd[idx] & 7) != k4d_targ_in_read_sense(ix, base, match_len, minus, idx)) { exact = 0; continue; }
    if (++exact == mfe) break;
  }
  if (!pe && ((idx + (uint32_t)min_trimmed) > match_len || exact < mfe)) {  // :1830-1843
    rr[i].num_hits = 0; rr[i].nar = K4_NAR_TRIM;
    atomicAdd(n_elim, 1ull);
    return;
  }
  const int left_ofs = (int)idx - (mfe - 1);
  exact = 0;
  const int pemin3 = !pe ? 0 : (int)(match_len * 2) / 3;
  for (idx = match_len - 1; idx >= (uint32_t)(left_ofs + min_trimmed) && idx > (uint32_t)pemin3; idx--) {  // :1855-1866
    if ((uint32_t)(rd[idx] & 7) != k4d_targ_in_read_sense(ix, base, match_len, minus, idx)) { exact = 0; continue; }
    if (++exact == mfe) break;
  }
  if (!pe && (exact != mfe || idx < (uint32_t)(left_ofs + min_trimmed))) {  // :1868-1881
    rr[i].num_hits = 0; rr[i].nar = K4_NAR_TRIM;
    atomicAdd(n_elim, 1ull);
    return;
  }
  const int right_ofs = (int)idx + mfe;
  const uint32_t tl = (uint32_t)left_ofs, tr = match_len - (uint32_t)right_ofs;
  hp->ext = (h.ext & ~0xFFFFFFu) | (tl & 0xFFFu) | ((tr & 0xFFFu) << 12);
}

// ---- orphan junctions ---------------------------------------------------------------------------------------------------
struct IsJunct {
  const k4_read_result* rr;
  const k4_hit* hits;
  int max_ml;
  uint32_t which;
  __device__ bool operator()(uint32_t i) const { return rr[i].nar == K4_NAR_ACCEPTED && (hits[(int64_t)i * max_ml].ext & which) != 0; }
};

// Starts = AdjEndLoci(Seg[0]), Ends = AdjStartLoci(Seg[1]) (:2446-2447; a two-segment hit carries no trimming)
__global__ void __launch_bounds__(256) k4k_junct_keys(uint64_t m, const uint32_t* __restrict__ idx, const k4_hit* __restrict__ hits, int max_ml,
                                                      const k4_seg2* __restrict__ seg2, uint64_t* __restrict__ major, uint32_t* __restrict__ minor) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const uint32_t i = idx[j];
  const k4_hit h = hits[(int64_t)i * max_ml];
  const uint32_t starts = h.match_loci + ((uint32_t)h.match_len - 1u);
  major[j] = ((uint64_t)h.chrom_id << 32) | starts;
  minor[j] = seg2[i].match_loci;
}
// neighbours in (chrom, start, end) order that agree within 3 bp at both ends support each other (:2456-2465; 32-bit
// unsigned arithmetic as there)
__global__ void __launch_bounds__(256) k4k_junct_mark(uint64_t m, const uint32_t* __restrict__ order, const uint64_t* __restrict__ major,
                                                      const uint32_t* __restrict__ minor, k4_hit* __restrict__ hits, int max_ml) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j + 1 >= m) return;
  const uint32_t ca = (uint32_t)(major[j] >> 32), cb = (uint32_t)(major[j + 1] >> 32);
  const uint32_t sa = (uint32_t)major[j], sb = (uint32_t)major[j + 1], ea = minor[j], eb = minor[j + 1];
  if (ca == cb && sa <= (uint32_t)(sb + 3u) && sa >= (uint32_t)(sb - 3u) && ea <= (uint32_t)(eb + 3u) && ea >= (uint32_t)(eb - 3u)) {
    atomicOr(&hits[(int64_t)order[j] * max_ml].ext, K4_EXT_NONORPHAN);
    atomicOr(&hits[(int64_t)order[j + 1] * max_ml].ext, K4_EXT_NONORPHAN);
  }
}
__global__ void __launch_bounds__(256) k4k_junct_drop(uint64_t m, const uint32_t* __restrict__ order, const k4_hit* __restrict__ hits, int max_ml,
                                                      k4_read_result* __restrict__ rr, int nar, unsigned long long* __restrict__ n_removed) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const uint32_t i = order[j];
  if (!(hits[(int64_t)i * max_ml].ext & K4_EXT_NONORPHAN)) {
    rr[i].nar = nar; rr[i].num_hits = 0; rr[i].inst = 0;
    atomicAdd(n_removed, 1ull);
  }
}

}  // namespace

extern "C" int k4_auto_trim_flanks_dev(k4_index* ix, int32_t min_flank_exacts, int pe, int64_t n_reads, int32_t max_ml, void* d_rr,
                                       void* d_hits, const void* d_reads, const void* d_offs, const void* d_lens,
                                       int64_t* n_eliminated, void* stream) {
  if (!ix) return K4_ERR_PARAMS;
  if (n_eliminated) *n_eliminated = 0;
  if (min_flank_exacts <= 0 || n_reads <= 0) return K4_OK;
  if (!d_rr || (!pe && (!d_hits || max_ml < 1)) || !d_reads || !d_offs || !d_lens) return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  K4_HIP(ix, hipSetDevice(ix->device));
  hipStream_t st = (hipStream_t)stream;
  Buf cnt;
  K4_HIP(ix, cnt.alloc(8));
  K4_HIP(ix, hipMemsetAsync(cnt.p, 0, 8, st));
  // PE: d_rr holds k4_pe_read records (hit inside); SE: k4_read_result + the hit slots
  hipLaunchKernelGGL(k4k_auto_trim, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, ix->d, (int)min_flank_exacts, pe ? 1 : 0,
                     n_reads, (int)max_ml, pe ? (k4_read_result*)nullptr : (k4_read_result*)d_rr, pe ? (k4_pe_read*)d_rr : (k4_pe_read*)nullptr,
                     (k4_hit*)d_hits, (const uint8_t*)d_reads, (const uint64_t*)d_offs, (const uint32_t*)d_lens,
                     cnt.as<unsigned long long>());
  K4_HIP(ix, hipGetLastError());
  unsigned long long c = 0;
  K4_HIP(ix, hipMemcpyAsync(&c, cnt.p, 8, hipMemcpyDeviceToHost, st));
  K4_HIP(ix, hipStreamSynchronize(st));
  if (n_eliminated) *n_eliminated = (int64_t)c;
  return K4_OK;
}

extern "C" int k4_remove_orphan_juncts_dev(k4_index* ix, uint32_t which, int64_t n_reads, int32_t max_ml, void* d_rr, void* d_hits,
                                           const void* d_seg2, int64_t* n_removed, void* stream) {
  if (!ix) return K4_ERR_PARAMS;
  if (n_removed) *n_removed = 0;
  if (which != K4_EXT_SPLICE && which != K4_EXT_INDEL) return k4_fail(ix, K4_ERR_PARAMS, "which must be K4_EXT_SPLICE or K4_EXT_INDEL");
  if (n_reads <= 0) return K4_OK;
  if (!d_rr || !d_hits || !d_seg2 || max_ml < 1) return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  if (n_reads >= 0xFFFFFF00ll) return k4_fail(ix, K4_ERR_PARAMS, "at most 2^32-256 reads per call");
  K4_HIP(ix, hipSetDevice(ix->device));
  hipStream_t st = (hipStream_t)stream;
  Buf idx0, idx1, cnt, tmp, ka, kb, ma, mb2;
  K4_HIP(ix, idx0.alloc((size_t)n_reads * 4));
  K4_HIP(ix, cnt.alloc(16));
  K4_HIP(ix, hipMemsetAsync(cnt.p, 0, 16, st));
  {
    rocprim::counting_iterator<uint32_t> all(0);
    IsJunct pred{(const k4_read_result*)d_rr, (const k4_hit*)d_hits, (int)max_ml, which};
    size_t tb = 0;
    K4_HIP(ix, rocprim::select(nullptr, tb, all, idx0.as<uint32_t>(), cnt.as<uint64_t>(), (size_t)n_reads, pred, st));
    K4_HIP(ix, tmp.alloc(tb));
    K4_HIP(ix, rocprim::select(tmp.p, tb, all, idx0.as<uint32_t>(), cnt.as<uint64_t>(), (size_t)n_reads, pred, st));
  }
  uint64_t m = 0;
  K4_HIP(ix, hipMemcpyAsync(&m, cnt.p, 8, hipMemcpyDeviceToHost, st));
  K4_HIP(ix, hipStreamSynchronize(st));
  if (m == 0) return K4_OK;
  K4_HIP(ix, idx1.alloc(m * 4));
  K4_HIP(ix, ka.alloc(m * 8));
  K4_HIP(ix, kb.alloc(m * 8));
  K4_HIP(ix, ma.alloc(m * 4));
  K4_HIP(ix, mb2.alloc(m * 4));
  const unsigned nb = (unsigned)((m + 255) / 256);
  const uint32_t* order = idx0.as<uint32_t>();
  if (m > 1) {
    // SortSegJuncts (KAligner.cpp:11104): chrom, start, end -- two stable radix sorts, the minor key first; the keys travel
    // with the values so that they are in sorted order for the neighbour test
    hipLaunchKernelGGL(k4k_junct_keys, dim3(nb), dim3(256), 0, st, m, idx0.as<uint32_t>(), (const k4_hit*)d_hits, (int)max_ml,
                       (const k4_seg2*)d_seg2, ka.as<uint64_t>(), ma.as<uint32_t>());
    rocprim::double_buffer<uint32_t> mk(ma.as<uint32_t>(), mb2.as<uint32_t>());
    rocprim::double_buffer<uint32_t> vb(idx0.as<uint32_t>(), idx1.as<uint32_t>());
    size_t tb = 0;
    K4_HIP(ix, rocprim::radix_sort_pairs(nullptr, tb, mk, vb, (size_t)m, 0u, 32u, st));
    Buf t2;
    K4_HIP(ix, t2.alloc(tb));
    K4_HIP(ix, rocprim::radix_sort_pairs(t2.p, tb, mk, vb, (size_t)m, 0u, 32u, st));
    // keys of the values in their new order
    hipLaunchKernelGGL(k4k_junct_keys, dim3(nb), dim3(256), 0, st, m, vb.current(), (const k4_hit*)d_hits, (int)max_ml,
                       (const k4_seg2*)d_seg2, ka.as<uint64_t>(), mk.alternate());
    rocprim::double_buffer<uint64_t> kk(ka.as<uint64_t>(), kb.as<uint64_t>());
    size_t tb2 = 0;
    K4_HIP(ix, rocprim::radix_sort_pairs(nullptr, tb2, kk, vb, (size_t)m, 0u, 64u, st));
    Buf t3;
    K4_HIP(ix, t3.alloc(tb2));
    K4_HIP(ix, rocprim::radix_sort_pairs(t3.p, tb2, kk, vb, (size_t)m, 0u, 64u, st));
    order = vb.current();
    uint32_t* minor_sorted = mk.alternate();
    hipLaunchKernelGGL(k4k_junct_keys, dim3(nb), dim3(256), 0, st, m, order, (const k4_hit*)d_hits, (int)max_ml, (const k4_seg2*)d_seg2,
                       kk.alternate(), minor_sorted);
    hipLaunchKernelGGL(k4k_junct_mark, dim3(nb), dim3(256), 0, st, m, order, (const uint64_t*)kk.alternate(), (const uint32_t*)minor_sorted,
                       (k4_hit*)d_hits, (int)max_ml);
    hipLaunchKernelGGL(k4k_junct_drop, dim3(nb), dim3(256), 0, st, m, order, (const k4_hit*)d_hits, (int)max_ml, (k4_read_result*)d_rr,
                       which == K4_EXT_SPLICE ? K4_NAR_SPLICEJCTN : K4_NAR_MICROINDEL, cnt.as<unsigned long long>() + 1);
    K4_HIP(ix, hipGetLastError());
    unsigned long long c = 0;
    K4_HIP(ix, hipMemcpyAsync(&c, cnt.as<unsigned long long>() + 1, 8, hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipStreamSynchronize(st));
    if (n_removed) *n_removed = (int64_t)c;
    return K4_OK;
  }
  // a lone junction is an orphan (:2482-2489)
  hipLaunchKernelGGL(k4k_junct_drop, dim3(nb), dim3(256), 0, st, m, order, (const k4_hit*)d_hits, (int)max_ml, (k4_read_result*)d_rr,
                     which == K4_EXT_SPLICE ? K4_NAR_SPLICEJCTN : K4_NAR_MICROINDEL, cnt.as<unsigned long long>() + 1);
  K4_HIP(ix, hipGetLastError());
  unsigned long long c = 0;
  K4_HIP(ix, hipMemcpyAsync(&c, cnt.as<unsigned long long>() + 1, 8, hipMemcpyDeviceToHost, st));
  K4_HIP(ix, hipStreamSynchronize(st));
  if (n_removed) *n_removed = (int64_t)c;
  return K4_OK;
}
