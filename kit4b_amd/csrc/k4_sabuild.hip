// kit4b_amd/csrc/k4_sabuild.hip -- suffix-array construction in HBM (`ngskit4b index`).
//
// Produces the order of CSfxArray::QSortSeq (libkit4b/SfxArray.cpp:9739-9834): suffixes compared bytewise on the low
// nibble, A<C<G<T<N<EOS(7), comparison stops after the first EOS; suffixes that are identical through their EOS are
// tied (the reference's parallel quicksort leaves them in arbitrary order) -- ties come out in offset order here.
//
// Method (MI355X-first: the whole problem lives in HBM, 288 GB make the 64-bit-key sort of every suffix affordable):
//   1. key = the first 21 symbols of each suffix at 3 bits each (everything after an EOS zeroed); one LSD radix sort of
//      (key, offset) pairs.  On i.i.d. sequence 4^21 buckets leave ~n^2/2^43 tied pairs.
//   2. refinement rounds on the (small) set of still-tied suffixes: key of the next 21 symbols, stable sort by
//      (tie group, key), scatter back, repeat until no tie group without an EOS remains.
// The radix sort, the compaction and the scan are rocPRIM device primitives (library calls for setup work, like
// hipBLASLt for a plain GEMM); key extraction and tie detection are the kernels below.
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <string>
#include "k4_internal.h"

#define K4_SYMS 21
#define K4_LOW3 0x1249249249249249ull  // bit 0 of every 3-bit group (21 groups)

__device__ __forceinline__ uint64_t k4d_eos_stop(uint64_t k) {  // zero everything after the first (most significant) EOS
  uint64_t t = k & (k >> 1) & (k >> 2) & K4_LOW3;
  if (t) {
    int top = 63 - __clzll(t);  // bit index 3g of the most significant group holding 7
    k &= ~((1ull << top) - 1ull);
  }
  return k;
}
__device__ __forceinline__ bool k4d_key_has_eos(uint64_t k) { return (k & (k >> 1) & (k >> 2) & K4_LOW3) != 0; }

__device__ __forceinline__ uint64_t k4d_key_at(const uint8_t* __restrict__ seq, uint64_t n, uint64_t pos) {
  uint64_t k = 0;
  for (int j = 0; j < K4_SYMS; j++) {
    uint64_t q = pos + j;
    uint64_t s = q < n ? (seq[q] & 7) : 7;
    k = (k << 3) | s;
  }
  return k4d_eos_stop(k);
}

// first-level keys: each thread slides over 16 consecutive suffixes
template <typename VT>
__global__ void __launch_bounds__(256) k4k_sa_keys0(const uint8_t* __restrict__ seq, uint64_t n, uint64_t* __restrict__ keys,
                                                    VT* __restrict__ vals) {
  uint64_t base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (base >= n) return;
  uint64_t k = 0;
  for (int j = 0; j < K4_SYMS - 1; j++) {
    uint64_t q = base + j;
    k = (k << 3) | (q < n ? (uint64_t)(seq[q] & 7) : 7ull);
  }
  for (int t = 0; t < 16; t++) {
    uint64_t i = base + t;
    if (i >= n) break;
    uint64_t q = i + K4_SYMS - 1;
    k = ((k << 3) | (q < n ? (uint64_t)(seq[q] & 7) : 7ull)) & 0x7FFFFFFFFFFFFFFFull;
    keys[i] = k4d_eos_stop(k);
    vals[i] = (VT)i;
  }
}

// tie flags over the fully sorted array: rank r is unresolved when it shares its key with a neighbour and the key
// holds no EOS.  head[r] marks the first rank of such a run.
__global__ void __launch_bounds__(256) k4k_sa_ties0(const uint64_t* __restrict__ keys, uint64_t n, uint8_t* __restrict__ flag) {
  uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  uint64_t k = keys[r];
  bool tie = !k4d_key_has_eos(k) && ((r > 0 && keys[r - 1] == k) || (r + 1 < n && keys[r + 1] == k));
  flag[r] = tie ? 1 : 0;
}

// list element j: rank Rk[j]; is it the head of its run (different key from the previous list element or not adjacent)?
template <typename VT>
__global__ void __launch_bounds__(256) k4k_sa_heads0(const uint64_t* __restrict__ keys, const VT* __restrict__ rk,
                                                     uint64_t m, VT* __restrict__ ghead) {
  uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  VT r = rk[j];
  bool head = (r == 0) || keys[r - 1] != keys[r];
  ghead[j] = head ? r : (VT)0;  // max-scan propagates the head rank (rank 0 can only be the very first head)
}

template <typename VT, typename IT>
__global__ void __launch_bounds__(256) k4k_sa_gather(const VT* __restrict__ sa32, const IT* __restrict__ rk,
                                                     uint64_t m, VT* __restrict__ sfx) {
  uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < m) sfx[j] = sa32[rk[j]];
}

__global__ void __launch_bounds__(256) k4k_sa_iota(uint32_t* __restrict__ p, uint64_t m) {
  uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < m) p[j] = (uint32_t)j;
}

template <typename VT>
__global__ void __launch_bounds__(256) k4k_sa_keysN(const uint8_t* __restrict__ seq, uint64_t n, const VT* __restrict__ sfx,
                                                    uint64_t m, uint64_t depth, uint64_t* __restrict__ keys) {
  uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < m) keys[j] = k4d_key_at(seq, n, (uint64_t)sfx[j] + depth);
}

template <typename VT>
__global__ void __launch_bounds__(256) k4k_sa_scatter(VT* __restrict__ sa32, const VT* __restrict__ rk,
                                                      const VT* __restrict__ sfx, uint64_t m) {
  uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < m) sa32[rk[j]] = sfx[j];
}

// after sorting the list by (group, key): still tied when a neighbour has the same group and key and no EOS
template <typename VT>
__global__ void __launch_bounds__(256) k4k_sa_tiesN(const uint64_t* __restrict__ keys, const VT* __restrict__ grp,
                                                    const VT* __restrict__ rk, uint64_t m, uint8_t* __restrict__ flag,
                                                    VT* __restrict__ newhead) {
  uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  uint64_t k = keys[j];
  VT g = grp[j];
  bool prev = j > 0 && grp[j - 1] == g && keys[j - 1] == k;
  bool next = j + 1 < m && grp[j + 1] == g && keys[j + 1] == k;
  bool tie = !k4d_key_has_eos(k) && (prev || next);
  flag[j] = tie ? 1 : 0;
  newhead[j] = (tie && !prev) ? rk[j] : (VT)0;
}

template <typename VT>
__global__ void __launch_bounds__(256) k4k_sa_to_el5(const VT* __restrict__ sa32, uint64_t n, uint8_t* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t v = sa32[i];
  uint8_t* p = out + i * 5;
  p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); p[4] = (uint8_t)(v >> 32);
}

namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  void release() { if (p) hipFree(p); p = nullptr; }
  template <typename T> T* as() { return (T*)p; }
};
#define SA_HIP(call)                                                                              \
  do {                                                                                            \
    hipError_t _e = (call);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      if (err) *err = std::string("HIP error in " #call ": ") + hipGetErrorString(_e);           \
      return _e == hipErrorOutOfMemory ? K4_ERR_MEM : K4_ERR_NO_DEVICE;                           \
    }                                                                                             \
  } while (0)
inline unsigned nblk(uint64_t n) { return (unsigned)((n + 255) / 256); }
struct MaxOp {
  template <typename T>
  __device__ __host__ T operator()(T a, T b) const { return a > b ? a : b; }
};
}  // namespace

// Level-0 result (keys sorted, sa32 = suffix offsets in that order, cnt items) -> fully ordered sa32: refinement rounds
// on the still-tied suffixes.  n is the length of the whole sequence block (for key extraction).
template <typename VT>
static int refine_sorted(const uint8_t* d_seq, uint64_t n, uint64_t cnt_items, uint64_t* keys, uint64_t* keys_alt, VT* sa32,
                         hipStream_t st, std::string* err) {
  constexpr unsigned VBITS = sizeof(VT) == 4 ? 32u : 40u;  // rank / offset bits that matter
  // ---- ties after level 0 -----------------------------------------------------------------------------------
  DevBuf flag, cnt;
  SA_HIP(flag.alloc(cnt_items));
  SA_HIP(cnt.alloc(8));
  hipLaunchKernelGGL(k4k_sa_ties0, dim3(nblk(cnt_items)), dim3(256), 0, st, keys, cnt_items, flag.as<uint8_t>());
  // compact the tied ranks; the alternate key buffer is free now and large enough for the rank list
  VT* rk_all = reinterpret_cast<VT*>(keys_alt);
  size_t sbytes = 0;
  rocprim::counting_iterator<VT> ranks(0);
  SA_HIP(rocprim::select(nullptr, sbytes, ranks, flag.as<uint8_t>(), rk_all, cnt.as<uint64_t>(), (size_t)cnt_items, st));
  DevBuf stmp;
  SA_HIP(stmp.alloc(sbytes));
  SA_HIP(rocprim::select(stmp.p, sbytes, ranks, flag.as<uint8_t>(), rk_all, cnt.as<uint64_t>(), (size_t)cnt_items, st));
  uint64_t m = 0;
  SA_HIP(hipMemcpy(&m, cnt.p, 8, hipMemcpyDeviceToHost));

  if (m >= 0xFFFFFF00ull) {  // (one thread per item below: a launch holds fewer than 2^32 threads)
    if (err) *err = "more than 2^32 tied suffixes after the first sort level";
    return K4_ERR_UNSUPPORTED;
  }
  if (m > 0) {
    // Refinement working set, one row per still-tied suffix, ordered by rank (tie groups are contiguous):
    //   rk   rank in the suffix array (fixed: a group keeps its rank range)     sf  suffix offset
    //   gh   rank of the group's first member (ascending along the list)
    DevBuf rk, rk2, sf, sf2, gh, gh2, key, key2, pa, pb, ga, gb, fl2, nh;
    SA_HIP(rk.alloc(m * sizeof(VT))); SA_HIP(rk2.alloc(m * sizeof(VT)));
    SA_HIP(sf.alloc(m * sizeof(VT))); SA_HIP(sf2.alloc(m * sizeof(VT)));
    SA_HIP(gh.alloc(m * sizeof(VT))); SA_HIP(gh2.alloc(m * sizeof(VT)));
    SA_HIP(key.alloc(m * 8)); SA_HIP(key2.alloc(m * 8));
    SA_HIP(pa.alloc(m * 4)); SA_HIP(pb.alloc(m * 4));
    SA_HIP(ga.alloc(m * sizeof(VT))); SA_HIP(gb.alloc(m * sizeof(VT)));
    SA_HIP(fl2.alloc(m)); SA_HIP(nh.alloc(m * sizeof(VT)));
    SA_HIP(hipMemcpyAsync(rk.p, rk_all, m * sizeof(VT), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL((k4k_sa_heads0<VT>), dim3(nblk(m)), dim3(256), 0, st, keys, rk.as<VT>(), m, gh2.as<VT>());
    size_t b_scan = 0, b_s1 = 0, b_s2 = 0, b_sel = 0;
    {
      rocprim::double_buffer<uint64_t> a(key.as<uint64_t>(), key2.as<uint64_t>());
      rocprim::double_buffer<uint32_t> v(pa.as<uint32_t>(), pb.as<uint32_t>());
      rocprim::double_buffer<VT> g(ga.as<VT>(), gb.as<VT>());
      SA_HIP(rocprim::inclusive_scan(nullptr, b_scan, gh2.as<VT>(), gh.as<VT>(), (size_t)m, MaxOp(), st));
      SA_HIP(rocprim::radix_sort_pairs(nullptr, b_s1, a, v, (size_t)m, 0u, 63u, st));
      SA_HIP(rocprim::radix_sort_pairs(nullptr, b_s2, g, v, (size_t)m, 0u, VBITS, st));
      SA_HIP(rocprim::select(nullptr, b_sel, rk.as<VT>(), fl2.as<uint8_t>(), rk2.as<VT>(), cnt.as<uint64_t>(), (size_t)m, st));
    }
    const size_t rtb = std::max(std::max(b_s1, b_s2), std::max(b_sel, b_scan));
    DevBuf rtmp;
    SA_HIP(rtmp.alloc(rtb));
    size_t tb = rtb;
    SA_HIP(rocprim::inclusive_scan(rtmp.p, tb, gh2.as<VT>(), gh.as<VT>(), (size_t)m, MaxOp(), st));
    hipLaunchKernelGGL((k4k_sa_gather<VT, VT>), dim3(nblk(m)), dim3(256), 0, st, sa32, rk.as<VT>(), m, sf.as<VT>());
    uint64_t depth = K4_SYMS;
    const uint64_t max_depth = 100010 + K4_SYMS;  // the reference compares at most gMaxBaseCmpLen+10 bases (SfxArray.cpp:9793)
    while (m > 0 && depth < max_depth) {
      // order every group by its next 21 symbols: permutation = stable sort by key, then stable sort by group
      hipLaunchKernelGGL((k4k_sa_keysN<VT>), dim3(nblk(m)), dim3(256), 0, st, d_seq, n, sf.as<VT>(), m, depth, key.as<uint64_t>());
      hipLaunchKernelGGL(k4k_sa_iota, dim3(nblk(m)), dim3(256), 0, st, pa.as<uint32_t>(), m);
      rocprim::double_buffer<uint64_t> a(key.as<uint64_t>(), key2.as<uint64_t>());
      rocprim::double_buffer<uint32_t> v(pa.as<uint32_t>(), pb.as<uint32_t>());
      tb = rtb;
      SA_HIP(rocprim::radix_sort_pairs(rtmp.p, tb, a, v, (size_t)m, 0u, 63u, st));
      hipLaunchKernelGGL((k4k_sa_gather<VT, uint32_t>), dim3(nblk(m)), dim3(256), 0, st, gh.as<VT>(), v.current(), m, ga.as<VT>());
      rocprim::double_buffer<VT> g(ga.as<VT>(), gb.as<VT>());
      tb = rtb;
      SA_HIP(rocprim::radix_sort_pairs(rtmp.p, tb, g, v, (size_t)m, 0u, VBITS, st));
      // apply: suffixes follow the permutation; group heads are unchanged (sorted list of the same values);
      // keys are recomputed for the permuted suffixes
      hipLaunchKernelGGL((k4k_sa_gather<VT, uint32_t>), dim3(nblk(m)), dim3(256), 0, st, sf.as<VT>(), v.current(), m, sf2.as<VT>());
      hipLaunchKernelGGL((k4k_sa_keysN<VT>), dim3(nblk(m)), dim3(256), 0, st, d_seq, n, sf2.as<VT>(), m, depth, key.as<uint64_t>());
      hipLaunchKernelGGL((k4k_sa_scatter<VT>), dim3(nblk(m)), dim3(256), 0, st, sa32, rk.as<VT>(), sf2.as<VT>(), m);
      hipLaunchKernelGGL((k4k_sa_tiesN<VT>), dim3(nblk(m)), dim3(256), 0, st, key.as<uint64_t>(), gh.as<VT>(), rk.as<VT>(), m,
                         fl2.as<uint8_t>(), nh.as<VT>());
      tb = rtb;
      SA_HIP(rocprim::select(rtmp.p, tb, rk.as<VT>(), fl2.as<uint8_t>(), rk2.as<VT>(), cnt.as<uint64_t>(), (size_t)m, st));
      tb = rtb;
      SA_HIP(rocprim::select(rtmp.p, tb, sf2.as<VT>(), fl2.as<uint8_t>(), sf.as<VT>(), cnt.as<uint64_t>(), (size_t)m, st));
      tb = rtb;
      SA_HIP(rocprim::select(rtmp.p, tb, nh.as<VT>(), fl2.as<uint8_t>(), gh2.as<VT>(), cnt.as<uint64_t>(), (size_t)m, st));
      uint64_t m2 = 0;
      SA_HIP(hipMemcpy(&m2, cnt.p, 8, hipMemcpyDeviceToHost));
      if (m2) {
        tb = rtb;
        SA_HIP(rocprim::inclusive_scan(rtmp.p, tb, gh2.as<VT>(), gh.as<VT>(), (size_t)m2, MaxOp(), st));
        SA_HIP(hipMemcpyAsync(rk.p, rk2.p, m2 * sizeof(VT), hipMemcpyDeviceToDevice, st));
      }
      m = m2;
      depth += K4_SYMS;
    }
  }
  return K4_OK;
}

template <typename VT>
static int build_sa_t(uint64_t n, uint32_t el, const uint8_t* d_seq, uint8_t* d_sa, int device, std::string* err) {
  SA_HIP(hipSetDevice(device));
  hipStream_t st = 0;
  DevBuf k0, k1, v0, v1, tmp;
  SA_HIP(k0.alloc(n * 8));
  SA_HIP(k1.alloc(n * 8));
  SA_HIP(v0.alloc(n * sizeof(VT)));
  if (el == 5 || sizeof(VT) != 4) SA_HIP(v1.alloc(n * sizeof(VT)));
  VT* vb = (el == 4 && sizeof(VT) == 4) ? (VT*)d_sa : v1.as<VT>();  // with 4-byte elements the output doubles as a buffer
  hipLaunchKernelGGL((k4k_sa_keys0<VT>), dim3(nblk((n + 15) / 16)), dim3(256), 0, st, d_seq, n, k0.as<uint64_t>(), v0.as<VT>());
  rocprim::double_buffer<uint64_t> kb(k0.as<uint64_t>(), k1.as<uint64_t>());
  rocprim::double_buffer<VT> vbuf(v0.as<VT>(), vb);
  size_t tbytes = 0;
  SA_HIP(rocprim::radix_sort_pairs(nullptr, tbytes, kb, vbuf, (size_t)n, 0u, 63u, st));
  SA_HIP(tmp.alloc(tbytes));
  SA_HIP(rocprim::radix_sort_pairs(tmp.p, tbytes, kb, vbuf, (size_t)n, 0u, 63u, st));
  SA_HIP(hipStreamSynchronize(st));
  uint64_t* keys = kb.current();
  VT* sa32 = vbuf.current();
  uint64_t* keys_alt = kb.alternate();

  {
    int rc = refine_sorted<VT>(d_seq, n, n, keys, keys_alt, sa32, st, err);
    if (rc != K4_OK) return rc;
  }
  // ---- emit -----------------------------------------------------------------------------------------------------------
  if (el == 4) {
    if ((void*)sa32 != (void*)d_sa) SA_HIP(hipMemcpy(d_sa, sa32, n * 4, hipMemcpyDeviceToDevice));
  } else {
    hipLaunchKernelGGL((k4k_sa_to_el5<VT>), dim3(nblk(n)), dim3(256), 0, st, sa32, n, d_sa);
  }
  SA_HIP(hipGetLastError());
  SA_HIP(hipDeviceSynchronize());
  return K4_OK;
}

// >= 2^32 symbols: one sort per leading symbol (A, C, G, T, N, then everything else incl. EOS), each below 2^32 items and
// ~8 bytes of scratch per symbol of the whole block, appended in symbol order.  Same order as one global sort.
__global__ void __launch_bounds__(256) k4k_sa_lead_flag(const uint8_t* __restrict__ seq, uint64_t n, uint32_t sym, uint8_t* __restrict__ flag) {
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = seq[i] & 7;
  flag[i] = sym < 5 ? (s == sym) : (s > 4);
}

// suffix counts per leading-symbol class (0..4, 5 = everything else)
__global__ void __launch_bounds__(256) k4k_sa_lead_hist(const uint8_t* __restrict__ seq, uint64_t n, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int h[6];
  if (threadIdx.x < 6) h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const uint32_t s = seq[i] & 7;
    atomicAdd(&h[s < 5 ? s : 5], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 6 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

static int build_sa_bucketed(uint64_t n, const uint8_t* d_seq, uint8_t* d_sa, int device, std::string* err) {
  typedef uint64_t VT;
  SA_HIP(hipSetDevice(device));
  hipStream_t st = 0;
  DevBuf flag, cnt, hist;
  const uint64_t PIECE = 1ull << 30;  // no rocPRIM call over the whole block: pieces of 2^30 positions
  SA_HIP(flag.alloc(PIECE));
  SA_HIP(cnt.alloc(8));
  SA_HIP(hist.alloc(6 * 8));
  SA_HIP(hipMemsetAsync(hist.p, 0, 6 * 8, st));
  hipLaunchKernelGGL(k4k_sa_lead_hist, dim3(8192), dim3(256), 0, st, d_seq, n, hist.as<unsigned long long>());
  uint64_t m_of[6];
  SA_HIP(hipMemcpy(m_of, hist.p, 6 * 8, hipMemcpyDeviceToHost));
  size_t selb = 0;
  SA_HIP(rocprim::select(nullptr, selb, rocprim::counting_iterator<VT>(0), flag.as<uint8_t>(), (VT*)nullptr, cnt.as<uint64_t>(), (size_t)PIECE, st));
  DevBuf selt;
  SA_HIP(selt.alloc(selb));
  uint64_t done = 0;
  for (uint32_t sym = 0; sym < 6; sym++) {
    const uint64_t m = m_of[sym];
    if (m == 0) continue;
    if (m >= 0xFFFFFF00ull) {  // (one thread per item below: a launch holds fewer than 2^32 threads)
      if (err) *err = "a leading-symbol bucket holds 2^32 or more suffixes";
      return K4_ERR_UNSUPPORTED;
    }
    DevBuf k0, k1, v0, v1, tmp;
    SA_HIP(k0.alloc(m * 8)); SA_HIP(k1.alloc(m * 8));
    SA_HIP(v0.alloc(m * 8)); SA_HIP(v1.alloc(m * 8));
    uint64_t got = 0;
    for (uint64_t base = 0; base < n; base += PIECE) {
      const uint64_t len = n - base < PIECE ? n - base : PIECE;
      hipLaunchKernelGGL(k4k_sa_lead_flag, dim3(nblk(len)), dim3(256), 0, st, d_seq + base, len, sym, flag.as<uint8_t>());
      size_t rb = selb;
      SA_HIP(rocprim::select(selt.p, rb, rocprim::counting_iterator<VT>(base), flag.as<uint8_t>(), v0.as<VT>() + got, cnt.as<uint64_t>(), (size_t)len, st));
      uint64_t c = 0;
      SA_HIP(hipMemcpyAsync(&c, cnt.p, 8, hipMemcpyDeviceToHost, st));
      SA_HIP(hipStreamSynchronize(st));
      got += c;
      if (got > m) break;
    }
    if (got != m) {
      if (err) *err = "bucketed suffix sort: candidate count mismatch";
      return K4_ERR_INTERNAL;
    }
    hipLaunchKernelGGL((k4k_sa_keysN<VT>), dim3(nblk(m)), dim3(256), 0, st, d_seq, n, v0.as<VT>(), m, (uint64_t)0, k0.as<uint64_t>());
    rocprim::double_buffer<uint64_t> kb(k0.as<uint64_t>(), k1.as<uint64_t>());
    rocprim::double_buffer<VT> vbuf(v0.as<VT>(), v1.as<VT>());
    size_t tbytes = 0;
    SA_HIP(rocprim::radix_sort_pairs(nullptr, tbytes, kb, vbuf, (size_t)m, 0u, 63u, st));
    SA_HIP(tmp.alloc(tbytes));
    SA_HIP(rocprim::radix_sort_pairs(tmp.p, tbytes, kb, vbuf, (size_t)m, 0u, 63u, st));
    SA_HIP(hipStreamSynchronize(st));
    tmp.release();
    int rc = refine_sorted<VT>(d_seq, n, m, kb.current(), kb.alternate(), vbuf.current(), st, err);
    if (rc != K4_OK) return rc;
    hipLaunchKernelGGL((k4k_sa_to_el5<VT>), dim3(nblk(m)), dim3(256), 0, st, vbuf.current(), m, d_sa + done * 5);
    SA_HIP(hipStreamSynchronize(st));
    done += m;
  }
  SA_HIP(hipGetLastError());
  SA_HIP(hipDeviceSynchronize());
  if (done != n) {
    if (err) *err = "bucketed suffix sort lost suffixes";
    return K4_ERR_INTERNAL;
  }
  return K4_OK;
}

int k4i_build_sa(uint64_t n, uint32_t el, const uint8_t* d_seq, uint8_t* d_sa, int device, std::string* err) {
  if (n == 0 || (el != 4 && el != 5) || (el == 4 && n > 0xFFFFFFFFull)) return K4_ERR_PARAMS;
  if (n >= (1ull << 40)) return K4_ERR_PARAMS;
  // below 2^32 symbols: one sort of every suffix with 32-bit offsets; above: one sort per leading symbol with 64-bit
  // offsets (each bucket stays below 2^32 items: rocPRIM's radix sort is not safe beyond that, and scratch stays
  // ~8 bytes per symbol of the block, so a 15 Gbp block needs ~125 GB of HBM next to the sequence and the output)
  if (n < 0xFFFFFF00ull) return build_sa_t<uint32_t>(n, el, d_seq, d_sa, device, err);
  return build_sa_bucketed(n, d_seq, d_sa, device, err);
}

extern "C" int k4_build_sa_device(uint64_t concat_len, uint32_t el, const void* d_seq, void* d_sa, int device) {
  std::string err;
  if (!d_seq || !d_sa) return K4_ERR_PARAMS;
  int rc = k4i_build_sa(concat_len, el, (const uint8_t*)d_seq, (uint8_t*)d_sa, device, &err);
  if (rc != K4_OK) k4_set_global_error("%s", err.c_str());
  return rc;
}
