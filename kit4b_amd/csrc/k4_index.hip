// kit4b_amd/csrc/k4_index.hip -- index life cycle: .sfx container I/O, upload, 2-bit packing, exception data and
// the direct-address k-mer table.  Replaces CSfxArray::Open/SetTargBlock/Close and the accessors CKAligner uses
// (libkit4b/SfxArray.h:524-1023; container layout SfxArray.h:95-123,191-223).
#include <fcntl.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <mutex>
#include <atomic>
#include <thread>
#include <vector>
#include "k4_device.h"
#include "k4_pool.h"

// ---- errors ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
void k4_set_global_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
int k4_fail(k4_index* ix, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ix) ix->err = buf;
  g_err = buf;
  return code;
}
int k4_check_hip(k4_index* ix, hipError_t e, const char* what) {
  if (e == hipSuccess) return K4_OK;
  return k4_fail(ix, e == hipErrorOutOfMemory ? K4_ERR_MEM : K4_ERR_NO_DEVICE, "HIP error %d (%s) in %s", (int)e,
                 hipGetErrorString(e), what);
}
extern "C" const char* k4_last_error(const k4_index* ix) { return ix ? ix->err.c_str() : g_err.c_str(); }
extern "C" const char* k4_global_error(void) { return g_err.c_str(); }
extern "C" int k4_abi_version(void) { return K4_ABI_VERSION; }

// ---- kernels: packing ---------------------------------------------------------------------------------------
// One thread per 64 bases: 4 packed words + an exception flag; the wave's ballot, folded 4:1, gives the bits of the
// wave's 16 bitmap blocks of 256 bases.
__global__ void __launch_bounds__(256) k4k_pack_ref(const uint8_t* __restrict__ seq, uint64_t n,
                                                    uint32_t* __restrict__ ref2, uint16_t* __restrict__ excbm16,
                                                    uint64_t n_blocks) {
  uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool exc = false;
  if (b < n_blocks) {
    uint64_t base = b * 64;
    uint32_t w[4] = {0, 0, 0, 0};
    if (base + 64 <= n) {
      const uint4* p = reinterpret_cast<const uint4*>(seq + base);  // hipMalloc'd + 64-byte stride: aligned
#pragma unroll
      for (int q = 0; q < 4; q++) {
        uint4 v = p[q];
        uint32_t d[4] = {v.x, v.y, v.z, v.w};
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            uint32_t s = (d[k] >> (8 * j)) & 0x0f;
            if (s > 3) { exc = true; s = 0; }
            acc = (acc << 2) | s;
          }
        }
        w[q] = acc;
      }
    } else {
      for (int q = 0; q < 4; q++) {
        uint32_t acc = 0;
        for (int j = 0; j < 16; j++) {
          uint64_t pos = base + q * 16 + j;
          uint32_t s = pos < n ? (seq[pos] & 0x0f) : 0;
          if (s > 3) { exc = true; s = 0; }
          acc = (acc << 2) | s;
        }
        w[q] = acc;
      }
    }
    uint4 o = {w[0], w[1], w[2], w[3]};
    *reinterpret_cast<uint4*>(ref2 + b * 4) = o;
  }
  unsigned long long m = __ballot(exc);
  if ((threadIdx.x & 63) == 0) {
    m |= m >> 1;
    m |= m >> 2;  // bit 4j = any of the four 64-base flags of 256-base block j
    uint32_t v = 0;
    for (int j = 0; j < 16; j++) v |= (uint32_t)((m >> (4 * j)) & 1) << j;
    excbm16[(uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = (uint16_t)v;  // a wave covers 64 * 64 = 16 * 256 bases
  }
}

// exact nibbles of the flagged blocks: one thread per (flagged block, word)
__global__ void __launch_bounds__(256) k4k_exc_nibbles(const uint8_t* __restrict__ seq, uint64_t n,
                                                       const uint32_t* __restrict__ excblk, uint32_t n_exc,
                                                       uint32_t* __restrict__ excnib) {
  uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t wpb = K4_EXC_BLOCK / 8;  // words per flagged block
  if (t >= (uint64_t)n_exc * wpb) return;
  uint32_t r = (uint32_t)(t / wpb), q = (uint32_t)(t % wpb);
  uint64_t base = (uint64_t)excblk[r] * K4_EXC_BLOCK + q * 8;
  uint32_t acc = 0;
  for (int j = 0; j < 8; j++) {
    uint64_t pos = base + j;
    uint32_t s = pos < n ? (seq[pos] & 0x0f) : 7u;  // beyond the end reads as EOS
    acc |= s << (4 * j);
  }
  excnib[t] = acc;
}

// ---- kernels: k-mer table -----------------------------------------------------------------------------------
// "Ceiling" of the suffix at pos among the k-mers: the smallest k-mer code whose k-mer sorts strictly after the suffix
// (A<C<G<T<N<EOS).  A suffix whose first k symbols are ACGT with code c has ceiling c + 1; one that meets N / EOS (or
// the end of the block) after j clean symbols with prefix code P is greater than every k-mer starting with P, so its
// ceiling is (P + 1) << 2(k - j).  lb[c] = number of suffixes with ceiling <= c, exact for every code, which is what makes
// prefix ranges of the table (cores shorter than k) an exact statement of "all suffixes that start with the core".
K4_DEV uint64_t k4d_kmer_ceiling(const K4DevIndex& ix, uint64_t pos, uint32_t k) {
  if (pos + k <= ix.n && !k4d_any_exc(ix, (int64_t)pos, (int64_t)pos + k))
    return (k4d_ref_chunk(ix, (int64_t)pos) >> (64 - 2 * k)) + 1;
  uint64_t c = 0;
  for (uint32_t j = 0; j < k; j++) {
    const uint32_t s = pos + j < ix.n ? k4d_ref_base(ix, pos + j) : 7u;
    if (s > 3) return (c + 1) << (2 * (k - j));
    c = (c << 2) | s;
  }
  return c + 1;
}

template <int EL, typename T>
__global__ void __launch_bounds__(256) k4k_ktab_mark(K4DevIndex ix, T* __restrict__ tab) {
  // grid-stride: a launch may not exceed 2^32 threads in total, the block may hold more symbols than that
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < ix.n; i += stride) {
    const uint64_t pos = k4d_sa_at<EL>(ix, i);
    const uint64_t cc = k4d_kmer_ceiling(ix, pos, ix.k);
    const uint64_t cp = i > 0 ? k4d_kmer_ceiling(ix, k4d_sa_at<EL>(ix, i - 1), ix.k) : 0;
    if (cc != cp) {  // first suffix with this ceiling: every code in [cp, cc) has lb = i; the min-scan fills downwards
      constexpr int ST = sizeof(T) == 4 ? K4_KTAB_STRIDE32 : K4_KTAB_STRIDE64;
      const uint64_t c = cc - 1;
      tab[ST * c] = (T)i;
      if (sizeof(T) == 4) {
        tab[ST * c + 1] = (T)pos;
        tab[ST * c + 2] = (T)(k4d_ref_chunk(ix, (int64_t)(pos + ix.k)) >> 32);  // the next 16 bases
      } else  // 40-bit pos0; the bits above it receive sub-bucket counts in k4k_ktab_subcounts
        tab[ST * c + 1] = (T)(pos & K4_KTAB64_MASK);
    }
  }
}

template <typename T>
K4_DEV T k4_tmin(T a, T b) { return b < a ? b : a; }  // (no reliance on which ::min overload a 64-bit T picks)
// reverse (suffix) min-scan over the lb fields (entry stride 3 or 2 words) in three passes: lb[c] = min over c' >= c of lb[c']; unset = max value.
template <typename T>
__global__ void __launch_bounds__(256) k4k_scan_block_min(const T* __restrict__ tab, uint64_t n, T* __restrict__ agg) {
  __shared__ T sh[256];
  uint64_t base = (uint64_t)blockIdx.x * 2048 + (uint64_t)threadIdx.x * 8;
  T m = (T)~(T)0;
  for (int j = 0; j < 8; j++)
    if (base + j < n) m = k4_tmin<T>(m, tab[(sizeof(T) == 4 ? K4_KTAB_STRIDE32 : K4_KTAB_STRIDE64) * (base + j)]);
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] = k4_tmin<T>(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) agg[blockIdx.x] = sh[0];
}
template <typename T>
__global__ void __launch_bounds__(1024) k4k_scan_agg(T* __restrict__ agg, uint64_t nb) {
  // single block; exclusive reverse min-scan of the block aggregates, processed from the top in tiles of 1024
  __shared__ T sh[1024];
  __shared__ T carry_s;
  if (threadIdx.x == 0) carry_s = (T)~(T)0;
  __syncthreads();
  uint64_t tiles = (nb + 1023) / 1024;
  for (uint64_t t = tiles; t-- > 0;) {
    uint64_t i = t * 1024 + threadIdx.x;
    T v = i < nb ? agg[i] : (T)~(T)0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {  // inclusive reverse scan (Hillis-Steele)
      T o = threadIdx.x + s < 1024 ? sh[threadIdx.x + s] : (T)~(T)0;
      __syncthreads();
      sh[threadIdx.x] = k4_tmin<T>(sh[threadIdx.x], o);
      __syncthreads();
    }
    T carry = carry_s;
    T excl = threadIdx.x + 1 < 1024 ? sh[threadIdx.x + 1] : (T)~(T)0;  // strictly-right within the tile
    T tile_min = sh[0];
    __syncthreads();
    if (i < nb) agg[i] = k4_tmin<T>(excl, carry);
    if (threadIdx.x == 0) carry_s = k4_tmin<T>(carry, tile_min);
    __syncthreads();
  }
}
template <typename T>
__global__ void __launch_bounds__(256) k4k_scan_apply(T* __restrict__ tab, uint64_t n, const T* __restrict__ agg) {
  __shared__ T sh[256];
  uint64_t base = (uint64_t)blockIdx.x * 2048 + (uint64_t)threadIdx.x * 8;
  T v[8];
  T m = (T)~(T)0;
  for (int j = 7; j >= 0; j--) {
    T x = base + j < n ? tab[(sizeof(T) == 4 ? K4_KTAB_STRIDE32 : K4_KTAB_STRIDE64) * (base + j)] : (T)~(T)0;
    m = k4_tmin<T>(m, x);
    v[j] = m;  // min over this thread's elements j..7
  }
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int s = 1; s < 256; s <<= 1) {
    T o = threadIdx.x + s < 256 ? sh[threadIdx.x + s] : (T)~(T)0;
    __syncthreads();
    sh[threadIdx.x] = k4_tmin<T>(sh[threadIdx.x], o);
    __syncthreads();
  }
  T right = threadIdx.x + 1 < 256 ? sh[threadIdx.x + 1] : (T)~(T)0;
  right = k4_tmin<T>(right, agg[blockIdx.x]);
  for (int j = 0; j < 8; j++)
    if (base + j < n) tab[(sizeof(T) == 4 ? K4_KTAB_STRIDE32 : K4_KTAB_STRIDE64) * (base + j)] = k4_tmin<T>(v[j], right);
}

// 64-bit form, after the lb scan: sixteen 3-bit counts per bucket (k4_device.h) -- one thread per k-mer code walks its
// bucket's suffixes (a handful on average: the block holds a few times 4^k of them) and looks at the two symbols behind the
// k-mer.  Packed into the 24 bits above lb and above pos0.
template <int EL>
__global__ void __launch_bounds__(256) k4k_ktab_subcounts(K4DevIndex ix, uint64_t* __restrict__ tab, uint64_t n_codes) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c < n_codes; c += stride) {
    const uint64_t w0 = tab[2 * c];
    const uint64_t lb = w0 & K4_KTAB64_MASK, lb1 = tab[2 * (c + 1)] & K4_KTAB64_MASK;  // (a neighbour only ever changes its high bits)
    const uint64_t size = lb1 - lb;
    uint64_t cnt = 0;
    if (size > 6 * 16)
      cnt = K4_KTAB64_IRREGULAR;
    else
      for (uint64_t j = 0; j < size; j++) {
        const uint64_t pos = k4d_sa_at<EL>(ix, lb + j);
        // (a suffix of the bucket that meets N / a separator within its first k symbols sorts behind the clean ones: irregular too)
        if (pos + ix.k + 2 > ix.n || k4d_any_exc(ix, (int64_t)pos, (int64_t)pos + ix.k + 2)) { cnt = K4_KTAB64_IRREGULAR; break; }
        const uint32_t e = (uint32_t)(k4d_ref_chunk(ix, (int64_t)(pos + ix.k)) >> 60);
        if (((cnt >> (3 * e)) & 7) == 6) { cnt = K4_KTAB64_IRREGULAR; break; }
        cnt += 1ull << (3 * e);
      }
    tab[2 * c] = lb | ((cnt & 0xFFFFFFull) << 40);
    tab[2 * c + 1] = (tab[2 * c + 1] & K4_KTAB64_MASK) | ((cnt >> 24) << 40);
  }
}

__global__ void k4k_unpack_range(K4DevIndex ix, uint64_t start, uint64_t len, uint8_t* __restrict__ out) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < len) out[t] = (uint8_t)k4d_ref_base(ix, start + t);
}

// ---- host: device structures ----------------------------------------------------------------------------------
static int choose_k(uint64_t n, int want) {
  if (want > 0) return std::min(16, std::max(4, want));
  int k = 1;
  while (k < 16 && (1ull << (2 * k)) < n) k++;  // smallest k with 4^k >= n
  return std::max(6, k);
}

// suffixes that sit in buckets deeper than K4_DEEP_BUCKET (after the lb scan, before the 64-bit form's counts go into the high
// bits): the share of repeat families in the index, which decides how the first alignment phase is launched (k4_align.hip)
template <typename T>
__global__ void __launch_bounds__(256) k4k_ktab_deep(const T* __restrict__ tab, uint64_t n_codes, unsigned long long* __restrict__ total) {
  constexpr int ST = sizeof(T) == 4 ? K4_KTAB_STRIDE32 : K4_KTAB_STRIDE64;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  unsigned long long mine = 0;
  for (uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x; c < n_codes; c += stride) {
    const uint64_t size = (uint64_t)tab[ST * (c + 1)] - (uint64_t)tab[ST * c];
    if (size > K4_DEEP_BUCKET) mine += size;
  }
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(total, mine);
}

template <int EL, typename T>
static int build_ktab(k4_index* ix) {
  uint64_t nent = (1ull << (2 * ix->d.k)) + 1;
  T* tab = nullptr;
  const int ST = sizeof(T) == 4 ? K4_KTAB_STRIDE32 : K4_KTAB_STRIDE64;
  K4_HIP(ix, hipMalloc(&tab, nent * ST * sizeof(T) + 32));
  ix->ktab = tab;
  ix->device_bytes += nent * ST * sizeof(T) + 32;
  K4_HIP(ix, hipMemset(tab, 0xFF, nent * ST * sizeof(T) + 32));
  T last = (T)ix->d.n;
  K4_HIP(ix, hipMemcpy(tab + (size_t)ST * (nent - 1), &last, sizeof(T), hipMemcpyHostToDevice));
  ix->d.ktab = tab;
  uint64_t nb = std::min<uint64_t>((ix->d.n + 255) / 256, 1ull << 23);
  hipLaunchKernelGGL((k4k_ktab_mark<EL, T>), dim3((unsigned)nb), dim3(256), 0, 0, ix->d, tab);
  K4_HIP(ix, hipGetLastError());
  uint64_t sb = (nent + 2047) / 2048;
  T* agg = nullptr;
  K4_HIP(ix, hipMalloc(&agg, sb * sizeof(T)));
  hipLaunchKernelGGL((k4k_scan_block_min<T>), dim3((unsigned)sb), dim3(256), 0, 0, tab, nent, agg);
  hipLaunchKernelGGL((k4k_scan_agg<T>), dim3(1), dim3(1024), 0, 0, agg, sb);
  hipLaunchKernelGGL((k4k_scan_apply<T>), dim3((unsigned)sb), dim3(256), 0, 0, tab, nent, agg);
  K4_HIP(ix, hipGetLastError());
  K4_HIP(ix, hipDeviceSynchronize());
  K4_HIP(ix, hipFree(agg));
  {
    unsigned long long* d_deep = nullptr;
    unsigned long long deep = 0;
    K4_HIP(ix, hipMalloc(&d_deep, 8));
    K4_HIP(ix, hipMemset(d_deep, 0, 8));
    hipLaunchKernelGGL((k4k_ktab_deep<T>), dim3(8192), dim3(256), 0, 0, (const T*)tab, nent - 1, d_deep);
    K4_HIP(ix, hipMemcpy(&deep, d_deep, 8, hipMemcpyDeviceToHost));
    K4_HIP(ix, hipFree(d_deep));
    ix->deep_bucket_frac = ix->d.n ? (double)deep / (double)ix->d.n : 0.0;
  }
  if (sizeof(T) == 8) {
    const uint64_t n_codes = nent - 1;
    hipLaunchKernelGGL((k4k_ktab_subcounts<EL>), dim3((unsigned)std::min<uint64_t>((n_codes + 255) / 256, 1ull << 22)), dim3(256), 0, 0, ix->d,
                       (uint64_t*)tab, n_codes);
    K4_HIP(ix, hipGetLastError());
    K4_HIP(ix, hipDeviceSynchronize());
  }
  return K4_OK;
}

// d_seq: concat_len bytes (1 byte/base) in HBM; ix->sa, entries and ix->d.{n,el} must already be set.
int k4i_build_device_structures(k4_index* ix, const void* d_seq, int kmer_k) {
  const uint64_t n = ix->d.n;
  const uint64_t n_blocks = (n + 63) / 64;        // 64-base packing units
  const uint64_t words = n_blocks * 4;
  const uint64_t n_eblocks = (n + K4_EXC_BLOCK - 1) >> K4_EXC_SHIFT;
  const uint64_t bm_words = ((n_blocks + 255) / 256) * 2 + 8;  // the pack kernel writes 16 bits per wave; +pad for the 8-byte fetch
  K4_HIP(ix, hipMalloc(&ix->ref2_alloc, (words + 2 * K4_PAD_WORDS) * 4));
  K4_HIP(ix, hipMemset(ix->ref2_alloc, 0, (words + 2 * K4_PAD_WORDS) * 4));
  K4_HIP(ix, hipMalloc(&ix->excbm, bm_words * 4));
  K4_HIP(ix, hipMemset(ix->excbm, 0, bm_words * 4));
  ix->device_bytes += (words + 2 * K4_PAD_WORDS) * 4 + bm_words * 4;
  uint32_t* ref2 = ix->ref2_alloc + K4_PAD_WORDS;
  hipLaunchKernelGGL(k4k_pack_ref, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, 0,
                     (const uint8_t*)d_seq, n, ref2, (uint16_t*)ix->excbm, n_blocks);
  K4_HIP(ix, hipGetLastError());
  // flagged-block list on the host (bitmap is n/512 bytes)
  std::vector<uint32_t> bm(bm_words);
  K4_HIP(ix, hipMemcpy(bm.data(), ix->excbm, bm_words * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> blk;
  for (uint64_t w = 0; w < bm_words; w++) {
    uint32_t v = bm[w];
    while (v) {
      int b = __builtin_ctz(v);
      v &= v - 1;
      uint64_t id = w * 32 + b;
      if (id < n_eblocks) blk.push_back((uint32_t)id);
    }
  }
  uint32_t n_exc = (uint32_t)blk.size();
  // coarse bitmap for LDS: the smallest power-of-two granularity at which the genome (plus slack for windows that
  // run past the end) fits K4_SUP_WORDS * 32 - 64 bits
  uint32_t sup_shift = K4_EXC_SHIFT;
  while (((n + (1ull << 16)) >> sup_shift) + 64 > (uint64_t)K4_SUP_WORDS * 32) sup_shift++;
  std::vector<uint32_t> sup(K4_SUP_WORDS, 0);
  for (uint32_t id : blk) {
    uint64_t b = (uint64_t)id >> (sup_shift - K4_EXC_SHIFT);
    sup[b >> 5] |= 1u << (b & 31);
  }
  K4_HIP(ix, hipMalloc(&ix->excsup, K4_SUP_WORDS * 4));
  K4_HIP(ix, hipMemcpy(ix->excsup, sup.data(), K4_SUP_WORDS * 4, hipMemcpyHostToDevice));
  K4_HIP(ix, hipMalloc(&ix->excblk, (size_t)(n_exc + 1) * 4));
  K4_HIP(ix, hipMalloc(&ix->excnib, (size_t)(n_exc + 1) * (K4_EXC_BLOCK / 2)));
  ix->device_bytes += (uint64_t)(n_exc + 1) * (4 + K4_EXC_BLOCK / 2);
  if (n_exc) {
    K4_HIP(ix, hipMemcpy(ix->excblk, blk.data(), (size_t)n_exc * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k4k_exc_nibbles, dim3((unsigned)(((uint64_t)n_exc * (K4_EXC_BLOCK / 8) + 255) / 256)), dim3(256), 0, 0,
                       (const uint8_t*)d_seq, n, ix->excblk, n_exc, ix->excnib);
    K4_HIP(ix, hipGetLastError());
  }
  // entries
  uint32_t ne = (uint32_t)ix->entries.size();
  std::vector<uint64_t> es(ne), ee(ne);
  std::vector<uint32_t> ei(ne);
  ix->tot_seqs_len = 0;
  for (uint32_t i = 0; i < ne; i++) {
    es[i] = ix->entries[i].start_ofs;
    ee[i] = ix->entries[i].end_ofs;
    ei[i] = ix->entries[i].entry_id;
    ix->tot_seqs_len += ix->entries[i].seq_len;
  }
  K4_HIP(ix, hipMalloc(&ix->ent_start, (size_t)(ne + 1) * 8));
  K4_HIP(ix, hipMalloc(&ix->ent_end, (size_t)(ne + 1) * 8));
  K4_HIP(ix, hipMalloc(&ix->ent_id, (size_t)(ne + 1) * 4));
  if (ne) {
    K4_HIP(ix, hipMemcpy(ix->ent_start, es.data(), (size_t)ne * 8, hipMemcpyHostToDevice));
    K4_HIP(ix, hipMemcpy(ix->ent_end, ee.data(), (size_t)ne * 8, hipMemcpyHostToDevice));
    K4_HIP(ix, hipMemcpy(ix->ent_id, ei.data(), (size_t)ne * 4, hipMemcpyHostToDevice));
  }
  K4_HIP(ix, hipMalloc(&ix->counters, sizeof(k4_counters) + K4_PROF_SLOTS * 8));  // (+ the profiling build's slots, k4_align.hip)
  K4_HIP(ix, hipMemset(ix->counters, 0, sizeof(k4_counters) + K4_PROF_SLOTS * 8));
  ix->d.ref2 = ref2;
  ix->d.excbm = ix->excbm;
  ix->d.excsup = ix->excsup;
  ix->d.sup_shift = sup_shift;
  ix->d.excblk = ix->excblk;
  ix->d.excnib = ix->excnib;
  ix->d.n_exc = n_exc;
  ix->d.sa = ix->sa;
  ix->d.ent_start = ix->ent_start;
  ix->d.ent_end = ix->ent_end;
  ix->d.ent_id = ix->ent_id;
  ix->d.n_entries = ne;
  ix->d.k = (uint32_t)choose_k(n, kmer_k);
  // 16-byte entries wherever they fit: besides lb and pos0 they hold the sixteen sub-bucket counts, with which the table answers
  // like one of k + 2 bases (C2: 4.1 -> 1.9 probes per read, 21.2 -> 18.1 ms per batch; 69 instead of 52 GB at 3 Gbp).  The
  // 12-byte form {lb, pos0, sig} stays for blocks below 2^32 symbols on a device that is short of memory.
  ix->d.ktab64 = 1;
  if (n < 0xFFFFFFFFull) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && (((uint64_t)1 << (2 * ix->d.k)) + 1) * 16 > fr / 2) ix->d.ktab64 = 0;
    if (getenv("K4_FORCE_KTAB64")) ix->d.ktab64 = atoi(getenv("K4_FORCE_KTAB64")) ? 1 : 0;  // test hook: either form on a small index
  }
  int rc;
  if (ix->d.el == 4)
    rc = ix->d.ktab64 ? build_ktab<4, uint64_t>(ix) : build_ktab<4, uint32_t>(ix);
  else
    rc = ix->d.ktab64 ? build_ktab<5, uint64_t>(ix) : build_ktab<5, uint32_t>(ix);
  if (rc != K4_OK) return rc;
  K4_HIP(ix, hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
  return K4_OK;
}

// ---- host: open / close ---------------------------------------------------------------------------------------
static int select_device(int device) {
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0) {
    k4_set_global_error("no usable HIP device (hipGetDeviceCount: %s); libk4sfx has no CPU fallback",
                        e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    return K4_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= cnt) {
    k4_set_global_error("device %d out of range (0..%d)", device, cnt - 1);
    return K4_ERR_PARAMS;
  }
  e = hipSetDevice(device);
  if (e != hipSuccess) {
    k4_set_global_error("hipSetDevice(%d): %s", device, hipGetErrorString(e));
    return K4_ERR_NO_DEVICE;
  }
  return K4_OK;
}

static int check_entries(uint64_t n, uint32_t ne, const k4_entry* e) {
  uint64_t prev_end = 0;
  for (uint32_t i = 0; i < ne; i++) {
    if (e[i].end_ofs < e[i].start_ofs || e[i].end_ofs >= n) return 0;
    if (i && e[i].start_ofs <= prev_end) return 0;
    prev_end = e[i].end_ofs;
  }
  return 1;
}

// Pageable host memory (the mapped .sfx file, 15 GB at 3 Gbp) -> HBM.  One hipMemcpy from pageable memory is staged by the runtime
// through its own small pinned buffers by ONE thread, page faults of the mapping included (~6 GB/s).  Here: three pinned pieces,
// several host threads fill one (touching the file's pages side by side) while the previous piece's copy is on its way.
// the .sfx files mapped by k4_sfx_map, each with a descriptor kept open: what lies in one of them is read with pread() -- the kernel
// copies from the page cache without a page fault per 4 KB of the mapping (about twice as fast here)
struct K4MappedFile { const uint8_t* base; size_t len; int fd; };
static std::mutex g_maps_m;
static std::vector<K4MappedFile> g_maps;
static int mapped_fd(const uint8_t* p, size_t bytes, off_t* off) {
  std::lock_guard<std::mutex> lk(g_maps_m);
  for (const K4MappedFile& m : g_maps)
    if (p >= m.base && p + bytes <= m.base + m.len) { *off = (off_t)(p - m.base); return m.fd; }
  return -1;
}

// The pages of a large pageable (or file-mapped) source registered with HIP in place, piece by piece, and copied straight from
// there: no staging copy through pinned buffers.  Registering costs about as much per byte as one host memcpy, but several
// pieces register side by side while the DMA engine (57 GB/s from registered pages) drains the ones before: a 15 GB .sfx in
// tmpfs reaches the device at more than twice the rate of the staged path below, which stays as the fallback (a source
// HIP refuses to register).  K4_NO_HOSTREG=1 forces the fallback.
static bool k4i_upload_registered(void* d_dst, const uint8_t* src, size_t bytes) {
  static const bool off = getenv("K4_NO_HOSTREG") != nullptr;
  if (off) return false;
  const size_t piece = (size_t)512 << 20, page = 4096;
  const size_t n_pieces = (bytes + piece - 1) / piece;
  const int NT = (int)std::min<size_t>(4, n_pieces);
  std::atomic<size_t> next(0);
  std::atomic<int> failed(0), copied(0);
  auto work = [&]() {
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { failed = 1; return; }
    for (size_t k; !failed && (k = next.fetch_add(1)) < n_pieces;) {
      const size_t a = k * piece, len = std::min(piece, bytes - a);
      // whole pages around the piece (the mapping holds them: it starts at the file's first byte and ends on a page boundary)
      const uintptr_t lo = (uintptr_t)(src + a) & ~(uintptr_t)(page - 1), hi = ((uintptr_t)(src + a + len) + page - 1) & ~(uintptr_t)(page - 1);
      if (hipHostRegister((void*)lo, (size_t)(hi - lo), hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); failed = 1; break; }
      const bool ok = hipMemcpyAsync((uint8_t*)d_dst + a, src + a, len, hipMemcpyHostToDevice, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
      (void)hipHostUnregister((void*)lo);
      if (!ok) { failed = 1; break; }
      copied++;
    }
    (void)hipStreamDestroy(st);
  };
  std::vector<std::thread> th;
  for (int t = 0; t < NT; t++) th.emplace_back(work);
  for (std::thread& x : th) x.join();
  return !failed && (size_t)copied.load() == n_pieces;  // (a failure part-way: the staged path writes everything again)
}

static int k4i_upload_pageable(k4_index* ix, void* d_dst, const uint8_t* src, size_t bytes) {
  const size_t piece = (size_t)128 << 20;
  if (bytes >= 2 * piece && k4i_upload_registered(d_dst, src, bytes)) return K4_OK;
  off_t file_off = 0;
  const int fd = bytes >= 2 * piece ? mapped_fd(src, bytes, &file_off) : -1;
  if (bytes < 2 * piece) return k4_check_hip(ix, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice), "hipMemcpy(index)");
  const int NB = 3;
  const int NT = (int)std::min(16u, std::max(4u, std::thread::hardware_concurrency()));
  uint8_t* buf[NB] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[NB] = {nullptr, nullptr, nullptr};
  bool used[NB] = {false, false, false};
  hipStream_t st = nullptr;
  int rc = K4_OK;
  auto done = [&]() {
    if (st) { hipStreamSynchronize(st); hipStreamDestroy(st); }
    for (int b = 0; b < NB; b++) { if (buf[b]) hipHostFree(buf[b]); if (ev[b]) hipEventDestroy(ev[b]); }
    return rc;
  };
  if ((rc = k4_check_hip(ix, hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "stream")) != K4_OK) return done();
  for (int b = 0; b < NB; b++) {
    if ((rc = k4_check_hip(ix, hipHostMalloc((void**)&buf[b], piece, hipHostMallocDefault), "pinned staging")) != K4_OK) return done();
    if ((rc = k4_check_hip(ix, hipEventCreateWithFlags(&ev[b], hipEventDisableTiming), "event")) != K4_OK) return done();
  }
  int k = 0;
  for (size_t off = 0; off < bytes; off += piece, k++) {
    const int b = k % NB;
    const size_t len = std::min(piece, bytes - off);
    if (used[b] && (rc = k4_check_hip(ix, hipEventSynchronize(ev[b]), "upload")) != K4_OK) return done();
    std::vector<std::thread> th;
    for (int t = 0; t < NT; t++)
      th.emplace_back([=] {
        const size_t a = len * (size_t)t / NT, z = len * (size_t)(t + 1) / NT;
        size_t got = a;
        while (fd >= 0 && got < z) {
          const ssize_t g = pread(fd, buf[b] + got, z - got, file_off + (off_t)(off + got));
          if (g <= 0) break;  // (whatever is missing comes through the mapping below)
          got += (size_t)g;
        }
        if (got < z) memcpy(buf[b] + got, src + off + got, z - got);
      });
    for (std::thread& x : th) x.join();
    if ((rc = k4_check_hip(ix, hipMemcpyAsync((uint8_t*)d_dst + off, buf[b], len, hipMemcpyHostToDevice, st), "upload")) != K4_OK) return done();
    if ((rc = k4_check_hip(ix, hipEventRecord(ev[b], st), "upload")) != K4_OK) return done();
    used[b] = true;
  }
  rc = k4_check_hip(ix, hipStreamSynchronize(st), "upload");
  return done();
}

// K4_TRACE=1: where the time of an index load goes, on stderr
static double k4i_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static bool k4i_trace() { static const bool on = getenv("K4_TRACE") != nullptr; return on; }

// the index object with what the file's tables say (entries, lengths): everything k4_info / k4_get_entry / k4_min_core_len answer
static int open_prepare(uint64_t n, uint32_t el, uint32_t ne, const k4_entry* entries, const char* dataset, int device, k4_index** out) {
  if (!out) return K4_ERR_PARAMS;
  *out = nullptr;
  if (n == 0 || (el != 4 && el != 5) || (el == 4 && n > 0xFFFFFFFFull) || !entries || ne == 0) {
    k4_set_global_error("bad index geometry: concat_len=%llu el=%u entries=%u", (unsigned long long)n, el, ne);
    return K4_ERR_PARAMS;
  }
  if (!check_entries(n, ne, entries)) {
    k4_set_global_error("entries table is not sorted / inside the block");
    return K4_ERR_PARAMS;
  }
  int rc = select_device(device);
  if (rc != K4_OK) return rc;
  k4_index* ix = new k4_index;
  ix->device = device;
  ix->d.n = n;
  ix->d.el = el;
  ix->entries.assign(entries, entries + ne);
  ix->dataset = dataset ? dataset : "";
  ix->d.max_iter = 50000;  // cDfltMaxIter, libkit4b/SfxArray.h:12 (set here: k4_set_max_iter may come while the arrays still load)
  ix->d.n_entries = ne;    // (k4_info answers from here on)
  ix->tot_seqs_len = 0;
  for (uint32_t i = 0; i < ne; i++) ix->tot_seqs_len += entries[i].seq_len;
  *out = ix;
  return K4_OK;
}
// ... and the arrays: suffix array and sequence onto the device, packed reference / exception tables / k-mer table built there
static int open_load(k4_index* ix, const uint8_t* h_seq, const void* d_seq_in, const uint8_t* h_sa, void* d_sa_in, int adopt_sa, int kmer_k) {
  const uint64_t n = ix->d.n;
  const uint32_t el = ix->d.el;
  int rc = k4_check_hip(ix, hipSetDevice(ix->device), "hipSetDevice");
  if (rc != K4_OK) return rc;
  // suffix array (padded so the 5-byte reader may touch 8 bytes past the end)
  if (d_sa_in && adopt_sa) {
    ix->sa = (uint8_t*)d_sa_in;
    ix->owns_sa = false;
  } else {
    const double t_m = k4i_now();
    if ((rc = k4_check_hip(ix, hipMalloc(&ix->sa, n * el + 16), "hipMalloc(sa)")) != K4_OK) return rc;
    if (k4i_trace()) fprintf(stderr, "[k4 trace] device selected and %.2f GB allocated for the suffix array in %.2fs\n", n * el / 1e9, k4i_now() - t_m);
    ix->device_bytes += n * el + 16;
    const double t_a = k4i_now();
    rc = d_sa_in ? k4_check_hip(ix, hipMemcpy(ix->sa, d_sa_in, n * el, hipMemcpyDeviceToDevice), "hipMemcpy(sa)")
                 : k4i_upload_pageable(ix, ix->sa, h_sa, (size_t)(n * el));
    if (rc != K4_OK) return rc;
    if (k4i_trace()) fprintf(stderr, "[k4 trace] suffix array %.2f GB on the device in %.2fs\n", n * el / 1e9, k4i_now() - t_a);
  }
  // sequence bytes: temporary on the device, only needed to derive the packed form
  uint8_t* d_tmp = nullptr;
  const void* d_seq = d_seq_in;
  if (!d_seq) {
    if ((rc = k4_check_hip(ix, hipMalloc(&d_tmp, n + 64), "hipMalloc(seq)")) != K4_OK) return rc;
    const double t_c = k4i_now();
    rc = k4i_upload_pageable(ix, d_tmp, h_seq, (size_t)n);
    if (k4i_trace()) fprintf(stderr, "[k4 trace] sequence %.2f GB on the device in %.2fs\n", n / 1e9, k4i_now() - t_c);
    if (rc != K4_OK) {
      hipFree(d_tmp);
      return rc;
    }
    d_seq = d_tmp;
  }
  const double t_b = k4i_now();
  rc = k4i_build_device_structures(ix, d_seq, kmer_k);
  if (k4i_trace()) { (void)hipDeviceSynchronize(); fprintf(stderr, "[k4 trace] packed reference, exception tables, k-mer table built in %.2fs\n", k4i_now() - t_b); }
  if (d_tmp) hipFree(d_tmp);
  return rc;
}
static int open_common(uint64_t n, uint32_t el, const uint8_t* h_seq, const void* d_seq_in, const uint8_t* h_sa,
                       void* d_sa_in, int adopt_sa, uint32_t ne, const k4_entry* entries, const char* dataset,
                       int device, int kmer_k, k4_index** out) {
  k4_index* ix = nullptr;
  int rc = open_prepare(n, el, ne, entries, dataset, device, &ix);
  if (rc != K4_OK) return rc;
  rc = open_load(ix, h_seq, d_seq_in, h_sa, d_sa_in, adopt_sa, kmer_k);
  if (rc != K4_OK) {
    const std::string keep = ix->err;
    if (d_sa_in && adopt_sa) { ix->sa = nullptr; ix->owns_sa = true; }  // (the caller keeps what it handed in)
    k4_close(ix);
    k4_set_global_error("%s", keep.c_str());
    *out = nullptr;
    return rc;
  }
  *out = ix;
  return K4_OK;
}

// host memory of the caller made DMA-able in place (page-aligned address and length): what a host program without a HIP runtime of
// its own needs to let the *_dev / pipeline entry points copy straight from or into a mapped file
extern "C" int k4_host_register(void* p, uint64_t bytes) {
  if (!p || !bytes) return K4_ERR_PARAMS;
  const hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault);
  if (e != hipSuccess) { (void)hipGetLastError(); return k4_check_hip(nullptr, e, "hipHostRegister"); }
  return K4_OK;
}
extern "C" int k4_host_unregister(void* p) {
  if (!p) return K4_ERR_PARAMS;
  return k4_check_hip(nullptr, hipHostUnregister(p), "hipHostUnregister");
}

// pageable (or file-mapped) host memory -> device through pinned 128 MB pieces filled by several threads: what k4_open does with
// the two big arrays of an .sfx file (10 GB/s where one pageable hipMemcpy gives 6), for callers that place an index themselves
extern "C" int k4_upload_pageable(int device, void* d_dst, const void* h_src, uint64_t bytes) {
  if (!d_dst || (!h_src && bytes)) return K4_ERR_PARAMS;
  int rc = k4_check_hip(nullptr, hipSetDevice(device), "hipSetDevice");
  if (rc != K4_OK) return rc;
  return k4i_upload_pageable(nullptr, d_dst, (const uint8_t*)h_src, (size_t)bytes);
}

extern "C" int k4_open_host(uint64_t n, uint32_t el, const uint8_t* seq, const uint8_t* sa, uint32_t ne,
                            const k4_entry* entries, const char* dataset, int device, int kmer_k, k4_index** out) {
  if (!seq || !sa) return K4_ERR_PARAMS;
  return open_common(n, el, seq, nullptr, sa, nullptr, 0, ne, entries, dataset, device, kmer_k, out);
}

extern "C" int k4_open_device(uint64_t n, uint32_t el, const void* d_seq, void* d_sa, int adopt_sa, uint32_t ne,
                              const k4_entry* entries, const char* dataset, int device, int kmer_k, k4_index** out) {
  if (!d_seq || !d_sa) return K4_ERR_PARAMS;
  return open_common(n, el, nullptr, d_seq, nullptr, d_sa, adopt_sa, ne, entries, dataset, device, kmer_k, out);
}

// .sfx container: header 1224 B pack(4), block header 20 B pack(1), entries 111 B pack(1)
template <typename T>
static T rd(const uint8_t* p) {
  T v;
  memcpy(&v, p, sizeof(T));
  return v;
}

// the .sfx container mapped read-only, its tables decoded (CSfxArray::Disk2Hdr / Disk2Entries, SfxArray.cpp:629-825)
extern "C" int k4_sfx_map(const char* path, k4_sfx_file* o) {
  if (!path || !*path || !o) return K4_ERR_PARAMS;
  memset(o, 0, sizeof(*o));
  int fd = open(path, O_RDONLY);
  if (fd < 0) {
    k4_set_global_error("unable to open %s", path);
    return K4_ERR_OPEN_FILE;
  }
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size < 1224) {
    close(fd);
    k4_set_global_error("%s: too short for a suffix array file header", path);
    return K4_ERR_NOT_SFX;
  }
  size_t len = (size_t)st.st_size;
  const uint8_t* f = (const uint8_t*)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
  if (f == MAP_FAILED) {
    close(fd);
    k4_set_global_error("mmap of %s failed", path);
    return K4_ERR_FILE_ACCESS;
  }
  auto done = [&](int code) {
    munmap((void*)f, len);
    close(fd);
    return code;
  };
  if (tolower(f[0]) != 's' || tolower(f[1]) != 'f' || tolower(f[2]) != 'x' || f[3] < '3' || f[3] > '5') {
    k4_set_global_error("%s opened but invalid magic signature - not a 'kit4b index' generated suffix array file", path);
    return done(K4_ERR_NOT_SFX);
  }
  uint32_t ver = rd<uint32_t>(f + 4);
  if (ver < 4 || ver > 5) {  // v3 (36-char names, SfxArray.h:127-141) is not produced by any current kit4b
    k4_set_global_error("%s: structure version %u is not supported (4..5)", path, ver);
    return done(K4_ERR_FILE_VER);
  }
  uint32_t attr = rd<uint32_t>(f + 8);
  if (attr & 3) {
    k4_set_global_error("%s: bisulfite / colourspace indexes are outside the accelerated path", path);
    return done(K4_ERR_UNSUPPORTED);
  }
  uint64_t entries_ofs = rd<uint64_t>(f + 20);
  uint32_t entries_size = rd<uint32_t>(f + 28);
  uint32_t n_blocks = rd<uint32_t>(f + 32);
  uint64_t block_ofs = rd<uint64_t>(f + 44);
  if (n_blocks != 1 || block_ofs + 20 > len || entries_ofs == 0 || entries_ofs + entries_size > len ||
      entries_size < 8) {
    k4_set_global_error("%s: inconsistent header (blocks=%u)", path, n_blocks);
    return done(K4_ERR_FILE_ACCESS);
  }
  memcpy(o->dataset, f + 52, 80);
  o->dataset[80] = 0;
  const uint8_t* b = f + block_ofs;
  uint64_t n = rd<uint64_t>(b + 8);
  uint32_t el = rd<uint32_t>(b + 16);
  if ((el != 4 && el != 5) || block_ofs + 20 + n + n * el > len) {
    k4_set_global_error("%s: suffix block truncated (n=%llu el=%u)", path, (unsigned long long)n, el);
    return done(K4_ERR_FILE_ACCESS);
  }
  const uint8_t* e = f + entries_ofs;
  uint32_t ne = rd<uint32_t>(e);
  if (8 + (uint64_t)ne * 111 > entries_size) {
    k4_set_global_error("%s: entries block truncated", path);
    return done(K4_ERR_FILE_ACCESS);
  }
  k4_entry* ents = (k4_entry*)calloc(ne ? ne : 1, sizeof(k4_entry));
  if (!ents) return done(K4_ERR_MEM);
  for (uint32_t i = 0; i < ne; i++) {
    const uint8_t* p = e + 8 + (size_t)i * 111;
    k4_entry& d = ents[i];
    d.entry_id = rd<uint32_t>(p);
    d.fblock_id = rd<uint32_t>(p + 4);
    memcpy(d.name, p + 8, 80);
    d.name_hash = rd<uint16_t>(p + 89);
    d.seq_len = rd<uint32_t>(p + 91);
    d.start_ofs = rd<uint64_t>(p + 95);
    d.end_ofs = rd<uint64_t>(p + 103);
  }
  o->map = f; o->map_len = len; o->concat_len = n; o->sfx_el_size = el; o->n_entries = ne; o->entries = ents;
  o->seq = b + 20; o->sa = b + 20 + n; o->header = f;
  { std::lock_guard<std::mutex> lk(g_maps_m); g_maps.push_back({f, len, fd}); }  // (closed by k4_sfx_unmap)
  return K4_OK;
}
extern "C" void k4_sfx_unmap(k4_sfx_file* o) {
  if (!o) return;
  if (o->map) {
    {
      std::lock_guard<std::mutex> lk(g_maps_m);
      for (size_t k = 0; k < g_maps.size(); k++)
        if (g_maps[k].base == (const uint8_t*)o->map) { close(g_maps[k].fd); g_maps.erase(g_maps.begin() + (long)k); break; }
    }
    munmap((void*)o->map, o->map_len);
  }
  free(o->entries);
  memset(o, 0, sizeof(*o));
}

// k4_open in two halves: k4_open_async returns once the file's header and entry table are read -- k4_info, k4_get_entry,
// k4_min_core_len, k4_set_max_iter, k4_set_fastq_quality and the ingest side of a pipeline (open, acquire / submit: upload,
// parse, length filter) work from then on -- while a thread of the library uploads the two big arrays and builds the device
// structures; k4_open_wait joins it (every alignment entry point of the pipeline does so itself before its first batch).  A
// program's reading of its reads thereby runs beside the index load (kalign overlaps its loader and its aligner threads
// likewise, KAligner.cpp:4786-4866).
extern "C" int k4_open_async(const char* path, int device, int kmer_k, k4_index** out) {
  if (!out) return K4_ERR_PARAMS;
  *out = nullptr;
  k4_sfx_file* m = new k4_sfx_file;
  int rc = k4_sfx_map(path, m);
  if (rc != K4_OK) { delete m; return rc; }
  k4_index* ix = nullptr;
  rc = open_prepare(m->concat_len, m->sfx_el_size, m->n_entries, m->entries, m->dataset, device, &ix);
  if (rc != K4_OK) { k4_sfx_unmap(m); delete m; return rc; }
  ix->raw_header.assign(m->header, m->header + 1224);
  ix->load_rc = K4_OK;
  ix->loader = std::thread([ix, m, kmer_k] {
    const double t0 = k4i_now();
    ix->load_rc = open_load(ix, m->seq, nullptr, m->sa, nullptr, 0, kmer_k);
    ix->load_seconds = k4i_now() - t0;
    k4_sfx_unmap(m);
    delete m;
  });
  *out = ix;
  return K4_OK;
}
extern "C" int k4_open_wait(k4_index* ix) {
  if (!ix) return K4_ERR_PARAMS;
  if (ix->loader.joinable()) ix->loader.join();
  if (ix->load_rc != K4_OK) k4_set_global_error("%s", ix->err.c_str());
  return ix->load_rc;
}
extern "C" double k4_open_seconds(const k4_index* ix) { return ix ? ix->load_seconds : 0.0; }
extern "C" int k4_open(const char* path, int device, int kmer_k, k4_index** out) {
  if (!out) return K4_ERR_PARAMS;
  int rc = k4_open_async(path, device, kmer_k, out);
  if (rc != K4_OK) return rc;
  rc = k4_open_wait(*out);
  if (rc != K4_OK) {
    const std::string keep = (*out)->err;
    k4_close(*out);
    *out = nullptr;
    k4_set_global_error("%s", keep.c_str());
  }
  return rc;
}

// the header text an index built from parts reports and writes (GetSfxHeader / k4_write_sfx): the file's own 1224 bytes
extern "C" int k4_set_raw_header(k4_index* ix, const void* hdr_1224) {
  if (!ix || !hdr_1224) return K4_ERR_PARAMS;
  ix->raw_header.assign((const uint8_t*)hdr_1224, (const uint8_t*)hdr_1224 + 1224);
  return K4_OK;
}

extern "C" void k4_close(k4_index* ix) {
  if (!ix) return;
  if (ix->loader.joinable()) ix->loader.join();
  hipSetDevice(ix->device);
  K4Workspace& w = ix->ws;
  void* ptrs[] = {ix->ref2_alloc, ix->excbm, ix->excsup, ix->excblk, ix->excnib, ix->owns_sa ? ix->sa : nullptr, ix->ktab,
                  ix->ent_start, ix->ent_end, ix->ent_id, ix->counters, w.ids[0], w.ids[1], w.rows[0], w.rows[1], w.slow_list, w.slow_step, w.huge_list, w.huge_step, w.ctl,
                  w.slow_probe, w.slow_hash, w.d_reads, w.d_offs, w.d_lens, w.d_out4, w.d_hits};
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (ix->d_qlut) hipFree(ix->d_qlut);
  if (w.d_small) hipFree(w.d_small);
  if (w.h_small) hipHostFree(w.h_small);
  for (void* p : {(void*)ix->pe_rr, (void*)ix->pe_hits, (void*)ix->pe_list, (void*)ix->pe_ctl, ix->rs_tasks, ix->rs_reads, ix->rs_res, ix->rs_hits})
    if (p) hipFree(p);
  if (ix->stream) hipStreamDestroy(ix->stream);
  k4_pool_trim_current_device();  // the ingest / emit stages' cached scratch (k4_pool.h)
  for (hipEvent_t e : ix->ev0) hipEventDestroy(e);
  for (hipEvent_t e : ix->ev1) hipEventDestroy(e);
  for (hipEvent_t e : ix->ev2) hipEventDestroy(e);
  delete ix;
}

// ---- accessors ------------------------------------------------------------------------------------------------
extern "C" int k4_info(const k4_index* ix, k4_info_t* o) {
  if (!ix || !o) return K4_ERR_PARAMS;
  memset(o, 0, sizeof(*o));
  o->concat_len = ix->d.n;
  o->tot_seqs_len = ix->tot_seqs_len;
  o->sfx_el_size = ix->d.el;
  o->n_entries = ix->d.n_entries;
  o->kmer_k = ix->d.k;
  o->n_exc_blocks = ix->d.n_exc;
  o->device_bytes = ix->device_bytes;
  o->device = ix->device;
  o->max_iter = ix->d.max_iter;
  strncpy(o->dataset, ix->dataset.c_str(), 80);
  return K4_OK;
}

extern "C" int k4_get_entry(const k4_index* ix, uint32_t entry_id, k4_entry* out) {
  if (!ix || !out) return K4_ERR_PARAMS;
  if (entry_id < 1 || entry_id > ix->entries.size()) return K4_ERR_ENTRY;
  *out = ix->entries[entry_id - 1];
  return K4_OK;
}

extern "C" int k4_get_ident(const k4_index* ix, const char* name) {  // CSfxArray::GetIdent: case-insensitive
  if (!ix || !name) return K4_ERR_PARAMS;
  for (const k4_entry& e : ix->entries)
    if (!strcasecmp(e.name, name)) return (int)e.entry_id;
  return K4_ERR_ENTRY;
}

// quality character -> 4-bit score, exactly as CKAligner::LoadRawReads scales it (KAligner.cpp:12096-12158), the clamps of
// out-of-range characters included (Solexa's lower clamp is to 64 although its range starts at 59: the reference's)
extern "C" int k4_set_fastq_quality(k4_index* ix, int method) {
  if (!ix || method < 0 || method > 3) return K4_ERR_PARAMS;
  int rc = k4_check_hip(ix, hipSetDevice(ix->device), "hipSetDevice");
  if (rc != K4_OK) return rc;
  ix->q_method = method;
  if (method == 3) return K4_OK;
  uint8_t lut[256];
  for (int c = 0; c < 256; c++) {
    uint8_t q = (uint8_t)c, Qphred = 0;
    switch (method) {
      case 0:
        if (q < 33 || q >= 126) q = q < 33 ? 33 : 125;
        Qphred = q - 33;
        break;
      case 1:
        if (q < 64 || q >= 126) q = q < 64 ? 64 : 125;
        Qphred = q - 64;
        break;
      default:
        if (q < 59 || q >= 126) q = q < 64 ? 64 : 125;
        Qphred = q - 59;
        Qphred = (uint8_t)(10 * log(1 + pow(10.0, ((double)Qphred / 10.0) / log(10.0))));
        break;
    }
    if (Qphred > 40) Qphred = 40;
    lut[c] = (uint8_t)((((uint32_t)Qphred + 2) * 15) / 40);
  }
  if (!ix->d_qlut && (rc = k4_check_hip(ix, hipMalloc(&ix->d_qlut, 256), "hipMalloc(quality table)")) != K4_OK) return rc;
  return k4_check_hip(ix, hipMemcpy(ix->d_qlut, lut, 256, hipMemcpyHostToDevice), "hipMemcpy(quality table)");
}

extern "C" int k4_set_max_iter(k4_index* ix, int max_iter) {  // CSfxArray::SetMaxIter, SfxArray.cpp:1501
  if (!ix) return K4_ERR_PARAMS;
  int prev = ix->d.max_iter;
  ix->d.max_iter = max_iter > 0 ? max_iter : 0;
  return prev;
}

// test hook (not part of the ABI header): k-mer table entry {lb, pos0, sig | sub-bucket counts} of code c
extern "C" int k4i_debug_ktab(const k4_index* ix, uint64_t c, uint64_t* out) {
  if (!ix || !out) return K4_ERR_PARAMS;
  hipSetDevice(ix->device);
  if (ix->d.ktab64) {
    if (hipMemcpy(out, (const uint64_t*)ix->ktab + K4_KTAB_STRIDE64 * c, 16, hipMemcpyDeviceToHost) != hipSuccess) return K4_ERR_NO_DEVICE;
    out[2] = (out[0] >> 40) | ((out[1] >> 40) << 24);  // the sixteen 3-bit sub-bucket counts
    out[0] &= K4_KTAB64_MASK;
    out[1] &= K4_KTAB64_MASK;
  } else {
    uint32_t v[3];
    if (hipMemcpy(v, (const uint32_t*)ix->ktab + K4_KTAB_STRIDE32 * c, 12, hipMemcpyDeviceToHost) != hipSuccess) return K4_ERR_NO_DEVICE;
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
  }
  return K4_OK;
}

static int unpack_to_host(const k4_index* cix, uint64_t start, uint64_t len, uint8_t* out) {
  k4_index* ix = const_cast<k4_index*>(cix);
  hipSetDevice(ix->device);
  const uint64_t chunk = 64ull << 20;
  uint8_t* d = nullptr;
  K4_HIP(ix, hipMalloc(&d, std::min(chunk, len ? len : 1)));
  for (uint64_t o = 0; o < len; o += chunk) {
    uint64_t c = std::min(chunk, len - o);
    hipLaunchKernelGGL(k4k_unpack_range, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, 0, ix->d, start + o, c, d);
    hipError_t e = hipMemcpy(out + o, d, c, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      hipFree(d);
      return k4_check_hip(ix, e, "unpack copy");
    }
  }
  hipFree(d);
  return K4_OK;
}

// CSfxArray::GetSeq (SfxArray.h:996): returns the number of bases written (may be shorter), 0 on error
extern "C" int k4_get_seq(const k4_index* ix, uint32_t entry_id, uint32_t loci, uint8_t* out, uint32_t len) {
  if (!ix || !out || entry_id < 1 || entry_id > ix->entries.size()) return 0;
  const k4_entry& e = ix->entries[entry_id - 1];
  if (loci >= e.seq_len) return 0;
  if (loci + (uint64_t)len > e.seq_len) len = e.seq_len - loci;
  if (unpack_to_host(ix, e.start_ofs + loci, len, out) != K4_OK) return 0;
  return (int)len;
}

// CSfxArray::Flush2Disk (SfxArray.cpp:892): header, block (sequence + suffix array), entries
// tsSfxHeaderV3 (libkit4b/SfxArray.h:194-207, 1224 bytes, pack(4)) for an index that did not come from a file
static void make_header(const k4_index* ix, uint8_t* hdr) {
  const uint64_t n = ix->d.n;
  const uint32_t el = ix->d.el, ne = ix->d.n_entries;
  memset(hdr, 0, 1224);
  memcpy(hdr, "sfx5", 4);
  uint32_t ver = 5, attr = 0, nblocks = 1, entries_size = 8 + 111 * ne;
  uint64_t block_ofs = 1224, block_size = 20 + n + n * el, entries_ofs = block_ofs + block_size;
  uint64_t file_len = entries_ofs + entries_size;
  memcpy(hdr + 4, &ver, 4); memcpy(hdr + 8, &attr, 4); memcpy(hdr + 12, &file_len, 8);
  memcpy(hdr + 20, &entries_ofs, 8); memcpy(hdr + 28, &entries_size, 4); memcpy(hdr + 32, &nblocks, 4);
  memcpy(hdr + 36, &block_size, 8); memcpy(hdr + 44, &block_ofs, 8);
  strncpy((char*)hdr + 52, ix->dataset.c_str(), 80);
  strncpy((char*)hdr + 133, ix->description.empty() ? "k4sfx MI355X index" : ix->description.c_str(), 1023);
  strncpy((char*)hdr + 1157, ix->title.empty() ? "k4sfx" : ix->title.c_str(), 63);
}

// CSfxArray::SetDescription / SetTitle (SfxArray.h:552-553): the free-text header fields k4_write_sfx will write
extern "C" int k4_set_description(k4_index* ix, const char* description, const char* title) {
  if (!ix) return K4_ERR_PARAMS;
  if (description) ix->description = description;
  if (title) ix->title = title;
  ix->raw_header.clear();
  return K4_OK;
}

// CSfxArray::GetSfxHeader (SfxArray.h:551): the 1224-byte file header -- as read for an index opened from a file
extern "C" int k4_get_sfx_header(const k4_index* ix, void* out_1224) {
  if (!ix || !out_1224) return K4_ERR_PARAMS;
  if (ix->raw_header.size() == 1224) memcpy(out_1224, ix->raw_header.data(), 1224);
  else make_header(ix, (uint8_t*)out_1224);
  return K4_OK;
}

extern "C" int k4_write_sfx(const k4_index* cix, const char* path) {
  if (!cix || !path) return K4_ERR_PARAMS;
  k4_index* ix = const_cast<k4_index*>(cix);
  const uint64_t n = ix->d.n;
  const uint32_t el = ix->d.el, ne = ix->d.n_entries;
  FILE* fp = fopen(path, "wb");
  if (!fp) return k4_fail(ix, K4_ERR_CREATE_FILE, "unable to create %s", path);
  uint8_t hdr[1224];
  make_header(ix, hdr);
  fwrite(hdr, 1, sizeof(hdr), fp);
  uint8_t bh[20];
  uint32_t bid = 1;
  memcpy(bh, &bid, 4); memcpy(bh + 4, &ne, 4); memcpy(bh + 8, &n, 8); memcpy(bh + 16, &el, 4);
  fwrite(bh, 1, sizeof(bh), fp);
  const uint64_t chunk = 256ull << 20;
  std::vector<uint8_t> buf(std::min<uint64_t>(chunk, std::max<uint64_t>(n * el, 1)));
  for (uint64_t o = 0; o < n; o += chunk) {
    uint64_t c = std::min(chunk, n - o);
    int rc = unpack_to_host(ix, o, c, buf.data());
    if (rc != K4_OK) { fclose(fp); return rc; }
    fwrite(buf.data(), 1, c, fp);
  }
  hipSetDevice(ix->device);
  for (uint64_t o = 0; o < n * el; o += chunk) {
    uint64_t c = std::min(chunk, n * el - o);
    hipError_t e = hipMemcpy(buf.data(), ix->sa + o, c, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { fclose(fp); return k4_check_hip(ix, e, "sa copy"); }
    fwrite(buf.data(), 1, c, fp);
  }
  uint32_t nee[2] = {ne, ne};
  fwrite(nee, 4, 2, fp);
  for (uint32_t i = 0; i < ne; i++) {
    uint8_t e[111];
    const k4_entry& s = ix->entries[i];
    memset(e, 0, sizeof(e));
    memcpy(e, &s.entry_id, 4); memcpy(e + 4, &s.fblock_id, 4);
    strncpy((char*)e + 8, s.name, 80);
    memcpy(e + 89, &s.name_hash, 2); memcpy(e + 91, &s.seq_len, 4);
    memcpy(e + 95, &s.start_ofs, 8); memcpy(e + 103, &s.end_ofs, 8);
    fwrite(e, 1, sizeof(e), fp);
  }
  int bad = ferror(fp);
  fclose(fp);
  return bad ? k4_fail(ix, K4_ERR_FILE_ACCESS, "write to %s failed", path) : K4_OK;
}

// LocateCoredApprox parameter derivation (ngskit4b/KAligner.cpp:9367-9393)
extern "C" int k4_min_core_len(const k4_index* ix, int pmode, int* max_num_slides) {
  if (!ix) return K4_ERR_PARAMS;
  uint64_t tot = ix->tot_seqs_len;
  int autolen = 1;
  while (tot >>= 2) autolen++;
  autolen -= 1;
  int mcl = std::max(4, autolen);  // cKAMinCoreLen, KAligner.h:39
  int slides;
  switch (pmode) {
    case 2: mcl -= 2; slides = 9; break;
    case 1: mcl -= 1; slides = 8; break;
    case 0: slides = 8; break;
    default: mcl += 2; slides = 6; break;
  }
  if (max_num_slides) *max_num_slides = slides;
  return mcl;
}
