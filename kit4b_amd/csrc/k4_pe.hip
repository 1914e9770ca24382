// kit4b_amd/csrc/k4_pe.hip -- paired-end pass: mate rescue and the CKAligner pairing logic as kernels.
//
//   CSfxArray::AlignPairedRead    libkit4b/SfxArray.cpp:8571-8767 (linear-scan branch) + AdaptiveTrim :5561-5639
//   CKAligner::ProcCoredApprox    ngskit4b/KAligner.cpp:10160-10239 (multi x multi pair resolution)
//   CKAligner::ProcessPairedEnds  ngskit4b/KAligner.cpp:3159-3596, AcceptProvPE :2799, PEInsertSize :2875
#include <string.h>
#include <algorithm>
#include <vector>
#include "k4_device.h"
#include "k4_trim.h"

// ---- mate rescue: one wave scans every locus of the insert window ------------------------------------------------
// A locus is acceptable when the full-length mismatch count is <= ((len*rate)+99)/100, no mismatch sits in the first
// three or the last two bases (AdaptiveTrim :5622-5631 with MinFlankMatches 3) and the count is < rate + 1
// (PrevBestMaxChimericMMs :8684).  The reference keeps the first strictly better locus and stops at a 0-mismatch one,
// i.e. the lexicographic minimum of (mismatches, locus).
// One lane per locus: the window comes in with 16-byte loads (neighbouring lanes share its cache lines) and is XORed
// against the packed mate 32 bases at a time -- popcount for the mismatches, two masks for the flank rule; a window that
// touches N / a separator, or a mate that holds N, is compared symbol by symbol through the exact store instead.
// Windows of 1000 loci or more: the reference switches to exact seeds there (:8685-8726), but outside its chimeric mode it
// passes a seed length of 0 to IterateExactsRange, whose loop then never meets a mismatch and walks off the end of the suffix
// array -- `ngskit4b kalign -U1 -d100 -D1500` dies with SIGSEGV at the first rescue (known reference defects).
// There is no reference behaviour to reproduce: the linear-scan rule is applied to windows of any size.
// Wave-cooperative; rs = K4_RESCUE_LDS bytes of LDS owned by this wave (the oriented mate as symbols, then packed).
// Returns the AlignPairedRead result (1 placed, 0 not, < 0 error), identical in every lane; h is filled when 1.
#define K4_RESCUE_LDS (K4_MAX_READ_LEN + 8 * (K4_MAX_READ_LEN / 32 + 2))
// chimeric mode: one mismatch bit vector per lane behind that (AdaptiveTrim takes reads of up to 2048 bases: 64 words per lane)
#define K4_RESCUE_MK_WORDS 64
#define K4_RESCUE_LDS_CHIM (K4_RESCUE_LDS + 64 * 4 * K4_RESCUE_MK_WORDS)

// the mate (symbols rs[0, len), packed pk when it holds no N) against the reference at g: mismatch bits into the lane's column
K4_DEV void k4d_rescue_mm_vector(const K4DevIndex& ix, const uint8_t* rs, const uint64_t* pk, bool packed, int len, uint64_t g, uint32_t* mk) {
  if (packed && g + (uint64_t)len <= ix.n && !k4d_any_exc(ix, (int64_t)g, (int64_t)g + len)) {
    const int al = (int)(g & 15);
    for (int c0 = 0; 32 * c0 < len; c0 += 4) {
      const int rem = len - 32 * c0;
      uint64_t rc[4];
      k4d_ref_chunks4(ix, (int64_t)g, c0, rem + al <= 128, rc);
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (32 * c < rem) mk[(c0 + c) * 64] = k4d_mm_bits((rc[c] ^ pk[c0 + c]) & k4d_range_mask(0, rem - 32 * c));
    }
    return;
  }
  K4Tb t;
  t.init(ix);
  for (int c = 0; 32 * c < len; c++) {
    uint32_t m = 0;
    for (int q = 0; q < 32 && 32 * c + q < len; q++)
      if ((uint32_t)rs[32 * c + q] != t.get((int64_t)g + 32 * c + q)) m |= 1u << q;
    mk[c * 64] = m;
  }
}
// CmpProbeTarg (SfxArray.cpp:2508-2525) of the core rs[ofs, ofs + cl) against the suffix at pos, on exact symbols
K4_DEV int k4d_rescue_cmp(const K4DevIndex& ix, const uint8_t* rs, int ofs, int cl, uint64_t pos) {
  K4Tb t;
  t.init(ix);
  for (int j = 0; j < cl; j++) {
    const uint32_t e2 = t.get((int64_t)pos + j);
    if (e2 == 7u) return -1;
    const uint32_t e1 = rs[ofs + j];
    if (e1 > e2) return 1;
    if (e1 < e2) return -1;
  }
  return 0;
}
K4_DEV uint64_t k4d_rescue_sa(const K4DevIndex& ix, uint64_t i) { return ix.el == 4 ? k4d_sa_at<4>(ix, i) : k4d_sa_at<5>(ix, i); }
K4_DEV int k4d_mate_rescue(const K4DevIndex& ix, const k4_rescue_task& tk, const uint8_t* __restrict__ reads, int lane,
                           uint8_t* rs, k4_hit& h, uint32_t* mk = nullptr /* chimeric mode: this lane's column of the mismatch vectors */) {
  int res = 0;
  unsigned long long best = ~0ull;  // (mm << 32) | (locus - sp)
  uint32_t sp = 0, ep = 0;
  const int len = (int)tk.read_len;
  bool run = false;
  if (tk.chrom_id >= 1 && tk.chrom_id <= ix.n_entries && len >= 1 && len <= K4_MAX_READ_LEN) {
    const uint64_t cs = ix.ent_start[tk.chrom_id - 1];
    const uint32_t chrom_len = (uint32_t)(ix.ent_end[tk.chrom_id - 1] - cs + 1);
    int min_ins = tk.min_insert, max_ins = tk.max_insert;
    if (tk.start_loci >= tk.end_loci || tk.end_loci >= chrom_len) res = -1;
    else if (min_ins > max_ins) res = 0;
    else {
      if (min_ins < len) { max_ins += len - min_ins; min_ins = len; }
      run = true;
      if (tk.b3prime_extend) {
        if ((uint32_t)(tk.start_loci + min_ins) >= chrom_len) run = false;
        else {
          sp = tk.start_loci + min_ins - len;
          ep = min(chrom_len - (uint32_t)len, (uint32_t)(tk.start_loci + max_ins - len));
        }
      } else {
        if (tk.end_loci < (uint32_t)min_ins) run = false;
        else {
          sp = tk.end_loci <= (uint32_t)max_ins ? 0 : tk.end_loci - max_ins;
          ep = tk.end_loci - min_ins;
        }
      }
      // AdaptiveTrim parameter validation (:5598-5603): failing it means no locus is ever accepted
      if (run && (len < 25 || len > 2048 || (uint32_t)tk.max_allowed_mm > (uint32_t)((15 * len + 99) / 100))) run = false;
    }
    // chimeric mode (:8610-8621): the placement may keep only part of the mate
    int min_put_len = len, core_len = (int)((tk.chimeric >> 8) & 0xFFF), core_delta = (int)(tk.chimeric >> 20);
    {
      const int mcl = (int)(tk.chimeric & 0xFF);
      if (core_len > 0 && mcl >= 15 && mcl <= 99) {
        min_put_len = (len * mcl + 50) / 100;
        if (core_len > min_put_len) core_len = min_put_len;
      }
      if (min_put_len == len) core_len = 0;
    }
    if (run && min_put_len != len) {
      if (!mk) { res = K4_ERR_PARAMS; run = false; }
      if (min_put_len < 15) run = false;  // (AdaptiveTrim refuses MinTrimLen below cMinATTrimmedLen: nothing is ever accepted)
    }
    if (run && min_put_len != len) {
      uint64_t* pk = reinterpret_cast<uint64_t*>(rs + K4_MAX_READ_LEN);
      __syncthreads();
      const uint8_t* src = reads + tk.read_off;
      for (int q = lane; q < len; q += 64) {
        uint8_t b = tk.antisense ? src[len - 1 - q] & 7 : src[q] & 7;
        if (tk.antisense && b <= 3) b = 3 - b;
        rs[q] = b;
      }
      __syncthreads();
      const int nw = (len + 31) >> 5;
      bool has_n = false;
      for (int w = lane; w < nw; w += 64) {
        uint64_t acc = 0;
        for (int q = 0; q < 32; q++) {
          const int j = 32 * w + q;
          uint32_t b = j < len ? rs[j] : 0u;
          if (b > 3) { has_n = true; b = 0; }
          acc = (acc << 2) | b;
        }
        pk[w] = acc;
      }
      const bool packed = __ballot(has_n) == 0;
      __syncthreads();
      // The reference takes the candidates one after the other and hands AdaptiveTrim the length of the best placement so far
      // as its minimum (which changes what it returns, not only whether it accepts): 64 candidates are tried against the
      // current state at once, the first one the state admits is taken, and the ones behind it are tried again.
      int cur_min = min_put_len;
      uint32_t prev_best = (uint32_t)tk.max_allowed_mm + 1;
      uint32_t b_loci = 0, b_t5 = 0, b_t3 = 0, b_mm = 0;
      bool have = false, done = false;
      if ((ep - sp) >= 1000) {  // :8685-8726 seeded with exact cores (IterateExactsRange, :3461-3553)
        for (int ofs = 0; core_len > 0 && core_delta > 0 && ofs + core_len <= len && !done; ofs += core_delta) {
          // lowest suffix that is not below the core: 64 evenly spaced pivots per round over the whole array
          int64_t lo = 0, hi = (int64_t)ix.n - 1, first = -1;
          while (lo <= hi) {
            const int64_t step = (hi - lo + 1 + 63) / 64;
            const int64_t pv = lo + (int64_t)lane * step;
            const bool hv = pv <= hi;
            int c = 1;
            if (hv) c = k4d_rescue_cmp(ix, rs, ofs, core_len, k4d_rescue_sa(ix, (uint64_t)pv));
            const unsigned long long hm = __ballot(hv), le = __ballot(hv && c <= 0);
            if (!le) { lo = lo + (int64_t)(__popcll(hm) - 1) * step + 1; continue; }
            const int f = __ffsll((long long)le) - 1;
            const int64_t pvf = lo + (int64_t)f * step;
            if (step == 1) { if (__shfl(c, f, 64) == 0) first = pvf; break; }
            if (f > 0) lo = lo + (int64_t)(f - 1) * step + 1;
            hi = pvf;
          }
          if (first < 0) continue;
          for (int64_t base = first; base < (int64_t)ix.n;) {
            const int64_t idx = base + lane;
            uint64_t pos = 0;
            bool same = false;
            if (idx < (int64_t)ix.n) {
              pos = k4d_rescue_sa(ix, (uint64_t)idx);
              same = k4d_rescue_cmp(ix, rs, ofs, core_len, pos) == 0;
            }
            const unsigned long long stop_m = __ballot(!same);
            const int n_run = stop_m ? __ffsll((long long)stop_m) - 1 : 64;  // lanes [0, n_run) still start with the core
            bool acc = false;
            K4Trim tr = {0, 0, 0, 0};
            uint32_t loci = 0;
            if (lane < n_run) {
              const int e = k4d_map_entry(ix, pos);
              if (e == (int)tk.chrom_id - 1) {
                const uint32_t hit_loci = (uint32_t)(pos - cs);
                if (hit_loci >= sp && hit_loci <= ep && (uint32_t)ofs <= hit_loci && hit_loci + (uint32_t)len - (uint32_t)ofs < chrom_len) {
                  loci = hit_loci - (uint32_t)ofs;
                  k4d_rescue_mm_vector(ix, rs, pk, packed, len, cs + loci, mk);
                  tr = k4d_adaptive_trim(mk, len, cur_min, tk.max_allowed_mm, 3);
                  acc = tr.len > cur_min || (tr.len == cur_min && (uint32_t)tr.mms < prev_best);
                }
              }
            }
            const unsigned long long am = __ballot(acc);
            if (am) {
              const int c = __ffsll((long long)am) - 1;
              cur_min = __shfl(tr.len, c, 64); prev_best = (uint32_t)__shfl(tr.mms, c, 64);
              b_loci = (uint32_t)__shfl((int)loci, c, 64); b_t5 = (uint32_t)__shfl(tr.t5, c, 64); b_t3 = (uint32_t)__shfl(tr.t3, c, 64);
              b_mm = prev_best;
              have = true;
              base += c + 1;
              continue;
            }
            if (n_run < 64) break;
            base += 64;
          }
        }
      } else {  // :8731-8766 every locus of the window in turn
        for (uint32_t start = sp; start <= ep && !done;) {
          const uint32_t loci = start + (uint32_t)lane;
          bool acc = false;
          K4Trim tr = {0, 0, 0, 0};
          if (loci <= ep && loci >= start) {
            k4d_rescue_mm_vector(ix, rs, pk, packed, len, cs + loci, mk);
            tr = k4d_adaptive_trim(mk, len, cur_min, tk.max_allowed_mm, 3);
            acc = tr.len > cur_min || (tr.len == cur_min && (uint32_t)tr.mms < prev_best);
          }
          const unsigned long long am = __ballot(acc);
          if (am) {
            const int c = __ffsll((long long)am) - 1;
            cur_min = __shfl(tr.len, c, 64); prev_best = (uint32_t)__shfl(tr.mms, c, 64);
            b_loci = start + (uint32_t)c; b_t5 = (uint32_t)__shfl(tr.t5, c, 64); b_t3 = (uint32_t)__shfl(tr.t3, c, 64);
            b_mm = prev_best;
            have = true;
            if (cur_min == len && prev_best == 0) done = true;
            start += (uint32_t)c + 1;
            if (start == 0) break;  // (wrapped)
            continue;
          }
          if (ep - start < 64) break;
          start += 64;
        }
      }
      memset(&h, 0, sizeof(h));
      if (have && prev_best <= (uint32_t)tk.max_allowed_mm) {
        h.chrom_id = tk.chrom_id;
        h.match_loci = b_loci;
        h.match_len = (uint16_t)len;
        h.strand = tk.antisense ? '-' : '+';
        h.mismatches = (uint8_t)b_mm;
        const uint32_t tl = tk.antisense ? b_t3 : b_t5, trr = tk.antisense ? b_t5 : b_t3;  // :8702-8711
        h.ext = (tl & 0xFFFu) | ((trr & 0xFFFu) << 12) | (cur_min != len ? K4_EXT_CHIMERIC : 0u);
        return 1;
      }
      return 0;
    }
    if (run) {
      uint64_t* pk = reinterpret_cast<uint64_t*>(rs + K4_MAX_READ_LEN);
      __syncthreads();  // (blocks are one wave)
      const uint8_t* src = reads + tk.read_off;
      for (int q = lane; q < len; q += 64) {
        uint8_t b = tk.antisense ? src[len - 1 - q] & 7 : src[q] & 7;
        if (tk.antisense && b <= 3) b = 3 - b;
        rs[q] = b;
      }
      __syncthreads();
      const int nw = (len + 31) >> 5;
      bool has_n = false;
      for (int w = lane; w < nw; w += 64) {
        uint64_t acc = 0;
        for (int q = 0; q < 32; q++) {
          const int j = 32 * w + q;
          uint32_t b = j < len ? rs[j] : 0u;
          if (b > 3) { has_n = true; b = 0; }
          acc = (acc << 2) | b;
        }
        pk[w] = acc;
      }
      const bool packed = __ballot(has_n) == 0;
      __syncthreads();
      const uint32_t max_allowed = ((uint32_t)len * (uint32_t)tk.max_allowed_mm + 99) / 100;
      for (uint32_t loci = sp + lane; loci <= ep; loci += 64) {
        const uint64_t g = cs + loci;
        uint32_t mm = 0;
        bool ok = true;
        if (packed && !k4d_any_exc(ix, (int64_t)g, (int64_t)g + len)) {
          const int al = (int)(g & 15);
          uint64_t flank = 0;
          for (int c0 = 0; 32 * c0 < len; c0 += 4) {
            const int rem = len - 32 * c0;
            uint64_t rc[4];
            k4d_ref_chunks4(ix, (int64_t)g, c0, rem + al <= 128, rc);
#pragma unroll
            for (int c = 0; c < 4; c++)
              if (32 * c < rem) {
                const int base = 32 * (c0 + c);
                const uint64_t x = (rc[c] ^ pk[c0 + c]) & k4d_range_mask(0, rem - 32 * c);
                mm += k4d_mm_count(x);
                if (base < 3) flank |= x & k4d_range_mask(0, 3 - base);                                   // read bases 0..2
                if (len - 2 - base < 32 && len - base > 0) flank |= x & k4d_range_mask(len - 2 - base, len - base);  // the last two
              }
          }
          ok = flank == 0 && mm <= max_allowed;
        } else {
          for (int o = 0; o < len; o++) {
            if (rs[o] != k4d_ref_base(ix, g + o)) {
              if (++mm > max_allowed) { ok = false; break; }
              if (o < 3 || (len - o) < 3) { ok = false; break; }
            }
          }
        }
        if (ok && mm <= (uint32_t)tk.max_allowed_mm) {
          const unsigned long long v = ((unsigned long long)mm << 32) | (loci - sp);
          best = v < best ? v : best;
        }
      }
    }
  } else
    res = -1;
  for (int d = 32; d > 0; d >>= 1) {
    const unsigned long long o = __shfl_xor(best, d, 64);
    best = o < best ? o : best;
  }
  memset(&h, 0, sizeof(h));
  if (run && best != ~0ull) {
    res = 1;
    h.chrom_id = tk.chrom_id;
    h.match_loci = sp + (uint32_t)(best & 0xFFFFFFFFu);
    h.match_len = (uint16_t)len;
    h.strand = tk.antisense ? '-' : '+';
    h.mismatches = (uint8_t)(best >> 32);
  }
  return res;
}

__global__ void __launch_bounds__(64) k4k_mate_rescue(K4DevIndex ix, const k4_rescue_task* __restrict__ tasks,
                                                      const uint8_t* __restrict__ reads, int64_t n_tasks,
                                                      int32_t* __restrict__ rslt, k4_hit* __restrict__ hits, int chim) {
  extern __shared__ __attribute__((aligned(8))) uint8_t rs[];  // the mate, oriented as it must align; with `chim`, the mismatch vectors behind it
  const int lane = threadIdx.x;
  uint32_t* mk = chim ? reinterpret_cast<uint32_t*>(rs + K4_RESCUE_LDS) + lane : nullptr;
  for (int64_t t = blockIdx.x; t < n_tasks; t += gridDim.x) {
    const k4_rescue_task tk = tasks[t];
    k4_hit h;
    const int res = k4d_mate_rescue(ix, tk, reads, lane, rs, h, mk);
    if (lane == 0) {
      rslt[t] = res;
      hits[t] = h;
    }
  }
}

extern "C" int k4_mate_rescue_batch(k4_index* ix, int64_t n, const k4_rescue_task* tasks, const uint8_t* reads,
                                    uint64_t reads_bytes, int32_t* rslt, k4_hit* hits) {
  if (!ix || n < 0 || (n > 0 && (!tasks || !reads || !rslt || !hits))) return K4_ERR_PARAMS;
  if (n == 0) return K4_OK;
  for (int64_t i = 0; i < n; i++)
    if (tasks[i].read_off + tasks[i].read_len > reads_bytes) return k4_fail(ix, K4_ERR_PARAMS, "rescue task %lld outside the read buffer", (long long)i);
  K4_HIP(ix, hipSetDevice(ix->device));
  // staging buffers live with the index and only grow (the facade's AlignPairedRead comes here once per orphan read)
  if ((size_t)n > ix->rs_cap_tasks) {
    for (void* p : {(void*)ix->rs_tasks, (void*)ix->rs_res, (void*)ix->rs_hits})
      if (p) hipFree(p);
    ix->rs_tasks = nullptr; ix->rs_res = nullptr; ix->rs_hits = nullptr; ix->rs_cap_tasks = 0;
    const size_t cap = std::max<size_t>((size_t)n, 256);
    K4_HIP(ix, hipMalloc(&ix->rs_tasks, cap * sizeof(k4_rescue_task)));
    K4_HIP(ix, hipMalloc(&ix->rs_res, cap * 4));
    K4_HIP(ix, hipMalloc(&ix->rs_hits, cap * sizeof(k4_hit)));
    ix->rs_cap_tasks = cap;
  }
  if (reads_bytes + 16 > ix->rs_cap_reads) {
    if (ix->rs_reads) hipFree(ix->rs_reads);
    ix->rs_reads = nullptr; ix->rs_cap_reads = 0;
    const size_t cap = std::max<size_t>(reads_bytes + 16, 1 << 16);
    K4_HIP(ix, hipMalloc(&ix->rs_reads, cap));
    ix->rs_cap_reads = cap;
  }
  k4_rescue_task* d_t = (k4_rescue_task*)ix->rs_tasks;
  uint8_t* d_r = (uint8_t*)ix->rs_reads;
  int32_t* d_res = (int32_t*)ix->rs_res;
  k4_hit* d_h = (k4_hit*)ix->rs_hits;
  hipStream_t st = ix->stream;
  hipMemcpyAsync(d_t, tasks, (size_t)n * sizeof(k4_rescue_task), hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_r, reads, reads_bytes, hipMemcpyHostToDevice, st);
  unsigned grid = (unsigned)std::min<int64_t>(n, 256 * 32);
  int chim = 0;
  for (int64_t i = 0; i < n; i++) chim |= tasks[i].chimeric != 0;
  hipLaunchKernelGGL(k4k_mate_rescue, dim3(grid), dim3(64), chim ? K4_RESCUE_LDS_CHIM : K4_RESCUE_LDS, st, ix->d, d_t, d_r, n, d_res, d_h, chim);
  hipMemcpyAsync(rslt, d_res, (size_t)n * 4, hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(hits, d_h, (size_t)n * sizeof(k4_hit), hipMemcpyDeviceToHost, st);
  int rc = k4_check_hip(ix, hipStreamSynchronize(st), "mate rescue");
  if (rc != K4_OK) return rc;
  return K4_OK;
}

// ---- the pairing logic (device) -------------------------------------------------------------------------------------
K4_DEV int k4d_pe_insert_size(const k4_pe_params& pe, uint8_t s1, uint32_t st1, uint32_t en1, uint8_t s2, uint32_t st2,
                              uint32_t en2) {  // PEInsertSize, KAligner.cpp:2875-2918
  if ((pe.pair_strand && s1 != s2) || (!pe.pair_strand && s1 == s2)) return -1;
  int frag = (int)(1 + max(en1, en2) - min(st1, st2));
  if (frag < 0) return -1;
  if (frag < pe.pair_min_len) return -6;
  if (frag > pe.pair_max_len) return -7;
  return frag;
}
// AdjStartLoci / AdjEndLoci (KAligner.cpp:1633-1650): a chimeric hit counts from / to its trimmed ends
K4_DEV uint32_t k4d_pe_adj_start(const k4_hit& h) { return h.match_loci + (h.strand == '+' ? K4_HIT_TRIM_LEFT(h) : K4_HIT_TRIM_RIGHT(h)); }
K4_DEV uint32_t k4d_pe_adj_end(const k4_hit& h) {
  return h.match_loci + ((uint32_t)h.match_len - (h.strand == '+' ? K4_HIT_TRIM_RIGHT(h) : K4_HIT_TRIM_LEFT(h)) - 1);
}
K4_DEV int k4d_accept_prov_pe(const k4_pe_params& pe, int nh1, const k4_hit& h1, int nh2, const k4_hit& h2) {  // :2799-2861
  if (!(nh1 == 1 && nh2 == 1)) return 0;
  if (h1.chrom_id != h2.chrom_id) return -2;
  return k4d_pe_insert_size(pe, h1.strand, k4d_pe_adj_start(h1), k4d_pe_adj_end(h1), h2.strand, k4d_pe_adj_start(h2), k4d_pe_adj_end(h2));
}
struct K4PeChim { int min_chimeric_len, min_core_len, slides_per100, min_edit_dist; };  // AlignPairedRead's chimeric mode (0: off)
K4_DEV bool k4d_pe_unaligned(int nar) { return nar == K4_NAR_NS || nar == K4_NAR_NOHIT || nar == K4_NAR_UNALIGNED; }

// what is left of an orphan pair that could not be accepted as a pair (KAligner.cpp:3538-3585)
K4_DEV void k4d_pe_leftover(const k4_pe_params& pe, k4_pe_read& f, k4_pe_read& r) {
  if (!(pe.pe_mode == 3 || pe.pe_mode == 4)) {
    f.num_hits = r.num_hits = 0; f.inst = r.inst = 0;
    if (f.nar == K4_NAR_ACCEPTED) f.nar = K4_NAR_PENOHIT;
    if (r.nar == K4_NAR_ACCEPTED) r.nar = K4_NAR_PENOHIT;
    return;
  }
  if (f.num_hits != 1) { f.num_hits = 0; f.inst = 0; if (f.nar == K4_NAR_ACCEPTED) f.nar = K4_NAR_PEUNALIGN; }
  else f.nar = K4_NAR_ACCEPTED;
  if (r.num_hits != 1) { r.num_hits = 0; r.inst = 0; if (r.nar == K4_NAR_ACCEPTED) r.nar = K4_NAR_PEUNALIGN; }
  else r.nar = K4_NAR_ACCEPTED;
}

// One thread per pair: AlignRead's PE view of each end, ProcCoredApprox's multi x multi resolution (KAligner.cpp:10185-10239)
// and ProcessPairedEnds (:3207-3318) up to the point where a mate rescue is needed; pairs that need one are listed.
__global__ void __launch_bounds__(256) k4k_pe_pair(k4_pe_params pe, int64_t n_pairs, int mh, const k4_read_result* __restrict__ rr,
                                                   const k4_hit* __restrict__ hits, k4_pe_read* __restrict__ out,
                                                   uint32_t* __restrict__ orphans, uint32_t* __restrict__ ctl) {
  // The block's 256 pairs read 12 KB of results and write 20 KB of records, both contiguous: they pass through LDS in 16-byte
  // pieces, a lane per piece (a thread copying its own 48 / 80 bytes touches forty lines per wave instruction: 4.4 ms per 50 M
  // pairs before, half the step time of a paired-end batch's last phase)
  __shared__ uint4 stage[256 * 5];
  static_assert(sizeof(k4_read_result) == 24 && sizeof(k4_pe_read) == 40, "staging sizes");
  const int64_t i0 = (int64_t)blockIdx.x * 256;
  const int n_here = (int)min((int64_t)256, n_pairs - i0);
  // (pair i0 starts 48 i0 / 80 i0 bytes into the arrays: 16-byte aligned whenever the arrays are -- a caller's oddly placed
  // buffer takes the plain copies)
  const bool al = ((reinterpret_cast<uintptr_t>(rr) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  const int64_t i = i0 + threadIdx.x;
  const bool live = threadIdx.x < n_here;
  k4_read_result q1, q2;
  memset(&q1, 0, sizeof(q1));
  memset(&q2, 0, sizeof(q2));
  if (al) {
    const uint4* src = reinterpret_cast<const uint4*>(rr + 2 * i0);
    for (int q = threadIdx.x; q < n_here * 3; q += 256) stage[q] = src[q];
    __syncthreads();
    if (live) {
      const k4_read_result* l = reinterpret_cast<const k4_read_result*>(stage) + 2 * threadIdx.x;
      q1 = l[0]; q2 = l[1];
    }
    __syncthreads();  // (the same LDS takes the records below)
  } else if (live) {
    q1 = rr[2 * i]; q2 = rr[2 * i + 1];
  }
  const k4_hit* h1 = hits + (size_t)(2 * (live ? i : i0)) * mh;
  const k4_hit* h2 = hits + (size_t)(2 * (live ? i : i0) + 1) * mh;
  k4_pe_read f, r;
  memset(&f, 0, sizeof(f));
  memset(&r, 0, sizeof(r));
  f.nar = q1.nar; f.num_hits = q1.num_hits; f.inst = q1.inst; f.low_mm = q1.low_mm;
  r.nar = q2.nar; r.num_hits = q2.num_hits; r.inst = q2.inst; r.low_mm = q2.low_mm;
  if (q1.nar == K4_NAR_ACCEPTED) f.hit = h1[0];
  if (q2.nar == K4_NAR_ACCEPTED) r.hit = h2[0];
  if (q1.hit_rslt == K4_HR_HITS && q2.hit_rslt == K4_HR_HITS && !(f.inst == 1 && r.inst == 1) && f.inst < 10 && r.inst < 10) {
    bool multi = false, accepted = false;
    k4_hit p1, p2;
    memset(&p1, 0, sizeof(p1));
    memset(&p2, 0, sizeof(p2));
    for (int a = 0; !(multi && !accepted) && a < f.inst; a++)
      for (int b = 0; b < r.inst; b++) {
        const k4_hit ha = h1[a], hb = h2[b];
        if (k4d_accept_prov_pe(pe, 1, ha, 1, hb) > 0) {
          if (!multi) { p1 = ha; p2 = hb; multi = true; accepted = true; }
          else { accepted = false; break; }
        }
      }
    if (accepted) {
      f.hit = p1; f.nar = K4_NAR_ACCEPTED; f.num_hits = 1;
      r.hit = p2; r.nar = K4_NAR_ACCEPTED; r.num_hits = 1;
    }
  }
  bool orphan = false;
  if (f.nar == K4_NAR_ACCEPTED || r.nar == K4_NAR_ACCEPTED) {
    bool strict_fail = pe.pe_mode == 2 && (k4d_pe_unaligned(f.nar) || k4d_pe_unaligned(r.nar));
    bool paired = false;
    if (!strict_fail && f.nar == K4_NAR_ACCEPTED && r.nar == K4_NAR_ACCEPTED) {
      const int frag = k4d_accept_prov_pe(pe, f.num_hits, f.hit, r.num_hits, r.hit);
      if (frag > 0) { f.pe_aligned = r.pe_aligned = 1; paired = true; }
      else {
        switch (frag) {
          case -1: f.nar = r.nar = K4_NAR_PESTRAND; break;
          case -2: f.nar = r.nar = K4_NAR_PECHROM; break;
          case -6: f.nar = r.nar = K4_NAR_PEINSERTMIN; break;
          case -7: f.nar = r.nar = K4_NAR_PEINSERTMAX; break;
          default: break;
        }
        if (pe.pe_mode == 2) strict_fail = true;
      }
    }
    if (!paired) {
      if (strict_fail) {
        f.num_hits = r.num_hits = 0; f.inst = r.inst = 0;
        if (f.nar == K4_NAR_ACCEPTED) f.nar = K4_NAR_PENOHIT;
        if (r.nar == K4_NAR_ACCEPTED) r.nar = K4_NAR_PENOHIT;
      } else if (pe.pe_mode == 1 || pe.pe_mode == 3) orphan = true;  // orphan recovery follows
      else k4d_pe_leftover(pe, f, r);
    }
  }
  if (al) {
    if (live) {
      k4_pe_read* l = reinterpret_cast<k4_pe_read*>(stage) + 2 * threadIdx.x;
      l[0] = f; l[1] = r;
    }
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(out + 2 * i0);
    for (int q = threadIdx.x; q < n_here * 5; q += 256) dst[q] = stage[q];
  } else if (live) {
    out[2 * i] = f;
    out[2 * i + 1] = r;
  }
  if (live && orphan) orphans[atomicAdd(&ctl[0], 1u)] = (uint32_t)i;
}

// One wave per orphan pair (modes 1 and 3): first the PE1 alignment as anchor (:3320-3418), then PE2 (:3424-3535),
// then the leftover rule.  Every lane carries the same copy of the two records; lane 0 writes them back.
__global__ void __launch_bounds__(64) k4k_pe_orphans(K4DevIndex ix, k4_pe_params pe, int max_subs, const uint8_t* __restrict__ reads,
                                                     const uint64_t* __restrict__ offs, const uint32_t* __restrict__ lens,
                                                     const uint32_t* __restrict__ orphans, k4_pe_read* __restrict__ out,
                                                     uint32_t* __restrict__ ctl, K4PeChim ch) {
  extern __shared__ __attribute__((aligned(8))) uint8_t rs[];
  const int lane = threadIdx.x;
  uint32_t* mk = ch.min_chimeric_len > 0 ? reinterpret_cast<uint32_t*>(rs + K4_RESCUE_LDS) + lane : nullptr;
  const uint32_t n_orph = ctl[0];
  for (uint32_t t = blockIdx.x; t < n_orph; t += gridDim.x) {
    const int64_t i = orphans[t];
    k4_pe_read f = out[2 * i], r = out[2 * i + 1];
    bool done = false;
    for (int round = 0; round < 2 && !done; round++) {
      const k4_pe_read& anchor = round == 0 ? f : r;
      const bool mate_unal = k4d_pe_unaligned(round == 0 ? r.nar : f.nar);
      if (!(anchor.num_hits == 1 && !mate_unal)) continue;
      const bool plus = anchor.hit.strand == '+';
      k4_rescue_task tk;
      memset(&tk, 0, sizeof(tk));
      if (round == 0) { tk.b3prime_extend = plus; tk.antisense = pe.pair_strand ? !plus : plus; }
      else {
        tk.b3prime_extend = plus; tk.antisense = plus;
        if (pe.pair_strand) { tk.b3prime_extend = !tk.b3prime_extend; tk.antisense = !tk.antisense; }
      }
      tk.chrom_id = anchor.hit.chrom_id;
      tk.start_loci = k4d_pe_adj_start(anchor.hit);  // OrphStartLoci / OrphEndLoci (:3354-3355)
      tk.end_loci = k4d_pe_adj_end(anchor.hit);
      const int64_t mate = round == 0 ? 2 * i + 1 : 2 * i;
      tk.read_len = lens[mate];
      tk.read_off = offs[mate];
      tk.min_insert = pe.pair_min_len;
      tk.max_insert = pe.pair_max_len;
      tk.max_allowed_mm = max_subs;  // the per-100 bp rate, as the reference passes it (KAligner.cpp:3379, Q15)
      if (ch.min_chimeric_len > 0) {   // the window core CKAligner derives for this mate (:3357-3370)
        const int pl = (int)tk.read_len;
        int tot_mm = max_subs == 0 ? 0 : max(1, (int)(0.5 + ((pl - 1) * max_subs) / 100.0));
        tot_mm = min(tot_mm, 63);
        const int cl = max(ch.min_core_len, pl / (ch.min_edit_dist == 1 ? tot_mm + 1 : tot_mm + 2));
        const int cd = max(pl / ch.slides_per100 - 1, cl);
        tk.chimeric = (uint32_t)ch.min_chimeric_len | ((uint32_t)min(cl, 4095) << 8) | ((uint32_t)min(cd, 4095) << 20);
      }
      k4_hit h;
      const int res = k4d_mate_rescue(ix, tk, reads, lane, rs, h, mk);
      if (res != 1) continue;
      const uint32_t hs = k4d_pe_adj_start(h), he = k4d_pe_adj_end(h);  // (:3390, :3492)
      const uint32_t as = tk.start_loci, ae = tk.end_loci;
      const int frag = round == 0 ? k4d_pe_insert_size(pe, anchor.hit.strand, as, ae, h.strand, hs, he)
                                  : k4d_pe_insert_size(pe, h.strand, hs, he, anchor.hit.strand, as, ae);
      if (frag <= 0) continue;
      k4_pe_read& m = round == 0 ? r : f;
      m.hit = h; m.num_hits = 1; m.low_mm = h.mismatches; m.inst = 1; m.rescued = 1;
      f.pe_aligned = r.pe_aligned = 1;
      f.nar = r.nar = K4_NAR_ACCEPTED;
      done = true;
    }
    if (!done) k4d_pe_leftover(pe, f, r);
    if (lane == 0) {
      out[2 * i] = f;
      out[2 * i + 1] = r;
    }
  }
}

static int pe_reserve(k4_index* ix, int64_t n_pairs, int mh) {
  if (n_pairs <= ix->pe_cap_pairs && mh <= ix->pe_cap_hits) return K4_OK;
  const int64_t cap = std::max(n_pairs, ix->pe_cap_pairs);
  const int h = std::max(mh, ix->pe_cap_hits);
  for (void* p : {(void*)ix->pe_rr, (void*)ix->pe_hits, (void*)ix->pe_list, (void*)ix->pe_ctl})
    if (p) hipFree(p);
  ix->pe_rr = nullptr; ix->pe_hits = nullptr; ix->pe_list = nullptr; ix->pe_ctl = nullptr;
  ix->pe_cap_pairs = 0;
  K4_HIP(ix, hipMalloc(&ix->pe_rr, (size_t)(2 * cap + 1) * sizeof(k4_read_result)));
  K4_HIP(ix, hipMalloc(&ix->pe_hits, (size_t)(2 * cap + 1) * h * sizeof(k4_hit)));
  K4_HIP(ix, hipMalloc(&ix->pe_list, (size_t)(cap + 1) * 4));
  K4_HIP(ix, hipMalloc(&ix->pe_ctl, 16));
  ix->pe_cap_pairs = cap;
  ix->pe_cap_hits = h;
  return K4_OK;
}

// Device buffers in, device records out; reads are interleaved (read 2i = PE1 of pair i, read 2i+1 = its PE2).
// Enqueues the SE pass over the 2n ends, the pairing kernel and the orphan kernel on `stream`, then waits for the
// stream (the insert-window error of the rescue can only be reported after the fact).
extern "C" int k4_kalign_pe_batch_dev(k4_index* ix, const k4_kalign_params* p, const k4_pe_params* pe_in, int64_t n_pairs,
                                      int32_t max_read_len, const void* d_reads, const void* d_offs, const void* d_lens,
                                      void* d_out, void* stream) {
  if (!ix || !p || !pe_in) return K4_ERR_PARAMS;
  if (n_pairs < 0 || (n_pairs > 0 && (!d_reads || !d_offs || !d_lens || !d_out))) return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  const k4_pe_params pe = *pe_in;
  if (pe.pe_mode < 1 || pe.pe_mode > 4 || pe.pair_min_len < 1 || pe.pair_max_len < pe.pair_min_len)
    return k4_fail(ix, K4_ERR_PARAMS, "PE parameters out of range");
  if (n_pairs == 0) return K4_OK;
  if (n_pairs >= 0x7FFFFFF0ll) return k4_fail(ix, K4_ERR_PARAMS, "at most 2^31-16 pairs per batch");
  K4_HIP(ix, hipSetDevice(ix->device));
  // both ends as SE reads with MaxHits = max(m_MaxMLmatches, cMaxMLPEmatches) and the PE classification
  k4_kalign_params kp = *p;
  kp.pe_mode = 1;
  kp.max_ml = std::max(p->max_ml, 10);
  const int mh = kp.max_ml;
  int rc = pe_reserve(ix, n_pairs, mh);
  if (rc != K4_OK) return rc;
  rc = k4_reserve(ix, 2 * n_pairs, max_read_len, mh);
  if (rc != K4_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  K4_HIP(ix, hipMemsetAsync(ix->pe_ctl, 0, 16, st));
  rc = k4i_kalign_batch_dev(ix, &kp, 2 * n_pairs, max_read_len, d_reads, d_offs, d_lens, ix->pe_rr, ix->pe_hits, stream, 1);
  if (rc != K4_OK) return rc;
  hipLaunchKernelGGL(k4k_pe_pair, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, st, pe, n_pairs, mh, ix->pe_rr,
                     ix->pe_hits, (k4_pe_read*)d_out, ix->pe_list, ix->pe_ctl);
  if (pe.pe_mode == 1 || pe.pe_mode == 3) {
    K4PeChim ch = {0, 0, 1, p->min_edit_dist};
    if (p->min_chimeric_len > 0) {
      int slides = 0;
      int mcl = k4_min_core_len(ix, p->pmode, &slides);
      if (p->min_core_len > 0) mcl = p->min_core_len;
      if (p->max_num_slides > 0) slides = p->max_num_slides;
      ch.min_chimeric_len = p->min_chimeric_len; ch.min_core_len = mcl; ch.slides_per100 = std::max(slides, 1);
    }
    hipLaunchKernelGGL(k4k_pe_orphans, dim3((unsigned)std::min<int64_t>(n_pairs, 256 * 32)), dim3(64),
                       ch.min_chimeric_len > 0 ? K4_RESCUE_LDS_CHIM : K4_RESCUE_LDS, st, ix->d, pe, p->max_subs,
                       (const uint8_t*)d_reads, (const uint64_t*)d_offs, (const uint32_t*)d_lens, ix->pe_list,
                       (k4_pe_read*)d_out, ix->pe_ctl, ch);
  }
  K4_HIP(ix, hipGetLastError());
  K4_HIP(ix, hipStreamSynchronize(st));  // (the call's contract, include/k4sfx.h: the stream is waited for)
  return K4_OK;
}

// Host buffers: interleave the two files' reads, run the device flow in slices, copy the records back.
extern "C" int k4_kalign_pe_batch(k4_index* ix, const k4_kalign_params* p, const k4_pe_params* pe_in, int64_t n_pairs,
                                  const uint8_t* reads1, const uint64_t* offs1, const uint32_t* lens1,
                                  const uint8_t* reads2, const uint64_t* offs2, const uint32_t* lens2, k4_pe_read* out) {
  if (!ix || !p || !pe_in) return K4_ERR_PARAMS;
  if (n_pairs < 0 || (n_pairs > 0 && (!reads1 || !offs1 || !lens1 || !reads2 || !offs2 || !lens2 || !out)))
    return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  if (pe_in->pe_mode < 1 || pe_in->pe_mode > 4 || pe_in->pair_min_len < 1 || pe_in->pair_max_len < pe_in->pair_min_len)
    return k4_fail(ix, K4_ERR_PARAMS, "PE parameters out of range");
  if (n_pairs == 0) return K4_OK;
  K4_HIP(ix, hipSetDevice(ix->device));
  const int64_t slice = 4 << 20;  // pairs per pass: bounds the staging buffers
  std::vector<uint8_t> cat;
  std::vector<uint64_t> offs;
  std::vector<uint32_t> lens;
  uint8_t* d_reads = nullptr;
  uint64_t* d_offs = nullptr;
  uint32_t* d_lens = nullptr;
  k4_pe_read* d_out = nullptr;
  size_t reads_cap = 0;
  auto cleanup = [&]() {
    for (void* q : {(void*)d_reads, (void*)d_offs, (void*)d_lens, (void*)d_out})
      if (q) hipFree(q);
  };
  int rc = K4_OK;
  const int64_t cap = std::min(n_pairs, slice);
  if ((rc = k4_check_hip(ix, hipMalloc(&d_offs, (size_t)(2 * cap + 1) * 8), "PE staging")) != K4_OK ||
      (rc = k4_check_hip(ix, hipMalloc(&d_lens, (size_t)(2 * cap + 1) * 4), "PE staging")) != K4_OK ||
      (rc = k4_check_hip(ix, hipMalloc(&d_out, (size_t)(2 * cap + 1) * sizeof(k4_pe_read)), "PE staging")) != K4_OK) {
    cleanup();
    return rc;
  }
  for (int64_t s0 = 0; s0 < n_pairs && rc == K4_OK; s0 += slice) {
    const int64_t n = std::min(slice, n_pairs - s0);
    uint64_t tot = 0;
    int max_len = 1;
    for (int64_t i = s0; i < s0 + n; i++) {
      tot += (uint64_t)lens1[i] + lens2[i];
      max_len = std::max<int>(max_len, (int)std::max(lens1[i], lens2[i]));
    }
    if (max_len > K4_MAX_READ_LEN) { rc = k4_fail(ix, K4_ERR_PARAMS, "read longer than %d bases", K4_MAX_READ_LEN); break; }
    cat.resize(tot + 16);
    offs.resize((size_t)2 * n);
    lens.resize((size_t)2 * n);
    uint64_t o = 0;
    for (int64_t i = 0; i < n; i++) {
      offs[2 * i] = o; lens[2 * i] = lens1[s0 + i];
      memcpy(cat.data() + o, reads1 + offs1[s0 + i], lens1[s0 + i]);
      o += lens1[s0 + i];
      offs[2 * i + 1] = o; lens[2 * i + 1] = lens2[s0 + i];
      memcpy(cat.data() + o, reads2 + offs2[s0 + i], lens2[s0 + i]);
      o += lens2[s0 + i];
    }
    if (tot + 64 > reads_cap) {
      if (d_reads) hipFree(d_reads);
      d_reads = nullptr;
      if ((rc = k4_check_hip(ix, hipMalloc(&d_reads, tot + 64), "PE staging")) != K4_OK) break;
      reads_cap = tot + 64;
    }
    hipStream_t st = ix->stream;
    if ((rc = k4_check_hip(ix, hipMemcpyAsync(d_reads, cat.data(), tot, hipMemcpyHostToDevice, st), "PE upload")) != K4_OK) break;
    if ((rc = k4_check_hip(ix, hipMemcpyAsync(d_offs, offs.data(), (size_t)2 * n * 8, hipMemcpyHostToDevice, st), "PE upload")) != K4_OK) break;
    if ((rc = k4_check_hip(ix, hipMemcpyAsync(d_lens, lens.data(), (size_t)2 * n * 4, hipMemcpyHostToDevice, st), "PE upload")) != K4_OK) break;
    rc = k4_kalign_pe_batch_dev(ix, p, pe_in, n, max_len, d_reads, d_offs, d_lens, d_out, st);
    if (rc != K4_OK) break;
    rc = k4_check_hip(ix, hipMemcpy(out + 2 * s0, d_out, (size_t)2 * n * sizeof(k4_pe_read), hipMemcpyDeviceToHost), "PE download");
  }
  cleanup();
  return rc;
}
