// kit4b_amd/csrc/k4_pe.hip -- paired-end pass: mate rescue kernel + the CKAligner pairing logic on the host.
//
//   CSfxArray::AlignPairedRead    libkit4b/SfxArray.cpp:8571-8767 (linear-scan branch) + AdaptiveTrim :5561-5639
//   CKAligner::ProcCoredApprox    ngskit4b/KAligner.cpp:10160-10239 (multi x multi pair resolution)
//   CKAligner::ProcessPairedEnds  ngskit4b/KAligner.cpp:3159-3596, AcceptProvPE :2799, PEInsertSize :2875
#include <string.h>
#include <algorithm>
#include <vector>
#include "k4_device.h"

// ---- mate rescue: one wave per task scans every locus of the insert window -------------------------------------
// A locus is acceptable when the full-length mismatch count is <= ((len*rate)+99)/100, no mismatch sits in the first
// three or the last two bases (AdaptiveTrim :5622-5631 with MinFlankMatches 3) and the count is < rate + 1
// (PrevBestMaxChimericMMs :8684).  The reference keeps the first strictly better locus and stops at a 0-mismatch one,
// i.e. the lexicographic minimum of (mismatches, locus).
__global__ void __launch_bounds__(64) k4k_mate_rescue(K4DevIndex ix, const k4_rescue_task* __restrict__ tasks,
                                                      const uint8_t* __restrict__ reads, int64_t n_tasks,
                                                      int32_t* __restrict__ rslt, k4_hit* __restrict__ hits) {
  __shared__ uint8_t rs[K4_MAX_READ_LEN];  // the mate, oriented as it must align ('-': reverse complemented)
  const int lane = threadIdx.x;
  for (int64_t t = blockIdx.x; t < n_tasks; t += gridDim.x) {
    const k4_rescue_task tk = tasks[t];
    int res = 0;
    uint32_t best = 0xFFFFFFFFu;  // (mm << 20) | (locus - sp)
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
        if (run && (ep - sp) >= 1000) { run = false; res = K4_ERR_UNSUPPORTED; }  // the reference's CoreLen==0 path
        // AdaptiveTrim parameter validation (:5598-5603): failing it means no locus is ever accepted
        if (run && (len < 25 || len > 2048 || (uint32_t)tk.max_allowed_mm > (uint32_t)((15 * len + 99) / 100))) run = false;
      }
      if (run) {
        __syncthreads();
        const uint8_t* src = reads + tk.read_off;
        for (int q = lane; q < len; q += 64) {
          uint8_t b = tk.antisense ? src[len - 1 - q] & 7 : src[q] & 7;
          if (tk.antisense && b <= 3) b = 3 - b;
          rs[q] = b;
        }
        __syncthreads();
        const uint32_t max_allowed = ((uint32_t)len * (uint32_t)tk.max_allowed_mm + 99) / 100;
        for (uint32_t loci = sp + lane; loci <= ep; loci += 64) {
          const uint64_t g = cs + loci;
          uint32_t mm = 0;
          bool ok = true;
          for (int o = 0; o < len; o++) {
            if (rs[o] != k4d_ref_base(ix, g + o)) {
              if (++mm > max_allowed) { ok = false; break; }
              if (o < 3 || (len - o) < 3) { ok = false; break; }
            }
          }
          if (ok && mm <= (uint32_t)tk.max_allowed_mm) best = min(best, (mm << 20) | (loci - sp));
        }
      }
    } else
      res = -1;
    for (int d = 32; d > 0; d >>= 1) best = min(best, (uint32_t)__shfl_down(best, d, 64));
    if (lane == 0) {
      k4_hit h;
      memset(&h, 0, sizeof(h));
      if (run && best != 0xFFFFFFFFu) {
        res = 1;
        h.chrom_id = tk.chrom_id;
        h.match_loci = sp + (best & 0xFFFFF);
        h.match_len = (uint16_t)len;
        h.strand = tk.antisense ? '-' : '+';
        h.mismatches = (uint8_t)(best >> 20);
      }
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
  k4_rescue_task* d_t = nullptr;
  uint8_t* d_r = nullptr;
  int32_t* d_res = nullptr;
  k4_hit* d_h = nullptr;
  auto cleanup = [&]() {
    for (void* p : {(void*)d_t, (void*)d_r, (void*)d_res, (void*)d_h})
      if (p) hipFree(p);
  };
  int rc;
  if ((rc = k4_check_hip(ix, hipMalloc(&d_t, (size_t)n * sizeof(k4_rescue_task)), "rescue alloc")) != K4_OK ||
      (rc = k4_check_hip(ix, hipMalloc(&d_r, reads_bytes + 16), "rescue alloc")) != K4_OK ||
      (rc = k4_check_hip(ix, hipMalloc(&d_res, (size_t)n * 4), "rescue alloc")) != K4_OK ||
      (rc = k4_check_hip(ix, hipMalloc(&d_h, (size_t)n * sizeof(k4_hit)), "rescue alloc")) != K4_OK) {
    cleanup();
    return rc;
  }
  hipStream_t st = ix->stream;
  hipMemcpyAsync(d_t, tasks, (size_t)n * sizeof(k4_rescue_task), hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_r, reads, reads_bytes, hipMemcpyHostToDevice, st);
  unsigned grid = (unsigned)std::min<int64_t>(n, 256 * 32);
  hipLaunchKernelGGL(k4k_mate_rescue, dim3(grid), dim3(64), 0, st, ix->d, d_t, d_r, n, d_res, d_h);
  hipMemcpyAsync(rslt, d_res, (size_t)n * 4, hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(hits, d_h, (size_t)n * sizeof(k4_hit), hipMemcpyDeviceToHost, st);
  rc = k4_check_hip(ix, hipStreamSynchronize(st), "mate rescue");
  cleanup();
  if (rc != K4_OK) return rc;
  for (int64_t i = 0; i < n; i++)
    if (rslt[i] == K4_ERR_UNSUPPORTED)
      return k4_fail(ix, K4_ERR_UNSUPPORTED, "insert window of 1000 or more loci (reference takes its CoreLen==0 seed path): keep -D minus -d below 1000");
  return K4_OK;
}

// ---- host: the pairing logic ----------------------------------------------------------------------------------------
static int pe_insert_size(const k4_pe_params& pe, uint8_t s1, uint32_t st1, uint32_t en1, uint8_t s2, uint32_t st2,
                          uint32_t en2) {  // PEInsertSize, KAligner.cpp:2875-2918
  if ((pe.pair_strand && s1 != s2) || (!pe.pair_strand && s1 == s2)) return -1;
  int frag = (int)(1 + std::max(en1, en2) - std::min(st1, st2));
  if (frag < 0) return -1;
  if (frag < pe.pair_min_len) return -6;
  if (frag > pe.pair_max_len) return -7;
  return frag;
}
static int accept_prov_pe(const k4_pe_params& pe, int nh1, const k4_hit& h1, int nh2, const k4_hit& h2) {  // :2799-2861
  if (!(nh1 == 1 && nh2 == 1)) return 0;
  if (h1.chrom_id != h2.chrom_id) return -2;
  return pe_insert_size(pe, h1.strand, h1.match_loci, h1.match_loci + h1.match_len - 1, h2.strand, h2.match_loci,
                        h2.match_loci + h2.match_len - 1);
}

extern "C" int k4_kalign_pe_batch(k4_index* ix, const k4_kalign_params* p, const k4_pe_params* pe_in, int64_t n_pairs,
                                  const uint8_t* reads1, const uint64_t* offs1, const uint32_t* lens1,
                                  const uint8_t* reads2, const uint64_t* offs2, const uint32_t* lens2, k4_pe_read* out) {
  if (!ix || !p || !pe_in) return K4_ERR_PARAMS;
  if (n_pairs < 0 || (n_pairs > 0 && (!reads1 || !offs1 || !lens1 || !reads2 || !offs2 || !lens2 || !out)))
    return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  const k4_pe_params pe = *pe_in;
  if (pe.pe_mode < 1 || pe.pe_mode > 4 || pe.pair_min_len < 1 || pe.pair_max_len < pe.pair_min_len)
    return k4_fail(ix, K4_ERR_PARAMS, "PE parameters out of range");
  if (n_pairs == 0) return K4_OK;
  const int64_t n = n_pairs;
  // 1. both ends as SE reads with MaxHits = max(m_MaxMLmatches, cMaxMLPEmatches) and the PE classification
  k4_kalign_params kp = *p;
  kp.pe_mode = 1;
  kp.max_ml = std::max(p->max_ml, 10);
  const int mh = kp.max_ml;
  std::vector<uint8_t> cat;
  std::vector<uint64_t> offs((size_t)2 * n);
  std::vector<uint32_t> lens((size_t)2 * n);
  uint64_t tot = 0;
  for (int64_t i = 0; i < n; i++) tot += (uint64_t)lens1[i] + lens2[i];
  cat.resize(tot + 16);
  uint64_t o = 0;
  for (int64_t i = 0; i < n; i++) {
    offs[2 * i] = o; lens[2 * i] = lens1[i];
    memcpy(cat.data() + o, reads1 + offs1[i], lens1[i]);
    o += lens1[i];
    offs[2 * i + 1] = o; lens[2 * i + 1] = lens2[i];
    memcpy(cat.data() + o, reads2 + offs2[i], lens2[i]);
    o += lens2[i];
  }
  std::vector<k4_read_result> rr((size_t)2 * n);
  std::vector<k4_hit> hits((size_t)2 * n * mh);
  int rc = k4_kalign_batch(ix, &kp, 2 * n, cat.data(), offs.data(), lens.data(), rr.data(), hits.data());
  if (rc != K4_OK) return rc;

  // 2. per pair: AlignRead's PE view of each end, ProcCoredApprox's multi x multi resolution, then ProcessPairedEnds up
  //    to the point where a mate rescue is needed
  enum { DONE = 0, ORPHAN = 1 };
  std::vector<uint8_t> state((size_t)n, DONE), f_unal((size_t)n), r_unal((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    k4_pe_read& f = out[2 * i];
    k4_pe_read& r = out[2 * i + 1];
    for (int e = 0; e < 2; e++) {
      const k4_read_result& q = rr[2 * i + e];
      k4_pe_read& d = out[2 * i + e];
      memset(&d, 0, sizeof(d));
      d.nar = q.nar; d.num_hits = q.num_hits; d.inst = q.inst; d.low_mm = q.low_mm;
      if (q.nar == K4_NAR_ACCEPTED) d.hit = hits[(size_t)(2 * i + e) * mh];
    }
    const k4_hit* h1 = &hits[(size_t)(2 * i) * mh];
    const k4_hit* h2 = &hits[(size_t)(2 * i + 1) * mh];
    if (rr[2 * i].hit_rslt == K4_HR_HITS && rr[2 * i + 1].hit_rslt == K4_HR_HITS && !(f.inst == 1 && r.inst == 1) &&
        f.inst < 10 && r.inst < 10) {  // KAligner.cpp:10185-10239
      bool multi = false, accepted = false;
      k4_hit p1{}, p2{};
      for (int a = 0; !(multi && !accepted) && a < f.inst; a++)
        for (int b = 0; b < r.inst; b++)
          if (accept_prov_pe(pe, 1, h1[a], 1, h2[b]) > 0) {
            if (!multi) { p1 = h1[a]; p2 = h2[b]; multi = true; accepted = true; }
            else { accepted = false; break; }
          }
      if (accepted) {
        f.hit = p1; f.nar = K4_NAR_ACCEPTED; f.num_hits = 1;
        r.hit = p2; r.nar = K4_NAR_ACCEPTED; r.num_hits = 1;
      }
    }
    // ProcessPairedEnds, KAligner.cpp:3207-3318
    f_unal[i] = f.nar == K4_NAR_NS || f.nar == K4_NAR_NOHIT || f.nar == K4_NAR_UNALIGNED;
    r_unal[i] = r.nar == K4_NAR_NS || r.nar == K4_NAR_NOHIT || r.nar == K4_NAR_UNALIGNED;
    if (!(f.nar == K4_NAR_ACCEPTED || r.nar == K4_NAR_ACCEPTED)) continue;
    bool strict_fail = pe.pe_mode == 2 && (f_unal[i] || r_unal[i]);
    if (!strict_fail && f.nar == K4_NAR_ACCEPTED && r.nar == K4_NAR_ACCEPTED) {
      int frag = accept_prov_pe(pe, f.num_hits, f.hit, r.num_hits, r.hit);
      if (frag > 0) { f.pe_aligned = r.pe_aligned = 1; continue; }
      switch (frag) {
        case -1: f.nar = r.nar = K4_NAR_PESTRAND; break;
        case -2: f.nar = r.nar = K4_NAR_PECHROM; break;
        case -6: f.nar = r.nar = K4_NAR_PEINSERTMIN; break;
        case -7: f.nar = r.nar = K4_NAR_PEINSERTMAX; break;
        default: break;
      }
      if (pe.pe_mode == 2) strict_fail = true;
    }
    if (strict_fail) {
      f.num_hits = r.num_hits = 0; f.inst = r.inst = 0;
      if (f.nar == K4_NAR_ACCEPTED) f.nar = K4_NAR_PENOHIT;
      if (r.nar == K4_NAR_ACCEPTED) r.nar = K4_NAR_PENOHIT;
      continue;
    }
    state[i] = ORPHAN;
  }

  // 3. orphan recovery (modes 1 and 3): first the PE1 alignment as anchor (:3320-3418), then PE2 (:3424-3535)
  if (pe.pe_mode == 1 || pe.pe_mode == 3) {
    for (int round = 0; round < 2; round++) {
      std::vector<k4_rescue_task> tasks;
      std::vector<int64_t> owner;
      for (int64_t i = 0; i < n; i++) {
        if (state[i] != ORPHAN) continue;
        const k4_pe_read& anchor = round == 0 ? out[2 * i] : out[2 * i + 1];
        const bool mate_unal = round == 0 ? r_unal[i] : f_unal[i];
        if (!(anchor.num_hits == 1 && !mate_unal)) continue;
        const bool plus = anchor.hit.strand == '+';
        k4_rescue_task t;
        memset(&t, 0, sizeof(t));
        if (round == 0) { t.b3prime_extend = plus; t.antisense = pe.pair_strand ? !plus : plus; }
        else {
          t.b3prime_extend = plus; t.antisense = plus;
          if (pe.pair_strand) { t.b3prime_extend = !t.b3prime_extend; t.antisense = !t.antisense; }
        }
        t.chrom_id = anchor.hit.chrom_id;
        t.start_loci = anchor.hit.match_loci;
        t.end_loci = anchor.hit.match_loci + anchor.hit.match_len - 1;
        const int64_t mate = round == 0 ? 2 * i + 1 : 2 * i;
        t.read_len = lens[mate];
        t.read_off = offs[mate];
        t.min_insert = pe.pair_min_len;
        t.max_insert = pe.pair_max_len;
        t.max_allowed_mm = p->max_subs;  // the per-100 bp rate, as the reference passes it (KAligner.cpp:3379, Q15)
        tasks.push_back(t);
        owner.push_back(i);
      }
      if (tasks.empty()) continue;
      std::vector<int32_t> res(tasks.size());
      std::vector<k4_hit> rh(tasks.size());
      rc = k4_mate_rescue_batch(ix, (int64_t)tasks.size(), tasks.data(), cat.data(), tot, res.data(), rh.data());
      if (rc != K4_OK) return rc;
      for (size_t q = 0; q < tasks.size(); q++) {
        if (res[q] != 1) continue;
        const int64_t i = owner[q];
        k4_pe_read& f = out[2 * i];
        k4_pe_read& r = out[2 * i + 1];
        const k4_hit& h = rh[q];
        const uint32_t hs = h.match_loci, he = h.match_loci + h.match_len - 1;
        const k4_pe_read& anchor = round == 0 ? f : r;
        const uint32_t as = anchor.hit.match_loci, ae = anchor.hit.match_loci + anchor.hit.match_len - 1;
        int frag = round == 0 ? pe_insert_size(pe, anchor.hit.strand, as, ae, h.strand, hs, he)
                              : pe_insert_size(pe, h.strand, hs, he, anchor.hit.strand, as, ae);
        if (frag <= 0) continue;
        k4_pe_read& m = round == 0 ? r : f;
        m.hit = h; m.num_hits = 1; m.low_mm = h.mismatches; m.inst = 1; m.rescued = 1;
        f.pe_aligned = r.pe_aligned = 1;
        f.nar = r.nar = K4_NAR_ACCEPTED;
        state[i] = DONE;
      }
    }
  }

  // 4. what is left could not be accepted as a pair (:3538-3585)
  for (int64_t i = 0; i < n; i++) {
    if (state[i] != ORPHAN) continue;
    k4_pe_read& f = out[2 * i];
    k4_pe_read& r = out[2 * i + 1];
    if (!(pe.pe_mode == 3 || pe.pe_mode == 4)) {
      f.num_hits = r.num_hits = 0; f.inst = r.inst = 0;
      if (f.nar == K4_NAR_ACCEPTED) f.nar = K4_NAR_PENOHIT;
      if (r.nar == K4_NAR_ACCEPTED) r.nar = K4_NAR_PENOHIT;
      continue;
    }
    for (k4_pe_read* e : {&f, &r}) {
      if (e->num_hits != 1) { e->num_hits = 0; e->inst = 0; if (e->nar == K4_NAR_ACCEPTED) e->nar = K4_NAR_PEUNALIGN; }
      else e->nar = K4_NAR_ACCEPTED;
    }
  }
  return K4_OK;
}
