// kit4b_amd/csrc/k4_trim.h -- CSfxArray::AdaptiveTrim (libkit4b/SfxArray.cpp:5561-5795) on a mismatch bit vector; shared by the chimeric pass of AlignReads (k4_ext.h, general kernel) and the chimeric mate
// rescue of AlignPairedRead (k4_pe.hip).  gfx950 only; needs k4_device.h.
#pragma once

// ---- AdaptiveTrim (SfxArray.cpp:5561-5795) over a mismatch bit vector: bit j of word j >> 5 (LSB first) is set when read
// base j differs from the target.  The vector of lane l sits at mk[w * 64 + l] (LDS).  The reference's regions are the runs
// of equal bits; its two floating-point tests compare fractions whose cross products are small integers, so they are
// evaluated exactly in integers ((M+1)/100 <= a/b  <=>  (M+1)*b <= 100*a: unequal fractions differ by >= 1/(100*2048)). ------
struct K4Trim { int len, t5, t3, mms; };

K4_DEV int k4d_mk_bit(const uint32_t* mk, int j) { return (int)((mk[(j >> 5) * 64] >> (j & 31)) & 1u); }
K4_DEV int k4d_run_end(const uint32_t* mk, int L, int pos) {  // end (exclusive) of the run of equal bits that starts at pos
  const uint32_t flip = k4d_mk_bit(mk, pos) ? 0xFFFFFFFFu : 0u;
  int w = pos >> 5, b = pos & 31;
  for (;;) {
    const uint32_t d = ((mk[w * 64] ^ flip) >> b);
    if (d) { const int e = (w << 5) + b + (__ffs((int)d) - 1); return e < L ? e : L; }
    w++; b = 0;
    if ((w << 5) >= L) return L;
  }
}

K4_DEV K4Trim k4d_adaptive_trim(const uint32_t* mk, int L, int min_trim, int max_mm, int min_flank) {
  K4Trim r = {0, 0, 0, 0};
  if (L < 25 || L > 2048 || min_trim < 15 || min_trim > L || max_mm > ((15 * L + 99) / 100) || min_flank > 10) return r;  // :5601-5605
  if (min_flank == 0) min_flank = 1;
  if (min_trim == L) {  // :5612-5639 the mismatch total, none of them inside the flanks
    const int allowed = (L * max_mm + 99) / 100;
    int mms = 0;
    for (int pos = 0; pos < L;) {
      const int e = k4d_run_end(mk, L, pos);
      if (k4d_mk_bit(mk, pos)) {
        for (int j = pos; j < e; j++) {
          if (++mms > allowed) return r;
          if (j < min_flank || (L - j) < min_flank) return r;
        }
      }
      pos = e;
    }
    r.len = L; r.mms = mms;
    return r;
  }
  // :5641-5710 one pass over the runs: is there an exact run of cMinATExactLen, which runs may start / end the result
  int n_min_exact = 0, first_start = -1, last_start = -1, last_end_end = -1;
  const int mt16 = (int)(uint16_t)min_trim;
  for (int pos = 0; pos < L;) {
    const int e = k4d_run_end(mk, L, pos);
    if (!k4d_mk_bit(mk, pos)) {
      const int rl = e - pos;
      if (rl >= 8) n_min_exact++;
      if (rl >= min_flank) {
        if (pos <= L - min_trim) { last_start = pos; if (first_start < 0) first_start = pos; }
        if (e >= mt16) last_end_end = e;
      }
    }
    pos = e;
  }
  if (!n_min_exact || first_start < 0 || last_end_end < 0) return r;
  // :5712-5780 from every start run extend over the following runs while the mismatch rate allows
  int best_len = 0, best_mm = 0, best_start = 0, best_end = 0;
  for (int s = first_start; s <= last_start;) {
    const int s_end = k4d_run_end(mk, L, s);
    const bool trim5 = !k4d_mk_bit(mk, s) && (s_end - s) >= min_flank && s <= L - min_trim;
    if (trim5) {
      int cur_len = 0, cur_mm = 0;
      for (int p = s; p < last_end_end;) {
        const int e = k4d_run_end(mk, L, p);
        const int rl = e - p;
        const bool mm = k4d_mk_bit(mk, p) != 0;
        const bool trim3 = !mm && rl >= min_flank && e >= mt16;
        cur_len += rl;
        p = e;
        if (mm) {
          if (max_mm == 0) break;
          cur_mm += rl;
          if ((max_mm + 1) * (L - s) <= 100 * cur_mm) break;
        } else if (best_len == 0) {
          best_start = s; best_end = L - (s + cur_len); best_len = cur_len; best_mm = 0;
          continue;
        }
        if (cur_len < min_trim || !trim3) continue;
        if ((max_mm + 1) * cur_len <= 100 * cur_mm) continue;
        if (best_len < cur_len || (best_len == cur_len && (best_mm == 0 || cur_mm < best_mm))) {
          best_start = s; best_end = L - (s + cur_len); best_len = cur_len; best_mm = cur_mm;
        }
      }
    }
    s = s_end;
  }
  if (best_len >= min_trim) { r.len = best_len; r.t5 = best_start; r.t3 = best_end; r.mms = best_mm; }
  return r;
}

// A cheap NECESSARY condition for k4d_adaptive_trim to return anything: what it returns is a stretch of cur_len >= min_trim bases
// with cur_mm mismatches where 100 cur_mm < (max_mm + 1) cur_len (the rate test of :5763-5775; the untrimmed form allows
// (L max_mm + 99) / 100).  A stretch that long holds a whole block of B = (min_trim + 1) / 2 bases aligned to a multiple of B,
// and the block has no more mismatches than the stretch -- so if every aligned block holds more than the largest admissible
// count, nothing can come back.  A chance locus of a short chimeric core fails this after a few popcounts instead of walking
// its fifty-odd runs of equal bits three times.
K4_DEV bool k4d_trim_possible(const uint32_t* mk, int L, int min_trim, int max_mm) {
  if (min_trim < 2 || min_trim > L) return true;  // (k4d_adaptive_trim sorts the odd parameters out itself)
  const int B = (min_trim + 1) / 2;
  const int thr = max(((max_mm + 1) * L - 1) / 100, (L * max_mm + 99) / 100);
  for (int lo = 0; lo + B <= L; lo += B) {
    const int hi = lo + B;
    int c = 0;
    for (int w = lo >> 5; w <= (hi - 1) >> 5; w++) {
      const int a = max(lo - 32 * w, 0), b = min(hi - 32 * w, 32);
      const uint32_t m = (b >= 32 ? 0xFFFFFFFFu : ((1u << b) - 1u)) & ~((1u << a) - 1u);
      c += __popc(mk[w * 64] & m);
    }
    if (c <= thr) return true;
  }
  return false;
}

// 32 bases of a 2-bit XOR (MSB first, two bits per base) -> one bit per base, base 0 in bit 0
K4_DEV uint32_t k4d_mm_bits(uint64_t x) {
  uint64_t y = (x | (x >> 1)) & 0x5555555555555555ull;
  y = (y | (y >> 1)) & 0x3333333333333333ull;
  y = (y | (y >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  y = (y | (y >> 4)) & 0x00FF00FF00FF00FFull;
  y = (y | (y >> 8)) & 0x0000FFFF0000FFFFull;
  y = (y | (y >> 16)) & 0x00000000FFFFFFFFull;
  return __brev((uint32_t)y);
}

