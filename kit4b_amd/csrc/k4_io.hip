// kit4b_amd/csrc/k4_io.hip -- the two ends of the read pipeline on the device.
//
//   k4_parse_fastx_dev   FASTA / FASTQ text in HBM -> etSeqBase reads, offsets, lengths, descriptor spans
//                        <- CKAligner::LoadRawReads  ngskit4b/KAligner.cpp:11648-12421 (descriptor = first token of the
//                           header line, bases a/c/g/t/u in either case -> 0..3, '-' -> InDel, other letters -> N, anything else sloughed)
//   k4_format_sam_dev    alignment records in HBM -> coordinate-sorted SAM text in HBM
//                        <- CKAligner::WriteBAMReadHits :5718-5914, ReportBAMread :5957-6320, SortHitMatch :10969,
//                           CSAMfile::AddAlignment libkit4b/SAMfile.cpp:2194-2377
// Both are bandwidth-bound byte shuffles: newline positions by a flagged select, one thread per record for the
// bookkeeping, one wave per record for the byte moves (coalesced along the line).
#include <string.h>
#include <cstring>
#include <algorithm>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "k4_device.h"
#include "k4_pool.h"
#include "k4_stages.h"

namespace {

struct IsNewline {
  __device__ __host__ uint8_t operator()(uint8_t c) const { return c == '\n'; }
};

K4_DEV bool k4d_is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }
// CFasta::ReadSequence (libkit4b/Fasta.cpp:1172-1173) keeps letters and '-' of a sequence line and sloughs everything else;
// Ascii2Sense (:1657-1703) maps a/c/g/t/u in either case to 0..3, '-' to eBaseInDel (6) and any other letter to eBaseN
K4_DEV bool k4d_seq_skip(uint8_t c) { return !(((c | 0x20) >= 'a' && (c | 0x20) <= 'z') || c == '-'); }
K4_DEV uint8_t k4d_base_code(uint8_t c) {
  if (c == '-') return 6;
  switch (c | 0x20) {  // lower-case the letters
    case 'a': return 0;
    case 'c': return 1;
    case 'g': return 2;
    case 't': case 'u': return 3;
    default: return 4;
  }
}

struct K4FastxArgs {
  const uint8_t* text;
  uint64_t text_bytes;   // bytes that belong to whole records (end of the last one)
  const uint32_t* nl;    // line j ends at nl[j] (offset of its '\n', or text end for an unterminated last line)
  const uint32_t* hdr;   // FASTA: line numbers of the header lines
  uint64_t n_hdr;        // FASTA: headers found (one more than n_rec when the last record is left for the next chunk)
  int64_t n_rec;
  int fastq;
  uint64_t text_base;
  uint32_t* seq_off;     // out: offset of the first sequence byte, span in bytes
  uint32_t* seq_span;
  uint32_t* lens;        // out: bases
  uint64_t* name_off;    // out
  uint32_t* name_len;
  unsigned long long* tot;  // [0] bases, [1] max len, [2] malformed records
};

K4_DEV uint32_t k4d_line_start(const uint32_t* nl, uint64_t j) { return j == 0 ? 0u : nl[j - 1] + 1; }

// one thread per record: header token, sequence span, number of bases
__global__ void __launch_bounds__(256) k4k_fastx_records(K4FastxArgs a) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= a.n_rec) return;
  const uint64_t hl = a.fastq ? (uint64_t)4 * r : a.hdr[r];
  uint32_t s0 = k4d_line_start(a.nl, hl), e0 = a.nl[hl];
  bool bad = a.text[s0] != (a.fastq ? '@' : '>');
  // the read's name: the descriptor up to its first white space, at most 79 characters (cMaxDescrIDLen - 1, KAligner.cpp:12268-12275);
  // a FASTA descriptor starts behind any blanks and tabs that follow the '>' (CFasta, Fasta.cpp:1069-1071; not so in FASTQ)
  uint32_t p0 = s0 + 1;
  if (!a.fastq)
    while (p0 < e0 && (a.text[p0] == ' ' || a.text[p0] == '\t')) p0++;
  uint32_t p = p0;
  while (p < e0 && !k4d_is_space(a.text[p])) p++;
  a.name_off[r] = a.text_base + p0;
  a.name_len[r] = min(p - p0, 79u);
  const uint32_t s1 = e0 + 1;
  uint32_t e1;
  if (a.fastq) e1 = a.nl[hl + 1];
  else e1 = (uint64_t)r + 1 < a.n_hdr ? k4d_line_start(a.nl, a.hdr[r + 1]) : (uint32_t)a.text_bytes;
  if (e1 < s1) e1 = s1;
  if (a.fastq && a.text[k4d_line_start(a.nl, hl + 2)] != '+') bad = true;
  a.seq_off[r] = s1;
  a.seq_span[r] = e1 - s1;
  if (bad) atomicAdd(&a.tot[2], 1ull);
}

// four records per wave (16 lanes each): bases = bytes of the sequence span that CFasta keeps
__global__ void __launch_bounds__(64) k4k_fastx_count(const uint8_t* __restrict__ text, const uint32_t* __restrict__ seq_off,
                                                      const uint32_t* __restrict__ seq_span, int64_t n_rec,
                                                      uint32_t* __restrict__ lens, unsigned long long* __restrict__ tot) {
  const int lane = threadIdx.x, grp = lane >> 4, gl = lane & 15;
  unsigned long long sum = 0;
  uint32_t mx = 0;
  for (int64_t r0 = (int64_t)blockIdx.x * 4; r0 < n_rec; r0 += (int64_t)gridDim.x * 4) {
    const int64_t r = r0 + grp;
    const bool on = r < n_rec;
    const uint8_t* src = text + (on ? seq_off[r] : 0u);
    const uint32_t span = on ? seq_span[r] : 0u;
    uint32_t span_max = span;
    span_max = max(span_max, (uint32_t)__shfl_xor(span_max, 16, 64));
    span_max = max(span_max, (uint32_t)__shfl_xor(span_max, 32, 64));
    uint32_t n = 0;
    for (uint32_t q = 0; q < span_max; q += 16) {
      const bool keep = q + gl < span && !k4d_seq_skip(src[q + gl]);
      n += (uint32_t)__popc((uint32_t)(__ballot(keep) >> (16 * grp)) & 0xFFFFu);
    }
    if (on && gl == 0) {
      lens[r] = n;
      sum += n;
      mx = max(mx, n);
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    sum += __shfl_down(sum, d, 64);
    mx = max(mx, (uint32_t)__shfl_down(mx, d, 64));
  }
  if (lane == 0) {
    if (sum) atomicAdd(&tot[0], sum);
    if (mx) atomicMax(&tot[1], (unsigned long long)mx);
  }
}

// FASTQ, the usual case: the sequence line holds bases only (and perhaps a '\r'), so its length is the base count -- one
// thread per record instead of a pass over the text; k4k_fastx_encode checks the assumption and the exact count is redone
// when it does not hold anywhere
__global__ void __launch_bounds__(256) k4k_fastq_lens(const uint8_t* __restrict__ text, const uint32_t* __restrict__ seq_off,
                                                      const uint32_t* __restrict__ seq_span, int64_t n_rec,
                                                      uint32_t* __restrict__ lens, unsigned long long* __restrict__ tot) {
  unsigned long long sum = 0;
  uint32_t mx = 0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rec; r += (int64_t)gridDim.x * 256) {
    uint32_t n = seq_span[r];
    if (n && text[seq_off[r] + n - 1] == '\r') n--;
    lens[r] = n;
    sum += n;
    mx = max(mx, n);
  }
  for (int d = 32; d > 0; d >>= 1) {
    sum += __shfl_down(sum, d, 64);
    mx = max(mx, (uint32_t)__shfl_down(mx, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    if (sum) atomicAdd(&tot[0], sum);
    if (mx) atomicMax(&tot[1], (unsigned long long)mx);
  }
}

// one wave per record: the sequence bytes of its span, white space squeezed out, as etSeqBase codes; expect (optional):
// the base counts the offsets were laid out with -- *mismatch is raised when a record holds a different number
__global__ void __launch_bounds__(64) k4k_fastx_encode(const uint8_t* __restrict__ text, const uint32_t* __restrict__ seq_off,
                                                       const uint32_t* __restrict__ seq_span, const uint64_t* __restrict__ offs,
                                                       int64_t n_rec, uint8_t* __restrict__ reads,
                                                       const uint32_t* __restrict__ expect, uint32_t* __restrict__ mismatch) {
  // four records per wave, 16 lanes each: a record is a chain of dependent loads (its offsets, then its bytes), and one
  // chain per wave at a time left the kernel waiting on them
  const int lane = threadIdx.x, grp = lane >> 4, gl = lane & 15;
  for (int64_t r0 = (int64_t)blockIdx.x * 4; r0 < n_rec; r0 += (int64_t)gridDim.x * 4) {
    const int64_t r = r0 + grp;
    const bool on = r < n_rec;
    const uint8_t* src = text + (on ? seq_off[r] : 0u);
    const uint32_t span = on ? seq_span[r] : 0u;
    uint8_t* dst = reads + (on ? offs[r] : 0ull);
    uint32_t span_max = span;
    span_max = max(span_max, (uint32_t)__shfl_xor(span_max, 16, 64));
    span_max = max(span_max, (uint32_t)__shfl_xor(span_max, 32, 64));
    uint32_t done = 0;
    for (uint32_t q = 0; q < span_max; q += 16) {
      const uint8_t c = q + gl < span ? src[q + gl] : (uint8_t)' ';
      const bool keep = !k4d_seq_skip(c);
      const uint32_t m = (uint32_t)(__ballot(keep) >> (16 * grp)) & 0xFFFFu;
      if (keep) dst[done + (uint32_t)__popc(m & ((1u << gl) - 1u))] = k4d_base_code(c);
      done += (uint32_t)__popc(m);
    }
    if (expect && on && gl == 0 && done != expect[r]) *mismatch = 1u;
  }
}

// FASTQ qualities (k4_set_fastq_quality 0..2): the 4-bit score of every base into bits 4..7 of its read byte, as LoadRawReads packs
// them (KAligner.cpp:12096-12163).  Four records per wave, 16 lanes each; a quality line that is not as long as its read is an
// error there (eBSFerrParse) and here.
__global__ void __launch_bounds__(64) k4k_fastq_quals(const uint8_t* __restrict__ text, const uint32_t* __restrict__ nl, const uint64_t* __restrict__ offs,
                                                      const uint32_t* __restrict__ lens, int64_t n_rec, uint8_t* __restrict__ reads,
                                                      const uint8_t* __restrict__ lut, uint32_t* __restrict__ mismatch) {
  const int lane = threadIdx.x, grp = lane >> 4, gl = lane & 15;
  for (int64_t r0 = (int64_t)blockIdx.x * 4; r0 < n_rec; r0 += (int64_t)gridDim.x * 4) {
    const int64_t r = r0 + grp;
    if (r >= n_rec) continue;
    const uint32_t qs = nl[4 * r + 2] + 1;
    uint32_t qe = nl[4 * r + 3];
    if (qe > qs && text[qe - 1] == '\r') qe--;
    const uint32_t len = lens[r];
    if (qe - qs != len) { if (gl == 0) atomicAdd(mismatch, 1u); continue; }
    uint8_t* dst = reads + offs[r];
    for (uint32_t q = gl; q < len; q += 16) dst[q] = (uint8_t)((dst[q] & 0x0f) | (lut[text[qs + q]] << 4));
  }
}

// Line ends in two passes over 4 KB tiles (256 threads x 16 bytes): newlines per tile, then -- after a scan of the tile
// counts -- their offsets, in text order (exclusive scan of the per-thread counts inside the block).
#define K4_NL_TILE 4096
K4_DEV uint32_t k4d_nl_mask(const uint8_t* __restrict__ text, uint64_t n, uint64_t base) {
  uint32_t m = 0;
  if (base + 16 <= n) {
#pragma unroll
    for (int q = 0; q < 16; q++) m |= (uint32_t)(text[base + q] == '\n') << q;
  } else
    for (int q = 0; q < 16; q++)
      if (base + q < n) m |= (uint32_t)(text[base + q] == '\n') << q;
  return m;
}
__global__ void __launch_bounds__(256) k4k_nl_tiles(const uint8_t* __restrict__ text, uint64_t n, uint32_t* __restrict__ counts) {
  __shared__ uint32_t ws[4];
  const uint64_t base = (uint64_t)blockIdx.x * K4_NL_TILE + (uint64_t)threadIdx.x * 16;
  uint32_t c = __popc(k4d_nl_mask(text, n, base));
  for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ void __launch_bounds__(256) k4k_nl_scatter(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ tile_off,
                                                      uint32_t* __restrict__ nl) {
  __shared__ uint32_t ws[4];
  const uint64_t base = (uint64_t)blockIdx.x * K4_NL_TILE + (uint64_t)threadIdx.x * 16;
  uint32_t m = k4d_nl_mask(text, n, base);
  const uint32_t c = __popc(m);
  uint32_t inc = c;  // inclusive scan inside the wave
  const int lane = threadIdx.x & 63;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t v = __shfl_up(inc, d, 64);
    if (lane >= d) inc += v;
  }
  if (lane == 63) ws[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t off = tile_off[blockIdx.x] + inc - c;
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++) off += ws[w];
  while (m) {
    const int q = __ffs(m) - 1;
    m &= m - 1;
    nl[off++] = (uint32_t)(base + q);
  }
}

struct IsHeaderLine {  // FASTA: the line starts with '>'
  const uint8_t* text;
  const uint32_t* nl;
  __device__ bool operator()(uint32_t j) const { return text[j == 0 ? 0u : nl[j - 1] + 1] == '>'; }
};

// scratch of one stage call, from the device's caching pool (k4_pool.h: no hipFree, which would wait for every stream of the
// device -- the copy streams of the overlapped pipeline included); PoolStream names the stream the call works on
thread_local hipStream_t tl_pool_stream = nullptr;
struct PoolStream {
  hipStream_t prev;
  explicit PoolStream(hipStream_t s) : prev(tl_pool_stream) { tl_pool_stream = s; }
  ~PoolStream() { tl_pool_stream = prev; }
};
struct Buf : K4PoolBuf {
  hipError_t alloc(size_t bytes) { return K4PoolBuf::alloc(bytes, tl_pool_stream); }
};

}  // namespace

extern "C" int k4_parse_fastx_dev(k4_index* ix, const void* d_text_v, uint64_t text_bytes, uint64_t text_base, int final_chunk,
                                  int format, int64_t max_records, void* d_reads, uint64_t reads_base, void* d_offs,
                                  void* d_lens, void* d_name_off, void* d_name_len, k4_parse_info* info, void* stream) {
  if (!ix || !info) return K4_ERR_PARAMS;
  memset(info, 0, sizeof(*info));
  if (text_bytes == 0 || max_records <= 0) return K4_OK;
  if (!d_text_v || !d_reads || !d_offs || !d_lens || !d_name_off || !d_name_len) return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  if (text_bytes >= 0xFFFFFF00ull) return k4_fail(ix, K4_ERR_PARAMS, "text chunks are limited to 2^32-256 bytes");
  K4_HIP(ix, hipSetDevice(ix->device));
  hipStream_t st = (hipStream_t)stream;
  PoolStream pool_scope(st);
  const uint8_t* text = (const uint8_t*)d_text_v;
  if (format == 0) {  // first byte decides, as CFasta does
    uint8_t c = 0;
    K4_HIP(ix, hipMemcpyAsync(&c, text, 1, hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipStreamSynchronize(st));
    format = c == '@' ? K4_FASTQ : c == '>' ? K4_FASTA : 0;
    if (!format) return k4_fail(ix, K4_ERR_NOT_FASTA, "input is neither FASTA ('>') nor FASTQ ('@')");
  }
  const bool fastq = format == K4_FASTQ;
  Buf tot, nlb, hdrb, tmp, cnt, so, ss, tcnt, toff;
  K4_HIP(ix, tot.alloc(32));
  K4_HIP(ix, cnt.alloc(8));
  K4_HIP(ix, hipMemsetAsync(tot.p, 0, 32, st));
  const uint64_t n_tiles = (text_bytes + K4_NL_TILE - 1) / K4_NL_TILE;
  K4_HIP(ix, tcnt.alloc((n_tiles + 1) * 4));
  K4_HIP(ix, toff.alloc((n_tiles + 1) * 4));
  K4_HIP(ix, hipMemsetAsync(tcnt.as<uint32_t>() + n_tiles, 0, 4, st));
  hipLaunchKernelGGL(k4k_nl_tiles, dim3((unsigned)n_tiles), dim3(256), 0, st, text, text_bytes, tcnt.as<uint32_t>());
  {
    size_t tb = 0;
    K4_HIP(ix, rocprim::exclusive_scan(nullptr, tb, tcnt.as<uint32_t>(), toff.as<uint32_t>(), 0u, (size_t)(n_tiles + 1),
                                       rocprim::plus<uint32_t>(), st));
    K4_HIP(ix, tmp.alloc(tb));
    K4_HIP(ix, rocprim::exclusive_scan(tmp.p, tb, tcnt.as<uint32_t>(), toff.as<uint32_t>(), 0u, (size_t)(n_tiles + 1),
                                       rocprim::plus<uint32_t>(), st));
  }
  uint32_t n_nl32 = 0;
  uint8_t last = 0;
  K4_HIP(ix, hipMemcpyAsync(&n_nl32, toff.as<uint32_t>() + n_tiles, 4, hipMemcpyDeviceToHost, st));
  K4_HIP(ix, hipMemcpyAsync(&last, text + text_bytes - 1, 1, hipMemcpyDeviceToHost, st));
  K4_HIP(ix, hipStreamSynchronize(st));
  const unsigned long long n_nl = n_nl32;
  // line ends: every '\n', plus the end of the text when the final chunk's last line is unterminated
  uint64_t n_lines = n_nl;
  K4_HIP(ix, nlb.alloc((n_nl + 2) * 4));
  hipLaunchKernelGGL(k4k_nl_scatter, dim3((unsigned)n_tiles), dim3(256), 0, st, text, text_bytes, toff.as<uint32_t>(), nlb.as<uint32_t>());
  if (final_chunk && last != '\n') {
    const uint32_t endp = (uint32_t)text_bytes;
    K4_HIP(ix, hipMemcpyAsync(nlb.as<uint32_t>() + n_nl, &endp, 4, hipMemcpyHostToDevice, st));
    K4_HIP(ix, hipStreamSynchronize(st));  // (endp is a stack variable)
    n_lines++;
  }
  if (n_lines == 0) return K4_OK;  // no complete line in this chunk yet
  int64_t n_rec = 0;
  uint64_t n_hdr = 0;
  uint64_t consumed = 0;
  if (fastq) {
    n_rec = (int64_t)std::min<uint64_t>(n_lines / 4, (uint64_t)max_records);
    if (n_rec == 0) return K4_OK;
    uint32_t e = 0;
    K4_HIP(ix, hipMemcpyAsync(&e, nlb.as<uint32_t>() + (4 * n_rec - 1), 4, hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipStreamSynchronize(st));
    consumed = std::min<uint64_t>((uint64_t)e + 1, text_bytes);
    if (final_chunk && (uint64_t)n_rec == n_lines / 4 && (uint64_t)n_rec < (uint64_t)max_records) consumed = text_bytes;
  } else {
    K4_HIP(ix, hdrb.alloc((n_lines + 1) * 4));
    rocprim::counting_iterator<uint32_t> lines(0);
    IsHeaderLine pred{text, nlb.as<uint32_t>()};
    size_t tb = 0;
    K4_HIP(ix, rocprim::select(nullptr, tb, lines, hdrb.as<uint32_t>(), cnt.as<uint64_t>(), (size_t)n_lines, pred, st));
    Buf tmp2;
    K4_HIP(ix, tmp2.alloc(tb));
    K4_HIP(ix, rocprim::select(tmp2.p, tb, lines, hdrb.as<uint32_t>(), cnt.as<uint64_t>(), (size_t)n_lines, pred, st));
    K4_HIP(ix, hipMemcpyAsync(&n_hdr, cnt.p, 8, hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipStreamSynchronize(st));
    if (n_hdr == 0) return K4_OK;
    // a non-final chunk keeps its last record for the next call (more of its sequence may follow)
    uint64_t avail = final_chunk ? n_hdr : n_hdr - 1;
    n_rec = (int64_t)std::min<uint64_t>(avail, (uint64_t)max_records);
    if (n_rec == 0) return K4_OK;
    if ((uint64_t)n_rec < n_hdr) {  // ends where the next header line starts
      uint32_t hl = 0, e = 0;
      K4_HIP(ix, hipMemcpyAsync(&hl, hdrb.as<uint32_t>() + n_rec, 4, hipMemcpyDeviceToHost, st));
      K4_HIP(ix, hipStreamSynchronize(st));
      if (hl > 0) {
        K4_HIP(ix, hipMemcpyAsync(&e, nlb.as<uint32_t>() + (hl - 1), 4, hipMemcpyDeviceToHost, st));
        K4_HIP(ix, hipStreamSynchronize(st));
        consumed = (uint64_t)e + 1;
      }
    } else
      consumed = text_bytes;
  }
  K4_HIP(ix, so.alloc((size_t)n_rec * 4));
  K4_HIP(ix, ss.alloc((size_t)n_rec * 4));
  K4FastxArgs a;
  a.text = text; a.text_bytes = consumed; a.nl = nlb.as<uint32_t>(); a.hdr = hdrb.as<uint32_t>(); a.n_hdr = n_hdr;
  a.n_rec = n_rec; a.fastq = fastq ? 1 : 0; a.text_base = text_base;
  a.seq_off = so.as<uint32_t>(); a.seq_span = ss.as<uint32_t>(); a.lens = (uint32_t*)d_lens;
  a.name_off = (uint64_t*)d_name_off; a.name_len = (uint32_t*)d_name_len; a.tot = tot.as<unsigned long long>();
  hipLaunchKernelGGL(k4k_fastx_records, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, st, a);
  // tot: [0] bases, [1] longest, [2] malformed records, [3] (as uint32) FASTQ fast-path mismatch
  bool exact = !fastq;
  for (;;) {
    if (exact)
      hipLaunchKernelGGL(k4k_fastx_count, dim3((unsigned)std::min<int64_t>((n_rec + 3) / 4, 1 << 14)), dim3(64), 0, st, text, so.as<uint32_t>(),
                         ss.as<uint32_t>(), n_rec, (uint32_t*)d_lens, tot.as<unsigned long long>());
    else
      hipLaunchKernelGGL(k4k_fastq_lens, dim3((unsigned)std::min<int64_t>((n_rec + 255) / 256, 4096)), dim3(256), 0, st, text,
                         so.as<uint32_t>(), ss.as<uint32_t>(), n_rec, (uint32_t*)d_lens, tot.as<unsigned long long>());
    size_t tb = 0;
    K4_HIP(ix, rocprim::exclusive_scan(nullptr, tb, (const uint32_t*)d_lens, (uint64_t*)d_offs, reads_base, (size_t)n_rec,
                                       rocprim::plus<uint64_t>(), st));
    Buf tmp3;
    K4_HIP(ix, tmp3.alloc(tb));
    K4_HIP(ix, rocprim::exclusive_scan(tmp3.p, tb, (const uint32_t*)d_lens, (uint64_t*)d_offs, reads_base, (size_t)n_rec,
                                       rocprim::plus<uint64_t>(), st));
    hipLaunchKernelGGL(k4k_fastx_encode, dim3((unsigned)std::min<int64_t>((n_rec + 3) / 4, 1 << 16)), dim3(64), 0, st, text, so.as<uint32_t>(),
                       ss.as<uint32_t>(), (const uint64_t*)d_offs, n_rec, (uint8_t*)d_reads,
                       exact ? (const uint32_t*)nullptr : (const uint32_t*)d_lens, reinterpret_cast<uint32_t*>(tot.as<unsigned long long>() + 3));
    K4_HIP(ix, hipGetLastError());
    K4_HIP(ix, hipStreamSynchronize(st));  // tmp3 is released here
    if (exact) break;
    uint32_t mis = 0;
    K4_HIP(ix, hipMemcpy(&mis, tot.as<unsigned long long>() + 3, 4, hipMemcpyDeviceToHost));
    if (!mis) break;
    K4_HIP(ix, hipMemsetAsync(tot.p, 0, 16, st));  // bases and longest are counted again, exactly
    exact = true;
  }
  if (fastq && ix->q_method != 3 && ix->d_qlut) {  // kalign -g0..2: the scores ride in the read bytes
    K4_HIP(ix, hipMemsetAsync(tot.as<unsigned long long>() + 3, 0, 4, st));
    hipLaunchKernelGGL(k4k_fastq_quals, dim3((unsigned)std::min<int64_t>((n_rec + 3) / 4, 1 << 16)), dim3(64), 0, st, text, nlb.as<uint32_t>(),
                       (const uint64_t*)d_offs, (const uint32_t*)d_lens, n_rec, (uint8_t*)d_reads, (const uint8_t*)ix->d_qlut,
                       reinterpret_cast<uint32_t*>(tot.as<unsigned long long>() + 3));
    uint32_t mis = 0;
    K4_HIP(ix, hipMemcpyAsync(&mis, tot.as<unsigned long long>() + 3, 4, hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipStreamSynchronize(st));
    if (mis) return k4_fail(ix, K4_ERR_PARSE, "%u FASTQ records whose quality line is not as long as the read", mis);
  }
  unsigned long long t[3] = {0, 0, 0};
  K4_HIP(ix, hipMemcpy(t, tot.p, 24, hipMemcpyDeviceToHost));
  info->n_records = (uint64_t)n_rec;
  info->consumed = consumed;
  info->n_bases = t[0];
  info->max_len = (uint32_t)t[1];
  info->format = (uint32_t)format;
  if (t[2]) return k4_fail(ix, K4_ERR_NOT_FASTA, "%llu malformed %s records (header / separator line not where expected)", t[2], fastq ? "FASTQ" : "FASTA");
  return K4_OK;
}

// ---- length filter + PE interleave --------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) k4k_prepare_reads(int pe, int64_t n, uint32_t min_len, uint32_t max_len,
                                                         const uint64_t* __restrict__ o1, const uint32_t* __restrict__ l1,
                                                         const uint64_t* __restrict__ o2, const uint32_t* __restrict__ l2,
                                                         uint64_t base2, uint64_t* __restrict__ oo, uint32_t* __restrict__ lo,
                                                         unsigned long long* __restrict__ tot, uint32_t trim5, uint32_t trim3, uint32_t sample_nth,
                                                         int64_t first_unit) {
  // grid-stride, tallies kept per thread and folded once per wave at the end: a same-address atomic per wave of a
  // one-thread-per-read launch (312 k of them for 20 M reads) costs more than the whole copy
  uint32_t n_u = 0, n_o = 0, mx = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    // the end trims of `-y` / `-Y` come off first (KAligner.cpp:12254-12260); sloughed in the reference's order of tests (:12040-12090):
    // PE1 under, PE1 over, PE2 under, PE2 over -- which tally a pair lands in depends on it
    const uint32_t trims = trim5 + trim3;
    const uint32_t ra = l1[i], rb = pe ? l2[i] : ra;
    const uint32_t a = ra > trims ? ra - trims : 0, b = rb > trims ? rb - trims : 0;
    // `-#<n>`: every n-th read (pair) of the file is loaded, the first one included; the others are never looked at (:11983-11989)
    const bool sampled = sample_nth <= 1 || (uint64_t)(first_unit + i) % sample_nth == 0;
    const bool under = sampled && (a < min_len || (a <= max_len && b < min_len));
    const bool over = sampled && !under && (a > max_len || b > max_len);
    const bool keep = sampled && !under && !over;
    if (pe) {
      oo[2 * i] = o1[i] + trim5; lo[2 * i] = keep ? a : 0;
      oo[2 * i + 1] = o2[i] + base2 + trim5; lo[2 * i + 1] = keep ? b : 0;
    } else {
      oo[i] = o1[i] + trim5; lo[i] = keep ? a : 0;
    }
    n_u += under; n_o += over;
    if (keep) mx = max(mx, max(a, b));
  }
  for (int d = 32; d > 0; d >>= 1) {
    n_u += __shfl_xor(n_u, d, 64);
    n_o += __shfl_xor(n_o, d, 64);
    mx = max(mx, (uint32_t)__shfl_xor(mx, d, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    if (n_u) atomicAdd(&tot[0], (unsigned long long)n_u);
    if (n_o) atomicAdd(&tot[1], (unsigned long long)n_o);
    if (mx) atomicMax(&tot[2], (unsigned long long)mx);
  }
}
}  // namespace

extern "C" int k4_prepare_reads_dev(k4_index* ix, int pe, int64_t n, int32_t min_len, int32_t max_len, const void* d_offs1,
                                    const void* d_lens1, const void* d_offs2, const void* d_lens2, uint64_t reads2_base,
                                    void* d_offs_out, void* d_lens_out, uint64_t* n_under, uint64_t* n_over,
                                    uint32_t* max_read_len, void* stream) {
  return k4_prepare_reads_trim_dev(ix, pe, n, min_len, max_len, 0, 0, 1, 0, d_offs1, d_lens1, d_offs2, d_lens2, reads2_base, d_offs_out, d_lens_out,
                                   n_under, n_over, max_read_len, stream);
}
// ... with kalign's end trims (`-y` / `-Y`, 0..50 bases off the 5' / 3' end of every read before the length filter) and its
// sampling (`-#<n>`: of the file's reads / pairs, counted from first_unit for the first of this call, every n-th is loaded)
extern "C" int k4_prepare_reads_trim_dev(k4_index* ix, int pe, int64_t n, int32_t min_len, int32_t max_len, int32_t trim5, int32_t trim3,
                                         int32_t sample_nth, int64_t first_unit, const void* d_offs1, const void* d_lens1, const void* d_offs2, const void* d_lens2,
                                         uint64_t reads2_base, void* d_offs_out, void* d_lens_out, uint64_t* n_under, uint64_t* n_over,
                                         uint32_t* max_read_len, void* stream) {
  if (!ix || n < 0 || trim5 < 0 || trim3 < 0 || sample_nth < 0 || first_unit < 0) return K4_ERR_PARAMS;
  if (n_under) *n_under = 0;
  if (n_over) *n_over = 0;
  if (max_read_len) *max_read_len = 0;
  if (n == 0) return K4_OK;
  if (!d_offs1 || !d_lens1 || !d_offs_out || !d_lens_out || (pe && (!d_offs2 || !d_lens2))) return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  K4_HIP(ix, hipSetDevice(ix->device));
  hipStream_t st = (hipStream_t)stream;
  PoolStream pool_scope(st);
  Buf tot;
  K4_HIP(ix, tot.alloc(24));
  K4_HIP(ix, hipMemsetAsync(tot.p, 0, 24, st));
  hipLaunchKernelGGL(k4k_prepare_reads, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, st, pe ? 1 : 0, n,
                     (uint32_t)std::max(min_len, 0), (uint32_t)std::max(max_len, 0), (const uint64_t*)d_offs1, (const uint32_t*)d_lens1,
                     (const uint64_t*)d_offs2, (const uint32_t*)d_lens2, reads2_base, (uint64_t*)d_offs_out, (uint32_t*)d_lens_out,
                     tot.as<unsigned long long>(), (uint32_t)trim5, (uint32_t)trim3, (uint32_t)sample_nth, first_unit);
  unsigned long long t[3];
  K4_HIP(ix, hipMemcpyAsync(t, tot.p, 24, hipMemcpyDeviceToHost, st));
  K4_HIP(ix, hipStreamSynchronize(st));
  if (n_under) *n_under = t[0];
  if (n_over) *n_over = t[1];
  if (max_read_len) *max_read_len = (uint32_t)t[2];
  return K4_OK;
}

// ---- SAM ----------------------------------------------------------------------------------------------------------------
namespace {

struct K4SamArgs {
  int pe;                       // 0: SE (rr + hits), 1: PE (k4_pe_read per read, reads interleaved)
  int64_t n_reads;              // SE reads or 2 * pairs
  const k4_read_result* rr;
  const k4_hit* hits;
  int max_ml;
  const k4_pe_read* pr;
  const k4_seg2* seg2;          // SE: second segments of microInDel / splice hits, one per read, or null
  const uint8_t* reads;
  const uint64_t* offs;
  const uint32_t* lens;
  const uint8_t* text[2];
  const uint64_t* name_off[2];
  const uint32_t* name_len[2];
  const char* cnames;           // n_entries x K4_SAM_NAME_STRIDE
  const uint8_t* cname_len;
  uint32_t n_entries;
  const int32_t* refid;         // BAM: chromosome id - 1 -> index in the header's reference dictionary
  int all_reads;                // `-M1` (eFMsamAll): the reads that were not accepted are reported too, as unaligned records
  int quals;                    // kalign -g0..2: bits 4..7 of the read bytes hold 4-bit scores; QUAL is written from them (KAligner.cpp:6120-6145)
};
#define K4_SAM_NAME_STRIDE 96

// does read i carry any non-zero score?  (SumScores of ReportBAMread: none -> QUAL `*` / 0xff)
K4_DEV bool k4d_sam_has_qual(const K4SamArgs& a, int64_t i) {
  if (!a.quals) return false;
  const uint8_t* s = a.reads + a.offs[i];
  const uint32_t len = a.lens[i];
  uint32_t acc = 0;
  for (uint32_t q = 0; q < len; q++) acc |= s[q];
  return (acc & 0xF0u) != 0;
}
// A SAM line is addressed by v = read * vm + instance (vm = max_ml for SE, 1 for PE): with MLMode eMLall a read with
// NumHits instances yields that many lines (CKAligner::WriteHitLoci, KAligner.cpp:6922-6990).
K4_DEV int k4d_sam_nar(const K4SamArgs& a, int64_t i) { return a.pe ? a.pr[i].nar : a.rr[i].nar; }
K4_DEV int64_t k4d_sam_read(const K4SamArgs& a, int64_t v) { return a.pe ? v : v / a.max_ml; }
K4_DEV k4_hit k4d_sam_hit(const K4SamArgs& a, int64_t v) { return a.pe ? a.pr[v].hit : a.hits[v]; }
K4_DEV bool k4d_sam_reported(const K4SamArgs& a, int64_t v) {
  if (a.pe) return a.pr[v].nar == K4_NAR_ACCEPTED || (a.all_reads && a.lens[v] != 0);
  const int64_t i = v / a.max_ml;
  const k4_read_result r = a.rr[i];
  if (r.nar != K4_NAR_ACCEPTED) return a.all_reads && v == i * a.max_ml && a.lens[i] != 0;  // one record, if the read was loaded
  return (int)(v - i * a.max_ml) < max(r.num_hits, 1);
}
// a line of a read that was not accepted (only with all_reads)
K4_DEV bool k4d_sam_unaligned(const K4SamArgs& a, int64_t v) { return a.all_reads && k4d_sam_nar(a, k4d_sam_read(a, v)) != K4_NAR_ACCEPTED; }
// the unaligned record of `-M1` (ReportBAMread's last branch, KAligner.cpp:6253-6276, as CSAMfile::AddAlignment prints it):
//   QNAME FLAG * 0 128 <len>M * 0 0 SEQ * <empty> YU:Z:<NAR code>     FLAG: 4, PE: 1 | 2 | 64 / 128 | 4 and the mate's 8 or 32
K4_DEV uint32_t k4d_sam_unaligned_flag(const K4SamArgs& a, int64_t i) {
  if (!a.pe) return 0x4u;
  const k4_pe_read me = a.pr[i], mt = a.pr[i ^ 1];
  uint32_t f = 0x1u | 0x2u | ((i & 1) ? 0x80u : 0x40u) | 0x4u;
  if (me.pe_aligned && mt.pe_aligned && mt.nar == K4_NAR_ACCEPTED) f |= mt.hit.strand != '+' ? 0x20u : 0u;
  else f |= 0x8u;
  return f;
}
#define K4_SAM_UNALIGNED_TAIL 12  // "\t*\t\tYU:Z:xx\n"
__device__ const char k4_nar_codes[] = "NAAAENNLMHMLETOJOMDPDSFCPRUIOIUPISITNPLC";  // m_NARdesc, KAligner.cpp:48-67

struct K4SamFields {
  uint32_t flag, pos, mapq, pnext;
  uint32_t aligned;  // AdjAlignHitLen: aligned bases of both segments
  int32_t tlen;
  bool mate_eq;
  uint32_t n_ops;
  uint32_t op_len[7];  // CIGAR, at most S M S  gap  S M S (ReportBAMread, KAligner.cpp:6148-6225)
  char op[7];
};
// CKAligner::AdjAlignStartLoci / AdjAlignHitLen / AdjStartLoci / AdjHitLen (KAligner.cpp:1633-1693) on the flat records
K4_DEV uint32_t k4d_adj_start(const k4_hit& h) { return h.match_loci + (h.strand == '+' ? K4_HIT_TRIM_LEFT(h) : K4_HIT_TRIM_RIGHT(h)); }
K4_DEV uint32_t k4d_adj_len0(const k4_hit& h) { return (uint32_t)h.match_len - K4_HIT_TRIM_LEFT(h) - K4_HIT_TRIM_RIGHT(h); }
K4_DEV bool k4d_two_segs(const k4_hit& h) { return (h.ext & (K4_EXT_INDEL | K4_EXT_SPLICE)) != 0; }
K4_DEV k4_seg2 k4d_sam_seg2(const K4SamArgs& a, int64_t i, const k4_hit& h) {
  k4_seg2 z = {0, 0, 0, 0, 0, 0, 0};
  return (!a.pe && a.seg2 && k4d_two_segs(h)) ? a.seg2[i] : z;
}
// ReportBAMread (KAligner.cpp:6041-6251): FLAG, POS, CIGAR, MAPQ = max(1, M * aligned / readlen) with M = 254, less 20 for
// a splice junction and 10 for a microInDel, mate fields
K4_DEV K4SamFields k4d_sam_fields(const K4SamArgs& a, int64_t v, const k4_hit& h) {
  const int64_t i = k4d_sam_read(a, v);
  K4SamFields f;
  const uint32_t rl = a.lens[i];
  const bool two = !a.pe && a.seg2 && k4d_two_segs(h);
  const k4_seg2 s2 = k4d_sam_seg2(a, i, h);
  const uint32_t tl = K4_HIT_TRIM_LEFT(h), tr = K4_HIT_TRIM_RIGHT(h);
  const uint32_t len0 = k4d_adj_len0(h), len1 = two ? s2.match_len : 0u;
  f.pos = k4d_adj_start(h) + 1;
  f.aligned = len0 + len1;
  f.n_ops = 0;
  const uint32_t lead = h.strand == '+' ? tl : tr, trail = h.strand == '+' ? tr : tl;
  if (lead) { f.op_len[f.n_ops] = lead; f.op[f.n_ops++] = 'S'; }
  f.op_len[f.n_ops] = len0; f.op[f.n_ops++] = 'M';
  if (trail) { f.op_len[f.n_ops] = trail; f.op[f.n_ops++] = 'S'; }
  int mq0 = 254;
  if (two) {
    const int gap_t = (int)s2.match_loci - (int)(h.match_loci + h.match_len);
    if (h.ext & K4_EXT_SPLICE) { mq0 -= 20; f.op_len[f.n_ops] = (uint32_t)gap_t; f.op[f.n_ops++] = 'N'; }
    else {
      mq0 -= 10;
      if (h.ext & K4_EXT_INSERT) { f.op_len[f.n_ops] = rl - ((uint32_t)h.match_len + s2.match_len); f.op[f.n_ops++] = 'I'; }
      else { f.op_len[f.n_ops] = (uint32_t)(gap_t < 0 ? -gap_t : gap_t); f.op[f.n_ops++] = 'D'; }
    }
    f.op_len[f.n_ops] = len1; f.op[f.n_ops++] = 'M';
  }
  int mq = (int)(mq0 * ((double)(len0 + len1) / (double)rl));
  mq = mq < 1 ? 1 : mq > 254 ? 254 : mq;
  f.mapq = (uint32_t)mq;
  f.pnext = 0;
  f.tlen = 0;
  f.mate_eq = false;
  if (!a.pe) {
    f.flag = h.strand == '+' ? 0u : 0x10u;
    return f;
  }
  const k4_pe_read me = a.pr[i], mt = a.pr[i ^ 1];
  f.flag = 0x1u | 0x2u | ((i & 1) ? 0x80u : 0x40u) | (h.strand == '+' ? 0u : 0x10u);
  if (me.pe_aligned && mt.pe_aligned && mt.nar == K4_NAR_ACCEPTED) {
    if (mt.hit.strand != '+') f.flag |= 0x20u;
    f.mate_eq = true;
    f.pnext = k4d_adj_start(mt.hit) + 1;
    const int64_t s = k4d_adj_start(h), e = k4d_adj_start(mt.hit);
    f.tlen = (int32_t)(s <= e ? (e - s) + k4d_adj_len0(mt.hit) : (s - e) + len0);
  } else
    f.flag |= 0x8u;
  return f;
}
K4_DEV uint32_t k4d_udigits(uint32_t v) {
  uint32_t d = 1;
  while (v >= 10) { v /= 10; d++; }
  return d;
}
K4_DEV uint32_t k4d_put_uint(char* p, uint32_t v) {  // returns digits written
  const uint32_t d = k4d_udigits(v);
  for (uint32_t q = d; q-- > 0;) { p[q] = (char)('0' + v % 10); v /= 10; }
  return d;
}

struct IsAccepted {
  K4SamArgs a;
  __device__ bool operator()(uint32_t v) const { return k4d_sam_reported(a, v); }
};

// secondary key of SortHitMatch (len, strand, mismatches) and primary key (chrom, start)
__global__ void __launch_bounds__(256) k4k_sam_key_minor(K4SamArgs a, const uint32_t* __restrict__ idx, uint64_t m, uint32_t* __restrict__ key) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const int64_t v = idx[j];
  if (k4d_sam_unaligned(a, v)) { key[j] = 0; return; }  // (SortHitMatch leaves them in no defined order within a NAR: load order here)
  const k4_hit h = k4d_sam_hit(a, v);
  // AdjHitLen(Seg[0]), Strand, then the READ's LowMMCnt (both segments' mismatches for a two-segment hit)
  const int64_t i = k4d_sam_read(a, v);
  const uint32_t mm = a.pe ? h.mismatches : (k4d_two_segs(h) ? (uint32_t)a.rr[i].low_mm & 0xFFu : h.mismatches);
  key[j] = (k4d_adj_len0(h) << 16) | ((uint32_t)h.strand << 8) | mm;
}
__global__ void __launch_bounds__(256) k4k_sam_key_major(K4SamArgs a, const uint32_t* __restrict__ idx, uint64_t m, uint64_t* __restrict__ key) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  if (k4d_sam_unaligned(a, idx[j])) {  // behind every accepted alignment, by NAR (SortHitMatch: NAR first; accepted is the lowest in use)
    const int nar = k4d_sam_nar(a, k4d_sam_read(a, idx[j]));
    key[j] = (uint64_t)(a.n_entries + 1u + (uint32_t)(nar & 31)) << 32;
    return;
  }
  const k4_hit h = k4d_sam_hit(a, idx[j]);
  key[j] = ((uint64_t)h.chrom_id << 32) | k4d_adj_start(h);
}

// statistics over every read (ReportAlignStats, KAligner.cpp:3600-3830): NAR histogram [0..20), '+' [20], '-' [21]
__global__ void __launch_bounds__(256) k4k_sam_stats(K4SamArgs a, unsigned long long* __restrict__ st, uint8_t* __restrict__ chrom_hit) {
  __shared__ unsigned int h[22];
  if (threadIdx.x < 22) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_reads; i += stride) {
    if (a.lens[i] == 0) continue;  // a slot whose read was not loaded (under / over length)
    const int nar = k4d_sam_nar(a, i);
    atomicAdd(&h[nar >= 0 && nar < 20 ? nar : 0], 1u);
    if (nar == K4_NAR_ACCEPTED) {
      const int vm = a.pe ? 1 : a.max_ml;
      for (int q = 0; q < vm; q++) {
        if (!k4d_sam_reported(a, i * vm + q)) break;
        const k4_hit hh = k4d_sam_hit(a, i * vm + q);
        atomicAdd(&h[hh.strand == '+' ? 20 : 21], 1u);
        if (chrom_hit && hh.chrom_id <= a.n_entries) chrom_hit[hh.chrom_id] = 1;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 22 && h[threadIdx.x]) atomicAdd(&st[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

K4_DEV uint32_t k4d_sam_line_len(const K4SamArgs& a, int64_t v) {
  if (k4d_sam_unaligned(a, v)) {
    const int64_t i = k4d_sam_read(a, v);
    const int w = a.pe ? (int)(i & 1) : 0;
    const int64_t rec = a.pe ? (i >> 1) : i;
    return a.name_len[w][rec] + 1 + k4d_udigits(k4d_sam_unaligned_flag(a, i)) + 1 + 2 /* "*\t" */ + 2 /* "0\t" */ + 4 /* "128\t" */ +
           k4d_udigits(a.lens[i]) + 2 /* "M\t" */ + 2 /* "*\t" */ + 2 + 2 /* "0\t0\t" */ + a.lens[i] + K4_SAM_UNALIGNED_TAIL +
           (k4d_sam_has_qual(a, i) ? a.lens[i] - 1 : 0);
  }
  const k4_hit h = k4d_sam_hit(a, v);
  const K4SamFields f = k4d_sam_fields(a, v, h);
  const int64_t i = k4d_sam_read(a, v);
  const int w = a.pe ? (int)(i & 1) : 0;
  const int64_t rec = a.pe ? (i >> 1) : i;
  uint32_t n = a.name_len[w][rec] + 1 + k4d_udigits(f.flag) + 1 + a.cname_len[h.chrom_id - 1] + 1 + k4d_udigits(f.pos) + 1 +
               k4d_udigits(f.mapq) + 1 + 1 + 1 /*RNEXT*/ + 1 + k4d_udigits(f.pnext) + 1;
  for (uint32_t q = 0; q < f.n_ops; q++) n += k4d_udigits(f.op_len[q]) + 1;
  n += (f.tlen < 0 ? 1 : 0) + k4d_udigits((uint32_t)(f.tlen < 0 ? -(int64_t)f.tlen : f.tlen)) + 1;
  n += a.lens[i] + 1 + (k4d_sam_has_qual(a, i) ? a.lens[i] : 1 /* '*' */) + 1 /* '\n' */;
  return n;
}
// the same alignment as a BAM record (CSAMfile::AddAlignment's BAM branch, SAMfile.cpp:2379-2640, over ReportBAMread's fields):
// block_size | refID pos bin_mq_nl flag_nc l_seq next_refID next_pos tlen | read_name\0 | cigar | seq (4-bit) | qual (0xff: none)
#define K4_BAM_UNALIGNED_AUX 6  // 'Y' 'U' 'Z' + the two-letter NAR code + NUL
K4_DEV uint32_t k4d_bam_n_ops(const K4SamArgs& a, int64_t v, const k4_hit& h) {
  const uint32_t tl = K4_HIT_TRIM_LEFT(h), tr = K4_HIT_TRIM_RIGHT(h);
  const bool two = !a.pe && a.seg2 && k4d_two_segs(h);
  return 1u + (tl ? 1u : 0u) + (tr ? 1u : 0u) + (two ? 2u : 0u);
}
K4_DEV uint32_t k4d_bam_rec_len(const K4SamArgs& a, int64_t v) {
  const k4_hit h = k4d_sam_hit(a, v);
  const int64_t i = k4d_sam_read(a, v);
  const int w = a.pe ? (int)(i & 1) : 0;
  const int64_t rec = a.pe ? (i >> 1) : i;
  const uint32_t len = a.lens[i];
  if (k4d_sam_unaligned(a, v)) return 4u + 32u + a.name_len[w][rec] + 1u + 4u + (len + 1) / 2 + len + K4_BAM_UNALIGNED_AUX;
  return 4u + 32u + a.name_len[w][rec] + 1u + 4u * k4d_bam_n_ops(a, v, h) + (len + 1) / 2 + len;
}
template <bool BAM>
__global__ void __launch_bounds__(256) k4k_sam_line_lens(K4SamArgs a, const uint32_t* __restrict__ order, uint64_t m, uint32_t* __restrict__ ll) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < m) ll[j] = BAM ? k4d_bam_rec_len(a, order[j]) : k4d_sam_line_len(a, order[j]);
}
// calculate bin given an alignment covering [beg, end) (SAM specification; CKAligner::BAMreg2bin, KAligner.cpp:5929-5938)
K4_DEV uint32_t k4d_bam_reg2bin(int32_t beg, int32_t end) {
  --end;
  if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (beg >> 14);
  if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (beg >> 17);
  if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (beg >> 20);
  if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (beg >> 23);
  if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (beg >> 26);
  return 0;
}
K4_DEV void k4d_put_le32(char* p, uint32_t v) { p[0] = (char)v; p[1] = (char)(v >> 8); p[2] = (char)(v >> 16); p[3] = (char)(v >> 24); }
// one record (or sub-lane `sub`'s share of its sequence and quality bytes) at `line`
K4_DEV void k4d_bam_put_rec(const K4SamArgs& a, int64_t v, const k4_hit& h, int64_t i, char* line, uint32_t line_len, int sub, int lpl) {
  const uint32_t len = a.lens[i];
  const uint32_t nseq = (len + 1) / 2;
  const bool unal = k4d_sam_unaligned(a, v);
  const uint32_t aux = unal ? K4_BAM_UNALIGNED_AUX : 0u;
  char* seq = line + line_len - aux - len - nseq;
  char* qual = line + line_len - aux - len;
  if (sub == 0 && unal) {  // ReportBAMread's unaligned branch (KAligner.cpp:6253-6276): refID / pos / mate -1, bin 0, MAPQ 128, <len>M, YU:Z:<NAR>
    const int w = a.pe ? (int)(i & 1) : 0;
    const int64_t rec = a.pe ? (i >> 1) : i;
    const uint32_t nl_ = a.name_len[w][rec];
    k4d_put_le32(line, line_len - 4);
    k4d_put_le32(line + 4, 0xFFFFFFFFu);
    k4d_put_le32(line + 8, 0xFFFFFFFFu);
    k4d_put_le32(line + 12, (128u << 8) | (nl_ + 1));
    k4d_put_le32(line + 16, (k4d_sam_unaligned_flag(a, i) << 16) | 1u);
    k4d_put_le32(line + 20, len);
    k4d_put_le32(line + 24, 0xFFFFFFFFu);
    k4d_put_le32(line + 28, 0xFFFFFFFFu);
    k4d_put_le32(line + 32, 0u);
    char* p = line + 36;
    const uint8_t* nm = a.text[w] + a.name_off[w][rec];
    for (uint32_t q = 0; q < nl_; q++) p[q] = (char)nm[q];
    p[nl_] = 0;
    k4d_put_le32(p + nl_ + 1, len << 4);
    const int nar = k4d_sam_nar(a, i);
    const int code = nar >= 0 && nar < 20 ? nar : 0;
    char* t = line + line_len - K4_BAM_UNALIGNED_AUX;
    t[0] = 'Y'; t[1] = 'U'; t[2] = 'Z'; t[3] = k4_nar_codes[2 * code]; t[4] = k4_nar_codes[2 * code + 1]; t[5] = 0;
  } else if (sub == 0) {
    const K4SamFields f = k4d_sam_fields(a, v, h);
    const int w = a.pe ? (int)(i & 1) : 0;
    const int64_t rec = a.pe ? (i >> 1) : i;
    const uint32_t nl_ = a.name_len[w][rec];
    const int32_t refid = a.refid[h.chrom_id - 1];
    const int32_t pos0 = (int32_t)f.pos - 1;
    k4d_put_le32(line, line_len - 4);
    k4d_put_le32(line + 4, (uint32_t)refid);
    k4d_put_le32(line + 8, (uint32_t)pos0);
    k4d_put_le32(line + 12, (k4d_bam_reg2bin(pos0, pos0 + (int32_t)f.aligned) << 16) | (f.mapq << 8) | (nl_ + 1));
    k4d_put_le32(line + 16, (f.flag << 16) | f.n_ops);
    k4d_put_le32(line + 20, len);
    k4d_put_le32(line + 24, f.mate_eq ? (uint32_t)refid : 0xFFFFFFFFu);
    k4d_put_le32(line + 28, f.mate_eq ? f.pnext - 1 : 0xFFFFFFFFu);
    k4d_put_le32(line + 32, (uint32_t)f.tlen);
    char* p = line + 36;
    const uint8_t* nm = a.text[w] + a.name_off[w][rec];
    for (uint32_t q = 0; q < nl_; q++) p[q] = (char)nm[q];
    p[nl_] = 0;
    p += nl_ + 1;
    for (uint32_t q = 0; q < f.n_ops; q++) {
      const char c = f.op[q];
      const uint32_t code = c == 'M' ? 0u : c == 'I' ? 1u : c == 'D' ? 2u : c == 'N' ? 3u : 4u;
      k4d_put_le32(p, (f.op_len[q] << 4) | code);
      p += 4;
    }
  }
  // sequence: `=ACMGRSVTWYHKDBN' codes, first base in the high nibble, reverse complemented for a Crick alignment (:6254-6300)
  const uint8_t* s = a.reads + a.offs[i];
  const uint32_t b0 = (uint32_t)((uint64_t)nseq * sub / lpl), b1 = (uint32_t)((uint64_t)nseq * (sub + 1) / lpl);
  const uint64_t fwd = 0x0F0F0F0F08040201ull, rev = 0x0F0F0F0F01020408ull;  // by symbol: A C G T N.. / their complements
  for (uint32_t b = b0; b < b1; b++) {
    uint32_t hi, lo = 0;
    if (unal || h.strand == '+') {
      hi = (uint32_t)(fwd >> (8 * (s[2 * b] & 7))) & 0xF;
      if (2 * b + 1 < len) lo = (uint32_t)(fwd >> (8 * (s[2 * b + 1] & 7))) & 0xF;
    } else {
      hi = (uint32_t)(rev >> (8 * (s[len - 1 - 2 * b] & 7))) & 0xF;
      if (2 * b + 1 < len) lo = (uint32_t)(rev >> (8 * (s[len - 2 - 2 * b] & 7))) & 0xF;
    }
    seq[b] = (char)((hi << 4) | lo);
  }
  const uint32_t q0 = (uint32_t)((uint64_t)len * sub / lpl), q1 = (uint32_t)((uint64_t)len * (sub + 1) / lpl);
  if (k4d_sam_has_qual(a, i)) {  // the reference stores the SAM characters themselves in the BAM record (SAMfile.cpp:2468 copies pBAMalign->qual)
    const uint8_t* rs = a.reads + a.offs[i];
    const bool rv = !unal && h.strand != '+';
    for (uint32_t q = q0; q < q1; q++) qual[q] = (char)(33 + (((uint32_t)(rs[rv ? len - 1 - q : q] >> 4) & 0x0f) * 40) / 15);
  } else
    for (uint32_t q = q0; q < q1; q++) qual[q] = (char)0xFF;
}

// QNAME FLAG RNAME POS MAPQ <len>M RNEXT PNEXT TLEN SEQ * (AddAlignment, SAMfile.cpp:2194-2377).  The lines of the sorted
// order are consecutive in the output, so a wave takes a tile of 64 / LPL of them, LPL lanes per line write the line
// into the wave's LDS buffer (sub-lane 0 the fields, all of them a share of SEQ), and the wave then streams the tile out
// with 16-byte stores (the buffer is offset so that LDS and global addresses agree modulo 16).  A tile that does not
// fit (very long reads or names) is written straight to global memory by the same code.
#define K4_SAM_WAVE_BUF 12288
// four bytes of a read starting at any byte offset p >= 0 (aligned word loads, as k4d_pack_read: never outside the 4-byte
// cells the read occupies); bytes past `len` come back as whatever the cell holds or zero
K4_DEV uint32_t k4d_read4(const uint32_t* __restrict__ s32, uint32_t sh, uint32_t span, uint32_t p) {
  const uint32_t bo = p + sh, wi = bo >> 2, r = bo & 3;
  const uint32_t lo = s32[wi];
  const uint32_t hi = (r && 4 * (wi + 1) < span) ? s32[wi + 1] : 0u;
  return __builtin_amdgcn_alignbyte(hi, lo, r);
}
// one line (or sub-lane `sub`'s share of it) at `line`; inlined once with an LDS and once with a global destination
K4_DEV void k4d_sam_put_line(const K4SamArgs& a, int64_t v, const k4_hit& h, int64_t i, char* line, uint32_t line_len, int sub,
                             int lpl) {
  const uint32_t len = a.lens[i];
  const bool unal = k4d_sam_unaligned(a, v);
  const bool hq = k4d_sam_has_qual(a, i);  // QUAL holds len characters instead of `*`
  char* seq = line + line_len - (unal ? K4_SAM_UNALIGNED_TAIL : 3) - len - (hq ? len - 1 : 0);
  if (sub == 0 && unal) {
    const int w = a.pe ? (int)(i & 1) : 0;
    const int64_t rec = a.pe ? (i >> 1) : i;
    char* p = line;
    const uint8_t* nm = a.text[w] + a.name_off[w][rec];
    const uint32_t nl_ = a.name_len[w][rec];
    for (uint32_t q = 0; q < nl_; q++) p[q] = (char)nm[q];
    p += nl_;
    *p++ = '\t'; p += k4d_put_uint(p, k4d_sam_unaligned_flag(a, i));
    const char mid[] = "\t*\t0\t128\t";
    for (int q = 0; q < 9; q++) *p++ = mid[q];
    p += k4d_put_uint(p, len);
    const char mid2[] = "M\t*\t0\t0\t";
    for (int q = 0; q < 8; q++) *p++ = mid2[q];
    const int nar = k4d_sam_nar(a, i);
    const int code = nar >= 0 && nar < 20 ? nar : 0;
    char* t = seq + len;
    t[0] = '\t';
    if (!hq) t[1] = '*';
    t += hq ? len - 1 : 0;  // (the scores themselves: below, with the bases)
    t[2] = '\t'; t[3] = '\t'; t[4] = 'Y'; t[5] = 'U'; t[6] = ':'; t[7] = 'Z'; t[8] = ':';
    t[9] = k4_nar_codes[2 * code]; t[10] = k4_nar_codes[2 * code + 1]; t[11] = '\n';
  } else if (sub == 0) {
    const K4SamFields f = k4d_sam_fields(a, v, h);
    const int w = a.pe ? (int)(i & 1) : 0;
    const int64_t rec = a.pe ? (i >> 1) : i;
    char* p = line;
    const uint8_t* nm = a.text[w] + a.name_off[w][rec];
    const uint32_t nl_ = a.name_len[w][rec];
    for (uint32_t q = 0; q < nl_; q++) p[q] = (char)nm[q];
    p += nl_;
    *p++ = '\t'; p += k4d_put_uint(p, f.flag); *p++ = '\t';
    const char* cn = a.cnames + (size_t)(h.chrom_id - 1) * K4_SAM_NAME_STRIDE;
    const uint32_t cl = a.cname_len[h.chrom_id - 1];
    for (uint32_t q = 0; q < cl; q++) p[q] = cn[q];
    p += cl;
    *p++ = '\t'; p += k4d_put_uint(p, f.pos);
    *p++ = '\t'; p += k4d_put_uint(p, f.mapq);
    *p++ = '\t';
    for (uint32_t q = 0; q < f.n_ops; q++) { p += k4d_put_uint(p, f.op_len[q]); *p++ = f.op[q]; }
    *p++ = '\t'; *p++ = f.mate_eq ? '=' : '*';
    *p++ = '\t'; p += k4d_put_uint(p, f.pnext);
    *p++ = '\t';
    if (f.tlen < 0) *p++ = '-';
    p += k4d_put_uint(p, (uint32_t)(f.tlen < 0 ? -(int64_t)f.tlen : f.tlen));
    *p++ = '\t';
    seq[len] = '\t';
    if (!hq) seq[len + 1] = '*';
    seq[len + 1 + (hq ? len : 1)] = '\n';
  }
  const uint8_t* s = a.reads + a.offs[i];
  const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(s) & 3);
  const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(s - sh);
  const uint32_t span = len + sh;
  const uint32_t q0 = (uint32_t)((uint64_t)len * sub / lpl), q1 = (uint32_t)((uint64_t)len * (sub + 1) / lpl);
  const uint64_t fwd = 0x4E4E4E4E54474341ull, rev = 0x4E4E4E4E41434754ull;  // "ACGTNNNN" / "TGCANNNN" by symbol (:6279)
  if (hq) {  // '!' + score * 40 / 15, in read order -- reversed with the bases of a Crick alignment (KAligner.cpp:6120-6145)
    char* qv = seq + len + 1;
    const bool rv = !unal && h.strand != '+';
    for (uint32_t q = q0; q < q1; q++) qv[q] = (char)(33 + (((uint32_t)(s[rv ? len - 1 - q : q] >> 4) & 0x0f) * 40) / 15);
  }
  if (unal || h.strand == '+') {
    for (uint32_t q = q0; q < q1; q += 4) {
      const uint32_t d = k4d_read4(s32, sh, span, q);
#pragma unroll
      for (uint32_t k = 0; k < 4; k++)
        if (q + k < q1) seq[q + k] = (char)(fwd >> (8 * ((d >> (8 * k)) & 7)));
    }
  } else {
    for (uint32_t q = q0; q < q1; q += 4) {  // seq[q + k] = complement of s[len - 1 - q - k]
      const uint32_t top = len - 1 - q;      // highest source byte of this group
      if (top >= 3) {
        const uint32_t d = k4d_read4(s32, sh, span, top - 3);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
          if (q + k < q1) seq[q + k] = (char)(rev >> (8 * ((d >> (8 * (3 - k))) & 7)));
      } else {
        for (uint32_t k = 0; k <= top && q + k < q1; k++) seq[q + k] = (char)(rev >> (8 * (s[top - k] & 7)));
      }
    }
  }
}

template <int LPL, bool BAM>
__global__ void __launch_bounds__(256) k4k_sam_write(K4SamArgs a, const uint32_t* __restrict__ order, const uint64_t* __restrict__ loff,
                                                     uint64_t m, char* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) char sm[4][K4_SAM_WAVE_BUF];
  constexpr int TL = 64 / LPL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane / LPL, sub = lane % LPL;
  char* buf = sm[wave];
  for (uint64_t jb = (uint64_t)blockIdx.x * 4 * TL; jb < m; jb += (uint64_t)gridDim.x * 4 * TL) {  // same trip count for the block
    const uint64_t j0 = jb + (uint64_t)wave * TL;
    const int nl = j0 < m ? (int)min((uint64_t)TL, m - j0) : 0;
    uint64_t b0 = 0, b1 = 0;
    if (nl) { b0 = loff[j0]; b1 = loff[j0 + nl]; }
    const uint32_t pad = (uint32_t)(b0 & 15);
    const bool fits = nl && (b1 - b0) + pad <= (uint64_t)K4_SAM_WAVE_BUF;
    if (li < nl) {
      const uint64_t j = j0 + li;
      const int64_t v = order[j];
      const k4_hit h = k4d_sam_hit(a, v);
      const int64_t i = k4d_sam_read(a, v);
      const uint64_t l0 = loff[j], l1 = loff[j + 1];
      char* dst_line = fits ? &sm[wave][pad + (uint32_t)(l0 - b0)] : out + l0;
      if (BAM) k4d_bam_put_rec(a, v, h, i, dst_line, (uint32_t)(l1 - l0), sub, LPL);
      else k4d_sam_put_line(a, v, h, i, dst_line, (uint32_t)(l1 - l0), sub, LPL);
    }
    __syncthreads();
    if (fits) {
      const uint32_t total = (uint32_t)(b1 - b0);
      const uint32_t head = pad ? min(16u - pad, total) : 0u;
      char* dst = out + b0;
      const char* src = buf + pad;
      if ((uint32_t)lane < head) dst[lane] = src[lane];
      const uint32_t body = (total - head) / 16;
      const uint4* s4 = reinterpret_cast<const uint4*>(src + head);
      uint4* d4 = reinterpret_cast<uint4*>(dst + head);
      for (uint32_t q = lane; q < body; q += 64) d4[q] = s4[q];
      const uint32_t done = head + body * 16;
      if ((uint32_t)lane < total - done) dst[done + lane] = src[done + lane];
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" void k4_free_device(void* p) { if (p) hipFree(p); }
// thin device-memory helpers so that a host program over this ABI needs no HIP of its own
extern "C" int k4_alloc_device(k4_index* ix, uint64_t bytes, void** p) {
  if (!ix || !p) return K4_ERR_PARAMS;
  *p = nullptr;
  K4_HIP(ix, hipSetDevice(ix->device));
  K4_HIP(ix, k4_malloc_retry(p, bytes ? bytes : 1));
  return K4_OK;
}
extern "C" int k4_copy_to_device(k4_index* ix, void* d_dst, const void* src, uint64_t bytes) {
  if (!ix || (bytes && (!d_dst || !src))) return K4_ERR_PARAMS;
  K4_HIP(ix, hipSetDevice(ix->device));
  K4_HIP(ix, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
  return K4_OK;
}
extern "C" int k4_copy_to_host(k4_index* ix, void* dst, const void* d_src, uint64_t bytes) {
  if (!ix || (bytes && (!dst || !d_src))) return K4_ERR_PARAMS;
  K4_HIP(ix, hipSetDevice(ix->device));
  K4_HIP(ix, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
  return K4_OK;
}

// MLMode eMLrand (`-r2`, KAligner.cpp:9945-9962): a read within the instance limit reports ONE of its instances, the one the
// caller's random stream names; the caller draws in load order (one draw per such read, unique ones included, as the
// reference does with one thread) and passes the draws, this moves the chosen instance into slot 0.
__global__ void k4k_select_hits(int64_t n, int32_t max_ml, k4_read_result* __restrict__ rr, k4_hit* __restrict__ hits,
                                const uint32_t* __restrict__ choice) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    k4_read_result r = rr[i];
    if (r.nar != K4_NAR_ACCEPTED || r.num_hits < 1) continue;
    const uint32_t c = choice[i] % (uint32_t)r.num_hits;
    if (c) hits[i * max_ml] = hits[i * max_ml + c];
    rr[i].num_hits = 1;
  }
}

// ---- kalign -j / -J: the reads that found no alignment, or too many, as FASTA ---------------------------------------------------
// CKAligner::ReportNoneAligned / ReportMultiAlign (KAligner.cpp:3833-4020): for every loaded read whose NAR is EN or NL (-j) / ML (-J),
//   >lcl|na|<ReadID> <name> <ReadID>|1|<length>      (-J: lcl|ml)         ReadID: 1.. over the loaded reads in load order
// and the read as loaded, 70 bases to the line; the reads in the order of the sorted index: by NAR, within one NAR undefined (here:
// load order).  These are a few per cent of the reads: one thread per read, no tiling.
namespace {
struct NarIs {
  K4SamArgs a;
  int nar;
  __device__ bool operator()(uint32_t i) const { return a.lens[i] != 0 && k4d_sam_nar(a, i) == nar; }
};
struct IsLoaded {
  const uint32_t* lens;
  __device__ uint32_t operator()(uint32_t i) const { return lens[i] != 0 ? 1u : 0u; }
};
K4_DEV uint32_t k4d_fasta_rec_len(const K4SamArgs& a, int64_t i, uint32_t id) {
  const int w = a.pe ? (int)(i & 1) : 0;
  const int64_t rec = a.pe ? (i >> 1) : i;
  const uint32_t len = a.lens[i];
  return 8u + k4d_udigits(id) + 1u + a.name_len[w][rec] + 1u + k4d_udigits(id) + 3u + k4d_udigits(len) + 1u + len + (len + 69u) / 70u;
}
__global__ void __launch_bounds__(256) k4k_fasta_lens(K4SamArgs a, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ ids, uint64_t m,
                                                      uint32_t* __restrict__ ll) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < m) ll[j] = k4d_fasta_rec_len(a, idx[j], ids[idx[j]] + 1);
}
__global__ void __launch_bounds__(256) k4k_fasta_write(K4SamArgs a, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ ids, uint64_t m,
                                                       const uint64_t* __restrict__ lo, int multi, char* __restrict__ out) {
  const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const int64_t i = idx[j];
  const uint32_t id = ids[i] + 1, len = a.lens[i];
  const int w = a.pe ? (int)(i & 1) : 0;
  const int64_t rec = a.pe ? (i >> 1) : i;
  char* p = out + lo[j];
  const char* head = multi ? ">lcl|ml|" : ">lcl|na|";
  for (int q = 0; q < 8; q++) *p++ = head[q];
  p += k4d_put_uint(p, id);
  *p++ = ' ';
  const uint8_t* nm = a.text[w] + a.name_off[w][rec];
  const uint32_t nl_ = a.name_len[w][rec];
  for (uint32_t q = 0; q < nl_; q++) *p++ = (char)nm[q];
  *p++ = ' ';
  p += k4d_put_uint(p, id);
  *p++ = '|'; *p++ = '1'; *p++ = '|';
  p += k4d_put_uint(p, len);
  *p++ = '\n';
  const uint8_t* s = a.reads + a.offs[i];
  for (uint32_t q = 0; q < len; q++) {
    *p++ = "ACGTNU-?"[s[q] & 7u];  // CSeqTrans::MapSeq2Ascii's defaults (N, undefined, InDel; SeqTrans.cpp:157-181)
    if (q % 70 == 69 || q + 1 == len) *p++ = '\n';
  }
}
}  // namespace

// which: 0 = the reads without an alignment (NAR EN, then NL), 1 = the multi-aligned reads (NAR ML).  *text: malloc'd host text
// (k4_free_host); *n_listed: records in it.
extern "C" int k4_unaligned_fasta_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_pe, const void* d_reads,
                                      const void* d_offs, const void* d_lens, const k4_sam_names* names, int32_t which, char** text,
                                      uint64_t* text_bytes, uint64_t* n_listed, void* stream) {
  if (!ix || !names || !text || !text_bytes || which < 0 || which > 1) return K4_ERR_PARAMS;
  *text = nullptr;
  *text_bytes = 0;
  if (n_listed) *n_listed = 0;
  if (n_units < 0) return k4_fail(ix, K4_ERR_PARAMS, "negative count");
  const int64_t n_reads = pe ? 2 * n_units : n_units;
  if (n_reads >= 0xFFFFFF00ll) return k4_fail(ix, K4_ERR_PARAMS, "at most 2^32-256 reads per call");
  auto empty = [&]() { *text = (char*)calloc(1, 1); return *text ? K4_OK : k4_fail(ix, K4_ERR_MEM, "out of memory"); };
  if (n_reads == 0) return empty();
  if ((pe && !d_pe) || (!pe && !d_rr) || !d_reads || !d_offs || !d_lens || !names->d_text[0] || !names->d_name_off[0] || !names->d_name_len[0] ||
      (pe && (!names->d_text[1] || !names->d_name_off[1] || !names->d_name_len[1])))
    return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  K4_HIP(ix, hipSetDevice(ix->device));
  hipStream_t st = (hipStream_t)stream;
  PoolStream pool_scope(st);
  K4SamArgs a;
  memset(&a, 0, sizeof(a));
  a.pe = pe ? 1 : 0; a.n_reads = n_reads; a.rr = (const k4_read_result*)d_rr; a.pr = (const k4_pe_read*)d_pe; a.max_ml = 1;
  a.reads = (const uint8_t*)d_reads; a.offs = (const uint64_t*)d_offs; a.lens = (const uint32_t*)d_lens;
  for (int w = 0; w < 2; w++) {
    a.text[w] = (const uint8_t*)names->d_text[w]; a.name_off[w] = (const uint64_t*)names->d_name_off[w];
    a.name_len[w] = (const uint32_t*)names->d_name_len[w];
  }
  Buf ids, idx, cnt, tmp, ll, lo, outb;
  K4_HIP(ix, ids.alloc((size_t)(n_reads + 1) * 4));
  K4_HIP(ix, idx.alloc((size_t)n_reads * 4));
  K4_HIP(ix, cnt.alloc(8));
  {  // ReadID - 1: the loaded reads before this one
    rocprim::counting_iterator<uint32_t> all(0);
    auto loaded = rocprim::make_transform_iterator(all, IsLoaded{a.lens});
    size_t tb = 0;
    K4_HIP(ix, rocprim::exclusive_scan(nullptr, tb, loaded, ids.as<uint32_t>(), 0u, (size_t)n_reads, rocprim::plus<uint32_t>(), st));
    K4_HIP(ix, tmp.alloc(tb));
    K4_HIP(ix, rocprim::exclusive_scan(tmp.p, tb, loaded, ids.as<uint32_t>(), 0u, (size_t)n_reads, rocprim::plus<uint32_t>(), st));
  }
  uint64_t m = 0;
  const int nars[2][2] = {{K4_NAR_NS, K4_NAR_NOHIT}, {K4_NAR_MULTIALIGN, -1}};
  for (int g = 0; g < 2; g++) {  // one group of the sorted index after the other
    if (nars[which][g] < 0) continue;
    rocprim::counting_iterator<uint32_t> all(0);
    NarIs pred{a, nars[which][g]};
    size_t tb = 0;
    K4_HIP(ix, rocprim::select(nullptr, tb, all, idx.as<uint32_t>() + m, cnt.as<uint64_t>(), (size_t)n_reads, pred, st));
    Buf t2;
    K4_HIP(ix, t2.alloc(tb));
    K4_HIP(ix, rocprim::select(t2.p, tb, all, idx.as<uint32_t>() + m, cnt.as<uint64_t>(), (size_t)n_reads, pred, st));
    uint64_t got = 0;
    K4_HIP(ix, hipMemcpyAsync(&got, cnt.p, 8, hipMemcpyDeviceToHost, st));
    K4_HIP(ix, hipStreamSynchronize(st));
    m += got;
  }
  if (n_listed) *n_listed = m;
  if (m == 0) return empty();
  K4_HIP(ix, ll.alloc((m + 1) * 4));
  K4_HIP(ix, lo.alloc((m + 1) * 8));
  K4_HIP(ix, hipMemsetAsync(ll.as<uint32_t>() + m, 0, 4, st));
  const unsigned mb = (unsigned)((m + 255) / 256);
  hipLaunchKernelGGL(k4k_fasta_lens, dim3(mb), dim3(256), 0, st, a, idx.as<uint32_t>(), ids.as<uint32_t>(), m, ll.as<uint32_t>());
  {
    size_t tb = 0;
    K4_HIP(ix, rocprim::exclusive_scan(nullptr, tb, ll.as<uint32_t>(), lo.as<uint64_t>(), (uint64_t)0, (size_t)(m + 1), rocprim::plus<uint64_t>(), st));
    Buf t3;
    K4_HIP(ix, t3.alloc(tb));
    K4_HIP(ix, rocprim::exclusive_scan(t3.p, tb, ll.as<uint32_t>(), lo.as<uint64_t>(), (uint64_t)0, (size_t)(m + 1), rocprim::plus<uint64_t>(), st));
  }
  uint64_t total = 0;
  K4_HIP(ix, hipMemcpyAsync(&total, lo.as<uint64_t>() + m, 8, hipMemcpyDeviceToHost, st));
  K4_HIP(ix, hipStreamSynchronize(st));
  K4_HIP(ix, outb.alloc(total + 16));
  hipLaunchKernelGGL(k4k_fasta_write, dim3(mb), dim3(256), 0, st, a, idx.as<uint32_t>(), ids.as<uint32_t>(), m, lo.as<uint64_t>(), (int)which, outb.as<char>());
  char* h = (char*)malloc(total + 1);
  if (!h) return k4_fail(ix, K4_ERR_MEM, "out of memory");
  int rc = k4_check_hip(ix, hipMemcpyAsync(h, outb.p, total, hipMemcpyDeviceToHost, st), "copy");
  if (rc == K4_OK) rc = k4_check_hip(ix, hipStreamSynchronize(st), "FASTA of the unaligned reads");
  if (rc != K4_OK) { free(h); return rc; }
  h[total] = 0;
  *text = h;
  *text_bytes = total;
  return K4_OK;
}


extern "C" int k4_select_hits_dev(k4_index* ix, int64_t n_reads, int32_t max_ml, void* d_rr, void* d_hits, const void* d_choice,
                                  void* stream) {
  if (!ix || n_reads < 0 || max_ml < 1 || (n_reads && (!d_rr || !d_hits || !d_choice))) return K4_ERR_PARAMS;
  if (!n_reads) return K4_OK;
  K4_HIP(ix, hipSetDevice(ix->device));
  const int64_t blocks = std::min<int64_t>((n_reads + 255) / 256, 65536);
  k4k_select_hits<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(n_reads, max_ml, (k4_read_result*)d_rr,
                                                                                  (k4_hit*)d_hits, (const uint32_t*)d_choice);
  K4_HIP(ix, hipGetLastError());
  return K4_OK;
}

extern "C" int k4_format_sam_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                                 const void* d_pe, const void* d_reads, const void* d_offs, const void* d_lens,
                                 const k4_sam_names* names, void** d_sam, uint64_t* sam_bytes, k4_sam_stats* stats,
                                 uint8_t* chrom_hit, void* stream) {
  return k4_format_sam_ext_dev(ix, pe, n_units, d_rr, d_hits, max_ml, d_pe, nullptr, d_reads, d_offs, d_lens, names, d_sam, sam_bytes,
                               stats, chrom_hit, stream);
}
extern "C" int k4_format_sam_ext_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                                     const void* d_pe, const void* d_seg2, const void* d_reads, const void* d_offs,
                                     const void* d_lens, const k4_sam_names* names, void** d_sam, uint64_t* sam_bytes,
                                     k4_sam_stats* stats, uint8_t* chrom_hit, void* stream) {
  return k4i_format_records(ix, 0, 0, pe, n_units, d_rr, d_hits, max_ml, d_pe, d_seg2, d_reads, d_offs, d_lens, names, d_sam, sam_bytes, stats,
                            chrom_hit, stream, nullptr, nullptr);
}
// `-M1` (eFMsamAll): the SAM body with the reads that were not accepted reported too, as unaligned records behind the alignments
extern "C" int k4_format_sam_all_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                                     const void* d_pe, const void* d_seg2, const void* d_reads, const void* d_offs,
                                     const void* d_lens, const k4_sam_names* names, void** d_sam, uint64_t* sam_bytes,
                                     k4_sam_stats* stats, uint8_t* chrom_hit, void* stream) {
  return k4i_format_records(ix, 2, 0, pe, n_units, d_rr, d_hits, max_ml, d_pe, d_seg2, d_reads, d_offs, d_lens, names, d_sam, sam_bytes, stats,
                            chrom_hit, stream, nullptr, nullptr);
}
// ... and the same as BAM records (refID / pos / mate -1, bin 0, MAPQ 128, <len>M, aux YU:Z:<NAR>)
extern "C" int k4_format_bam_all_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                                     const void* d_pe, const void* d_seg2, const void* d_reads, const void* d_offs,
                                     const void* d_lens, const k4_sam_names* names, int32_t sq_all, void** d_bam, uint64_t* bam_bytes,
                                     k4_sam_stats* stats, uint8_t* chrom_hit, void* stream) {
  return k4i_format_records(ix, 3, sq_all, pe, n_units, d_rr, d_hits, max_ml, d_pe, d_seg2, d_reads, d_offs, d_lens, names, d_bam, bam_bytes, stats,
                            chrom_hit, stream, nullptr, nullptr);
}
extern "C" int k4_format_bam_dev(k4_index* ix, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                                 const void* d_pe, const void* d_seg2, const void* d_reads, const void* d_offs,
                                 const void* d_lens, const k4_sam_names* names, int32_t sq_all, void** d_bam, uint64_t* bam_bytes,
                                 k4_sam_stats* stats, uint8_t* chrom_hit, void* stream) {
  return k4i_format_records(ix, 1, sq_all, pe, n_units, d_rr, d_hits, max_ml, d_pe, d_seg2, d_reads, d_offs, d_lens, names, d_bam, bam_bytes, stats,
                            chrom_hit, stream, nullptr, nullptr);
}
int k4i_format_records(k4_index* ix, int bam, int sq_all, int pe, int64_t n_units, const void* d_rr, const void* d_hits, int32_t max_ml,
                       const void* d_pe, const void* d_seg2, const void* d_reads, const void* d_offs,
                       const void* d_lens, const k4_sam_names* names, void** d_sam, uint64_t* sam_bytes,
                       k4_sam_stats* stats, uint8_t* chrom_hit, void* stream, K4SamSlices* slices, K4PoolBuf* out_buf) {
  if (!ix || !names || !d_sam || !sam_bytes) return K4_ERR_PARAMS;
  if (slices) slices->clear();
  const bool all_reads = bam >= 2;  // 2: SAM text, 3: BAM records with the reads that were not accepted behind the alignments (`-M1`)
  if (all_reads) bam -= 2;
  *d_sam = nullptr;
  *sam_bytes = 0;
  if (stats) memset(stats, 0, sizeof(*stats));
  if (chrom_hit) memset(chrom_hit, 0, ix->d.n_entries + 1);
  if (n_units < 0) return k4_fail(ix, K4_ERR_PARAMS, "negative count");
  if (n_units == 0) return K4_OK;
  const int64_t n_reads = pe ? 2 * n_units : n_units;
  const int64_t n_virt = pe ? n_reads : n_reads * (int64_t)std::max(max_ml, 1);  // addressable SAM lines
  if (n_virt >= 0xFFFFFF00ll) return k4_fail(ix, K4_ERR_PARAMS, "at most 2^32-256 reads x instances per call");
  if ((pe && !d_pe) || (!pe && (!d_rr || !d_hits || max_ml < 1)) || !d_reads || !d_offs || !d_lens || !names->d_text[0] ||
      !names->d_name_off[0] || !names->d_name_len[0] || (pe && (!names->d_text[1] || !names->d_name_off[1] || !names->d_name_len[1])))
    return k4_fail(ix, K4_ERR_PARAMS, "null buffer");
  K4_HIP(ix, hipSetDevice(ix->device));
  hipStream_t st = (hipStream_t)stream;
  PoolStream pool_scope(st);
  // chromosome names on the device
  const uint32_t ne = ix->d.n_entries;
  Buf cn, cl;
  {
    std::vector<char> hn((size_t)ne * K4_SAM_NAME_STRIDE, 0);
    std::vector<uint8_t> hl(ne);
    for (uint32_t e = 0; e < ne; e++) {
      const size_t l = strnlen(ix->entries[e].name, 80);
      memcpy(&hn[(size_t)e * K4_SAM_NAME_STRIDE], ix->entries[e].name, l);
      hl[e] = (uint8_t)l;
    }
    K4_HIP(ix, cn.alloc(hn.size()));
    K4_HIP(ix, cl.alloc(hl.size()));
    K4_HIP(ix, hipMemcpy(cn.p, hn.data(), hn.size(), hipMemcpyHostToDevice));
    K4_HIP(ix, hipMemcpy(cl.p, hl.data(), hl.size(), hipMemcpyHostToDevice));
  }
  K4SamArgs a;
  memset(&a, 0, sizeof(a));
  a.pe = pe ? 1 : 0; a.n_reads = n_reads; a.rr = (const k4_read_result*)d_rr; a.hits = (const k4_hit*)d_hits; a.max_ml = max_ml;
  a.pr = (const k4_pe_read*)d_pe; a.seg2 = pe ? nullptr : (const k4_seg2*)d_seg2; a.reads = (const uint8_t*)d_reads; a.offs = (const uint64_t*)d_offs; a.lens = (const uint32_t*)d_lens;
  for (int w = 0; w < 2; w++) {
    a.text[w] = (const uint8_t*)names->d_text[w]; a.name_off[w] = (const uint64_t*)names->d_name_off[w];
    a.name_len[w] = (const uint32_t*)names->d_name_len[w];
  }
  a.cnames = cn.as<char>(); a.cname_len = cl.as<uint8_t>(); a.n_entries = ne;
  a.all_reads = all_reads ? 1 : 0;
  a.quals = ix->q_method != 3 ? 1 : 0;

  Buf stb, chb, cnt, idx0, idx1, k32a, k32b, k64a, k64b, tmp, ll, lo;
  K4_HIP(ix, stb.alloc(22 * 8));
  K4_HIP(ix, chb.alloc(ne + 1));
  K4_HIP(ix, cnt.alloc(8));
  K4_HIP(ix, hipMemsetAsync(stb.p, 0, 22 * 8, st));
  K4_HIP(ix, hipMemsetAsync(chb.p, 0, ne + 1, st));
  hipLaunchKernelGGL(k4k_sam_stats, dim3(2048), dim3(256), 0, st, a, stb.as<unsigned long long>(), chb.as<uint8_t>());
  // accepted reads, in load order
  K4_HIP(ix, idx0.alloc((size_t)n_virt * 4));
  {
    rocprim::counting_iterator<uint32_t> all(0);
    IsAccepted pred{a};
    size_t tb = 0;
    K4_HIP(ix, rocprim::select(nullptr, tb, all, idx0.as<uint32_t>(), cnt.as<uint64_t>(), (size_t)n_virt, pred, st));
    K4_HIP(ix, tmp.alloc(tb));
    K4_HIP(ix, rocprim::select(tmp.p, tb, all, idx0.as<uint32_t>(), cnt.as<uint64_t>(), (size_t)n_virt, pred, st));
  }
  uint64_t m = 0;
  K4_HIP(ix, hipMemcpyAsync(&m, cnt.p, 8, hipMemcpyDeviceToHost, st));
  K4_HIP(ix, hipStreamSynchronize(st));
  unsigned long long hs[22];
  K4_HIP(ix, hipMemcpy(hs, stb.p, sizeof(hs), hipMemcpyDeviceToHost));
  if (stats) {
    for (int k = 0; k < 20; k++) stats->nar[k] = hs[k];
    stats->plus = hs[20];
    stats->minus = hs[21];
    stats->n_lines = m;
  }
  std::vector<uint8_t> ch_host(ne + 1, 0);
  K4_HIP(ix, hipMemcpy(ch_host.data(), chb.p, ne + 1, hipMemcpyDeviceToHost));
  if (chrom_hit) memcpy(chrom_hit, ch_host.data(), ne + 1);
  if (m == 0) return K4_OK;
  Buf refid;
  if (bam) {  // the reference dictionary of the header the caller writes: every sequence, or those with a hit, in index order
    std::vector<int32_t> map(ne, -1);
    int32_t next = 0;
    for (uint32_t e = 0; e < ne; e++)
      if (sq_all || ch_host[e + 1]) map[e] = next++;
    K4_HIP(ix, refid.alloc((size_t)ne * 4));
    K4_HIP(ix, hipMemcpy(refid.p, map.data(), (size_t)ne * 4, hipMemcpyHostToDevice));
    a.refid = refid.as<int32_t>();
  }
  // SortHitMatch (KAligner.cpp:10969): chrom, start, len, strand, mismatches; equal keys keep load order.
  // Two stable radix sorts: minor key first, then (chrom, start).
  K4_HIP(ix, idx1.alloc(m * 4));
  K4_HIP(ix, k32a.alloc(m * 4));
  K4_HIP(ix, k32b.alloc(m * 4));
  K4_HIP(ix, k64a.alloc(m * 8));
  K4_HIP(ix, k64b.alloc(m * 8));
  const unsigned mb = (unsigned)((m + 255) / 256);
  hipLaunchKernelGGL(k4k_sam_key_minor, dim3(mb), dim3(256), 0, st, a, idx0.as<uint32_t>(), m, k32a.as<uint32_t>());
  const uint32_t* order = nullptr;
  {
    rocprim::double_buffer<uint32_t> kb(k32a.as<uint32_t>(), k32b.as<uint32_t>());
    rocprim::double_buffer<uint32_t> vb(idx0.as<uint32_t>(), idx1.as<uint32_t>());
    size_t tb = 0;
    K4_HIP(ix, rocprim::radix_sort_pairs(nullptr, tb, kb, vb, (size_t)m, 0u, 32u, st));
    Buf t2;
    K4_HIP(ix, t2.alloc(tb));
    K4_HIP(ix, rocprim::radix_sort_pairs(t2.p, tb, kb, vb, (size_t)m, 0u, 32u, st));
    hipLaunchKernelGGL(k4k_sam_key_major, dim3(mb), dim3(256), 0, st, a, vb.current(), m, k64a.as<uint64_t>());
    rocprim::double_buffer<uint64_t> kb2(k64a.as<uint64_t>(), k64b.as<uint64_t>());
    size_t tb2 = 0;
    unsigned top = 33;  // key = chrom << 32 | start: only the bits chromosome ids can reach are sorted on
    while (top < 64 && ((a.n_entries + (all_reads ? 33u : 0u)) >> (top - 32)) != 0) top++;
    K4_HIP(ix, rocprim::radix_sort_pairs(nullptr, tb2, kb2, vb, (size_t)m, 0u, top, st));
    Buf t3;
    K4_HIP(ix, t3.alloc(tb2));
    K4_HIP(ix, rocprim::radix_sort_pairs(t3.p, tb2, kb2, vb, (size_t)m, 0u, top, st));
    K4_HIP(ix, hipStreamSynchronize(st));
    order = vb.current();
  }
  // line lengths -> offsets -> text
  K4_HIP(ix, ll.alloc((m + 1) * 4));
  K4_HIP(ix, lo.alloc((m + 1) * 8));
  K4_HIP(ix, hipMemsetAsync(ll.as<uint32_t>() + m, 0, 4, st));
  if (bam) hipLaunchKernelGGL(k4k_sam_line_lens<true>, dim3(mb), dim3(256), 0, st, a, order, m, ll.as<uint32_t>());
  else hipLaunchKernelGGL(k4k_sam_line_lens<false>, dim3(mb), dim3(256), 0, st, a, order, m, ll.as<uint32_t>());
  {
    size_t tb = 0;
    K4_HIP(ix, rocprim::exclusive_scan(nullptr, tb, ll.as<uint32_t>(), lo.as<uint64_t>(), (uint64_t)0, (size_t)(m + 1),
                                       rocprim::plus<uint64_t>(), st));
    Buf t4;
    K4_HIP(ix, t4.alloc(tb));
    K4_HIP(ix, rocprim::exclusive_scan(t4.p, tb, ll.as<uint32_t>(), lo.as<uint64_t>(), (uint64_t)0, (size_t)(m + 1),
                                       rocprim::plus<uint64_t>(), st));
    K4_HIP(ix, hipStreamSynchronize(st));
  }
  // slices of consecutive lines (one when the caller waits for the whole body anyway): their byte bounds come down with the total
  const int n_sl = slices ? (int)std::min<uint64_t>(16, std::max<uint64_t>(m >> 16, 1)) : 1;
  std::vector<uint64_t> line_end((size_t)n_sl), byte_end((size_t)n_sl);
  for (int k = 0; k < n_sl; k++) {
    line_end[(size_t)k] = k + 1 == n_sl ? m : m * (uint64_t)(k + 1) / (uint64_t)n_sl;
    K4_HIP(ix, hipMemcpyAsync(&byte_end[(size_t)k], lo.as<uint64_t>() + line_end[(size_t)k], 8, hipMemcpyDeviceToHost, st));
  }
  K4_HIP(ix, hipStreamSynchronize(st));
  const uint64_t total = byte_end[(size_t)n_sl - 1];
  char* out = nullptr;
  if (out_buf) {
    K4_HIP(ix, out_buf->alloc(total + 16, st));
    out = out_buf->as<char>();
  } else
    K4_HIP(ix, hipMalloc(&out, total + 16));
  int rc = K4_OK;
  {  // lines per wave tile: as many as fit the wave's LDS buffer at 1.25 x the average line
    const uint64_t avg = total / m + 1;
    const int lpl = avg * 64 * 5 / 4 <= K4_SAM_WAVE_BUF ? 1 : avg * 32 * 5 / 4 <= K4_SAM_WAVE_BUF ? 2 : 4;
    const uint64_t per_block = 4 * (64 / lpl);
    uint64_t b0 = 0;
    for (int k = 0; k < n_sl && rc == K4_OK; k++) {
      const uint64_t mk = line_end[(size_t)k] - b0;
      if (mk) {
        const dim3 grid((unsigned)std::min<uint64_t>((mk + per_block - 1) / per_block, 1u << 16));
        const uint32_t* ord = order + b0;
        const uint64_t* lok = lo.as<uint64_t>() + b0;
        if (bam) {
          if (lpl == 1) hipLaunchKernelGGL((k4k_sam_write<1, true>), grid, dim3(256), 0, st, a, ord, lok, mk, out);
          else if (lpl == 2) hipLaunchKernelGGL((k4k_sam_write<2, true>), grid, dim3(256), 0, st, a, ord, lok, mk, out);
          else hipLaunchKernelGGL((k4k_sam_write<4, true>), grid, dim3(256), 0, st, a, ord, lok, mk, out);
        } else {
          if (lpl == 1) hipLaunchKernelGGL((k4k_sam_write<1, false>), grid, dim3(256), 0, st, a, ord, lok, mk, out);
          else if (lpl == 2) hipLaunchKernelGGL((k4k_sam_write<2, false>), grid, dim3(256), 0, st, a, ord, lok, mk, out);
          else hipLaunchKernelGGL((k4k_sam_write<4, false>), grid, dim3(256), 0, st, a, ord, lok, mk, out);
        }
        rc = k4_check_hip(ix, hipGetLastError(), "SAM write");
      }
      b0 = line_end[(size_t)k];
      if (slices && rc == K4_OK) {
        hipEvent_t ev = nullptr;
        rc = k4_check_hip(ix, hipEventCreateWithFlags(&ev, hipEventDisableTiming), "event");
        if (rc == K4_OK) {
          slices->ev.push_back(ev);
          slices->end.push_back(byte_end[(size_t)k]);
          rc = k4_check_hip(ix, hipEventRecord(ev, st), "event");
        }
      }
    }
  }
  if (rc == K4_OK && !slices) rc = k4_check_hip(ix, hipStreamSynchronize(st), "SAM write");
  if (rc != K4_OK) {
    (void)hipStreamSynchronize(st);
    if (slices) slices->clear();
    if (out_buf) out_buf->release(); else hipFree(out);
    return rc;
  }
  *d_sam = out;
  *sam_bytes = total;
  return K4_OK;
}
