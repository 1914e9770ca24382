// kit4b_amd/csrc/k4_pipeline.hip -- the overlapped host <-> device pipeline around the hot path.
//
// The reference loads reads on a background thread while its workers align (CKAligner::InitiateLoadingReads /
// ProcLoadReadFiles, ngskit4b/KAligner.cpp:4786-4866,11323-11496; ThreadedIterReads :10370-10438) and writes the sorted
// result at the end (WriteBAMReadHits :5718).  Here the same three stages run on three HIP streams:
//
//   reader thread(s) of the caller -> pinned ring buffers --copy stream--> text arena in HBM       (k4_pipeline_acquire/submit)
//   worker thread of this library:   parse -> length filter -> align, chunk by chunk, on the compute stream, while the
//                                    next chunk is still on its way up; results accumulate in HBM arenas
//   k4_pipeline_format:              ONE global coordinate sort + SAM text over all chunks (the output is identical to a
//                                    single batch by construction)
//   k4_pipeline_next_sam:            SAM text --copy-out stream--> pinned ring -> the caller's writer, piece k+1 coming
//                                    down while piece k is written
// Nothing here computes: every kernel is launched through the *_dev entry points of this library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include "k4_internal.h"
#include "k4_pool.h"
#include "k4_stages.h"

namespace {

struct Arena {  // grow-only device array from the device's block pool (k4_pool.h); only the worker thread touches it while the pipeline runs
  K4PoolBuf b;
  uint8_t* p = nullptr;
  size_t cap = 0, used = 0;
  hipStream_t st = nullptr;      // the stream whose kernels read and write the array (the compute stream)
  hipStream_t writer = nullptr;  // a second stream that writes into it (the text arrays: the copy-in stream)
  // Growing never blocks the host: the new block is taken behind its last user, the old contents are copied on `st` (behind what
  // `writer` had queued into the old block), and the old block goes back to the pool behind that copy.
  int reserve(k4_index* ix, size_t need, size_t slack_div = 2) {
    if (need <= cap) return K4_OK;
    const size_t ncap = std::max(need + need / slack_div + 4096, cap * 2);
    K4PoolBuf nb;
    K4_HIP(ix, nb.alloc(ncap + 64, st));
    if (writer) K4_HIP(ix, hipStreamWaitEvent(writer, nb.ev, 0));  // (a fresh block's event was never recorded: no wait)
    if (used) {
      if (writer) {
        K4_HIP(ix, hipEventRecord(b.ev, writer));  // the old block's own event: free to use while the block is held
        K4_HIP(ix, hipStreamWaitEvent(st, b.ev, 0));
      }
      K4_HIP(ix, hipMemcpyAsync(nb.p, p, used, hipMemcpyDeviceToDevice, st));
    }
    std::swap(b.p, nb.p); std::swap(b.cap, nb.cap); std::swap(b.ev, nb.ev);
    b.st = st; nb.st = st;  // nb now holds the old block: back to the pool behind the copy, at the end of this scope
    p = (uint8_t*)b.p; cap = b.cap - 64;
    return K4_OK;
  }
  void release() { b.release(); p = nullptr; cap = used = 0; }
};

struct PinBuf {
  uint8_t* h = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;  // recorded behind the copy that uses the buffer
  bool in_flight = false;   // handed to submit(), copy not yet known complete
  bool lent = false;        // handed out by acquire(), not yet submitted
};

struct Job {
  int end;
  int buf;            // ring index, or -1 for caller memory
  const void* src;    // caller memory (buf == -1)
  uint64_t bytes;
  int final_chunk;
};

struct End {
  Arena text;                 // every byte of the file(s) of this end, in order
  uint64_t parsed = 0;        // bytes that belong to complete records already parsed
  uint64_t uploaded = 0;      // bytes whose copy has been enqueued
  hipEvent_t up_ev = nullptr; // behind the last enqueued copy
  Arena offs, lens, noff, nlen;  // per record: 8 / 4 / 8 / 4 bytes
  int64_t n_rec = 0;
  int fmt = 0;
  bool final_seen = false;
  std::vector<PinBuf> ring;
};

}  // namespace

struct k4_pipeline {
  k4_index* ix = nullptr;
  k4_pipeline_params prm{};
  hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
  End end[2];
  Arena reads;           // etSeqBase bytes of both ends
  Arena c_offs, c_lens;  // per aligned read (PE: interleaved), 8 / 4 bytes
  Arena rr, hits, seg2, pe;
  int64_t units_done = 0;
  uint32_t max_read_len = 0;
  uint64_t n_under = 0, n_over = 0;
  int n_ends = 1;
  int32_t trim5 = 0, trim3 = 0;  // kalign -y / -Y (k4_pipeline_set_trims)
  int32_t sample_nth = 1;        // kalign -#<n> (k4_pipeline_set_sampling)
  // worker
  std::thread worker;
  std::mutex m;
  std::condition_variable cv;
  std::deque<Job> jobs;
  bool closing = false, worker_done = false;
  int err = K4_OK;
  // output
  void* d_sam = nullptr;        // the body, in sam_buf
  K4PoolBuf sam_buf;
  K4SamSlices slices;           // it is written slice by slice: the copy-out stream follows behind the slices' events
  size_t slice_waited = 0;      // slices the copy-out stream already waits behind
  uint64_t sam_bytes = 0, sam_next = 0, sam_given = 0;
  std::vector<PinBuf> out_ring;
  std::deque<std::pair<int, uint64_t>> out_q;  // (ring index, bytes) of pieces on their way down, oldest first
  int out_held = -1;
};

namespace {

int pin_alloc(k4_index* ix, PinBuf& b, size_t cap) {
  K4_HIP(ix, hipHostMalloc((void**)&b.h, cap, hipHostMallocDefault));
  K4_HIP(ix, hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
  b.cap = cap;
  return K4_OK;
}

// parse whatever complete records the uploaded text of `e` holds beyond `parsed` (chunks of < 4 GiB per call)
int parse_more(k4_pipeline* pl, int e) {
  k4_index* ix = pl->ix;
  End& E = pl->end[e];
  // (the compute stream already waits behind the copy of everything below E.uploaded: worker_main)
  const uint64_t piece = 3ull << 30;
  while (E.parsed < E.uploaded) {
    const uint64_t len = std::min(piece, E.uploaded - E.parsed);
    const int final_chunk = E.final_seen && E.parsed + len == E.uploaded;
    // room: a record takes at least ~ 2 lines; 1 per 24 bytes is generous for real reads -- until the first records show how long
    // they are (then: twice as many as their mean length suggests); a shortfall only costs another round
    int64_t cap_rec = (int64_t)(len / 24 + 1024);
    if (E.n_rec > 4096 && E.parsed > 0) cap_rec = std::min<int64_t>(cap_rec, (int64_t)((long double)len * 2 * E.n_rec / E.parsed) + 1024);
    // with the size of the whole input known, an array that has to grow grows once: to what the records seen so far project
    const uint64_t expect = pl->prm.expect_text_bytes[e];
    const long double scale = expect > E.parsed && E.parsed > 0 ? (long double)expect / E.parsed * 1.03L : 0.0L;
    auto want = [&](size_t now, size_t have) { return std::max(now, scale > 0 ? (size_t)(have * scale) : (size_t)0); };
    int rc;
    if ((rc = E.offs.reserve(ix, want((size_t)(E.n_rec + cap_rec) * 8, (size_t)E.n_rec * 8), 8)) != K4_OK) return rc;
    if ((rc = E.lens.reserve(ix, want((size_t)(E.n_rec + cap_rec) * 4, (size_t)E.n_rec * 4), 8)) != K4_OK) return rc;
    if ((rc = E.noff.reserve(ix, want((size_t)(E.n_rec + cap_rec) * 8, (size_t)E.n_rec * 8), 8)) != K4_OK) return rc;
    if ((rc = E.nlen.reserve(ix, want((size_t)(E.n_rec + cap_rec) * 4, (size_t)E.n_rec * 4), 8)) != K4_OK) return rc;
    if ((rc = pl->reads.reserve(ix, want(pl->reads.used + len + 64, pl->reads.used * (size_t)pl->n_ends), 8)) != K4_OK) return rc;
    k4_parse_info info;
    rc = k4_parse_fastx_dev(ix, E.text.p + E.parsed, len, E.parsed, final_chunk, E.fmt, cap_rec, pl->reads.p, pl->reads.used,
                            E.offs.p + 8 * E.n_rec, E.lens.p + 4 * E.n_rec, E.noff.p + 8 * E.n_rec, E.nlen.p + 4 * E.n_rec, &info,
                            pl->s_comp);
    if (rc != K4_OK) return rc;
    if (info.format) E.fmt = (int)info.format;
    if (info.consumed == 0) {
      if (len == E.uploaded - E.parsed) break;  // an incomplete record: more text needed
      return k4_fail(ix, K4_ERR_PARAMS, "a record longer than 3 GiB");
    }
    E.n_rec += (int64_t)info.n_records;
    E.offs.used = (size_t)E.n_rec * 8; E.lens.used = (size_t)E.n_rec * 4; E.noff.used = (size_t)E.n_rec * 8; E.nlen.used = (size_t)E.n_rec * 4;
    pl->reads.used += info.n_bases;
    E.parsed += info.consumed;
  }
  return K4_OK;
}

// align the units (reads / pairs) both ends have delivered and nobody has aligned yet
int align_more(k4_pipeline* pl, bool flush) {
  k4_index* ix = pl->ix;
  const bool pe = pl->n_ends == 2;
  const int64_t avail = pe ? std::min(pl->end[0].n_rec, pl->end[1].n_rec) : pl->end[0].n_rec;
  const int64_t n = avail - pl->units_done;
  // large batches while there is input to come (the align kernels want millions of reads per launch); with the size of the
  // input known, the last eighth is aligned as it arrives, so that little is left to do behind the last upload
  bool endgame = false;
  if (pl->prm.expect_text_bytes[0]) endgame = pl->end[0].uploaded >= pl->prm.expect_text_bytes[0] - pl->prm.expect_text_bytes[0] / 8;
  if (n <= 0 || (!flush && !endgame && n < (int64_t)pl->prm.min_batch_units)) return K4_OK;
  const int64_t r0 = pe ? 2 * pl->units_done : pl->units_done, nr = pe ? 2 * n : n;
  const int max_ml = std::max(pl->prm.kp.max_ml, 1);
  int rc;
  if ((rc = k4_open_wait(ix)) != K4_OK) return rc;  // (an index opened with k4_open_async: its arrays are needed from here on)
  // reads the whole input will hold, projected from the records of its first end so far (0: unknown) -- see parse_more
  const End& E0 = pl->end[0];
  const uint64_t expect0 = pl->prm.expect_text_bytes[0];
  const int64_t all_reads = expect0 > E0.parsed && E0.parsed > 0 && E0.n_rec > 4096
                                ? (int64_t)((long double)E0.n_rec * expect0 / E0.parsed * 1.03L) * (pe ? 2 : 1) : 0;
  auto rows = [&](int64_t now) { return (size_t)std::max(now, all_reads); };
  if ((rc = pl->c_offs.reserve(ix, rows(r0 + nr + 1) * 8, 8)) != K4_OK) return rc;
  if ((rc = pl->c_lens.reserve(ix, rows(r0 + nr + 1) * 4, 8)) != K4_OK) return rc;
  uint64_t under = 0, over = 0;
  uint32_t max_len = 0;
  const int64_t d = pl->units_done;
  rc = k4_prepare_reads_trim_dev(ix, pe ? 1 : 0, n, pl->prm.min_len, pl->prm.max_len, pl->trim5, pl->trim3, pl->sample_nth, d, pl->end[0].offs.p + 8 * d, pl->end[0].lens.p + 4 * d,
                            pe ? pl->end[1].offs.p + 8 * d : nullptr, pe ? pl->end[1].lens.p + 4 * d : nullptr, 0,
                            pl->c_offs.p + 8 * r0, pl->c_lens.p + 4 * r0, &under, &over, &max_len, pl->s_comp);
  if (rc != K4_OK) return rc;
  pl->c_offs.used = (size_t)(r0 + nr) * 8; pl->c_lens.used = (size_t)(r0 + nr) * 4;
  pl->n_under += under; pl->n_over += over;
  pl->max_read_len = std::max(pl->max_read_len, max_len);
  if (pe) {
    if ((rc = pl->pe.reserve(ix, rows(r0 + nr) * sizeof(k4_pe_read), 8)) != K4_OK) return rc;
    pl->pe.used = (size_t)(r0 + nr) * sizeof(k4_pe_read);
    if (max_len > 0)
      rc = k4_kalign_pe_batch_dev(ix, &pl->prm.kp, &pl->prm.pe, n, (int32_t)max_len, pl->reads.p, pl->c_offs.p + 8 * r0, pl->c_lens.p + 4 * r0,
                                  pl->pe.p + (size_t)r0 * sizeof(k4_pe_read), pl->s_comp);
    else
      rc = k4_check_hip(ix, hipMemsetAsync(pl->pe.p + (size_t)r0 * sizeof(k4_pe_read), 0, (size_t)nr * sizeof(k4_pe_read), pl->s_comp), "memset");
  } else {
    const bool two = pl->prm.kp.micro_indel_len > 0 || pl->prm.kp.max_splice_junct_len > 0;
    if ((rc = pl->rr.reserve(ix, rows(r0 + nr) * sizeof(k4_read_result), 8)) != K4_OK) return rc;
    if ((rc = pl->hits.reserve(ix, rows(r0 + nr) * max_ml * sizeof(k4_hit), 8)) != K4_OK) return rc;
    if (two && (rc = pl->seg2.reserve(ix, rows(r0 + nr) * sizeof(k4_seg2), 8)) != K4_OK) return rc;
    pl->rr.used = (size_t)(r0 + nr) * sizeof(k4_read_result);
    pl->hits.used = (size_t)(r0 + nr) * max_ml * sizeof(k4_hit);
    if (two) pl->seg2.used = (size_t)(r0 + nr) * sizeof(k4_seg2);
    if (max_len > 0) {
      if ((rc = k4_reserve(ix, n, (int32_t)max_len, max_ml)) != K4_OK) return rc;
      rc = k4_kalign_ext_batch_dev(ix, &pl->prm.kp, n, (int32_t)max_len, pl->reads.p, pl->c_offs.p + 8 * r0, pl->c_lens.p + 4 * r0,
                                   pl->rr.p + (size_t)r0 * sizeof(k4_read_result), pl->hits.p + (size_t)r0 * max_ml * sizeof(k4_hit),
                                   two ? pl->seg2.p + (size_t)r0 * sizeof(k4_seg2) : nullptr, pl->s_comp);
    } else {
      K4_HIP(ix, hipMemsetAsync(pl->rr.p + (size_t)r0 * sizeof(k4_read_result), 0, (size_t)nr * sizeof(k4_read_result), pl->s_comp));
      K4_HIP(ix, hipMemsetAsync(pl->hits.p + (size_t)r0 * max_ml * sizeof(k4_hit), 0, (size_t)nr * max_ml * sizeof(k4_hit), pl->s_comp));
    }
  }
  if (rc != K4_OK) return rc;
  pl->units_done = avail;
  return K4_OK;
}

// a chunk whose copy has been enqueued and that nobody has parsed yet
struct Sent {
  int end;
  uint64_t upto;     // E.text.used behind this chunk
  hipEvent_t ev;     // behind its copy on the copy-in stream
  int final_chunk;
};

void worker_main(k4_pipeline* pl) {
  k4_index* ix = pl->ix;
  hipSetDevice(ix->device);
  int rc = K4_OK;
  std::deque<Sent> sent;
  std::vector<hipEvent_t> spare;  // one event per chunk, destroyed when the compute stream has drained (never re-recorded while a wait on it may be queued)
  for (;;) {
    // 1. every chunk that is queued starts its way up at once: the copy-in stream never waits for the parsing and aligning below
    std::deque<Job> take;
    {
      std::unique_lock<std::mutex> lk(pl->m);
      pl->cv.wait(lk, [&] { return !pl->jobs.empty() || pl->closing || !sent.empty(); });
      take.swap(pl->jobs);
      if (take.empty() && sent.empty()) break;  // closing, nothing left
    }
    for (const Job& j : take) {
      End& E = pl->end[j.end];
      if (rc == K4_OK && j.bytes) {
        rc = E.text.reserve(ix, E.text.used + j.bytes + 64, 4);
        if (rc == K4_OK) {
          const void* src = j.buf >= 0 ? (const void*)E.ring[(size_t)j.buf].h : j.src;
          rc = k4_check_hip(ix, hipMemcpyAsync(E.text.p + E.text.used, src, j.bytes, hipMemcpyHostToDevice, pl->s_in), "upload");
        }
        if (rc == K4_OK) {
          E.text.used += j.bytes;
          if (j.buf >= 0) rc = k4_check_hip(ix, hipEventRecord(E.ring[(size_t)j.buf].ev, pl->s_in), "event");
        }
      }
      if (rc == K4_OK) {
        hipEvent_t ev = nullptr;
        rc = k4_check_hip(ix, hipEventCreateWithFlags(&ev, hipEventDisableTiming), "event");
        if (rc == K4_OK) rc = k4_check_hip(ix, hipEventRecord(ev, pl->s_in), "event");
        if (rc == K4_OK) sent.push_back({j.end, E.text.used, ev, j.final_chunk});
        else if (ev) spare.push_back(ev);
      }
      if (j.buf >= 0) {  // the ring buffer is free once its copy has completed: acquire() waits on the buffer's own event
        std::lock_guard<std::mutex> lk(pl->m);
        E.ring[(size_t)j.buf].in_flight = false;
        pl->cv.notify_all();
      }
    }
    if (rc != K4_OK) {  // drain
      for (Sent& c : sent) spare.push_back(c.ev);
      sent.clear();
      continue;
    }
    // 2. the oldest chunk on its way becomes parseable (the compute stream waits for its copy): parse and align what it completes
    if (!sent.empty()) {
      const Sent c = sent.front();
      sent.pop_front();
      End& E = pl->end[c.end];
      rc = k4_check_hip(ix, hipStreamWaitEvent(pl->s_comp, c.ev, 0), "event");
      spare.push_back(c.ev);
      E.uploaded = c.upto;
      if (c.final_chunk) E.final_seen = true;
      if (rc == K4_OK) rc = parse_more(pl, c.end);
      bool all_final = true;
      for (int e = 0; e < pl->n_ends; e++) all_final &= pl->end[e].final_seen;
      if (rc == K4_OK) rc = align_more(pl, all_final && sent.empty());
    }
  }
  if (rc == K4_OK) {  // whatever is left (the caller closed the input without a final chunk on some end: wait_aligned reports it)
    bool all_final = true;
    for (int e = 0; e < pl->n_ends; e++) all_final &= pl->end[e].final_seen;
    if (all_final) rc = align_more(pl, true);
  }
  if (rc == K4_OK) rc = k4_check_hip(ix, hipStreamSynchronize(pl->s_comp), "pipeline");
  else (void)hipStreamSynchronize(pl->s_comp);
  for (hipEvent_t ev : spare) hipEventDestroy(ev);
  std::lock_guard<std::mutex> lk(pl->m);
  pl->err = rc;
  pl->worker_done = true;
  pl->cv.notify_all();
}

int join_worker(k4_pipeline* pl) {
  {
    std::lock_guard<std::mutex> lk(pl->m);
    pl->closing = true;
    pl->cv.notify_all();
  }
  if (pl->worker.joinable()) pl->worker.join();
  return pl->err;
}

}  // namespace

extern "C" int k4_pipeline_open(k4_index* ix, const k4_pipeline_params* p, k4_pipeline** out) {
  if (!ix || !p || !out) return K4_ERR_PARAMS;
  *out = nullptr;
  if (p->kp.max_ml < 1) return k4_fail(ix, K4_ERR_PARAMS, "max_ml must be >= 1");
  K4_HIP(ix, hipSetDevice(ix->device));
  k4_pipeline* pl = new k4_pipeline;
  pl->ix = ix;
  pl->prm = *p;
  pl->n_ends = p->paired ? 2 : 1;
  if (pl->prm.chunk_bytes == 0) pl->prm.chunk_bytes = 256ull << 20;
  pl->prm.chunk_bytes = std::min<uint64_t>(std::max<uint64_t>(pl->prm.chunk_bytes, 1ull << 20), 2ull << 30);
  if (pl->prm.min_batch_units == 0) pl->prm.min_batch_units = 1u << 22;  // (the align kernels want millions of reads per launch)
  if (pl->prm.n_buffers < 2) pl->prm.n_buffers = 3;
  int rc = K4_OK;
  auto ck = [&](hipError_t e, const char* what) { if (rc == K4_OK) rc = k4_check_hip(ix, e, what); };
  ck(hipStreamCreateWithFlags(&pl->s_in, hipStreamNonBlocking), "stream");
  ck(hipStreamCreateWithFlags(&pl->s_comp, hipStreamNonBlocking), "stream");
  ck(hipStreamCreateWithFlags(&pl->s_out, hipStreamNonBlocking), "stream");
  for (int e = 0; e < 2; e++) {
    End& E = pl->end[e];
    for (Arena* a : {&E.text, &E.offs, &E.lens, &E.noff, &E.nlen}) a->st = pl->s_comp;
    E.text.writer = pl->s_in;
  }
  for (Arena* a : {&pl->reads, &pl->c_offs, &pl->c_lens, &pl->rr, &pl->hits, &pl->seg2, &pl->pe}) a->st = pl->s_comp;
  for (int e = 0; e < pl->n_ends && rc == K4_OK; e++) {
    pl->end[e].ring.resize((size_t)pl->prm.n_buffers);
    if (p->expect_text_bytes[e]) rc = pl->end[e].text.reserve(ix, (size_t)p->expect_text_bytes[e] + 64, 64);
  }
  if (rc != K4_OK) { k4_pipeline_close(pl); return rc; }
  pl->worker = std::thread(worker_main, pl);
  *out = pl;
  return K4_OK;
}

extern "C" int k4_pipeline_set_trims(k4_pipeline* pl, int32_t trim5, int32_t trim3) {
  if (!pl || trim5 < 0 || trim3 < 0 || trim5 > 50 || trim3 > 50) return K4_ERR_PARAMS;
  std::lock_guard<std::mutex> lk(pl->m);
  pl->trim5 = trim5; pl->trim3 = trim3;  // (read by the worker when it prepares a batch: set them before the first submit)
  return K4_OK;
}

extern "C" int k4_pipeline_set_sampling(k4_pipeline* pl, int32_t sample_nth) {
  if (!pl || sample_nth < 1 || sample_nth > 10000) return K4_ERR_PARAMS;
  std::lock_guard<std::mutex> lk(pl->m);
  pl->sample_nth = sample_nth;
  return K4_OK;
}

// a pinned buffer of the ring of `end` for the caller to fill; blocks while all of them are on their way up
extern "C" int k4_pipeline_acquire(k4_pipeline* pl, int end, void** buf, uint64_t* cap) {
  if (!pl || end < 0 || end >= pl->n_ends || !buf || !cap) return K4_ERR_PARAMS;
  k4_index* ix = pl->ix;
  End& E = pl->end[end];
  for (;;) {
    int pick = -1;
    {
      std::unique_lock<std::mutex> lk(pl->m);
      if (pl->err != K4_OK) return pl->err;
      for (size_t b = 0; b < E.ring.size(); b++)
        if (!E.ring[b].in_flight && !E.ring[b].lent) { pick = (int)b; break; }
      if (pick < 0) { pl->cv.wait(lk); continue; }
      E.ring[(size_t)pick].lent = true;
    }
    PinBuf& B = E.ring[(size_t)pick];
    if (!B.h) {
      K4_HIP(ix, hipSetDevice(ix->device));
      int rc = pin_alloc(ix, B, (size_t)pl->prm.chunk_bytes);
      if (rc != K4_OK) return rc;
    } else
      K4_HIP(ix, hipEventSynchronize(B.ev));  // its previous copy has left the buffer
    *buf = B.h;
    *cap = B.cap;
    return K4_OK;
  }
}

static int push_job(k4_pipeline* pl, const Job& j) {
  std::lock_guard<std::mutex> lk(pl->m);
  if (pl->err != K4_OK) return pl->err;
  if (pl->closing) return K4_ERR_PARAMS;
  pl->jobs.push_back(j);
  pl->cv.notify_all();
  return K4_OK;
}

extern "C" int k4_pipeline_submit(k4_pipeline* pl, int end, uint64_t bytes, int final_chunk) {
  if (!pl || end < 0 || end >= pl->n_ends) return K4_ERR_PARAMS;
  End& E = pl->end[end];
  int pick = -1;
  {
    std::lock_guard<std::mutex> lk(pl->m);
    for (size_t b = 0; b < E.ring.size(); b++)
      if (E.ring[b].lent) { pick = (int)b; break; }
    if (pick < 0) {
      if (bytes) return K4_ERR_PARAMS;  // nothing acquired
    } else {
      if (bytes > E.ring[(size_t)pick].cap) return K4_ERR_PARAMS;
      E.ring[(size_t)pick].lent = false;
      E.ring[(size_t)pick].in_flight = bytes != 0;
    }
  }
  Job j = {end, bytes ? pick : -1, nullptr, bytes, final_chunk};
  return push_job(pl, j);
}

// text in the caller's own memory (pinned for the full PCIe rate); it must stay valid until k4_pipeline_wait_aligned returns
extern "C" int k4_pipeline_submit_host(k4_pipeline* pl, int end, const void* text, uint64_t bytes, int final_chunk) {
  if (!pl || end < 0 || end >= pl->n_ends || (bytes && !text)) return K4_ERR_PARAMS;
  const uint64_t piece = pl->prm.chunk_bytes;
  uint64_t pos = 0;
  do {
    const uint64_t len = std::min(piece, bytes - pos);
    Job j = {end, -1, (const uint8_t*)text + pos, len, final_chunk && pos + len == bytes};
    int rc = push_job(pl, j);
    if (rc != K4_OK) return rc;
    pos += len;
  } while (pos < bytes);
  return K4_OK;
}

extern "C" int k4_pipeline_wait_aligned(k4_pipeline* pl, k4_pipeline_view* v) {
  if (!pl) return K4_ERR_PARAMS;
  int rc = join_worker(pl);
  if (rc != K4_OK) return rc;
  k4_index* ix = pl->ix;
  for (int e = 0; e < pl->n_ends; e++)
    if (!pl->end[e].final_seen) return k4_fail(ix, K4_ERR_PARAMS, "k4_pipeline_wait_aligned before the final chunk of end %d", e);
  if (pl->n_ends == 2 && pl->end[0].n_rec != pl->end[1].n_rec) return k4_fail(ix, K4_ERR_PARAMS, "the PE1 and PE2 inputs hold different numbers of reads (%lld, %lld)", (long long)pl->end[0].n_rec, (long long)pl->end[1].n_rec);
  for (int e = 0; e < pl->n_ends; e++)
    if (pl->end[e].parsed != pl->end[e].uploaded) return k4_fail(ix, K4_ERR_NOT_FASTA, "input %d ends inside a record", e);
  if (v) {
    memset(v, 0, sizeof(*v));
    v->n_units = pl->units_done;
    v->n_reads = pl->n_ends == 2 ? 2 * pl->units_done : pl->units_done;
    v->max_read_len = pl->max_read_len;
    v->n_under = pl->n_under; v->n_over = pl->n_over;
    v->max_ml = std::max(pl->prm.kp.max_ml, 1);
    v->d_rr = pl->rr.p; v->d_hits = pl->hits.p; v->d_seg2 = pl->seg2.p; v->d_pe = pl->pe.p;
    v->d_reads = pl->reads.p; v->d_offs = pl->c_offs.p; v->d_lens = pl->c_lens.p;
    for (int e = 0; e < pl->n_ends; e++) {
      v->names.d_text[e] = pl->end[e].text.p; v->names.d_name_off[e] = pl->end[e].noff.p; v->names.d_name_len[e] = pl->end[e].nlen.p;
    }
  }
  return K4_OK;
}

static int pipeline_format(k4_pipeline* pl, int bam, int sq_all, k4_sam_stats* stats, uint8_t* chrom_hit, uint64_t* sam_bytes);
extern "C" int k4_pipeline_format(k4_pipeline* pl, k4_sam_stats* stats, uint8_t* chrom_hit, uint64_t* sam_bytes) {
  return pipeline_format(pl, 0, 0, stats, chrom_hit, sam_bytes);
}
// the same alignments as BAM records (uncompressed, coordinate order): the caller deflates them into BGZF blocks as the pieces
// come down (k4_pipeline_next_sam) and writes header and index
// `-M1`: SAM text with the reads that were not accepted behind the alignments (k4_format_sam_all_dev)
extern "C" int k4_pipeline_format_all(k4_pipeline* pl, k4_sam_stats* stats, uint8_t* chrom_hit, uint64_t* sam_bytes) {
  return pipeline_format(pl, 2, 0, stats, chrom_hit, sam_bytes);
}
extern "C" int k4_pipeline_format_bam_all(k4_pipeline* pl, int32_t sq_all, k4_sam_stats* stats, uint8_t* chrom_hit, uint64_t* bam_bytes) {
  return pipeline_format(pl, 3, sq_all, stats, chrom_hit, bam_bytes);
}
extern "C" int k4_pipeline_format_bam(k4_pipeline* pl, int32_t sq_all, k4_sam_stats* stats, uint8_t* chrom_hit, uint64_t* bam_bytes) {
  return pipeline_format(pl, 1, sq_all, stats, chrom_hit, bam_bytes);
}
static int pipeline_format(k4_pipeline* pl, int bam, int sq_all, k4_sam_stats* stats, uint8_t* chrom_hit, uint64_t* sam_bytes) {
  if (!pl) return K4_ERR_PARAMS;
  k4_pipeline_view v;
  int rc = k4_pipeline_wait_aligned(pl, &v);
  if (rc != K4_OK) return rc;
  if (pl->d_sam) {  // a second format call: the first body's way down may still be running
    K4_HIP(pl->ix, hipStreamSynchronize(pl->s_out));
    pl->sam_buf.st = pl->s_comp;
    pl->sam_buf.release();
    pl->d_sam = nullptr;
  }
  pl->slices.clear();
  pl->slice_waited = 0;
  pl->sam_bytes = pl->sam_next = pl->sam_given = 0;
  if (stats) memset(stats, 0, sizeof(*stats));
  if (v.n_units > 0 && v.max_read_len > 0)  // returns with the last slices still being written (k4_stages.h)
    rc = k4i_format_records(pl->ix, bam, sq_all, pl->n_ends == 2 ? 1 : 0, v.n_units, v.d_rr, v.d_hits, v.max_ml, v.d_pe, v.d_seg2, v.d_reads,
                            v.d_offs, v.d_lens, &v.names, &pl->d_sam, &pl->sam_bytes, stats, chrom_hit, pl->s_comp, &pl->slices, &pl->sam_buf);
  if (sam_bytes) *sam_bytes = pl->sam_bytes;
  return rc;
}

// the copy-out stream may read the body up to `upto`: it waits behind the slice that holds the last of those bytes
static int wait_slices(k4_pipeline* pl, uint64_t upto) {
  K4SamSlices& S = pl->slices;
  size_t k = pl->slice_waited;
  while (k < S.end.size() && (k == 0 ? 0 : S.end[k - 1]) < upto) k++;
  if (k > pl->slice_waited) {
    K4_HIP(pl->ix, hipStreamWaitEvent(pl->s_out, S.ev[k - 1], 0));  // (the slices are written in order on one stream)
    pl->slice_waited = k;
  }
  return K4_OK;
}

// the next piece of the SAM body in a pinned buffer that stays valid until the following call; *bytes == 0 at the end.
// Two further pieces are already on their way down while the caller consumes this one.
extern "C" int k4_pipeline_next_sam(k4_pipeline* pl, const void** ptr, uint64_t* bytes) {
  if (!pl || !ptr || !bytes) return K4_ERR_PARAMS;
  k4_index* ix = pl->ix;
  *ptr = nullptr;
  *bytes = 0;
  K4_HIP(ix, hipSetDevice(ix->device));
  if (pl->out_ring.empty()) {
    pl->out_ring.resize(3);
    const size_t cap = (size_t)std::min<uint64_t>(std::max<uint64_t>(pl->prm.chunk_bytes, 8ull << 20), 256ull << 20);
    for (PinBuf& b : pl->out_ring) {
      int rc = pin_alloc(ix, b, cap);
      if (rc != K4_OK) return rc;
    }
  }
  if (pl->out_held >= 0) { pl->out_ring[(size_t)pl->out_held].in_flight = false; pl->out_held = -1; }
  // keep the copy-out stream busy: every free buffer gets the next piece
  for (size_t b = 0; b < pl->out_ring.size() && pl->sam_next < pl->sam_bytes; b++) {
    PinBuf& B = pl->out_ring[b];
    if (B.in_flight) continue;
    const uint64_t len = std::min<uint64_t>(B.cap, pl->sam_bytes - pl->sam_next);
    int rcw = wait_slices(pl, pl->sam_next + len);
    if (rcw != K4_OK) return rcw;
    K4_HIP(ix, hipMemcpyAsync(B.h, (const uint8_t*)pl->d_sam + pl->sam_next, len, hipMemcpyDeviceToHost, pl->s_out));
    K4_HIP(ix, hipEventRecord(B.ev, pl->s_out));
    B.in_flight = true;
    pl->out_q.push_back({(int)b, len});
    pl->sam_next += len;
  }
  if (pl->out_q.empty()) return K4_OK;
  const std::pair<int, uint64_t> f = pl->out_q.front();
  pl->out_q.pop_front();
  K4_HIP(ix, hipEventSynchronize(pl->out_ring[(size_t)f.first].ev));
  pl->out_held = f.first;
  *ptr = pl->out_ring[(size_t)f.first].h;
  *bytes = f.second;
  return K4_OK;
}

// the whole SAM body into the caller's memory (pinned for the full PCIe rate)
extern "C" int k4_pipeline_read_sam(k4_pipeline* pl, void* dst, uint64_t cap, uint64_t* bytes) {
  if (!pl || !bytes) return K4_ERR_PARAMS;
  *bytes = pl->sam_bytes;
  if (pl->sam_bytes == 0) return K4_OK;
  if (!dst || cap < pl->sam_bytes) return k4_fail(pl->ix, K4_ERR_PARAMS, "the SAM body takes %llu bytes", (unsigned long long)pl->sam_bytes);
  K4_HIP(pl->ix, hipSetDevice(pl->ix->device));
  uint64_t pos = 0;
  for (size_t k = 0; k < pl->slices.end.size(); k++) {  // slice k goes down while the slices behind it are still being written
    const uint64_t end = pl->slices.end[k];
    if (end == pos) continue;
    K4_HIP(pl->ix, hipStreamWaitEvent(pl->s_out, pl->slices.ev[k], 0));
    K4_HIP(pl->ix, hipMemcpyAsync((uint8_t*)dst + pos, (const uint8_t*)pl->d_sam + pos, end - pos, hipMemcpyDeviceToHost, pl->s_out));
    pos = end;
  }
  K4_HIP(pl->ix, hipStreamSynchronize(pl->s_out));
  return K4_OK;
}

extern "C" void k4_pipeline_close(k4_pipeline* pl) {
  if (!pl) return;
  join_worker(pl);
  hipSetDevice(pl->ix->device);
  hipDeviceSynchronize();
  for (int e = 0; e < 2; e++) {
    End& E = pl->end[e];
    for (PinBuf& b : E.ring) { if (b.h) hipHostFree(b.h); if (b.ev) hipEventDestroy(b.ev); }
    if (E.up_ev) hipEventDestroy(E.up_ev);
    for (Arena* a : {&E.text, &E.offs, &E.lens, &E.noff, &E.nlen}) a->release();
  }
  for (PinBuf& b : pl->out_ring) { if (b.h) hipHostFree(b.h); if (b.ev) hipEventDestroy(b.ev); }
  for (Arena* a : {&pl->reads, &pl->c_offs, &pl->c_lens, &pl->rr, &pl->hits, &pl->seg2, &pl->pe}) a->release();
  pl->slices.clear();
  pl->sam_buf.st = pl->s_comp;  // (everything is idle: hipDeviceSynchronize above)
  pl->sam_buf.release();
  for (hipStream_t s : {pl->s_in, pl->s_comp, pl->s_out}) if (s) hipStreamDestroy(s);
  // what a run of a few ten million reads (or pairs) needs stays cached for the next one -- up to a quarter of the device's
  // memory, 32 GB at least; more goes back (freeing and allocating tens of gigabytes anew costs a second per run)
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  k4_pool_trim_to(std::max<size_t>(32ull << 30, total_b / 4));
  delete pl;
}
