// kit4b_amd/csrc/k4_comm.hip -- libk4comm.so: RCCL over xGMI for the two exchange steps the path has (include/k4comm.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>
#include <stdlib.h>
#include <time.h>
#include "../../include/k4comm.h"

static double comm_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

struct k4_comm {
  ncclComm_t comm = nullptr;
  hipStream_t st = nullptr;
  int device = 0, rank = 0, n = 1;
  std::string err;
  // ncclCommInitRank takes seconds (5 s for ONE rank on the test box) and needs nothing the caller has: it runs on a thread of its
  // own from k4_comm_init on and is joined by the first call that talks to the peers -- rank 0 brings the index file up meanwhile
  std::thread former;
  int form_rc = K4_OK;
};

namespace {

int comm_ready(k4_comm* c) {  // the communicator is formed (or could not be)
  if (c->former.joinable()) c->former.join();
  if (c->form_rc != K4_OK && c->err.empty()) c->err = "ncclCommInitRank failed";
  return c->form_rc;
}

int fail(k4_comm* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}
#define CK_HIP(c, call)                                                                      \
  do {                                                                                       \
    hipError_t _e = (call);                                                                  \
    if (_e != hipSuccess) return fail((c), K4_ERR_NO_DEVICE, "%s: %s", #call, hipGetErrorString(_e)); \
  } while (0)
#define CK_NCCL(c, call)                                                                     \
  do {                                                                                       \
    ncclResult_t _r = (call);                                                                \
    if (_r != ncclSuccess) return fail((c), K4_ERR_NO_DEVICE, "%s: %s", #call, ncclGetErrorString(_r)); \
  } while (0)

// A device buffer that only rank 0 holds -> every rank, with every direct xGMI link in use: the root deals piece r to rank r
// (N-1 concurrent point-to-point sends over N-1 different links), then every rank passes its piece to each peer that lacks it
// (the root, which has everything, only gives its own piece 0).  Per link: two transfers of bytes / N.  The list of transfers
// is k4_comm_bcast_schedule's (a pure function, tested on the CPU for every rank count); here it is only issued.
int bcast_all_links(k4_comm* c, uint8_t* buf, uint64_t bytes) {
  const int N = c->n, me = c->rank;
  if (N == 1 || bytes == 0) return K4_OK;
  std::vector<k4_comm_xfer> xf((size_t)2 * N * N);
  const int nx = k4_comm_bcast_schedule(N, bytes, xf.data(), (int)xf.size());
  for (int phase = 0; phase < 2; phase++) {
    CK_NCCL(c, ncclGroupStart());
    for (int k = 0; k < nx; k++) {
      const k4_comm_xfer& x = xf[(size_t)k];
      if (x.phase != phase) continue;
      if (x.src == me) CK_NCCL(c, ncclSend(buf + x.off, x.len, ncclUint8, x.dst, c->comm, c->st));
      if (x.dst == me) CK_NCCL(c, ncclRecv(buf + x.off, x.len, ncclUint8, x.src, c->comm, c->st));
    }
    CK_NCCL(c, ncclGroupEnd());
  }
  CK_HIP(c, hipStreamSynchronize(c->st));
  return K4_OK;
}

// device allocations of k4_comm_open_index: released on every way out unless handed on
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  void* release() { void* q = p; p = nullptr; return q; }
};

struct Meta {  // what the peers need before they can size their buffers
  uint64_t n;
  uint32_t el, ne;
  int32_t rc;
  char dataset[81];
  uint8_t header[1224];
};

}  // namespace

extern "C" int k4_comm_bcast_schedule(int n_ranks, uint64_t bytes, k4_comm_xfer* out, int cap) {
  if (n_ranks < 1 || (!out && cap > 0)) return K4_ERR_PARAMS;
  const int N = n_ranks;
  if (N == 1 || bytes == 0) return 0;
  const uint64_t piece = ((bytes + N - 1) / N + 255) & ~255ull;  // 256-byte aligned pieces; the last ones may be short or empty
  auto off = [&](int r) { return std::min<uint64_t>((uint64_t)r * piece, bytes); };
  auto len = [&](int r) { return off(r + 1) - off(r); };
  int n = 0;
  auto put = [&](int phase, int src, int dst, int pc) {
    if (!len(pc)) return;
    if (n < cap) { out[n].phase = phase; out[n].src = src; out[n].dst = dst; out[n].off = off(pc); out[n].len = len(pc); }
    n++;
  };
  for (int r = 1; r < N; r++) put(0, 0, r, r);  // the root deals piece r to rank r
  for (int p = 1; p < N; p++)                   // receivers: everyone but the root, which holds everything
    for (int s = 0; s < N; s++)
      if (s != p) put(1, s, p, s);              // rank s passes its own piece (the root: piece 0) to p
  return n;
}

extern "C" int k4_comm_unique_id(uint8_t id[K4_COMM_ID_BYTES]) {
  if (!id) return K4_ERR_PARAMS;
  ncclUniqueId u;
  static_assert(sizeof(u) == K4_COMM_ID_BYTES, "ncclUniqueId size");
  if (ncclGetUniqueId(&u) != ncclSuccess) return K4_ERR_NO_DEVICE;
  memcpy(id, &u, sizeof(u));
  return K4_OK;
}

extern "C" int k4_comm_init(int device, int rank, int n_ranks, const uint8_t id[K4_COMM_ID_BYTES], k4_comm** out) {
  if (!out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return K4_ERR_PARAMS;
  *out = nullptr;
  k4_comm* c = new k4_comm;
  c->device = device; c->rank = rank; c->n = n_ranks;
  auto bail = [&](int rc) { delete c; return rc; };
  if (hipSetDevice(device) != hipSuccess) return bail(K4_ERR_NO_DEVICE);
  if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) return bail(K4_ERR_NO_DEVICE);
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  c->former = std::thread([c, u]() {
    const double t_i = comm_now();
    if (hipSetDevice(c->device) != hipSuccess || ncclCommInitRank(&c->comm, c->n, u, c->rank) != ncclSuccess) { c->form_rc = K4_ERR_NO_DEVICE; return; }
    if (getenv("K4_TRACE")) fprintf(stderr, "[k4 trace] rank %d: RCCL communicator of %d formed in %.2fs\n", c->rank, c->n, comm_now() - t_i);
  });
  *out = c;
  return K4_OK;
}

extern "C" int k4_comm_rank(const k4_comm* c) { return c ? c->rank : -1; }
extern "C" int k4_comm_size(const k4_comm* c) { return c ? c->n : 0; }
extern "C" const char* k4_comm_last_error(const k4_comm* c) { return c ? c->err.c_str() : "no communicator"; }

extern "C" int k4_comm_open_index(k4_comm* c, const char* sfx_path, int kmer_k, k4_index** out) {
  if (!c || !out) return K4_ERR_PARAMS;
  *out = nullptr;
  CK_HIP(c, hipSetDevice(c->device));
  k4_sfx_file f;
  memset(&f, 0, sizeof(f));
  Meta m;
  memset(&m, 0, sizeof(m));
  if (c->rank == 0) {
    m.rc = sfx_path ? k4_sfx_map(sfx_path, &f) : K4_ERR_PARAMS;
    if (m.rc == K4_OK) {
      m.n = f.concat_len; m.el = f.sfx_el_size; m.ne = f.n_entries;
      memcpy(m.dataset, f.dataset, sizeof(m.dataset));
      memcpy(m.header, f.header, sizeof(m.header));
    } else
      c->err = k4_global_error();
  }
  struct Unmap {  // rank 0's mapping: released on every way out
    k4_sfx_file* f;
    bool on;
    ~Unmap() { if (on) k4_sfx_unmap(f); }
  } unmap{&f, c->rank == 0 && m.rc == K4_OK};
  // rank 0: the two big arrays go up to the device now, while the communicator is still being formed
  DevBuf d_seq, d_sa;
  const bool trace = getenv("K4_TRACE") != nullptr;
  if (c->rank == 0 && m.rc == K4_OK) {
    const double t_u = comm_now();
    if (hipMalloc(&d_seq.p, m.n + 64) != hipSuccess || hipMalloc(&d_sa.p, m.n * m.el + 16) != hipSuccess) m.rc = K4_ERR_MEM;
    if (m.rc == K4_OK) m.rc = k4_upload_pageable(c->device, d_seq.p, f.seq, m.n);
    if (m.rc == K4_OK) m.rc = k4_upload_pageable(c->device, d_sa.p, f.sa, m.n * m.el);
    if (m.rc != K4_OK) c->err = k4_global_error();
    else if (trace) fprintf(stderr, "[k4 trace] rank 0: %.2f GB of the index file on the device in %.2fs\n", (m.n + m.n * m.el) / 1e9, comm_now() - t_u);
  }
  {
    const int frc = comm_ready(c);
    if (frc != K4_OK) return frc;
  }
  // geometry first (a small ncclBroadcast), then the entries table, then the two big arrays over every link
  {
    DevBuf d_m;
    CK_HIP(c, hipMalloc(&d_m.p, sizeof(Meta)));
    if (c->rank == 0) CK_HIP(c, hipMemcpy(d_m.p, &m, sizeof(m), hipMemcpyHostToDevice));
    CK_NCCL(c, ncclBroadcast(d_m.p, d_m.p, sizeof(Meta), ncclUint8, 0, c->comm, c->st));
    CK_HIP(c, hipStreamSynchronize(c->st));
    CK_HIP(c, hipMemcpy(&m, d_m.p, sizeof(m), hipMemcpyDeviceToHost));
  }
  if (m.rc != K4_OK) {  // every rank learns that the root could not read the file
    if (c->rank != 0) fail(c, m.rc, "rank 0 could not read the index");
    return m.rc;
  }
  std::vector<k4_entry> ents(m.ne);
  {
    DevBuf d_e;
    const size_t eb = (size_t)m.ne * sizeof(k4_entry);
    CK_HIP(c, hipMalloc(&d_e.p, eb));
    if (c->rank == 0) CK_HIP(c, hipMemcpy(d_e.p, f.entries, eb, hipMemcpyHostToDevice));
    CK_NCCL(c, ncclBroadcast(d_e.p, d_e.p, eb, ncclUint8, 0, c->comm, c->st));
    CK_HIP(c, hipStreamSynchronize(c->st));
    CK_HIP(c, hipMemcpy(ents.data(), d_e.p, eb, hipMemcpyDeviceToHost));
  }
  if (c->rank != 0) {
    CK_HIP(c, hipMalloc(&d_seq.p, m.n + 64));
    CK_HIP(c, hipMalloc(&d_sa.p, m.n * m.el + 16));
  }
  const double t_x = comm_now();
  int rc = bcast_all_links(c, (uint8_t*)d_seq.p, m.n);
  if (rc == K4_OK) rc = bcast_all_links(c, (uint8_t*)d_sa.p, m.n * m.el);
  if (rc != K4_OK) return rc;
  if (trace) fprintf(stderr, "[k4 trace] rank %d: exchange over the links done in %.2fs\n", c->rank, comm_now() - t_x);
  // the suffix array stays where it arrived (adopted; released with the index), the byte sequence is only the source of the
  // packed form
  rc = k4_open_device(m.n, m.el, d_seq.p, d_sa.p, 1, m.ne, ents.data(), m.dataset, c->device, kmer_k, out);
  if (rc != K4_OK) return fail(c, rc, "%s", k4_global_error());
  d_sa.release();
  k4_set_raw_header(*out, m.header);
  return K4_OK;
}

extern "C" int k4_comm_allreduce_sum_u64(k4_comm* c, uint64_t* vals, int n) {
  if (!c || !vals || n < 1) return K4_ERR_PARAMS;
  if (const int frc = comm_ready(c)) return frc;
  CK_HIP(c, hipSetDevice(c->device));
  DevBuf d;
  CK_HIP(c, hipMalloc(&d.p, (size_t)n * 8));
  CK_HIP(c, hipMemcpy(d.p, vals, (size_t)n * 8, hipMemcpyHostToDevice));
  CK_NCCL(c, ncclAllReduce(d.p, d.p, (size_t)n, ncclUint64, ncclSum, c->comm, c->st));
  CK_HIP(c, hipStreamSynchronize(c->st));
  CK_HIP(c, hipMemcpy(vals, d.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return K4_OK;
}

extern "C" int k4_comm_barrier(k4_comm* c) {
  uint64_t one = 1;
  return k4_comm_allreduce_sum_u64(c, &one, 1);
}

extern "C" void k4_comm_close(k4_comm* c) {
  if (!c) return;
  (void)comm_ready(c);
  hipSetDevice(c->device);
  if (c->comm) ncclCommDestroy(c->comm);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}
