// kit4b_amd/csrc/k4_comm.hip -- libk4comm.so: RCCL over xGMI for the two exchange steps the path has (include/k4comm.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/k4comm.h"

struct k4_comm {
  ncclComm_t comm = nullptr;
  hipStream_t st = nullptr;
  int device = 0, rank = 0, n = 1;
  std::string err;
};

namespace {

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
// (the root, which has everything, only gives its own piece 0).  Per link: two transfers of bytes / N.
int bcast_all_links(k4_comm* c, uint8_t* buf, uint64_t bytes) {
  const int N = c->n, me = c->rank;
  if (N == 1 || bytes == 0) return K4_OK;
  const uint64_t piece = ((bytes + N - 1) / N + 255) & ~255ull;
  auto off = [&](int r) { return std::min<uint64_t>((uint64_t)r * piece, bytes); };
  auto len = [&](int r) { return off(r + 1) - off(r); };
  CK_NCCL(c, ncclGroupStart());
  if (me == 0) {
    for (int r = 1; r < N; r++)
      if (len(r)) CK_NCCL(c, ncclSend(buf + off(r), len(r), ncclUint8, r, c->comm, c->st));
  } else if (len(me))
    CK_NCCL(c, ncclRecv(buf + off(me), len(me), ncclUint8, 0, c->comm, c->st));
  CK_NCCL(c, ncclGroupEnd());
  CK_NCCL(c, ncclGroupStart());
  for (int p = 1; p < N; p++) {  // receivers: everyone but the root
    if (p == me) {
      for (int s = 0; s < N; s++)
        if (s != me && len(s)) CK_NCCL(c, ncclRecv(buf + off(s), len(s), ncclUint8, s, c->comm, c->st));
    } else if (len(me))
      CK_NCCL(c, ncclSend(buf + off(me), len(me), ncclUint8, p, c->comm, c->st));
  }
  CK_NCCL(c, ncclGroupEnd());
  CK_HIP(c, hipStreamSynchronize(c->st));
  return K4_OK;
}

struct Meta {  // what the peers need before they can size their buffers
  uint64_t n;
  uint32_t el, ne;
  int32_t rc;
  char dataset[81];
  uint8_t header[1224];
};

}  // namespace

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
  if (ncclCommInitRank(&c->comm, n_ranks, u, rank) != ncclSuccess) { hipStreamDestroy(c->st); return bail(K4_ERR_NO_DEVICE); }
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
  // geometry first (a small ncclBroadcast), then the entries table, then the two big arrays over every link
  Meta* d_m = nullptr;
  CK_HIP(c, hipMalloc(&d_m, sizeof(Meta)));
  if (c->rank == 0) CK_HIP(c, hipMemcpy(d_m, &m, sizeof(m), hipMemcpyHostToDevice));
  CK_NCCL(c, ncclBroadcast(d_m, d_m, sizeof(Meta), ncclUint8, 0, c->comm, c->st));
  CK_HIP(c, hipStreamSynchronize(c->st));
  CK_HIP(c, hipMemcpy(&m, d_m, sizeof(m), hipMemcpyDeviceToHost));
  hipFree(d_m);
  if (m.rc != K4_OK) {  // every rank learns that the root could not read the file
    if (c->rank == 0) k4_sfx_unmap(&f);
    else fail(c, m.rc, "rank 0 could not read the index");
    return m.rc;
  }
  std::vector<k4_entry> ents(m.ne);
  {
    k4_entry* d_e = nullptr;
    const size_t eb = (size_t)m.ne * sizeof(k4_entry);
    CK_HIP(c, hipMalloc(&d_e, eb));
    if (c->rank == 0) CK_HIP(c, hipMemcpy(d_e, f.entries, eb, hipMemcpyHostToDevice));
    CK_NCCL(c, ncclBroadcast(d_e, d_e, eb, ncclUint8, 0, c->comm, c->st));
    CK_HIP(c, hipStreamSynchronize(c->st));
    CK_HIP(c, hipMemcpy(ents.data(), d_e, eb, hipMemcpyDeviceToHost));
    hipFree(d_e);
  }
  uint8_t *d_seq = nullptr, *d_sa = nullptr;
  CK_HIP(c, hipMalloc(&d_seq, m.n + 64));
  CK_HIP(c, hipMalloc(&d_sa, m.n * m.el + 16));
  if (c->rank == 0) {
    CK_HIP(c, hipMemcpy(d_seq, f.seq, m.n, hipMemcpyHostToDevice));
    CK_HIP(c, hipMemcpy(d_sa, f.sa, m.n * m.el, hipMemcpyHostToDevice));
    k4_sfx_unmap(&f);
  }
  int rc = bcast_all_links(c, d_seq, m.n);
  if (rc == K4_OK) rc = bcast_all_links(c, d_sa, m.n * m.el);
  if (rc != K4_OK) { hipFree(d_seq); hipFree(d_sa); return rc; }
  // the suffix array stays where it arrived (adopted; released with the index), the byte sequence is only the source of the
  // packed form
  rc = k4_open_device(m.n, m.el, d_seq, d_sa, 1, m.ne, ents.data(), m.dataset, c->device, kmer_k, out);
  hipFree(d_seq);
  if (rc != K4_OK) { hipFree(d_sa); return fail(c, rc, "%s", k4_global_error()); }
  k4_set_raw_header(*out, m.header);
  return K4_OK;
}

extern "C" int k4_comm_allreduce_sum_u64(k4_comm* c, uint64_t* vals, int n) {
  if (!c || !vals || n < 1) return K4_ERR_PARAMS;
  CK_HIP(c, hipSetDevice(c->device));
  uint64_t* d = nullptr;
  CK_HIP(c, hipMalloc(&d, (size_t)n * 8));
  CK_HIP(c, hipMemcpy(d, vals, (size_t)n * 8, hipMemcpyHostToDevice));
  CK_NCCL(c, ncclAllReduce(d, d, (size_t)n, ncclUint64, ncclSum, c->comm, c->st));
  CK_HIP(c, hipStreamSynchronize(c->st));
  CK_HIP(c, hipMemcpy(vals, d, (size_t)n * 8, hipMemcpyDeviceToHost));
  hipFree(d);
  return K4_OK;
}

extern "C" int k4_comm_barrier(k4_comm* c) {
  uint64_t one = 1;
  return k4_comm_allreduce_sum_u64(c, &one, 1);
}

extern "C" void k4_comm_close(k4_comm* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->comm) ncclCommDestroy(c->comm);
  if (c->st) (void)hipStreamDestroy(c->st);
  delete c;
}
