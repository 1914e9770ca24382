// kit4b_amd/csrc/k4_pool.h -- a caching pool for the short-lived device buffers of the ingest / emit stages.
//
// hipFree waits for every stream of the device, so a stage that frees its scratch at exit stalls behind whatever the copy
// streams still have queued (the overlapped pipeline, k4_pipeline.hip: parse and align of chunk k would wait for the uploads
// of chunks k+1.. to finish).  Blocks handed back here are kept per device and given out again; every block carries an event
// recorded on the stream of its last user, and the stream of its next user waits for that event (no host-side wait).  Nothing is
// freed before k4_pool_trim_current_device (k4_close, or an allocation that fails for want of memory) or k4_pool_trim_to (the
// pipeline keeps at most 32 GB cached when it closes).
#pragma once
#include <stddef.h>
#include <mutex>
#include <vector>
#include <hip/hip_runtime.h>

struct K4PoolBlock { void* p; size_t cap; hipEvent_t ev; };
struct K4Pool {
  std::mutex m;
  std::vector<K4PoolBlock> free_;
};
inline K4Pool g_k4_pool[32];  // per device

inline K4Pool& k4_pool_of_current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return g_k4_pool[(unsigned)dev & 31u];
}

inline void k4_pool_trim_current_device() {
  K4Pool& P = k4_pool_of_current_device();
  std::vector<K4PoolBlock> take;
  { std::lock_guard<std::mutex> lk(P.m); take.swap(P.free_); }
  for (K4PoolBlock& b : take) { (void)hipFree(b.p); (void)hipEventDestroy(b.ev); }
}

// keep at most `limit` bytes cached: the largest blocks go first (hipFree waits for the device: call it where that is harmless)
inline void k4_pool_trim_to(size_t limit) {
  K4Pool& P = k4_pool_of_current_device();
  std::vector<K4PoolBlock> drop;
  {
    std::lock_guard<std::mutex> lk(P.m);
    size_t held = 0;
    for (const K4PoolBlock& b : P.free_) held += b.cap;
    while (held > limit && !P.free_.empty()) {
      size_t big = 0;
      for (size_t k = 1; k < P.free_.size(); k++)
        if (P.free_[k].cap > P.free_[big].cap) big = k;
      held -= P.free_[big].cap;
      drop.push_back(P.free_[big]);
      P.free_[big] = P.free_.back();
      P.free_.pop_back();
    }
  }
  for (K4PoolBlock& b : drop) { (void)hipFree(b.p); (void)hipEventDestroy(b.ev); }
}

// capacity classes: eight per power of two, so that buffers sized by slightly different batches find each other
inline size_t k4_pool_class(size_t bytes) {
  if (bytes < 4096) return 4096;
  size_t top = (size_t)1 << (63 - __builtin_clzll((unsigned long long)bytes));
  const size_t step = top >> 3;
  return (bytes + step - 1) / step * step;
}

// hipMalloc for everything that is not a pool block: when the device is out of memory while gigabytes sit cached in the pool,
// the cache is given back and the allocation tried once more
inline hipError_t k4_malloc_retry(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    k4_pool_trim_current_device();
    e = hipMalloc(p, bytes);
  }
  return e;
}

inline hipError_t k4_pool_get(void** p, size_t bytes, hipStream_t st, size_t* cap_out, hipEvent_t* ev_out) {
  const size_t want = k4_pool_class(bytes ? bytes : 1);
  K4Pool& P = k4_pool_of_current_device();
  K4PoolBlock pick = {nullptr, 0, nullptr};
  {
    std::lock_guard<std::mutex> lk(P.m);
    size_t best = (size_t)-1;
    for (size_t k = 0; k < P.free_.size(); k++) {
      const K4PoolBlock& b = P.free_[k];
      if (b.cap < want || b.cap > want + want / 2) continue;
      if (best == (size_t)-1 || b.cap < P.free_[best].cap) best = k;
    }
    if (best != (size_t)-1) { pick = P.free_[best]; P.free_[best] = P.free_.back(); P.free_.pop_back(); }
  }
  if (pick.p) {
    hipError_t e = hipStreamWaitEvent(st, pick.ev, 0);  // behind the block's last user, whichever stream that was
    if (e != hipSuccess) {  // (the block goes back on the list: nothing is leaked)
      std::lock_guard<std::mutex> lk(P.m);
      P.free_.push_back(pick);
      return e;
    }
    *p = pick.p; *cap_out = pick.cap; *ev_out = pick.ev;
    return hipSuccess;
  }
  hipError_t e = hipMalloc(p, want);
  if (e == hipErrorOutOfMemory) {  // give the cached blocks back and try once more
    (void)hipGetLastError();
    k4_pool_trim_current_device();
    e = hipMalloc(p, want);
  }
  if (e != hipSuccess) return e;
  e = hipEventCreateWithFlags(ev_out, hipEventDisableTiming);
  if (e != hipSuccess) { (void)hipFree(*p); *p = nullptr; return e; }
  *cap_out = want;
  return hipSuccess;
}

inline void k4_pool_put(void* p, size_t cap, hipEvent_t ev, hipStream_t st) {
  if (!p) return;
  (void)hipEventRecord(ev, st);
  K4Pool& P = k4_pool_of_current_device();
  std::lock_guard<std::mutex> lk(P.m);
  P.free_.push_back({p, cap, ev});
}

// RAII buffer of one stage call; the stream is the one every kernel and copy of that call runs on
struct K4PoolBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  hipStream_t st = nullptr;
  K4PoolBuf() = default;
  K4PoolBuf(const K4PoolBuf&) = delete;
  K4PoolBuf& operator=(const K4PoolBuf&) = delete;
  ~K4PoolBuf() { release(); }
  void release() { if (p) k4_pool_put(p, cap, ev, st); p = nullptr; cap = 0; }
  hipError_t alloc(size_t bytes, hipStream_t s) {
    release();
    st = s;
    return k4_pool_get(&p, bytes, s, &cap, &ev);
  }
  template <typename T> T* as() { return (T*)p; }
};
