// tools/randgather.hip -- calibration microbenchmark (not part of the product): how many random cache-line touches per
// second does one MI355X sustain for the access shapes of k4k_align_fast?  Usage: randgather [table_GiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

// MODE 0: ILP independent 4-byte loads per iteration;  MODE 1: dependent chain of DEP loads (index from previous value)
// W = dwords per touch (1 -> 4 B, 4 -> 16 B, 9 -> 36 B unaligned window)
template <int ILP, int W>
__global__ void __launch_bounds__(256) k_indep(const uint32_t* __restrict__ tab, uint64_t n_words, int iters, uint32_t* out) {
  uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t v[ILP];
#pragma unroll
    for (int j = 0; j < ILP; j++) {
      uint64_t idx = mix(t * 1315423911ull + (uint64_t)it * ILP + j) % (n_words - 16);
      uint32_t s = 0;
#pragma unroll
      for (int w = 0; w < W; w++) s += tab[idx + w];
      v[j] = s;
    }
#pragma unroll
    for (int j = 0; j < ILP; j++) acc += v[j];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int DEP>
__global__ void __launch_bounds__(256) k_chain(const uint32_t* __restrict__ tab, uint64_t n_words, int iters, uint32_t* out) {
  uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    uint64_t idx = mix(t * 1315423911ull + it) % n_words;
#pragma unroll
    for (int d = 0; d < DEP; d++) {
      uint32_t v = tab[idx];
      acc += v;
      idx = mix(idx + v + d) % n_words;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
static void run(const char* name, F launch, double touches) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  printf("%-28s %8.2f ms  %7.2f G touches/s  (x64B = %6.2f TB/s)\n", name, ms, touches / ms / 1e6, touches * 64 / ms / 1e9);
}

int main(int argc, char** argv) {
  double gib = argc > 1 ? atof(argv[1]) : 12.0;
  uint64_t n_words = (uint64_t)(gib * (1ull << 30)) / 4;
  uint32_t *tab, *out;
  CK(hipMalloc(&tab, n_words * 4 + 256));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(tab, 1, n_words * 4 + 256));
  const int blocks = 256 * 8, iters = 64;
  const double lanes = (double)blocks * 256;
  printf("table %.1f GiB, %d blocks x 256 lanes (one resident wave set), %d iterations\n", gib, blocks, iters);
  run("indep ILP1 4B", [&] { hipLaunchKernelGGL((k_indep<1, 1>), dim3(blocks), dim3(256), 0, 0, tab, n_words, iters, out); }, lanes * iters * 1);
  run("indep ILP4 4B", [&] { hipLaunchKernelGGL((k_indep<4, 1>), dim3(blocks), dim3(256), 0, 0, tab, n_words, iters, out); }, lanes * iters * 4);
  run("indep ILP8 4B", [&] { hipLaunchKernelGGL((k_indep<8, 1>), dim3(blocks), dim3(256), 0, 0, tab, n_words, iters, out); }, lanes * iters * 8);
  run("indep ILP4 16B", [&] { hipLaunchKernelGGL((k_indep<4, 4>), dim3(blocks), dim3(256), 0, 0, tab, n_words, iters, out); }, lanes * iters * 4);
  run("indep ILP4 36B window", [&] { hipLaunchKernelGGL((k_indep<4, 9>), dim3(blocks), dim3(256), 0, 0, tab, n_words, iters, out); }, lanes * iters * 4);
  run("chain of 3 (ktab->SA->ref)", [&] { hipLaunchKernelGGL((k_chain<3>), dim3(blocks), dim3(256), 0, 0, tab, n_words, iters, out); }, lanes * iters * 3);
  const int blocks2 = 256 * 4;  // the occupancy k4k_align_fast gets today (16 waves/CU)
  run("chain of 3 @16 waves/CU", [&] { hipLaunchKernelGGL((k_chain<3>), dim3(blocks2), dim3(256), 0, 0, tab, n_words, iters * 2, out); }, (double)blocks2 * 256 * iters * 2 * 3);
  run("indep ILP4 4B @16 waves/CU", [&] { hipLaunchKernelGGL((k_indep<4, 1>), dim3(blocks2), dim3(256), 0, 0, tab, n_words, iters * 2, out); }, (double)blocks2 * 256 * iters * 2 * 4);
  return 0;
}
