// ubench_granule.hip -- what HBM gives the access patterns of the two-pass 2^20 NTT (csrc/p2mt_commit.hip k_ntt20_pass), with no
// arithmetic: every workgroup moves one tile of 1024 rows x Q columns (8-byte words, row pitch 8 KB) through registers.
//   col->col   read Q-word granules one row apart, write them back the same way          (pass 1)
//   row->col   read Q whole rows (8 KB runs), write Q-word granules one row apart        (pass 2, natural-order output)
//   linear     plain coalesced copy of the same bytes                                    (the ceiling)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_granule tools/ubench_granule.hip ; run: tools/ubench_granule [polys]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

template <unsigned Q, bool ROW_IN>
__global__ __launch_bounds__(64 * Q) void k_tile_copy(const u64* __restrict__ in, u64* __restrict__ out) {
  const unsigned t = threadIdx.x, q0 = blockIdx.y * Q;
  const u64* src0 = in + ((size_t)blockIdx.x << 20);
  u64* dst0 = out + ((size_t)blockIdx.x << 20);
  constexpr unsigned LOGQ = Q == 16 ? 4 : 3;
  const unsigned qa = ROW_IN ? (t >> 6) : (t & (Q - 1)), p_lo = ROW_IN ? (t & 63) : (t >> LOGQ);
  const u64* src = ROW_IN ? src0 + ((size_t)(q0 + qa) << 10) + p_lo : src0 + ((size_t)p_lo << 10) + q0 + qa;
  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = ROW_IN ? src[64 * k] : src[(size_t)(64 * k) << 10];
  const unsigned q = t & (Q - 1), r = t >> LOGQ;  // 64 row groups
#pragma unroll
  for (int k = 0; k < 16; ++k) dst0[((size_t)(64 * k + r) << 10) + q0 + q] = x[k] + 1;
}

__global__ __launch_bounds__(256) void k_linear(const u64* __restrict__ in, u64* __restrict__ out, size_t n2) {
  const ulonglong2* i2 = (const ulonglong2*)in;
  ulonglong2* o2 = (ulonglong2*)out;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    ulonglong2 v = i2[i];
    v.x += 1;
    o2[i] = v;
  }
}

template <typename F>
float time_ms(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char** argv) {
  const size_t polys = argc > 1 ? atoi(argv[1]) : 128;
  const size_t words = polys << 20, bytes = words * 8;
  u64 *a, *b;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
  hipMemset(a, 1, bytes);
  hipMemset(b, 2, bytes);
  const double gb = 2.0 * bytes / 1e9;
  float ms;
  ms = time_ms([&] { hipLaunchKernelGGL(k_linear, dim3(4096), dim3(256), 0, 0, a, b, words / 2); }, 10);
  printf("linear copy 16 B/lane         %8.3f ms  %8.1f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL((k_tile_copy<16, false>), dim3(polys, 64), dim3(1024), 0, 0, a, b); }, 10);
  printf("col->col  128-B granules      %8.3f ms  %8.1f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL((k_tile_copy<16, true>), dim3(polys, 64), dim3(1024), 0, 0, a, b); }, 10);
  printf("row->col  128-B granules out  %8.3f ms  %8.1f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL((k_tile_copy<8, false>), dim3(polys, 128), dim3(512), 0, 0, a, b); }, 10);
  printf("col->col   64-B granules      %8.3f ms  %8.1f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL((k_tile_copy<8, true>), dim3(polys, 128), dim3(512), 0, 0, a, b); }, 10);
  printf("row->col   64-B granules out  %8.3f ms  %8.1f GB/s\n", ms, gb / ms * 1e3);
  return 0;
}
