// tools/ubench_mfma32.hip -- the Poseidon MDS layer on v_mfma_i32_32x32x32_i8 with one hash per lane: lane-local layout + issue cost.
//
// Round 3's first matrix-pipe experiment (tools/ubench_mfma.hip, profiles/r03_mds_mfma_ab.txt) used the 4x4x4 16-block form: 72 MFMAs
// per MDS layer, each keeping the SIMD's VALU port for 4.4 cycles -- a wash.  The port cost is per INSTRUCTION, not per MAC, so the
// question here is whether ONE large MFMA per 8-bit limb (8 per layer) can stay lane-local.  It can, with a block-structured A:
//   * B (16 bytes per lane) = byte l of the 12 state words of the lane's OWN hash, xor 0x80 (signed bytes); bytes 12..15 = -128 against
//     -rowsum / 4 in A's K slots 12..15, which adds 128 rowsum back: the result registers hold the UNSIGNED limb sums;
//   * lanes 0..31 feed the first 16 K slots of column n = lane, lanes 32..63 the other 16 K slots of column n = lane - 32;
//   * result register v of lane (n, half) is row 8*(v/4) + 4*half + v%4 of column n.  So rows {0-3, 8-11, 16-19} (v = 0..11 of half 0)
//     carry M in the FIRST 16 K slots only and rows {4-7, 12-15, 20-23} carry M in the SECOND 16 K slots only:
//     register v < 12 of lane n      = sum_j M[v][j] * byte_l(word j of hash n),
//     register v < 12 of lane n + 32 = sum_j M[v][j] * byte_l(word j of hash n + 32).        No cross-lane movement; 28 % of the MACs useful.
// This tool (1) checks that claim against a host model with the real MDS matrix and random bytes and (2) measures how a stream of
// v_mad_u64_u32 slows down when 1, 2, 4 such MFMAs per 16 mads share the instruction stream, at 1..4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma32.hip -o tools/ubench_mfma32 ; run: tools/ubench_mfma32
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

static const int MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static int mds(int r, int c) { return MDS_CIRC[(c - r + 12) % 12] + ((r == 0 && c == 0) ? 8 : 0); }

__global__ void k_probe(const v4i* a, const v4i* b, v16i* d) {
  v16i c = {};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  d[threadIdx.x] = c;
}

// MODE 0: C/D in VGPRs (accumulating)   1: C/D in AGPRs   2: srcC = inline 0, D in VGPRs (a fresh product, what one limb of the MDS needs)
// MODE 3: v_mfma_i32_16x16x64_i8, C/D (4 registers) in VGPRs   4: the same with srcC = 0
template <int MODE>
__device__ __forceinline__ void mfma_op(v16i& acc, const v4i& va, const v4i& vb) {
  if constexpr (MODE == 0) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(va), "v"(vb));
  if constexpr (MODE == 1) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(va), "v"(vb));
  if constexpr (MODE == 2) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=v"(acc) : "v"(va), "v"(vb));
  if constexpr (MODE == 3) {
    v4i c = {acc[0], acc[1], acc[2], acc[3]};
    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(va), "v"(vb));
    acc[0] = c[0]; acc[1] = c[1]; acc[2] = c[2]; acc[3] = c[3];
  }
  if constexpr (MODE == 4) {
    v4i c;
    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=v"(c) : "v"(va), "v"(vb));
    acc[0] = c[0]; acc[1] = c[1]; acc[2] = c[2]; acc[3] = c[3];
  }
}

template <int MADS, int MFMAS, int MODE = 0>
__global__ __launch_bounds__(256) void k_mix(uint32_t* out, int iters, uint32_t seed) {
  uint64_t q[16];
  v16i acc[4];
  const uint32_t a = threadIdx.x * 2654435761u + seed, b = (threadIdx.x ^ seed) * 40503u + 17u;
  v4i va = {(int)a, (int)b, (int)(a ^ b), (int)(a + b)}, vb = {(int)b, (int)a, (int)(a * 3u), (int)(b * 5u)};
#pragma unroll
  for (int j = 0; j < 16; ++j) q[j] = ((uint64_t)(b + j) << 32) | (a ^ (j * 131u));
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[j][t] = (int)(a + j * 16 + t);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < (MADS > MFMAS ? MADS : MFMAS); ++j) {
      if (j < MADS) {
        uint64_t unused;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(q[j & 15]), "=s"(unused) : "v"(a), "v"(b));
      }
      if constexpr (MADS > 0 && MFMAS > 0) {  // spread the MFMAs evenly over the mad stream
        constexpr int stride = MADS / (MFMAS < MADS ? MFMAS : MADS);
        if (j % stride == 0) mfma_op<MODE>(acc[(j / stride) & 3], va, vb);
      } else if constexpr (MFMAS > 0) {
        if (j < MFMAS) mfma_op<MODE>(acc[j & 3], va, vb);
      }
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) r ^= (uint32_t)q[j] ^ (uint32_t)(q[j] >> 32);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 16; ++t) r ^= acc[j][t];
  if (r == 0x12345678u) out[0] = r;
}

template <int MADS, int MFMAS, int MODE = 0>
static void run(const char* name, int waves_per_simd, int n_cu) {
  uint32_t* d;
  CHECK(hipMalloc(&d, 4));
  const int iters = 20000;
  const int blocks = n_cu * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mix<MADS, MFMAS, MODE>), dim3(blocks), dim3(256), 0, 0, d, 100, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_mix<MADS, MFMAS, MODE>), dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double ns_per_iter_per_simd = ms * 1e6 / iters / waves_per_simd;
  printf("%-36s waves/SIMD=%d  %.3f ms  %7.2f ns per (iteration x wave)", name, waves_per_simd, ms, ns_per_iter_per_simd);
  if (MADS) printf("  [%.3f mad/SIMD/ns]", MADS / ns_per_iter_per_simd);
  if (MFMAS) printf("  [%.4f mfma32/SIMD/ns]", MFMAS / ns_per_iter_per_simd);
  printf("\n");
  CHECK(hipFree(d));
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int n_cu = p.multiProcessorCount;
  printf("device: %s  CUs=%d  clock=%d MHz\n", p.gcnArchName, n_cu, p.clockRate / 1000);
  {  // 1. lane-local layout probe with the real MDS matrix
    std::vector<int8_t> A(64 * 16, 0), B(64 * 16);
    srand(11);
    for (auto& x : B) x = (int8_t)(rand() % 256 - 128);
    for (int l = 0; l < 64; ++l) {
      const int i = l % 32, half = l / 32, sub = (i / 4) % 2, v = 4 * (i / 8) + i % 4;
      if (sub == half && v < 12) {
        int rowsum = 0;
        for (int j = 0; j < 12; ++j) A[l * 16 + j] = (int8_t)mds(v, j), rowsum += mds(v, j);
        for (int j = 12; j < 16; ++j) A[l * 16 + j] = (int8_t)(-rowsum / 4);  // x the constant byte -128 in B: + 128 rowsum (the shipped layout)
      }
    }
    for (int l = 0; l < 64; ++l)
      for (int j = 12; j < 16; ++j) B[l * 16 + j] = (int8_t)-128;
    v4i *da, *db;
    v16i* dd;
    CHECK(hipMalloc(&da, 1024));
    CHECK(hipMalloc(&db, 1024));
    CHECK(hipMalloc(&dd, 4096));
    CHECK(hipMemcpy(da, A.data(), 1024, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, B.data(), 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<int> d(64 * 16);
    CHECK(hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost));
    int bad = 0, bad_spare = 0;
    for (int l = 0; l < 64; ++l)
      for (int v = 0; v < 16; ++v) {
        int ref = 0;
        if (v < 12)  // the UNSIGNED limb sum: bytes were xor-ed with 0x80 (b - 128 as a signed byte), K slots 12..15 add 128 rowsum back
          for (int j = 0; j < 12; ++j) ref += mds(v, j) * ((int)B[l * 16 + j] + 128);
        if (ref != d[l * 16 + v]) { if (v < 12) ++bad; else ++bad_spare; }
      }
    printf("lane-local layout probe (register v < 12 of lane n = row v of M x UNSIGNED bytes of lane n's own B, offset via K slots 12..15; registers 12..15 = 0): %s (%d mismatches, %d in the spare registers)\n",
           (bad || bad_spare) ? "WRONG" : "confirmed", bad, bad_spare);
    if (bad) {
      for (int l = 0; l < 64; l += 9) {
        printf("  lane %2d got:", l);
        for (int v = 0; v < 16; ++v) printf(" %d", d[l * 16 + v]);
        printf("\n");
      }
    }
  }
  for (int w = 1; w <= 4; ++w) {
    run<16, 0>("16 mad", w, n_cu);
    run<0, 4>("4 mfma32", w, n_cu);
    run<16, 1>("16 mad + 1 mfma32", w, n_cu);
    run<16, 2>("16 mad + 2 mfma32", w, n_cu);
    run<16, 4>("16 mad + 4 mfma32", w, n_cu);
    run<16, 8>("16 mad + 8 mfma32", w, n_cu);
  }
  printf("-- operand placement and shape, 4 waves per SIMD\n");
  run<0, 4, 1>("4 mfma32 (AGPR acc)", 4, n_cu);
  run<16, 1, 1>("16 mad + 1 mfma32 (AGPR acc)", 4, n_cu);
  run<16, 2, 1>("16 mad + 2 mfma32 (AGPR acc)", 4, n_cu);
  run<16, 4, 1>("16 mad + 4 mfma32 (AGPR acc)", 4, n_cu);
  run<0, 4, 2>("4 mfma32 (srcC=0)", 4, n_cu);
  run<16, 1, 2>("16 mad + 1 mfma32 (srcC=0)", 4, n_cu);
  run<16, 2, 2>("16 mad + 2 mfma32 (srcC=0)", 4, n_cu);
  run<16, 4, 2>("16 mad + 4 mfma32 (srcC=0)", 4, n_cu);
  run<0, 4, 3>("4 mfma16x16x64", 4, n_cu);
  run<16, 1, 3>("16 mad + 1 mfma16x16x64", 4, n_cu);
  run<16, 2, 3>("16 mad + 2 mfma16x16x64", 4, n_cu);
  run<16, 4, 3>("16 mad + 4 mfma16x16x64", 4, n_cu);
  run<16, 8, 3>("16 mad + 8 mfma16x16x64", 4, n_cu);
  run<16, 2, 4>("16 mad + 2 mfma16x16x64 (srcC=0)", 4, n_cu);
  run<16, 4, 4>("16 mad + 4 mfma16x16x64 (srcC=0)", 4, n_cu);
  run<16, 8, 4>("16 mad + 8 mfma16x16x64 (srcC=0)", 4, n_cu);
  return 0;
}
