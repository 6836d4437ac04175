// tools/ubench_mfma.hip -- does the matrix pipe help an integer-issue-bound kernel on gfx950?
//
// Round 3, VERDICT item 7 ("measure the MFMA MDS instead of dismissing it").  The Poseidon MDS layer is a 12 x 12 contraction with
// 6-bit constants; on 8-bit limbs it fits v_mfma_i32_4x4x4_16b_i8 with NO cross-lane movement: the instruction is 16 independent
// 4x4x4 blocks, block b = lanes 4b..4b+3, so with lane = hash the B operand of lane n is four bytes of hash n (the same byte of four
// state words), the A operand is a per-lane constant (4 MDS entries of row i = lane % 4) and lane n receives rows r0..r0+3 of ITS OWN
// hash in its four result registers.  What decides whether that pays is how the MFMA shares the issue port with the VALU
// instructions that remain (byte transposes, limb recombination, S-boxes).  This tool measures:
//   1. the lane layout of the instruction (probe against a host model; asserts the claim above),
//   2. issue rates: v_mad_u64_u32 alone, the MFMA alone, and mixes of the two in one instruction stream,
//      at 1..3 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/ubench_mfma ; run: tools/ubench_mfma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k_probe(const uint32_t* a, const uint32_t* b, int* d) {
  const unsigned l = threadIdx.x;
  v4i c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_4x4x4i8((int)a[l], (int)b[l], c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[4 * l + i] = c[i];
}

// MADS v_mad_u64_u32 and MFMAS v_mfma_i32_4x4x4_16b_i8 per loop iteration, all chains independent (16 mad chains, 8 MFMA chains)
template <int MADS, int MFMAS>
__global__ __launch_bounds__(256) void k_mix(uint32_t* out, int iters, uint32_t seed) {
  uint64_t q[16];
  v4i acc[8];
  const uint32_t a = threadIdx.x * 2654435761u + seed, b = (threadIdx.x ^ seed) * 40503u + 17u;
#pragma unroll
  for (int j = 0; j < 16; ++j) q[j] = ((uint64_t)(b + j) << 32) | (a ^ (j * 131u));
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = v4i{(int)a, (int)b, j, 1};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < (MADS > MFMAS ? MADS : MFMAS); ++j) {
      if (j < MADS) {
        uint64_t unused;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(q[j & 15]), "=s"(unused) : "v"(a), "v"(b));
      }
      if (j < MFMAS) {
        asm volatile("v_mfma_i32_4x4x4_16b_i8 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(a), "v"(b));
      }
    }
  }
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) r ^= (uint32_t)q[j] ^ (uint32_t)(q[j] >> 32);
#pragma unroll
  for (int j = 0; j < 8; ++j) r ^= acc[j][0] ^ acc[j][1] ^ acc[j][2] ^ acc[j][3];
  if (r == 0x12345678u) out[0] = r;
}

template <int MADS, int MFMAS>
static void run(const char* name, int waves_per_simd, int n_cu) {
  uint32_t* d;
  CHECK(hipMalloc(&d, 4));
  const int iters = 20000;
  const int blocks = n_cu * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_mix<MADS, MFMAS>), dim3(blocks), dim3(256), 0, 0, d, 100, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_mix<MADS, MFMAS>), dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  // per SIMD: waves_per_simd waves x iters x (MADS + MFMAS) instructions
  const double ns_per_iter_per_simd = ms * 1e6 / iters / waves_per_simd;
  printf("%-28s waves/SIMD=%d  %.3f ms  %7.2f ns per (iteration x wave)  = %.2f ns/mad-equivalent slot", name, waves_per_simd, ms,
         ns_per_iter_per_simd, ns_per_iter_per_simd / (MADS + MFMAS ? MADS + MFMAS : 1));
  if (MADS) printf("  [%.3f mad/SIMD/ns]", MADS / ns_per_iter_per_simd);
  if (MFMAS) printf("  [%.3f mfma/SIMD/ns]", MFMAS / ns_per_iter_per_simd);
  printf("\n");
  CHECK(hipFree(d));
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int n_cu = p.multiProcessorCount;
  printf("device: %s  CUs=%d  clock=%d MHz\n", p.gcnArchName, n_cu, p.clockRate / 1000);
  {  // 1. lane layout probe
    std::vector<uint32_t> a(64), b(64);
    std::vector<int8_t> A(16 * 4 * 4), B(16 * 4 * 4);  // A[blk][i][k], B[blk][k][j]
    srand(7);
    for (auto& x : A) x = (int8_t)(rand() % 200 - 100);
    for (auto& x : B) x = (int8_t)(rand() % 256 - 128);
    for (int l = 0; l < 64; ++l) {
      const int blk = l / 4, ij = l % 4;
      uint32_t wa = 0, wb = 0;
      for (int k = 0; k < 4; ++k) {
        wa |= (uint32_t)(uint8_t)A[(blk * 4 + ij) * 4 + k] << (8 * k);  // lane (blk, i): A[i][k], k in byte k
        wb |= (uint32_t)(uint8_t)B[(blk * 4 + k) * 4 + ij] << (8 * k);  // lane (blk, j): B[k][j], k in byte k
      }
      a[l] = wa;
      b[l] = wb;
    }
    uint32_t *da, *db;
    int* dd;
    CHECK(hipMalloc(&da, 256));
    CHECK(hipMalloc(&db, 256));
    CHECK(hipMalloc(&dd, 1024));
    CHECK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dd);
    std::vector<int> d(256);
    CHECK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 4; ++i) {  // claim: lane (blk, j) register i = D[i][j]
        const int blk = l / 4, j = l % 4;
        int ref = 0;
        for (int k = 0; k < 4; ++k) ref += (int)A[(blk * 4 + i) * 4 + k] * (int)B[(blk * 4 + k) * 4 + j];
        if (ref != d[4 * l + i]) ++bad;
      }
    printf("layout probe (lane 4b+j holds D_b[0..3][j]; A lane 4b+i = A_b[i][0..3]; B lane 4b+j = B_b[0..3][j]): %s (%d mismatches)\n",
           bad ? "WRONG" : "confirmed", bad);
  }
  for (int w = 1; w <= 3; ++w) {
    run<16, 0>("16 mad", w, n_cu);
    run<0, 8>("8 mfma", w, n_cu);
    run<16, 2>("16 mad + 2 mfma", w, n_cu);
    run<16, 4>("16 mad + 4 mfma", w, n_cu);
    run<16, 8>("16 mad + 8 mfma", w, n_cu);
    run<16, 16>("16 mad + 16 mfma", w, n_cu);
    run<8, 16>("8 mad + 16 mfma", w, n_cu);
  }
  return 0;
}
