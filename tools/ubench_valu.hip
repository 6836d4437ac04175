// tools/ubench_valu.hip -- VALU issue-rate micro-benchmark for gfx950 (MI355X).
//
// Poseidon-over-Goldilocks is integer-ALU bound: ~1e3 64-bit modular multiplies per
// 72 algorithmic bytes.  The honest roofline for it is the *integer issue rate*, so this tool measures,
// per instruction, how many wave64 instructions a CU retires per second with every SIMD saturated
// (16 independent chains per wave, 8 waves/SIMD).  Results feed DESIGN.md's integer roofline and the
// choice of Goldilocks-multiply / MDS formulation.
//
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
// run:   tools/ubench_valu            (prints one line per instruction)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

enum Op {
  OP_ADD_U32, OP_ADD_CO_PAIR, OP_ADD3_U32, OP_LSHL_ADD_U32, OP_AND_OR, OP_CNDMASK, OP_PERM, OP_ALIGNBIT,
  OP_MUL_LO_U32, OP_MUL_HI_U32, OP_MAD_U64_U32, OP_MAD_U32_U24, OP_MUL_U32_U24, OP_MUL_HI_U32_U24,
  OP_MAD_U32_U16, OP_DOT4_U32_U8, OP_DOT2_U32_U16, OP_LSHLREV_B64, OP_LSHL_ADD_U64, OP_FMA_F64, OP_FMA_F32,
  OP_PK_FMA_F32, OP_PK_ADD_U16, OP_PK_MAD_U16,
  OP_ADD_U32_E64, OP_MOV, OP_XOR, OP_SUB_U32, OP_LSHLREV_B32, OP_BFE, OP_MIN_U32, OP_CNDMASK_E64_SGPR, OP_CMP_CNDMASK,
  OP_CMP_U32_VCC, OP_CMP_U64_SGPR, OP_ADDC_SGPR_CHAIN, OP_SUBB_VCC, OP_MAD_U64_SGPR_MUL, OP_MAD_U64_INLINE_ZERO, OP_MUL_LO_SGPR,
  OP_ADD_CO_ONLY, OP_COUNT
};

static const char* kNames[OP_COUNT] = {
  "v_add_u32", "v_add_co_u32+v_addc_co_u32 (pair)", "v_add3_u32", "v_lshl_add_u32", "v_and_or_b32", "v_cndmask_b32",
  "v_perm_b32", "v_alignbit_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mad_u32_u24", "v_mul_u32_u24",
  "v_mul_hi_u32_u24", "v_mad_u32_u16", "v_dot4_u32_u8", "v_dot2_u32_u16", "v_lshlrev_b64", "v_lshl_add_u64",
  "v_fma_f64", "v_fma_f32", "v_pk_fma_f32", "v_pk_add_u16", "v_pk_mad_u16",
  "v_add_u32_e64 (VOP3 enc)", "v_mov_b32", "v_xor_b32", "v_sub_u32", "v_lshlrev_b32", "v_bfe_u32", "v_min_u32", "v_cndmask_b32_e64 (sgpr mask)",
  "v_cmp_lt_u32+v_cndmask (pair)", "v_cmp_lt_u32_e32 (vcc)", "v_cmp_lt_u64_e64 (sgpr)", "v_addc_co_u32 e64 sgpr chain", "v_subb_co_u32 (vcc)",
  "v_mad_u64_u32 (sgpr mul)", "v_mad_u64_u32 (+0)", "v_mul_lo_u32 (sgpr)", "v_add_co_u32 (vcc out only)"};

#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <int OP>
__global__ __launch_bounds__(256) void bench(uint32_t* out, int iters, uint32_t seed, uint64_t smask) {
  uint32_t r[16];
  uint64_t q[16];
  uint64_t sm[4] = {smask, smask ^ 5, smask + 3, smask * 3};
  uint32_t a = threadIdx.x * 2654435761u + seed, b = (threadIdx.x ^ seed) * 40503u + 17u;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    r[j] = a + j * 977u;
    q[j] = ((uint64_t)(b + j) << 32) | (a ^ (j * 131u));
  }
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == OP_ADD_U32) {
#define M(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_ADD_CO_PAIR) {
#define M(j) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(r[j]), "+v"(r[(j + 8) & 15]) : "v"(a), "v"(b) : "vcc");
      M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
    } else if constexpr (OP == OP_ADD3_U32) {
#define M(j) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[j]) : "v"(a), "v"(b));
      R16(M)
#undef M
    } else if constexpr (OP == OP_LSHL_ADD_U32) {
#define M(j) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_AND_OR) {
#define M(j) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[j]) : "v"(a), "v"(b));
      R16(M)
#undef M
    } else if constexpr (OP == OP_CNDMASK) {
#define M(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[j]) : "v"(a) : );
      R16(M)
#undef M
    } else if constexpr (OP == OP_PERM) {
#define M(j) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[j]) : "v"(a), "v"(b));
      R16(M)
#undef M
    } else if constexpr (OP == OP_ALIGNBIT) {
#define M(j) asm volatile("v_alignbit_b32 %0, %0, %1, 11" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_MUL_LO_U32) {
#define M(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_MUL_HI_U32) {
#define M(j) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_MAD_U64_U32) {
#define M(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[j]) : "v"(a), "v"(b) : "vcc");
      R16(M)
#undef M
    } else if constexpr (OP == OP_MAD_U32_U24) {
#define M(j) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[j]) : "v"(a), "v"(b));
      R16(M)
#undef M
    } else if constexpr (OP == OP_MUL_U32_U24) {
#define M(j) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_MUL_HI_U32_U24) {
#define M(j) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_MAD_U32_U16) {
#define M(j) asm volatile("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(r[j]) : "v"(a), "v"(b));
      R16(M)
#undef M
    } else if constexpr (OP == OP_DOT4_U32_U8) {
#define M(j) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r[j]) : "v"(a), "v"(b));
      R16(M)
#undef M
    } else if constexpr (OP == OP_DOT2_U32_U16) {
#define M(j) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(r[j]) : "v"(a), "v"(b));
      R16(M)
#undef M
    } else if constexpr (OP == OP_LSHLREV_B64) {
#define M(j) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[j]));
      R16(M)
#undef M
    } else if constexpr (OP == OP_LSHL_ADD_U64) {
#define M(j) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[j]) : "v"(q[(j + 1) & 15]));
      R16(M)
#undef M
    } else if constexpr (OP == OP_FMA_F64) {
#define M(j) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(q[j]) : "v"(q[(j + 1) & 15]));
      R16(M)
#undef M
    } else if constexpr (OP == OP_FMA_F32) {
#define M(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[j]) : "v"(a), "v"(b));
      R16(M)
#undef M
    } else if constexpr (OP == OP_PK_FMA_F32) {
#define M(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q[j]) : "v"(q[(j + 1) & 15]));
      R16(M)
#undef M
    } else if constexpr (OP == OP_PK_ADD_U16) {
#define M(j) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_PK_MAD_U16) {
#define M(j) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(r[j]) : "v"(a), "v"(b));
      R16(M)
#undef M

    } else if constexpr (OP == OP_ADD_U32_E64) {
#define M(j) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_MOV) {
#define M(j) asm volatile("v_mov_b32 %0, %1" : "+v"(r[j]) : "v"(r[(j + 1) & 15]));
      R16(M)
#undef M
    } else if constexpr (OP == OP_XOR) {
#define M(j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_SUB_U32) {
#define M(j) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_LSHLREV_B32) {
#define M(j) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[j]));
      R16(M)
#undef M
    } else if constexpr (OP == OP_BFE) {
#define M(j) asm volatile("v_bfe_u32 %0, %0, 3, 22" : "+v"(r[j]));
      R16(M)
#undef M
    } else if constexpr (OP == OP_MIN_U32) {
#define M(j) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[j]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_CNDMASK_E64_SGPR) {
#define M(j) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[j]) : "v"(a), "s"(smask));
      R16(M)
#undef M
    } else if constexpr (OP == OP_CMP_CNDMASK) {
#define M(j) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[j]) : "v"(a), "v"(b) : "vcc");
      M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#undef M
    } else if constexpr (OP == OP_CMP_U32_VCC) {
#define M(j) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[j]), "v"(a) : "vcc");
      R16(M)
#undef M
    } else if constexpr (OP == OP_CMP_U64_SGPR) {
#define M(j) asm volatile("v_cmp_lt_u64_e64 %0, %1, %2" : "=s"(sm[j & 3]) : "v"(q[j]), "v"(q[(j + 1) & 15]));
      R16(M)
#undef M
    } else if constexpr (OP == OP_ADDC_SGPR_CHAIN) {
#define M(j) asm volatile("v_addc_co_u32_e64 %0, %1, %0, %2, %1" : "+v"(r[j]), "+s"(sm[j & 3]) : "v"(a));
      R16(M)
#undef M
    } else if constexpr (OP == OP_SUBB_VCC) {
#define M(j) asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[j]) : "v"(a) : "vcc");
      R16(M)
#undef M
    } else if constexpr (OP == OP_MAD_U64_SGPR_MUL) {
#define M(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[j]) : "v"(a), "s"(seed) : "vcc");
      R16(M)
#undef M
    } else if constexpr (OP == OP_MAD_U64_INLINE_ZERO) {
#define M(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[j]) : "v"(r[j]), "v"(b) : "vcc");
      R16(M)
#undef M
    } else if constexpr (OP == OP_MUL_LO_SGPR) {
#define M(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[j]) : "s"(seed));
      R16(M)
#undef M
    } else if constexpr (OP == OP_ADD_CO_ONLY) {
#define M(j) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[j]) : "v"(a) : "vcc");
      R16(M)
#undef M
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc ^= r[j] ^ (uint32_t)q[j] ^ (uint32_t)(q[j] >> 32) ^ (uint32_t)sm[j & 3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
static void run(uint32_t* d_out, int n_cu, double* base_rate) {
  const int blocks = n_cu * 8, threads = 256, iters = 4096;  // 8 blocks x 4 waves = 32 waves/CU = 8 waves/SIMD
  const int per_iter = (OP == OP_ADD_CO_PAIR || OP == OP_CMP_CNDMASK) ? 8 : 16;      // pairs counted once
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  bench<OP><<<blocks, threads>>>(d_out, 64, 1u, 0x5555aaaa5555aaaaull);  // warm-up
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    bench<OP><<<blocks, threads>>>(d_out, iters, 7u + rep, 0x5555aaaa5555aaaaull);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double wave_instr = (double)blocks * (threads / 64) * iters * per_iter;
  double rate = wave_instr / (best * 1e-3);           // wave-instructions / s, whole chip
  double per_cu_per_ns = rate / n_cu / 1e9;           // wave-instr per CU per ns
  double lane_ops = rate * 64;                         // lane-ops / s
  if (OP == OP_ADD_U32) *base_rate = rate;
  printf("%-36s %8.3f ms  %7.2f Gwave-instr/s  %6.3f /CU/ns  %7.2f Tlane-op/s  rel=%.3f\n", kNames[OP], best,
         rate / 1e9, per_cu_per_ns, lane_ops / 1e12, rate / *base_rate);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

template <int OP>
static void run_all(uint32_t* d_out, int n_cu, double* base) {
  if constexpr (OP < OP_COUNT) {
    run<OP>(d_out, n_cu, base);
    run_all<OP + 1>(d_out, n_cu, base);
  }
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  int n_cu = prop.multiProcessorCount;
  printf("device: %s  CUs=%d  clock=%d MHz\n", prop.name, n_cu, prop.clockRate / 1000);
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4));
  double base = 1;
  run_all<0>(d_out, n_cu, &base);
  CHECK(hipFree(d_out));
  return 0;
}
