// p2mt_plonk.hip -- the permutation-argument stage of the prover: Z and the partial products.
//
// Replaces, inside CircuitData::prove (reference call sites /root/reference/src/mmr/mmr_plonky2_verifier.rs:148 and
// mmr_plonky2_verifier_1_recursion.rs:192,218), plonky2's plonk/prover.rs all_wires_permutation_partial_products
// (wires_permutation_partial_products_and_zs, quotient_chunk_products, partial_products_and_z_gx) [plonky2 source is not in
// the reference tree: parity unpinned; the tests check the permutation argument itself -- the grand product closes
// exactly when the copy constraints hold].
//
// Per challenge (beta, gamma) and row i of the 2^degree_bits-row trace:
//   q[k]   = prod_{j in chunk k} (w_j + beta k_j x_i + gamma) / (w_j + beta sigma_j(x_i) + gamma),  x_i = w^i
//   pp[k]  = Z(x_i) q[0] .. q[k]   (k < num_prods),      Z(x_{i+1}) = Z(x_i) q[0] .. q[num_prods],  Z(x_0) = 1
// Two kernels: the chunk quotients are independent per (challenge, row, chunk) -- one lane each, one modular inversion
// per lane (a 74-multiplication addition chain for x^(p-2)); the running product over rows is a multiplicative scan in
// LDS on one workgroup per challenge.  Columns are read and written column-major ([column][row]), coalesced over rows.
#include "tree_common.hip.h"

#include <vector>

using namespace p2mt_dev;
using p2mt::DevBuf;
using p2mt::rt;

namespace {

// x^(p-2), p - 2 = (2^31 - 1) 2^33 + (2^32 - 1); inverse of 0 is 0
GL_DEV u64 gl_inv(u64 x) {
  auto sqn = [](u64 v, int k) {
    for (int i = 0; i < k; ++i) v = gl::sqr(v);
    return v;
  };
  const u64 t2 = gl::mul(gl::sqr(x), x);         // x^(2^2 - 1)
  const u64 t4 = gl::mul(sqn(t2, 2), t2);        // 2^4 - 1
  const u64 t8 = gl::mul(sqn(t4, 4), t4);
  const u64 t16 = gl::mul(sqn(t8, 8), t8);
  const u64 t24 = gl::mul(sqn(t16, 8), t8);
  const u64 t28 = gl::mul(sqn(t24, 4), t4);
  const u64 t30 = gl::mul(sqn(t28, 2), t2);
  const u64 t31 = gl::mul(gl::sqr(t30), x);      // x^(2^31 - 1)
  const u64 t32 = gl::mul(gl::sqr(t31), x);      // x^(2^32 - 1)
  return gl::mul(sqn(t31, 33), t32);
}

// q[(c * num_chunks + k) * n + i]; *zero_den is set when a denominator product vanishes (plonky2 panics there)
__global__ __launch_bounds__(kBlock) void k_pp_chunks(const u64* __restrict__ wires, const u64* __restrict__ sigmas,
                                                      const u64* __restrict__ k_is, const u64* __restrict__ betas,
                                                      const u64* __restrict__ gammas, u32 num_routed, u32 log_n, u32 chunk,
                                                      u32 num_chunks, u64 w, u64* __restrict__ q, int* __restrict__ zero_den, BatchArg ba) {
  wires = bp(wires, ba);
  sigmas = bp(sigmas, ba);
  k_is = bp(k_is, ba);
  betas = bp(betas, ba);
  gammas = bp(gammas, ba);
  q = bp(q, ba);
  zero_den = bp(zero_den, ba);
  const u32 n = 1u << log_n, i = blockIdx.x * kBlock + threadIdx.x, k = blockIdx.y % num_chunks, c = blockIdx.y / num_chunks;
  if (i >= n) return;
  const u64 beta = betas[c], gamma = gammas[c];
  const u64 bx = gl::mul(beta, gl::pow(w, i));  // beta * x_i
  u64 num = 1, den = 1;
  const u32 j1 = min(num_routed, (k + 1) * chunk);
  for (u32 j = k * chunk; j < j1; ++j) {
    const u64 wg = gl::add(wires[(size_t)j * n + i], gamma);
    num = gl::mul(num, gl::mul_add(bx, k_is[j], wg));
    den = gl::mul(den, gl::mul_add(beta, sigmas[(size_t)j * n + i], wg));
  }
  den = gl::canon(den);
  if (den == 0) *zero_den = 1;
  q[((size_t)c * num_chunks + k) * n + i] = gl::canon(gl::mul(num, gl_inv(den)));
}

// One workgroup per challenge: Z (exclusive running product of the row totals) and the partial products.
constexpr int kScanBlock = 1024;
__global__ __launch_bounds__(kScanBlock) void k_pp_scan(const u64* __restrict__ q, u32 log_n, u32 num_chunks, u32 num_challenges,
                                                        u64* __restrict__ out, BatchArg ba) {
  q = bp(q, ba);
  out = bp(out, ba);
  __shared__ u64 S[kScanBlock];
  const u32 n = 1u << log_n, c = blockIdx.x, t = threadIdx.x;
  const u32 len = n >= (u32)kScanBlock ? n / kScanBlock : 1, T = n / len;
  const u64* qc = q + (size_t)c * num_chunks * n;
  u64 loc = 1;
  if (t < T)
    for (u32 i = t * len; i < (t + 1) * len; ++i)
      for (u32 k = 0; k < num_chunks; ++k) loc = gl::mul(loc, qc[(size_t)k * n + i]);
  S[t] = loc;
  __syncthreads();
  for (u32 d = 1; d < T; d *= 2) {  // inclusive multiplicative scan
    u64 v = S[t];
    if (t >= d && t < T) v = gl::mul(v, S[t - d]);
    __syncthreads();
    S[t] = v;
    __syncthreads();
  }
  if (t >= T) return;
  u64 z = t ? S[t - 1] : 1;  // Z at this thread's first row
  u64* z_col = out + (size_t)c * n;
  u64* pp = out + (size_t)num_challenges * n + (size_t)c * (num_chunks - 1) * n;
  for (u32 i = t * len; i < (t + 1) * len; ++i) {
    z_col[i] = gl::canon(z);
    u64 acc = z;
    for (u32 k = 0; k < num_chunks; ++k) {
      acc = gl::mul(acc, qc[(size_t)k * n + i]);
      if (k + 1 < num_chunks) pp[(size_t)k * n + i] = gl::canon(acc);
    }
    z = acc;  // the last product is Z(g x)
  }
}

inline u64 h_mul(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % gl::P); }
inline u64 h_pow(u64 a, u64 e) {
  u64 r = 1;
  for (; e; e >>= 1, a = h_mul(a, a))
    if (e & 1) r = h_mul(r, a);
  return r;
}

}  // namespace

int p2mt::partial_products_async_dev(const uint64_t* d_wires, const uint64_t* d_sigmas, const uint64_t* d_k_is,
                                     const uint64_t* d_betas, const uint64_t* d_gammas, size_t num_challenges,
                                     size_t num_routed, unsigned degree_bits, unsigned chunk, uint64_t* d_q, uint64_t* d_out,
                                     int* d_zero_den) {
  const size_t n = (size_t)1 << degree_bits;
  const size_t num_chunks = (num_routed + chunk - 1) / chunk;
  hipStream_t st = rt().stream;
  u64 w = h_pow(7, (gl::P - 1) >> 32);  // primitive 2^degree_bits-th root of unity
  for (unsigned i = degree_bits; i < 32; ++i) w = h_mul(w, w);
  hipLaunchKernelGGL(k_pp_chunks, bgrid(grid_for(n), (unsigned)(num_chunks * num_challenges)), dim3(kBlock), 0, st, d_wires,
                     d_sigmas, d_k_is, d_betas, d_gammas, (u32)num_routed, degree_bits, chunk, (u32)num_chunks, w, d_q, d_zero_den,
                     barg());
  P2MT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pp_scan, bgrid((unsigned)num_challenges), dim3(kScanBlock), 0, st, (const u64*)d_q, degree_bits,
                     (u32)num_chunks, (u32)num_challenges, d_out, barg());
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}

extern "C" int p2mt_permutation_partial_products_dev(const uint64_t* d_wires, const uint64_t* d_sigmas, const uint64_t* k_is,
                                                     const uint64_t* betas, const uint64_t* gammas, size_t num_challenges,
                                                     size_t num_routed, unsigned degree_bits, unsigned chunk, uint64_t* d_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!d_wires || !d_sigmas || !k_is || !betas || !gammas || !d_out) return p2mt::fail(P2MT_EINVAL, "partial_products: null pointer");
  if (chunk < 2 || num_routed == 0 || num_routed > 4096 || degree_bits > 24 || num_challenges == 0 || num_challenges > 16)
    return p2mt::fail(P2MT_EINVAL, "partial_products: bad shape (max_degree must be > 1)");
  const size_t n = (size_t)1 << degree_bits;
  const size_t num_chunks = (num_routed + chunk - 1) / chunk;
  // scratch: q | k_is | betas | gammas | zero-denominator flag
  const size_t q_words = num_challenges * num_chunks * n;
  u64* ws;
  P2MT_TRY(p2mt::scratch_get(p2mt::kScratchPlonk, (q_words + num_routed + 2 * num_challenges + 1) * 8, (void**)&ws));
  u64 *d_q = ws, *d_k = ws + q_words, *d_bg = d_k + num_routed;
  int* d_flag = reinterpret_cast<int*>(d_bg + 2 * num_challenges);
  std::vector<u64> h(num_routed + 2 * num_challenges + 1, 0);
  for (size_t j = 0; j < num_routed; ++j) h[j] = k_is[j] % gl::P;
  for (size_t c = 0; c < num_challenges; ++c) {
    h[num_routed + c] = betas[c] % gl::P;
    h[num_routed + num_challenges + c] = gammas[c] % gl::P;
  }
  hipStream_t st = rt().stream;
  P2MT_HIP(hipMemcpyAsync(d_k, h.data(), h.size() * 8, hipMemcpyHostToDevice, st));
  P2MT_TRY(p2mt::partial_products_async_dev(d_wires, d_sigmas, d_k, d_bg, d_bg + num_challenges, num_challenges, num_routed,
                                            degree_bits, chunk, d_q, d_out, d_flag));
  int flag = 0;
  P2MT_HIP(hipMemcpyAsync(&flag, d_flag, sizeof flag, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  if (flag) return p2mt::fail(P2MT_EINVAL, "partial_products: zero denominator (plonky2 panics on this division)");
  return P2MT_OK;
  });
}

extern "C" int p2mt_permutation_partial_products(const uint64_t* wires, const uint64_t* sigmas, const uint64_t* k_is,
                                                 const uint64_t* betas, const uint64_t* gammas, size_t num_challenges,
                                                 size_t num_routed, unsigned degree_bits, unsigned chunk, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!wires || !sigmas || !out) return p2mt::fail(P2MT_EINVAL, "partial_products: null pointer");
  if (chunk < 2 || num_routed == 0 || num_routed > 4096 || degree_bits > 24 || num_challenges == 0 || num_challenges > 16)
    return p2mt::fail(P2MT_EINVAL, "partial_products: bad shape (max_degree must be > 1)");
  const size_t n = (size_t)1 << degree_bits;
  const size_t in_bytes = num_routed * n * 8;
  const size_t out_words = num_challenges * ((num_routed + chunk - 1) / chunk) * n;
  DevBuf bw, bs, bo;
  P2MT_TRY(bw.alloc(in_bytes));
  P2MT_TRY(bs.alloc(in_bytes));
  P2MT_TRY(bo.alloc(out_words * 8));
  hipStream_t st = rt().stream;
  P2MT_HIP(hipMemcpyAsync(bw.p, wires, in_bytes, hipMemcpyHostToDevice, st));
  P2MT_HIP(hipMemcpyAsync(bs.p, sigmas, in_bytes, hipMemcpyHostToDevice, st));
  P2MT_TRY(p2mt_permutation_partial_products_dev(bw.as<u64>(), bs.as<u64>(), k_is, betas, gammas, num_challenges, num_routed,
                                                 degree_bits, chunk, bo.as<u64>()));
  P2MT_HIP(hipMemcpyAsync(out, bo.p, out_words * 8, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  return P2MT_OK;
  });
}
