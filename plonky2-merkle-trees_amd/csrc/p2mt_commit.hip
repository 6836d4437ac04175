// p2mt_commit.hip -- the commit step of the Plonky2 prover: Goldilocks NTT / IFFT, x2^r coset LDE,
// wide-leaf Poseidon sponge and Merkle-cap tree (PolynomialBatch::from_values / from_coeffs).
//
// Replaces, inside CircuitData::prove (call sites /root/reference/src/mmr/mmr_plonky2_verifier.rs:148,
// /root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:192,218), the third-party (absent)
// plonky2_field 0.1.0 fft.rs (fft_with_options / ifft_with_options), plonky2 @3b21b87 fri/oracle.rs
// (PolynomialBatch) and hash/merkle_tree.rs (MerkleTree::new).  Conventions (parity
// unpinned by the reference): fft(c)[i] = f(w_n^i) natural order; LDE point i = shift * w_N^i; leaf i = all
// polynomials at one point; leaves stored bit-reversed; cap = the 2^cap_height subtree roots.
//
// MI355X mapping
//   * A transform of up to 2^12 points lives in one workgroup's LDS (32 KB): decimation-in-frequency radix-2
//     butterflies, natural order in, bit-reversed out, twiddles read from an L2-resident table.  Larger
//     transforms run their first (log_n - 12) stages as global passes, then the same LDS kernel per 2^12 chunk.
//   * The x2^r LDE is never one big transform: with i = 2^r k + j, f(s w_N^i) = NTT_n[c_m (s w_N^j)^m](k), i.e.
//     2^r independent size-n LDS transforms per polynomial, and after the leaf bit-reversal coset j is the
//     CONTIGUOUS block brev_r(j) in bit-reversed-k order -- exactly what the DIF kernel emits.  So the LDE
//     writes coalesced, poly-major, already in leaf order; no separate bit-reversal pass exists.
//   * The leaf sponge reads that poly-major matrix column-wise (lane i reads point i of every polynomial:
//     coalesced 8-byte loads), so hashing never needs the transposed leaf-major matrix; the transpose to
//     plonky2's leaf-major `leaves` is only produced when the caller asks for it.
//   All of it is HBM/LDS-bound integer work: no MFMA.
#include "tree_common.hip.h"
#include "ntt_arith.hip.h"

#include <map>
#include <mutex>
#include <tuple>
#include <utility>

using gl::u32;
using gl::u64;
using p2mt::BatchArg;
using p2mt_dev::barg;
using p2mt_dev::bgrid;
using p2mt_dev::bp;

namespace {

constexpr int kBlock = 256;
constexpr unsigned kLdsLog = 12;  // 2^12 x 8 B = 32 KB per workgroup

// canonical-form helpers (values in LDS are kept < p)
GL_DEV u64 cadd(u64 a, u64 b) {
  const u64 s = a + b;
  return (s < a || s >= gl::P) ? s - gl::P : s;
}
GL_DEV u64 csub(u64 a, u64 b) { return a >= b ? a - b : a - b + gl::P; }
GL_DEV u64 cmul(u64 a, u64 b) { return gl::canon(gl::mul(a, b)); }

GL_DEV u32 brev32(u32 x, unsigned bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// ---------------------------------------------------------------- tables
// tw[i] = root^i, i in [0, count)
__global__ __launch_bounds__(kBlock) void k_powers(u64* __restrict__ tw, u64 root, size_t count) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= count) return;
  tw[i] = gl::canon(gl::pow(root, i));
}

// ---------------------------------------------------------------- DIF butterflies
// One DIF stage in global memory: n-point transforms, stage s pairs elements `half = n >> (s+1)` apart.
__global__ __launch_bounds__(kBlock) void k_ntt_global_stage(u64* __restrict__ data, unsigned log_n, unsigned s,
                                                             const u64* __restrict__ tw, size_t n_polys, BatchArg ba) {
  data = bp(data, ba);
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t half_n = (size_t)1 << (log_n - 1);
  if (t >= half_n * n_polys) return;
  const size_t poly = t >> (log_n - 1), b = t & (half_n - 1);
  const size_t half = (size_t)1 << (log_n - 1 - s);
  const size_t blk = b / half, j = b & (half - 1);
  u64* p = data + (poly << log_n) + blk * 2 * half + j;
  const u64 x = gl::canon(p[0]), y = gl::canon(p[half]);
  p[0] = cadd(x, y);
  p[half] = cmul(csub(x, y), tw[j << s]);
}

// In-LDS DIF: buf[0 .. 2^c) natural -> bit-reversed; the chunk is the tail (stages first_stage .. log_n-1) of an
// n-point transform, so twiddles are tw_n[(j << s_local) << first_stage].
GL_DEV void lds_dif(u64* buf, unsigned c, unsigned first_stage, const u64* __restrict__ tw) {
  const unsigned half_c = 1u << (c - 1);
  for (unsigned s = 0; s < c; ++s) {
    const unsigned half = half_c >> s;
    for (unsigned t = threadIdx.x; t < half_c; t += kBlock) {
      const unsigned blk = t / half, j = t & (half - 1);
      const unsigned i0 = blk * 2 * half + j, i1 = i0 + half;
      const u64 x = buf[i0], y = buf[i1];
      buf[i0] = cadd(x, y);
      buf[i1] = cmul(csub(x, y), tw[((size_t)j << s) << first_stage]);
    }
    __syncthreads();
  }
}

// Tail of a (possibly large) transform: one workgroup per 2^c chunk.  Output stays in DIF (bit-reversed) order.
__global__ __launch_bounds__(kBlock) void k_ntt_lds_tail(u64* __restrict__ data, unsigned log_n, unsigned c,
                                                         const u64* __restrict__ tw, BatchArg ba) {
  data = bp(data, ba);
  extern __shared__ __attribute__((aligned(16))) u64 buf[];
  u64* chunk = data + ((size_t)blockIdx.x << c);
  const unsigned m = 1u << c;
  for (unsigned i = threadIdx.x; i < m; i += kBlock) buf[i] = gl::canon(chunk[i]);
  __syncthreads();
  lds_dif(buf, c, log_n - c, tw);
  for (unsigned i = threadIdx.x; i < m; i += kBlock) chunk[i] = buf[i];
}

// out[brev(q)] = in[q] * scale  (bit-reversal back to natural order, optional 1/n scaling for the inverse)
__global__ __launch_bounds__(kBlock) void k_bitrev_scale(const u64* __restrict__ in, u64* __restrict__ out, unsigned log_n,
                                                         size_t n_polys, u64 scale, BatchArg ba) {
  in = bp(in, ba);
  out = bp(out, ba);
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= (n_polys << log_n)) return;
  const size_t poly = t >> log_n, q = t & (((size_t)1 << log_n) - 1);
  const size_t r = log_n ? (__brevll(q) >> (64 - log_n)) : 0;
  u64 v = in[t];
  if (scale != 1) v = cmul(v, scale);
  out[(poly << log_n) + r] = v;
}

// Whole inverse transform of one polynomial (n <= 2^12) in one launch: values in natural order -> LDS, inverse DIF, and
// the bit-reversal back to natural order fused with the 1/n (and, for coset_ifft, shift^-k) scaling.  Replaces the
// copy + DIF + bit-reversal launches of the small circuits, where the launch count is what limits concurrent provers.
__global__ __launch_bounds__(kBlock) void k_ifft_small(const u64* __restrict__ in, u64* __restrict__ out, unsigned log_n,
                                                       const u64* __restrict__ tw_inv, u64 n_inv, u64 shift_inv, BatchArg ba) {
  in = bp(in, ba);
  out = bp(out, ba);
  extern __shared__ __attribute__((aligned(16))) u64 buf[];
  const unsigned n = 1u << log_n;
  const u64* src = in + ((size_t)blockIdx.x << log_n);
  u64* dst = out + ((size_t)blockIdx.x << log_n);
  for (unsigned i = threadIdx.x; i < n; i += kBlock) buf[i] = gl::canon(src[i]);
  __syncthreads();
  if (log_n) lds_dif(buf, log_n, 0, tw_inv);
  for (unsigned q = threadIdx.x; q < n; q += kBlock) {
    const unsigned r = brev32(q, log_n);
    u64 v = cmul(buf[q], n_inv);
    if (shift_inv != 1) v = cmul(v, gl::pow(shift_inv, r));
    dst[r] = v;
  }
}

// ---------------------------------------------------------------- coset LDE, one workgroup per (poly, coset)
// coeffs [n_polys][n] natural; out [n_polys][n << r] in LEAF ORDER: out[p][brev_r(j) * n + q] = DIF position q of
// NTT_n[c_m * cp[j][m]], cp[j][m] = (shift * w_N^j)^m.
__global__ __launch_bounds__(kBlock) void k_coset_lde(const u64* __restrict__ coeffs, unsigned log_n, unsigned rate_bits,
                                                      const u64* __restrict__ coset_pow, const u64* __restrict__ tw,
                                                      u64* __restrict__ out, BatchArg ba) {
  coeffs = bp(coeffs, ba);
  out = bp(out, ba);
  extern __shared__ __attribute__((aligned(16))) u64 buf[];
  const unsigned n = 1u << log_n;
  const unsigned poly = blockIdx.x >> rate_bits, j = blockIdx.x & ((1u << rate_bits) - 1);
  const u64* c = coeffs + ((size_t)poly << log_n);
  const u64* cp = coset_pow + ((size_t)j << log_n);
  for (unsigned m = threadIdx.x; m < n; m += kBlock) buf[m] = cmul(c[m], cp[m]);
  __syncthreads();
  if (log_n) lds_dif(buf, log_n, 0, tw);
  u64* o = out + ((size_t)poly << (log_n + rate_bits)) + ((size_t)brev32(j, rate_bits) << log_n);
  for (unsigned q = threadIdx.x; q < n; q += kBlock) o[q] = buf[q];
}

// from_values for small polynomials (n <= 2^9) in ONE launch: every (poly, coset) workgroup redoes the cheap inverse
// transform of its polynomial in LDS (8x redundant, negligible at this size), the coset-0 workgroup also writes the
// coefficients, then the coset's forward transform goes out in leaf order as in k_coset_lde.
__global__ __launch_bounds__(kBlock) void k_ifft_coset_lde(const u64* __restrict__ vals, unsigned log_n, unsigned rate_bits,
                                                           const u64* __restrict__ coset_pow, const u64* __restrict__ tw,
                                                           const u64* __restrict__ tw_inv, u64 n_inv, u64* __restrict__ coeffs,
                                                           u64* __restrict__ out, BatchArg ba) {
  vals = bp(vals, ba);
  coeffs = bp(coeffs, ba);
  out = bp(out, ba);
  extern __shared__ __attribute__((aligned(16))) u64 buf[];  // 2 n words
  const unsigned n = 1u << log_n;
  u64* cf = buf + n;
  const unsigned poly = blockIdx.x >> rate_bits, j = blockIdx.x & ((1u << rate_bits) - 1);
  const u64* v = vals + ((size_t)poly << log_n);
  for (unsigned m = threadIdx.x; m < n; m += kBlock) buf[m] = gl::canon(v[m]);
  __syncthreads();
  if (log_n) lds_dif(buf, log_n, 0, tw_inv);
  for (unsigned q = threadIdx.x; q < n; q += kBlock) cf[brev32(q, log_n)] = cmul(buf[q], n_inv);
  __syncthreads();
  const u64* cp = coset_pow + ((size_t)j << log_n);
  for (unsigned m = threadIdx.x; m < n; m += kBlock) {
    const u64 c = cf[m];
    if (j == 0) coeffs[((size_t)poly << log_n) + m] = c;
    buf[m] = cmul(c, cp[m]);
  }
  __syncthreads();
  if (log_n) lds_dif(buf, log_n, 0, tw);
  u64* o = out + ((size_t)poly << (log_n + rate_bits)) + ((size_t)brev32(j, rate_bits) << log_n);
  for (unsigned q = threadIdx.x; q < n; q += kBlock) o[q] = buf[q];
}

// The same for n = 64 (the 64-row circuits, i.e. every proof of the batched prover) with ONE WAVEFRONT per polynomial: lane m holds
// point m, a DIF stage is a lane exchange (ds_bpermute) and a butterfly -- no LDS array, no barrier, and the inverse transform is done
// once per polynomial instead of once per coset.  (k_ifft_coset_lde starts 8 workgroups of 256 threads per polynomial, 64 of them
// busy, through 18 barriers each: 340 us for the 34 560 polynomials of a 256-proof pass, 0.4 TB/s.)  Same values, same order.
__global__ __launch_bounds__(kBlock) void k_ifft_coset_lde_wave64(const u64* __restrict__ vals, unsigned rate_bits, size_t n_polys,
                                                                  const u64* __restrict__ coset_pow, const u64* __restrict__ tw,
                                                                  const u64* __restrict__ tw_inv, u64 n_inv, u64* __restrict__ coeffs,
                                                                  u64* __restrict__ out, BatchArg ba) {
  vals = bp(vals, ba);
  coeffs = bp(coeffs, ba);
  out = bp(out, ba);
  const unsigned m = threadIdx.x & 63;
  const size_t poly = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (poly >= n_polys) return;  // wave-uniform
  auto xchg = [&](u64 v, unsigned half) -> u64 {  // lane m ^ half's value
    const int addr = (int)((m ^ half) << 2);
    const u32 lo = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)v), hi = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)(v >> 32));
    return ((u64)hi << 32) | lo;
  };
  u64 twf[6], twi[6];  // stage s: tw[(m mod half) << s], half = 32 >> s (used by the lanes of the upper half of each block)
#pragma unroll
  for (unsigned s = 0; s < 6; ++s) {
    const unsigned j = m & ((32u >> s) - 1);
    twf[s] = tw[j << s];
    twi[s] = tw_inv[j << s];
  }
  auto dif64 = [&](u64 v, const u64 (&t)[6]) -> u64 {  // lds_dif over the wave: natural -> bit-reversed
#pragma unroll
    for (unsigned s = 0; s < 6; ++s) {
      const unsigned half = 32u >> s;
      const u64 p = xchg(v, half);
      const bool upper = (m & half) != 0;
      const u64 sum = cadd(v, p), dif = cmul(csub(p, v), t[s]);  // upper lane: (x - y) tw with x = the partner's, y = its own
      v = upper ? dif : sum;
    }
    return v;
  };
  u64 v = dif64(gl::canon(vals[(poly << 6) + m]), twi);
  v = cmul(v, n_inv);
  {  // natural order: lane m takes lane brev6(m)'s value
    const int addr = (int)((__brev(m) >> 26) << 2);
    const u32 lo = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)v), hi = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)(v >> 32));
    v = ((u64)hi << 32) | lo;
  }
  coeffs[(poly << 6) + m] = v;
  const unsigned cosets = 1u << rate_bits;
#pragma unroll 1
  for (unsigned j = 0; j < cosets; ++j) {
    const u64 y = dif64(cmul(v, coset_pow[((size_t)j << 6) + m]), twf);
    out[(poly << (6 + rate_bits)) + ((size_t)brev32(j, rate_bits) << 6) + m] = y;
  }
}

// The scaling half of the x2^r coset LDE for transforms above 2^12 points: row = p * 2^r + brev_r(j) of `out` gets
// c_p[m] * (shift * w_N^j)^m, m = 0 .. n-1 (natural order), ready for the in-place DIF.  The power is built per thread from two
// exponentiations (base^(256 * (m / 256)) once per block, times base^(m % 256)): ~40 multiplications per element against the
// transform's ~100.
__global__ __launch_bounds__(kBlock) void k_coset_scale_rows(const u64* __restrict__ coeffs, unsigned log_n, unsigned rate_bits, u64 shift,
                                                             u64 w_big, u64* __restrict__ out) {
  // (the row rides in grid.x, which goes to 2^31 - 1: grid.y stops at 65 535 and 8192 polynomials x 8 cosets are more rows than that)
  const unsigned bpr = (unsigned)((((size_t)1 << log_n) + kBlock - 1) / kBlock);  // blocks per row
  const unsigned row = blockIdx.x / bpr, blk = blockIdx.x % bpr;
  const size_t m = (size_t)blk * kBlock + threadIdx.x;
  const unsigned poly = row >> rate_bits, c = row & ((1u << rate_bits) - 1), j = brev32(c, rate_bits);
  if (m >= ((size_t)1 << log_n)) return;
  const u64 base = gl::mul(shift, gl::pow(w_big, j));
  const u64 pw = gl::mul(gl::pow(base, (u64)blk * kBlock), gl::pow(base, threadIdx.x));
  out[((size_t)row << log_n) + m] = cmul(coeffs[((size_t)poly << log_n) + m], pw);
}

// ---------------------------------------------------------------- register-blocked 2^12 coset LDE
// The streaming version of k_coset_lde for n = 4096 (the d = 12 circuits of config 4).  Each of the 256 threads
// keeps 16 points in registers and the transform is three radix-16 passes (4096 = 16 x 16 x 16), so the data
// crosses LDS three times instead of twelve and there are 5 barriers instead of 13:
//   pass A  thread t: points t + 256 k; DIF-16 over k with the 16th roots of unity as compile-time constants
//           (zeta16 = w_4096^256 = 2^156), then slot r *= w_4096^(t * brev4(r));       -> LDS [r][t]
//   pass B  thread (r, u): points [r][u + 16 v]; DIF-16 over v, slot r' *= w_4096^(16 u brev4(r'))  -> LDS [r][r'][u]
//   pass C  thread (r, r'): 16 contiguous points; DIF-16 over u; results are DIF position 256 r + 16 r' + r'',
//           transposed through LDS so the global stores are contiguous.
// The result is the same exact field values as the radix-2 DIF of lds_dif (same DFT factorisation).  Arithmetic is
// "loose" u64 with single-fix add/sub whose (astronomically rare) second carry, and the multiply's rare borrow,
// go to a sticky wave flag; a flagged workgroup redoes its transform with the exact radix-2 code.
// Algorithmic traffic: 8 B read + 64 B written per coefficient (72 B); ~180 VALU instructions per coefficient.
namespace lde12 {

constexpr unsigned kRowA = 256 + 16;  // [r][t] rows padded by 16 words: lanes 16 apart land in disjoint banks
constexpr unsigned kRowC = 17;        // [r][r'][u] rows of 16 padded to 17: conflict-free ds_read_b64 at stride 136 B

GL_DEV u64 add_l(u64 a, u64 b, u64& sticky) {
  u32 lo, hi;
  u64 c1, c2;
  asm("v_add_co_u32_e64 %0, %2, %3, %5\n\tv_addc_co_u32_e64 %1, %2, %4, %6, %2"
      : "=&v"(lo), "=v"(hi), "=&s"(c1)
      : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32)));
  const u64 s = ((u64)hi << 32) | lo;
  const u64 r = poseidon_fast::mad_carry(poseidon_fast::eps_if(c1), 1u, s, c2);  // + EPS where it wrapped
  sticky |= c2;
  return r;
}
GL_DEV u64 sub_l(u64 a, u64 b, u64& sticky) {
  u32 lo, hi;
  u64 b1, b2;
  asm("v_sub_co_u32_e64 %0, %2, %3, %5\n\tv_subb_co_u32_e64 %1, %2, %4, %6, %2"
      : "=&v"(lo), "=v"(hi), "=&s"(b1)
      : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32)));
  const u64 d = ((u64)hi << 32) | lo;
  const u64 r = poseidon_fast::sub32_borrow(d, poseidon_fast::eps_if(b1), b2);  // - EPS where it wrapped
  sticky |= b2;
  return r;
}

// powers of zeta16 = 2^156 (plonky2's primitive 16th root of unity)
constexpr u64 kZeta16[8] = {0x1ull, 0xefffffff00000001ull, 0xfffffffeff000001ull, 0xffffffff00000ull,
                            0x1000000000000ull, 0x1000ull, 0xfffffeff00000101ull, 0xffffffef00000001ull};
constexpr unsigned brev4(unsigned r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }

// in-register DIF over 16 points (natural slots in, bit-reversed slots out), twiddles zeta16^(e << s)
GL_DEV void dif16(u64 (&x)[16], u64& sticky) {
  poseidon::static_for<0, 4>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int half = 8 >> s;
    poseidon::static_for<0, 8>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      constexpr int blk = b / half, e = b % half;
      constexpr int i0 = blk * 2 * half + e, i1 = i0 + half;
      const u64 a0 = x[i0], a1 = x[i1];
      x[i0] = add_l(a0, a1, sticky);
      const u64 d = sub_l(a0, a1, sticky);
      if constexpr ((e << s) == 0) x[i1] = d;
      else x[i1] = poseidon_fast::mul(d, kZeta16[e << s], sticky);
    });
  });
}

}  // namespace lde12

__global__ __launch_bounds__(kBlock) void k_coset_lde12(const u64* __restrict__ coeffs, unsigned rate_bits,
                                                        const u64* __restrict__ coset_pow, const u64* __restrict__ tw_full,
                                                        const u64* __restrict__ tw_half, u64* __restrict__ out, BatchArg ba) {
  coeffs = bp(coeffs, ba);
  out = bp(out, ba);
  using namespace lde12;
  __shared__ __attribute__((aligned(16))) u64 buf[16 * kRowA];  // 34 KB; also holds the 256 x 17 layout (4352 words)
  const unsigned t = threadIdx.x;
  const unsigned poly = blockIdx.x >> rate_bits, j = blockIdx.x & ((1u << rate_bits) - 1);
  const u64* c = coeffs + ((size_t)poly << 12);
  const u64* cp = coset_pow + ((size_t)j << 12);
  u64* o = out + ((size_t)poly << (12 + rate_bits)) + ((size_t)brev32(j, rate_bits) << 12);
  u64 sticky = 0;
  u64 x[16];
  // ---- pass A
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = poseidon_fast::mul(c[t + 256 * k], cp[t + 256 * k], sticky);
  dif16(x, sticky);
  poseidon::static_for<1, 16>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    x[r] = poseidon_fast::mul(x[r], tw_full[t * brev4(r)], sticky);
  });
#pragma unroll
  for (int r = 0; r < 16; ++r) buf[r * kRowA + t] = x[r];
  __syncthreads();
  // ---- pass B: thread (r, u)
  const unsigned rr = t >> 4, u = t & 15;
#pragma unroll
  for (int v = 0; v < 16; ++v) x[v] = buf[rr * kRowA + u + 16 * v];
  __syncthreads();
  dif16(x, sticky);
  poseidon::static_for<1, 16>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    x[r] = poseidon_fast::mul(x[r], tw_full[16 * u * brev4(r)], sticky);
  });
#pragma unroll
  for (int r = 0; r < 16; ++r) buf[((rr * 16 + r) * kRowC) + u] = x[r];  // [r][r'][u], rows of 17
  __syncthreads();
  // ---- pass C: thread (r, r') = t
#pragma unroll
  for (int v = 0; v < 16; ++v) x[v] = buf[t * kRowC + v];
  __syncthreads();
  dif16(x, sticky);
  // results of thread t are DIF positions 16 t + r'': transpose through LDS for contiguous stores
#pragma unroll
  for (int r = 0; r < 16; ++r) buf[t * kRowC + r] = gl::canon(x[r]);
  __syncthreads();
  if (__builtin_expect(__syncthreads_or(sticky != 0), 0)) {  // rare: the whole workgroup redoes it exactly
    for (unsigned m = t; m < 4096; m += kBlock) buf[m] = cmul(c[m], cp[m]);
    __syncthreads();
    lds_dif(buf, 12, 0, tw_half);
    for (unsigned q = t; q < 4096; q += kBlock) o[q] = buf[q];
    return;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned q = i * 256 + t;
    o[q] = buf[(q >> 4) * kRowC + (q & 15)];
  }
}

// Round-4 form of the same kernel: identical data flow and LDS layouts, but
//   * the radix-16 butterflies use no multiplier: plonky2's w_16 is 2^156 = -2^60, so all their twiddles are +-2^(12 e) and
//     cost a shift + fold (5-7 issue slots instead of 12; ntt_arith.hip.h);
//   * every field operation is one asm block that keeps its carries to itself (no SGPR spills: the round-3 kernel spent a
//     quarter of its VALU instructions on v_readlane / v_writelane of spilled carry masks);
//   * the pass twiddles come from tables laid out in access order -- TA[r][t] = w^(t brev4(r)), TB[r][u] = w^(16 u brev4(r)) -- so a
//     wave's load is 512 contiguous bytes (round 3 gathered w^(t brev4(r)) from the natural-order table: up to 60 cache lines per
//     wave-load, 30 such loads per thread, which is what the kernel actually waited for);
//   * the eight cosets of a polynomial go to the SAME XCD (blocks b, b + 8, ... share an L2 under the round-robin dispatch), so
//     its 32 KB of coefficients leave HBM / the fabric once instead of once per coset.  Placement is for speed only;
//   * global (not flat) loads and stores, 16 bytes per lane on the way out.
typedef const u64 __attribute__((address_space(1)))* gcptr;
typedef u64 __attribute__((address_space(1)))* gptr;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u64x2 __attribute__((address_space(1)))* gptr2;
GL_DEV gcptr as_global(const u64* p) { return (gcptr)(unsigned long long)p; }
GL_DEV gptr as_global(u64* p) { return (gptr)(unsigned long long)p; }
// a wave-uniform base (an SGPR pair) plus a 32-bit per-lane byte offset: one VGPR of address per access instead of two
GL_DEV u64 ld_at(gcptr base, u32 byte_off) { return *(gcptr)((const char __attribute__((address_space(1)))*)base + byte_off); }
GL_DEV void st_at(gptr base, u32 byte_off, u64 v) { *(gptr)((char __attribute__((address_space(1)))*)base + byte_off) = v; }

// ta[(r - 1) * 256 + t] = w_4096^(t brev4(r)), tb[(r - 1) * 16 + u] = w_4096^(16 u brev4(r)), r = 1..15
__global__ __launch_bounds__(kBlock) void k_lde12_tables(const u64* __restrict__ tw_full, u64* __restrict__ ta, u64* __restrict__ tb) {
  const unsigned t = threadIdx.x, r = blockIdx.x + 1;
  ta[(r - 1) * 256 + t] = tw_full[(t * lde12::brev4(r)) & 4095];
  if (t < 16) tb[(r - 1) * 16 + t] = tw_full[(16 * t * lde12::brev4(r)) & 4095];
}

__global__ __launch_bounds__(kBlock, 4) void k_coset_lde12_v2(const u64* __restrict__ coeffs_, unsigned rate_bits, unsigned n_polys,
                                                              const u64* __restrict__ coset_pow, const u64* __restrict__ ta_,
                                                              const u64* __restrict__ tb_, const u64* __restrict__ tw_half,
                                                              u64* __restrict__ out_, unsigned force, BatchArg ba) {
  using namespace lde12;
  __shared__ __attribute__((aligned(16))) u64 buf[16 * kRowA];
  const unsigned t = threadIdx.x;
  // block -> (poly, coset): groups of 8 polynomials x 2^rate_bits cosets; inside a group block g = j * 8 + (poly % 8), so the
  // cosets of one polynomial are 8 blocks apart.  The last (partial) group keeps the plain order.
  unsigned poly, j;
  {
    const unsigned cosets = 1u << rate_bits, group = 8u << rate_bits;
    const unsigned g0 = (blockIdx.x / group) * 8, r = blockIdx.x % group;
    if (g0 + 8 <= n_polys) {
      poly = g0 + (r & 7);
      j = r >> 3;
    } else {
      poly = g0 + r / cosets;
      j = r % cosets;
    }
  }
  const gcptr c = as_global(bp(coeffs_, ba)) + ((size_t)poly << 12);
  const gcptr cp = as_global(coset_pow) + ((size_t)j << 12);
  const gcptr ta = as_global(ta_), tb = as_global(tb_);
  const gptr o = as_global(bp(out_, ba)) + ((size_t)poly << (12 + rate_bits)) + ((size_t)brev32(j, rate_bits) << 12);
  u64 sticky = 0;
  u64 x[16];
  // ---- pass A.  All loads of a pass are issued before its arithmetic (the field operations are ordered asm blocks: left to
  // itself the scheduler sinks each load to its use and the wave pays one memory round trip per coefficient).
  u64 tw[16];
  {
    u64 pw[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      x[k] = c[t + 256 * k];
      pw[k] = cp[t + 256 * k];
    }
#pragma unroll
    for (int r = 1; r < 16; ++r) tw[r] = ta[(r - 1) * 256 + t];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = ntt::mul(x[k], pw[k], sticky);
  }
  ntt::dif16<156>(x, sticky);
  poseidon::static_for<1, 16>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    x[r] = ntt::mul(x[r], tw[r], sticky);
  });
#pragma unroll
  for (int r = 0; r < 16; ++r) buf[r * kRowA + t] = x[r];
  __syncthreads();
  // ---- pass B: thread (r, u)
  const unsigned rr = t >> 4, u = t & 15;
#pragma unroll
  for (int v = 0; v < 16; ++v) x[v] = buf[rr * kRowA + u + 16 * v];
#pragma unroll
  for (int r = 1; r < 16; ++r) tw[r] = tb[(r - 1) * 16 + u];
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  ntt::dif16<156>(x, sticky);
  poseidon::static_for<1, 16>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    x[r] = ntt::mul(x[r], tw[r], sticky);
  });
#pragma unroll
  for (int r = 0; r < 16; ++r) buf[((rr * 16 + r) * kRowC) + u] = x[r];  // [r][r'][u], rows of 17
  __syncthreads();
  // ---- pass C: thread (r, r') = t
#pragma unroll
  for (int v = 0; v < 16; ++v) x[v] = buf[t * kRowC + v];
  __syncthreads();
  ntt::dif16<156>(x, sticky);
  // results of thread t are DIF positions 16 t + r'': transpose through LDS for contiguous stores
#pragma unroll
  for (int r = 0; r < 16; ++r) buf[t * kRowC + r] = gl::canon(x[r]);
  __syncthreads();
  if (__builtin_expect(__syncthreads_or(sticky != 0) || force, 0)) {  // rare (or forced by the tests): the workgroup redoes it exactly
    for (unsigned m = t; m < 4096; m += kBlock) buf[m] = cmul(c[m], cp[m]);
    __syncthreads();
    lds_dif(buf, 12, 0, tw_half);
    for (unsigned q = t; q < 4096; q += kBlock) o[q] = buf[q];
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned q = i * 512 + 2 * t;  // q even: q and q + 1 sit in the same padded row of 16
    const u64* src = &buf[(q >> 4) * kRowC + (q & 15)];
    u64x2 v;
    v.x = src[0];
    v.y = src[1];
    *(gptr2)(o + q) = v;
  }
}

// ---------------------------------------------------------------- 2^13 .. 2^20-point transforms: two passes over HBM (four-step)
// (described for n = 2^20 = 1024 x 1024; smaller n = 1024 x 2^m keep the column pass -- with rows of 2^m points -- and take the
//  general row pass k_ntt_rows_small)   i = i1 1024 + i2, k = k1 + 1024 k2:
//   pass 1  for every column i2:  A[k1][i2] = w_n^(i2 k1) sum_i1 x[i1][i2] w_1024^(i1 k1)      (1024-point transforms down the columns)
//   pass 2  for every row k1:     X[k1 + 1024 k2] = sum_i2 A[k1][i2] w_1024^(i2 k2)            (1024-point transforms along the rows)
// LDS holds 2^14 points at most, so two passes is the minimum for this size: 32 B of HBM traffic per point against the 16 B of a
// transform that fits; each pass kernel below moves its 16 B per point exactly once.
// One kernel does both passes.  A workgroup of 1024 threads owns a tile of 1024 points x 16 transforms (128 KB: one per CU), every
// thread 16 points, and the 1024-point DIF is radix 16 x 16 x 4 in registers with two exchanges through LDS:
//   (numbers for Q = 16 transforms per workgroup; Tile<Q> below)
//   step A  thread (p_lo, q): points p = 64 k + p_lo, DIF-16 over k; slot a *= w_1024^(p_lo brev4(a))           -> LDS
//   step B  thread (d, a, q): points p = 64 a + 4 b + d, DIF-16 over b; slot b' *= w_64^(d brev4(b')), d = t >> 8 wave-uniform
//                                                                                                                -> LDS, in place
//   step C  thread (beta, a, q): four DIF-4 over d for b' = 4 beta .. 4 beta + 3; slot (a, b', d') is frequency
//           k = brev2(d') 256 + brev4(b') 16 + brev4(a); pass 1 multiplies by the four-step twiddle; out[k][q].
// What differs between the passes is only which index is contiguous in memory on the way IN: pass 1 reads 16 neighbouring columns
// (128-byte granules, one row apart), pass 2 reads 16 whole rows (8 KB runs).  Both write out[k][q0 + q]: 128-byte granules, so
// pass 2 lands the result in natural order (X[k1 + 1024 k2] with k1 = q0 + q) with no transpose pass.  The two arrays must differ
// for pass 2 (it reads rows and writes columns); pass 1 may run in place.
// LDS layout for both exchanges: word(a, p_lo, q) = a * 1104 + p_lo * 17 + q (p_lo = 4 b + d): conflict-free for the writers of
// step A in either load mapping (lanes along q or along p_lo: stride 17) and for the (a, q)-lane readers / writers of steps B
// and C (1104 = 16 mod 32: the two values of a in a half-wave take the two halves of the banks).
// Algorithmic traffic per launch: 16 B per point (+ 8 B per point of twiddles in pass 1, from a table all transforms share).
namespace ntt20 {

// Q transforms per workgroup (64 Q threads): LDS word(a, p_lo, q) = a * rowA + p_lo * (Q + 1) + q, rowA = 64 (Q + 1) + Q
//   Q = 16: 128-byte granules, 141 KB of LDS, one workgroup of 1024 threads per CU (its memory phases are covered by nothing; sending
//           the exchanges through LDS one 32-bit half at a time halves the array, but a second workgroup of 1024 threads also needs
//           the kernel in 64 VGPRs and it has 128 + 34 spilled: built in round 4, 102 spills and 412 B of scratch, dropped);
//   Q = 8:   64-byte granules,  75 KB of LDS, two workgroups of 512 threads per CU (one computes while the other loads / stores).
// The row pass's step-A writers have lanes along p_lo (odd stride Q + 1); the column pass's have lanes along q, then p_lo: stride Q puts a
// half-wave's 32 words on 32 different double-banks (with Q + 1 it measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.20; the launch
// time did not move with it: LDS is 5.6 M of the pass's 227 M wave instructions).  Either
// way rowA = Q mod 32 keeps the (a, q)-lane accesses of steps B and C conflict-free.
template <unsigned Q, bool ROW_IN>
struct Tile {
  static constexpr unsigned kThreads = 64 * Q, kStride = ROW_IN ? Q + 1 : Q, kRowA = 64 * kStride + Q, kLdsWords = 16 * kRowA;
  static_assert(kLdsWords >= Q * 1024, "the redo path lays the tile out flat in the same array");
};

constexpr unsigned brev4(unsigned r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3); }
// the compile-time part of step C's output row: slot (bb, dd) of a thread
constexpr unsigned kc(int bb, int dd) { return (((dd & 1) << 1) | (dd >> 1)) * 256u + (((bb & 1) << 1) | (bb >> 1)) * 64u; }

// x *= 2^E for an exponent modulo 192 (2^96 = -1: the upper half costs a negation p - y, y > p flags)
template <int E>
GL_DEV u64 mul_pow2_mod192(u64 x, u64& sticky) {
  constexpr int e = ((E % 192) + 192) % 192;
  if constexpr (e < 96) {
    return ntt::mul_pow2<e>(x, sticky);
  } else {
    const u64 y = ntt::mul_pow2<e - 96>(x, sticky);
    u32 lo, hi;
    u64 w;
    asm("v_sub_co_u32_e64 %[lo], %[w], 1, %[y0]\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32_e64 %[hi], %[w], -1, %[y1], %[w]\n\t"
        "s_or_b64 %[st], %[st], %[w]"
        : [lo] "=&v"(lo), [hi] "=&v"(hi), [w] "=&s"(w), [st] "+s"(sticky)
        : [y0] "v"((u32)y), [y1] "v"((u32)(y >> 32))
      : "scc");
    return ((u64)hi << 32) | lo;
  }
}

}  // namespace ntt20

// ta1[(a - 1) * 64 + p_lo] = w_1024^(p_lo brev4(a)) (a = 1..15), then ta1[960 + d * 16 + b] = w_64^(d brev4(b))
__global__ __launch_bounds__(kBlock) void k_ntt1024_tables(u64 w1024, u64* __restrict__ ta1) {
  const unsigned i = blockIdx.x * kBlock + threadIdx.x;
  if (i < 15 * 64) {
    const unsigned a = (i >> 6) + 1, p_lo = i & 63;
    ta1[i] = gl::canon(gl::pow(w1024, (u64)p_lo * ntt20::brev4(a)));
  } else if (i < 16 * 64) {
    const unsigned d = (i >> 4) & 3, b = i & 15;
    ta1[i] = gl::canon(gl::pow(w1024, (u64)16 * d * ntt20::brev4(b)));
  }
}
// the four-step twiddles of a 2^(10 + log_c)-point transform: t4[k1 * 2^log_c + i2] = scale * w_n^(i2 k1)
__global__ __launch_bounds__(kBlock) void k_fourstep_twiddles(u64 wn, u64 scale, unsigned log_c, u64* __restrict__ t4) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= ((size_t)1 << (10 + log_c))) return;
  const u64 k = i >> log_c, i2 = i & (((size_t)1 << log_c) - 1);
  t4[i] = gl::canon(gl::mul(scale, gl::pow(wn, i2 * k)));
}

// DIR 0 forward (w), 1 inverse (w^-1; the 1/n rides in t4).  ROW_IN: the 16 transforms are 16 contiguous rows of `in` (pass 2);
// otherwise 16 neighbouring columns (pass 1).  TW: multiply the outputs by t4[k][q0 + q] (pass 1).
// grid: x = polynomial, y = tile (q0 = 16 y).  in / out: [n_polys][1024][1024].
template <int DIR, bool ROW_IN, bool TW, unsigned Q>
__global__ __launch_bounds__(64 * Q) void k_ntt20_pass(const u64* __restrict__ in_, u64* __restrict__ out_, const u64* __restrict__ ta1_,
                                                     const u64* __restrict__ t4_, const u64* __restrict__ tw_half, unsigned log_c, unsigned force) {
  using namespace ntt20;
  using T = Tile<Q, ROW_IN>;  // (ROW_IN: rows of 1024 points, i.e. log_c == 10)
  constexpr unsigned kRowA = T::kRowA, kStride = T::kStride, kTile = Q, kLogQ = Q == 16 ? 4 : 3;
  constexpr int Z16 = DIR ? 36 : 156, Z4 = DIR ? 144 : 48;
  __shared__ __attribute__((aligned(16))) u64 buf[T::kLdsWords];
  const unsigned t = threadIdx.x;
  // workgroup -> (polynomial, tile).  The column pass with 8-column tiles reads 64-byte halves of 128-byte lines: the two tiles that
  // share the lines run back to back on the SAME XCD (workgroups go to the XCDs round-robin in linear order), so the second half is
  // an L2 hit instead of a second fetch; consecutive pairs of an XCD are the same tiles of other polynomials (they share t4's rows).
  unsigned poly = blockIdx.x, tile = blockIdx.y;
  if constexpr (!ROW_IN && Q == 8) {
    if (((gridDim.x * gridDim.y) & 15) == 0 && (gridDim.y & 1) == 0) {
      const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, seq = lin >> 3, idx = (seq >> 1) * 8 + xcd;
      poly = idx % gridDim.x;
      tile = 2 * (idx / gridDim.x) + (seq & 1);
    }
  }
  const unsigned q0 = tile * kTile;
  // log_c: log2 of the row pitch of in / out (the number of columns): 10 for a 2^20-point transform; the column pass of a
  // 2^(10 + log_c)-point transform runs with 3 <= log_c <= 10 (its rows are 2^log_c long, its columns 1024)
  const gcptr in = as_global(in_) + ((size_t)poly << (10 + log_c));
  const gptr out = as_global(out_) + ((size_t)poly << (10 + log_c));
  const gcptr ta1 = as_global(ta1_), t4 = as_global(t4_);
  u64 sticky = 0;
  u64 x[16];
  // ---- load + step A
  const unsigned qa = ROW_IN ? (t >> 6) : (t & (Q - 1)), p_lo = ROW_IN ? (t & 63) : (t >> kLogQ);
  // every address of the tile is (uniform base) + (one 32-bit lane offset): the tile spans < 2^23 bytes, and the 16 strided
  // accesses of a thread differ by uniform amounts.  (Sixteen 64-bit lane addresses, kept for the redo path, were what spilled.)
  const u32 src_off = (ROW_IN ? ((q0 + qa) << 10) + p_lo : (p_lo << log_c) + q0 + qa) * 8u;
  {
    u64 tw[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = ld_at(ROW_IN ? in + 64 * k : in + ((size_t)(64 * k) << log_c), src_off);
#pragma unroll
    for (int a = 1; a < 16; ++a) tw[a] = ld_at(ta1 + (a - 1) * 64, p_lo * 8u);
    __builtin_amdgcn_sched_barrier(0);
    ntt::dif16<Z16>(x, sticky);
    poseidon::static_for<1, 16>([&](auto rc) {
      constexpr int a = decltype(rc)::value;
      x[a] = ntt::mul(x[a], tw[a], sticky);
    });
  }
#pragma unroll
  for (int a = 0; a < 16; ++a) buf[a * kRowA + p_lo * kStride + qa] = x[a];
  __syncthreads();
  // ---- step B: thread (d, a, q), d wave-uniform
  const unsigned d = __builtin_amdgcn_readfirstlane(t >> (kLogQ + 4)), ab = (t >> kLogQ) & 15, q = t & (Q - 1);
  u64* const plane = buf + ab * kRowA + q;
#pragma unroll
  for (int b = 0; b < 16; ++b) x[b] = plane[(4 * b + d) * kStride];
  // slot b' *= w_64^(d brev4(b')), d wave-uniform: w_64 = 2^39 (2^-39 = 2^153 for the inverse), so the factors are shifts picked by a
  // scalar branch -- 0 slots for d = 0, 5..8 per point otherwise against 12 for a table multiply.  (Round 4 had this and saw wrong
  // values: its field-arithmetic asm ran s_or_b64 without an "scc" clobber, and the compiler's s_cmp / s_cselect pair straddled it.)
  ntt::dif16<Z16>(x, sticky);
  if (d != 0) {
    auto arm = [&](auto dc) {
      asm volatile("" ::: "memory");  // (an arm is taken or skipped as a whole, never computed speculatively and selected)
      poseidon::static_for<1, 16>([&](auto rc) {
        constexpr int b = decltype(rc)::value, e = (DIR ? -39 : 39) * decltype(dc)::value * (int)brev4(b);
        x[b] = mul_pow2_mod192<e>(x[b], sticky);
      });
    };
    if (d == 1) arm(std::integral_constant<int, 1>{});
    else if (d == 2) arm(std::integral_constant<int, 2>{});
    else arm(std::integral_constant<int, 3>{});
  }
#pragma unroll
  for (int b = 0; b < 16; ++b) plane[(4 * b + d) * kStride] = x[b];  // in place: the words this thread read
  __syncthreads();
  // ---- step C: thread (beta, a, q): b' = 4 beta + bb
  // output row k = kc(bb, dd) + kl: kc = brev2(dd) * 256 + brev2(bb) * 64 known at compile time, kl = brev2(beta) * 16 + brev4(ab)
  // the lane's part (brev4(4 beta + bb) = brev2(bb) * 4 + brev2(beta))
  const unsigned beta = d;
  const unsigned kl = (((beta & 1) << 1) | (beta >> 1)) * 16 + (__builtin_bitreverse32(ab) >> 28);
  const u32 dst_off = ((kl << log_c) + q0 + q) * 8u;
  u64 tw[16];
  if constexpr (TW) {
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) tw[4 * bb + dd] = ld_at(t4 + ((size_t)kc(bb, dd) << log_c), dst_off);
  }
#pragma unroll
  for (int bb = 0; bb < 4; ++bb)
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) x[4 * bb + dd] = plane[(4 * (4 * beta + bb) + dd) * kStride];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int bb = 0; bb < 4; ++bb) {
    u64 y[4] = {x[4 * bb], x[4 * bb + 1], x[4 * bb + 2], x[4 * bb + 3]};
    ntt::dif4<Z4>(y, sticky);
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) x[4 * bb + dd] = TW ? ntt::mul(y[dd], tw[4 * bb + dd], sticky) : y[dd];
  }
  if (__builtin_expect(__syncthreads_or(sticky != 0) || force, 0)) {
    // rare (or forced by the tests): the workgroup redoes its tile with the exact radix-2 code.  flat[q * 1024 + p]; the in-place
    // DIF leaves frequency k at position brev10(k).
    u64* const flat = buf;
    __syncthreads();
    // (offsets made opaque here: otherwise the 16 + 16 lane addresses of the fast path are kept alive -- spilled -- for this one)
    u32 src_off2 = src_off, dst_off2 = dst_off;
    asm volatile("" : "+v"(src_off2), "+v"(dst_off2));
#pragma unroll
    for (int k = 0; k < 16; ++k)
      flat[qa * 1024 + 64 * k + p_lo] = gl::canon(ld_at(ROW_IN ? in + 64 * k : in + ((size_t)(64 * k) << log_c), src_off2));
    __syncthreads();
    for (unsigned s = 0; s < 10; ++s) {
      const unsigned half = 512u >> s;
      for (unsigned i = t; i < Q * 512; i += T::kThreads) {
        const unsigned col = i >> 9, bf = i & 511, blk = bf / half, j = bf & (half - 1);
        const unsigned i0 = col * 1024 + blk * 2 * half + j, i1 = i0 + half;
        const u64 u = flat[i0], v = flat[i1];
        flat[i0] = cadd(u, v);
        flat[i1] = cmul(csub(u, v), tw_half[(size_t)j << s]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {
        const unsigned k = kc(bb, dd) + kl;
        u64 v = flat[q * 1024 + brev32(k, 10)];
        if constexpr (TW) v = cmul(v, ld_at(t4 + ((size_t)kc(bb, dd) << log_c), dst_off2));
        st_at(out + ((size_t)kc(bb, dd) << log_c), dst_off2, v);
      }
    return;
  }
#pragma unroll
  for (int bb = 0; bb < 4; ++bb)
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
      // pass 1 writes the scratch array pass 2 reads: any u64 representative will do there (the arithmetic is the same loose one)
      st_at(out + ((size_t)kc(bb, dd) << log_c), dst_off, TW ? x[4 * bb + dd] : gl::canon(x[4 * bb + dd]));
    }
}

// The row pass for rows shorter than 1024 points (transforms of 2^13 .. 2^19 points): a workgroup takes 16 whole rows of 2^m points
// (m = log_c, 3..9), transforms them in LDS with the exact radix-2 butterflies and writes out[k2 * 1024 + r0 + r] -- the same
// 128-byte granules as the 1024-point row pass, natural order.  General rather than fast: the column pass above carries ten of the
// transform's stages, this one the remaining m.
template <int DIR>
__global__ __launch_bounds__(kBlock) void k_ntt_rows_small(const u64* __restrict__ in_, u64* __restrict__ out_, const u64* __restrict__ tw_half,
                                                           unsigned m) {
  extern __shared__ __attribute__((aligned(16))) u64 rows[];  // [16][2^m]
  const unsigned t = threadIdx.x, r0 = blockIdx.y * 16, len = 1u << m;
  const gcptr in = as_global(in_) + ((size_t)blockIdx.x << (10 + m)) + ((size_t)r0 << m);
  const gptr out = as_global(out_) + ((size_t)blockIdx.x << (10 + m));
  for (unsigned i = t; i < 16 * len; i += kBlock) rows[i] = gl::canon(in[i]);
  __syncthreads();
  for (unsigned s = 0; s < m; ++s) {
    const unsigned half = (len >> 1) >> s;
    for (unsigned i = t; i < 8 * len; i += kBlock) {
      const unsigned row = i >> (m - 1), bf = i & ((len >> 1) - 1), blk = bf / half, j = bf & (half - 1);
      const unsigned i0 = row * len + blk * 2 * half + j, i1 = i0 + half;
      const u64 u = rows[i0], v = rows[i1];
      rows[i0] = cadd(u, v);
      rows[i1] = cmul(csub(u, v), tw_half[(size_t)j << s]);
    }
    __syncthreads();
  }
  for (unsigned i = t; i < 16 * len; i += kBlock) {
    const unsigned r = i & 15, k2 = i >> 4;
    out[((size_t)k2 << 10) + r0 + r] = rows[r * len + brev32(k2, m)];  // the in-place DIF leaves frequency k2 at position brev(k2)
  }
}

// ---------------------------------------------------------------- leaves
// Poly-major [w][n_pts] -> leaf-major [n_pts][w] through a 32x32 LDS tile (+1 pad: conflict-free column reads).
__global__ __launch_bounds__(kBlock) void k_transpose(const u64* __restrict__ in, u64* __restrict__ out, size_t w,
                                                      size_t n_pts, BatchArg ba) {
  in = bp(in, ba);
  out = bp(out, ba);
  __shared__ u64 tile[32][33];
  const size_t p0 = (size_t)blockIdx.y * 32, i0 = (size_t)blockIdx.x * 32;
  const unsigned tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (unsigned r = ty; r < 32; r += 8)
    if (p0 + r < w && i0 + tx < n_pts) tile[r][tx] = in[(p0 + r) * n_pts + i0 + tx];
  __syncthreads();
  for (unsigned r = ty; r < 32; r += 8)
    if (i0 + r < n_pts && p0 + tx < w) out[(i0 + r) * w + p0 + tx] = tile[tx][r];
}

// hash_or_noop of leaf i = column i of the poly-major matrix: lane i reads in[p * n_pts + i] (coalesced).
template <int M, int PR>
__global__ __launch_bounds__(kBlock, 4) void k_hash_columns(const u64* __restrict__ in, size_t w, size_t n_pts,
                                                         u64* __restrict__ digests, BatchArg ba, p2mt::PermCtx ctx) {
  in = bp(in, ba);
  digests = bp(digests, ba);
  poseidon_fast::MfmaCtx mc;  // PR == 5: dense MDS layers on the matrix pipe; every lane stays in the sponge (an MFMA ignores EXEC),
  if constexpr (PR == 5) poseidon_fast::mfma32_ctx_init(mc);  // lanes past the end redo the last column and store nothing
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = i < n_pts;
  if constexpr (PR == 5) i = live ? i : n_pts - 1;
  else if (!live) return;
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) s[k] = 0;
  if (w <= 4) {  // no permutation: zero-padded copy
    for (size_t p = 0; p < w; ++p) s[p] = in[p * n_pts + i];
  } else {
    // A flagged wave redoes THAT permutation with the exact reference form, from the copy of the state each lane parks in LDS in front
    // of every permutation.  (With the exact folds of PR == 5 a flag is a 2^-32 event per operation; with flag-form folds it was
    // ~0.07 % of wave-permutations.)  (Until round 4 the flag was collected over the whole sponge and a
    // flagged wave -- one in ~45 for a 135-column leaf -- redid all 17 permutations at the end, alone on its SIMD for ~1 ms after
    // every other wave of the launch had finished: that tail was 15 % of the launch.)
    __shared__ u64 stash[M == 2 ? 12 : 1][kBlock];
#pragma unroll 1
    for (size_t off = 0; off < w; off += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (off + k < w) s[k] = in[(off + k) * n_pts + i];
      if constexpr (M == 2) {
#pragma unroll
        for (int k = 0; k < 12; ++k) stash[k][threadIdx.x] = s[k];
        const u64 sticky = poseidon_fast::permute<false, 12, false, false, false, (PR == 5 ? 3 : 0), (PR == 5 ? 2 : 1), 0, false, -1, PR == 5>(s, ctx.rc, &mc) | ctx.force_fallback;
        if (__builtin_expect(sticky != 0, 0)) {  // wave-uniform
#pragma unroll
          for (int k = 0; k < 12; ++k) s[k] = stash[k][threadIdx.x];
          poseidon::permute<poseidon::MDS_MAD64, poseidon::PARTIAL_NAIVE>(s);
        }
      } else {
        poseidon::permute<M, PR>(s);
      }
    }
  }
  if (!live) return;
  ulonglong2* q = reinterpret_cast<ulonglong2*>(digests + 4 * i);
  q[0] = make_ulonglong2(gl::canon(s[0]), gl::canon(s[1]));
  q[1] = make_ulonglong2(gl::canon(s[2]), gl::canon(s[3]));
}

// The same leaf sponge on four lanes per column (latency path for <= 2^16 leaves): lane q of a quad owns state words
// 3q..3q+2, so of each 8-word chunk it loads the (up to 3) words it owns.
__global__ __launch_bounds__(kBlock) void k_hash_columns_quad(const u64* __restrict__ in, size_t w, size_t n_pts,
                                                              u64* __restrict__ digests, BatchArg ba, p2mt::PermCtx ctx) {
  in = bp(in, ba);
  digests = bp(digests, ba);
  const size_t col = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 2;
  if (col >= n_pts) return;  // quad-uniform
  poseidon_quad::Lane ln;
  poseidon_quad::lane_init(ln, ctx.rc);
  u64 x[3] = {0, 0, 0};
#pragma unroll 1
  for (size_t off = 0; off < w; off += 8) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const u32 wd = 3 * ln.q + i;
      if (wd < 8 && off + wd < w) x[i] = in[(off + wd) * n_pts + col];
    }
    poseidon_quad::permute(x, ln);
  }
  u64* out = digests + 4 * col;
  if (ln.q == 0) {
    out[0] = gl::canon(x[0]);
    out[1] = gl::canon(x[1]);
    out[2] = gl::canon(x[2]);
  } else if (ln.q == 1) {
    out[3] = gl::canon(x[0]);
  }
}

// The same leaf sponge with one wavefront per column (latency path for <= 2^12 leaves -- the 64-row circuits): lane k < 8
// loads word k of each 8-word chunk, the chunk's permutation runs on 12 lanes (permute_wave, ~11 us instead of ~18 us
// on a quad), so a 135-wide leaf costs 17 x 11 us.
__global__ __launch_bounds__(kBlock) void k_hash_columns_wave(const u64* __restrict__ in, size_t w, size_t n_pts,
                                                              u64* __restrict__ digests, u64* __restrict__ leaves,
                                                              BatchArg ba, p2mt::PermCtx ctx) {
  in = bp(in, ba);
  digests = bp(digests, ba);
  leaves = bp(leaves, ba);
  __shared__ u64 rc_lds[p2mt_dev::kWaveRcWords];
  ctx = p2mt_dev::stage_round_constants(rc_lds, ctx);
  const size_t col = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (col >= n_pts) return;  // wave-uniform
  const unsigned lane = threadIdx.x & 63;
  u64 x = 0;
  u64 nx = (lane < 8 && lane < w) ? in[lane * n_pts + col] : 0;  // the next chunk's word is fetched under the current permutation
#pragma unroll 1
  for (size_t off = 0; off < w; off += 8) {
    if (lane < 8 && off + lane < w) {
      x = nx;
      if (leaves) leaves[col * w + off + lane] = x;  // the leaf-major copy the FRI queries read (saves the transpose launch)
    }
    if (lane < 8 && off + 8 + lane < w) nx = in[(off + 8 + lane) * n_pts + col];
    x = p2mt_dev::permute_wave(x, ctx);
  }
  if (lane < 4) digests[4 * col + lane] = gl::canon(x);
}

// The last <= 5 levels below the cap in ONE launch: a workgroup of 16 wavefronts owns one cap entry's subtree (<= 32 input
// nodes), a wavefront hashes one node per level, levels hand over through LDS.  Same latency as one launch per level (a
// level is one wave-permutation either way) but a fifth of the launches -- what limits concurrent provers is the number of
// dispatch packets, not the CUs.  in: 2^cap subtrees of 2^log_sub nodes each; next: where the level after `in` goes in the
// level-major digest array (null = the caller keeps no digests); cap: the 2^cap results.
__global__ __launch_bounds__(1024) void k_merkle_top(const u64* __restrict__ in, unsigned log_sub, u64* __restrict__ next,
                                                     size_t n_in, u64* __restrict__ cap, BatchArg ba, p2mt::PermCtx ctx) {
  in = bp(in, ba);
  next = bp(next, ba);
  cap = bp(cap, ba);
  __shared__ u64 rc_lds[p2mt_dev::kWaveRcWords];
  __shared__ u64 buf[2][16][4];
  ctx = p2mt_dev::stage_round_constants(rc_lds, ctx);
  const unsigned c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u64* src = in + ((size_t)c << log_sub) * 4;
  size_t lvl_off = 0;  // offset of level t in `next`, in nodes
  // four nodes per wavefront (one per 16-lane row, permute_wave4): the 16 nodes of the first level take four wavefronts -- one per
  // SIMD of the CU -- instead of sixteen sharing them
  const unsigned rl = lane & 15, row = lane >> 4;
  for (unsigned t = 1; t <= log_sub; ++t) {
    const unsigned cnt = 1u << (log_sub - t);
    if (4 * wave < cnt) {  // wave-uniform
      const unsigned node = 4 * wave + row;
      u64 x = 0;
      if (rl < 8 && node < cnt) {
        const unsigned child = 2 * node + (rl >> 2), word = rl & 3;
        x = t == 1 ? src[4 * child + word] : buf[t & 1][child][word];
      }
      x = gl::canon(p2mt_dev::permute_wave4(x, ctx));
      if (rl < 4 && node < cnt) {
        buf[(t + 1) & 1][node][rl] = x;
        if (t == log_sub) cap[4 * (size_t)c + rl] = x;
        else if (next) next[4 * (lvl_off + ((size_t)c << (log_sub - t)) + node) + rl] = x;
      }
    }
    lvl_off += n_in >> t;
    __syncthreads();
  }
}

inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// ---------------------------------------------------------------- host-side table cache (per process / device)
struct Tables {
  std::map<std::pair<unsigned, int>, u64*> twiddles;                          // (log_n, inverse) -> w^i, i < n/2
  std::map<std::tuple<unsigned, unsigned, u64>, u64*> coset_pows;             // (log_n, rate_bits, shift)
};
Tables& tables() {
  static Tables t;
  return t;
}
// The tables are built once, on whichever thread needs them first, on that thread's stream: creation is serialised and
// the stream is synchronised before the table is published to the other threads (concurrent provers, runtime.h).
std::mutex& tables_mutex() {
  static std::mutex m;
  return m;
}

// host Goldilocks (table roots only: a handful of multiplications per call, no hashing)
inline u64 h_mul(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % gl::P); }
inline u64 h_pow(u64 a, u64 e) {
  u64 r = 1;
  for (; e; e >>= 1, a = h_mul(a, a))
    if (e & 1) r = h_mul(r, a);
  return r;
}
inline u64 h_root_of_unity(unsigned log_n) {  // 7^((p-1)/2^32) squared down (plonky2_field: generator 7, 2-adicity 32)
  u64 g = h_pow(7, (gl::P - 1) >> 32);
  for (unsigned i = log_n; i < 32; ++i) g = h_mul(g, g);
  return g;
}

int get_twiddles(unsigned log_n, int inverse, const u64** out) {
  std::lock_guard<std::mutex> lock(tables_mutex());
  auto key = std::make_pair(log_n, inverse);
  auto it = tables().twiddles.find(key);
  if (it == tables().twiddles.end()) {
    const size_t count = log_n ? ((size_t)1 << (log_n - 1)) : 1;
    u64* d = nullptr;
    if (hipMalloc((void**)&d, count * 8) != hipSuccess) return p2mt::fail(P2MT_ENOMEM, "hipMalloc(twiddles) failed");
    u64 root = h_root_of_unity(log_n);
    if (inverse) root = h_pow(root, gl::P - 2);
    hipLaunchKernelGGL(k_powers, dim3(grid_for(count)), dim3(kBlock), 0, p2mt::rt().stream, d, root, count);
    P2MT_LAUNCH_CHECK();
    P2MT_HIP(hipStreamSynchronize(p2mt::rt().stream));
    it = tables().twiddles.emplace(key, d).first;
  }
  *out = it->second;
  return P2MT_OK;
}

// w^i for i in [0, n): the register-blocked kernels index it with products t * o that exceed n/2
int get_full_twiddles(unsigned log_n, const u64** out) {
  std::lock_guard<std::mutex> lock(tables_mutex());
  static std::map<unsigned, u64*> full;
  auto it = full.find(log_n);
  if (it == full.end()) {
    const size_t count = (size_t)1 << log_n;
    u64* d = nullptr;
    if (hipMalloc((void**)&d, count * 8) != hipSuccess) return p2mt::fail(P2MT_ENOMEM, "hipMalloc(twiddles) failed");
    hipLaunchKernelGGL(k_powers, dim3(grid_for(count)), dim3(kBlock), 0, p2mt::rt().stream, d, h_root_of_unity(log_n), count);
    P2MT_LAUNCH_CHECK();
    P2MT_HIP(hipStreamSynchronize(p2mt::rt().stream));
    it = full.emplace(log_n, d).first;
  }
  *out = it->second;
  return P2MT_OK;
}

// the pass twiddles of k_coset_lde12_v2 in access order (built once from the natural-order table)
int get_lde12_tables(const u64* tw_full, const u64** ta, const u64** tb) {
  std::lock_guard<std::mutex> lock(tables_mutex());
  static u64* d = nullptr;
  if (!d) {
    u64* p = nullptr;
    if (hipMalloc((void**)&p, (15 * 256 + 15 * 16) * 8) != hipSuccess) return p2mt::fail(P2MT_ENOMEM, "hipMalloc(twiddles) failed");
    hipLaunchKernelGGL(k_lde12_tables, dim3(15), dim3(kBlock), 0, p2mt::rt().stream, tw_full, p, p + 15 * 256);
    P2MT_LAUNCH_CHECK();
    P2MT_HIP(hipStreamSynchronize(p2mt::rt().stream));
    d = p;
  }
  *ta = d;
  *tb = d + 15 * 256;
  return P2MT_OK;
}

int get_coset_pows(unsigned log_n, unsigned rate_bits, u64 shift, const u64** out) {
  std::lock_guard<std::mutex> lock(tables_mutex());
  auto key = std::make_tuple(log_n, rate_bits, shift);
  auto it = tables().coset_pows.find(key);
  if (it == tables().coset_pows.end()) {
    const size_t n = (size_t)1 << log_n, cosets = (size_t)1 << rate_bits;
    u64* d = nullptr;
    if (hipMalloc((void**)&d, n * cosets * 8) != hipSuccess) return p2mt::fail(P2MT_ENOMEM, "hipMalloc(coset powers) failed");
    const u64 w_big = h_root_of_unity(log_n + rate_bits);
    for (size_t j = 0; j < cosets; ++j) {
      const u64 base = h_mul(shift % gl::P, h_pow(w_big, j));
      hipLaunchKernelGGL(k_powers, dim3(grid_for(n)), dim3(kBlock), 0, p2mt::rt().stream, d + j * n, base, n);
      P2MT_LAUNCH_CHECK();
    }
    P2MT_HIP(hipStreamSynchronize(p2mt::rt().stream));
    it = tables().coset_pows.emplace(key, d).first;
  }
  *out = it->second;
  return P2MT_OK;
}

// tables of the four-step transforms (built once per direction / size)
int get_fourstep_tables(unsigned log_n, int inverse, const u64** ta1, const u64** t4) {
  std::lock_guard<std::mutex> lock(tables_mutex());
  static u64* d_ta1[2] = {nullptr, nullptr};
  static std::map<std::pair<unsigned, int>, u64*> d_t4;
  hipStream_t st = p2mt::rt().stream;
  if (!d_ta1[inverse]) {
    u64* p = nullptr;
    if (hipMalloc((void**)&p, 16 * 64 * 8) != hipSuccess) return p2mt::fail(P2MT_ENOMEM, "hipMalloc(twiddles) failed");
    u64 w1024 = h_root_of_unity(10);
    if (inverse) w1024 = h_pow(w1024, gl::P - 2);
    hipLaunchKernelGGL(k_ntt1024_tables, dim3(4), dim3(kBlock), 0, st, w1024, p);
    P2MT_LAUNCH_CHECK();
    P2MT_HIP(hipStreamSynchronize(st));
    d_ta1[inverse] = p;
  }
  auto key = std::make_pair(log_n, inverse);
  auto it = d_t4.find(key);
  if (it == d_t4.end()) {
    u64* p = nullptr;
    if (hipMalloc((void**)&p, ((size_t)8) << log_n) != hipSuccess) return p2mt::fail(P2MT_ENOMEM, "hipMalloc(twiddles) failed");
    u64 wn = h_root_of_unity(log_n), scale = 1;
    if (inverse) {
      wn = h_pow(wn, gl::P - 2);
      scale = h_pow(((u64)1 << log_n) % gl::P, gl::P - 2);
    }
    hipLaunchKernelGGL(k_fourstep_twiddles, dim3(grid_for((size_t)1 << log_n)), dim3(kBlock), 0, st, wn, scale, log_n - 10, p);
    P2MT_LAUNCH_CHECK();
    P2MT_HIP(hipStreamSynchronize(st));
    it = d_t4.emplace(key, p).first;
  }
  *ta1 = d_ta1[inverse];
  *t4 = it->second;
  return P2MT_OK;
}

// fft_with_options / ifft_with_options for 2^13 .. 2^20 points, natural order in and out: n = 1024 x 2^m (four-step).
// d_data -> d_tmp (pass 1: 1024-point transforms down the columns, times w_n^(i2 k1)) -> d_data (pass 2: 2^m-point transforms along the
// rows, the result written transposed = natural order).  Two launches, 16 B of HBM traffic per point each.
int ntt_fourstep_natural_dev(u64* d_data, u64* d_tmp, unsigned log_n, size_t n_polys, int inverse) {
  const unsigned m = log_n - 10;
  const u64 *ta1, *t4, *twh, *twm;
  P2MT_TRY(get_fourstep_tables(log_n, inverse, &ta1, &t4));
  P2MT_TRY(get_twiddles(10, inverse, &twh));
  P2MT_TRY(get_twiddles(m, inverse, &twm));
  hipStream_t st = p2mt::rt().stream;
  const unsigned force = p2mt::rt().force_fallback ? 1u : 0u;
  // Tile width, measured at 2^20 x 128 (profiles/r05_ntt_passes.txt): 8 transforms per workgroup for both passes -- two 512-thread
  // workgroups per CU cover each other's memory phases (column pass 0.77 ms against 0.82-0.88 with 16, row pass 0.58 against 0.74).
  // Until round 5 the column pass took 159 VGPRs at this width (one workgroup per CU, 1.16 ms) and 16 was the faster one.
  // P2MT_LDE12=3 forces 16 for both (A/B).
  const int mode = p2mt::rt().use_lde12;
  {
    const bool wide = m >= 4 && mode == 3;
    const dim3 grid((unsigned)n_polys, (1u << m) / (wide ? 16 : 8)), block(wide ? 1024 : 512);
    const int slot = p2mt::prof_begin();
    if (wide && !inverse) hipLaunchKernelGGL((k_ntt20_pass<0, false, true, 16>), grid, block, 0, st, (const u64*)d_data, d_tmp, ta1, t4, twh, m, force);
    if (wide && inverse) hipLaunchKernelGGL((k_ntt20_pass<1, false, true, 16>), grid, block, 0, st, (const u64*)d_data, d_tmp, ta1, t4, twh, m, force);
    if (!wide && !inverse) hipLaunchKernelGGL((k_ntt20_pass<0, false, true, 8>), grid, block, 0, st, (const u64*)d_data, d_tmp, ta1, t4, twh, m, force);
    if (!wide && inverse) hipLaunchKernelGGL((k_ntt20_pass<1, false, true, 8>), grid, block, 0, st, (const u64*)d_data, d_tmp, ta1, t4, twh, m, force);
    p2mt::prof_end(slot);
    P2MT_LAUNCH_CHECK();
  }
  if (m == 10) {
    const bool wide = mode == 3;
    const dim3 grid((unsigned)n_polys, wide ? 64 : 128), block(wide ? 1024 : 512);
    const int slot = p2mt::prof_begin();
    if (wide && !inverse) hipLaunchKernelGGL((k_ntt20_pass<0, true, false, 16>), grid, block, 0, st, (const u64*)d_tmp, d_data, ta1, t4, twh, 10u, force);
    if (wide && inverse) hipLaunchKernelGGL((k_ntt20_pass<1, true, false, 16>), grid, block, 0, st, (const u64*)d_tmp, d_data, ta1, t4, twh, 10u, force);
    if (!wide && !inverse) hipLaunchKernelGGL((k_ntt20_pass<0, true, false, 8>), grid, block, 0, st, (const u64*)d_tmp, d_data, ta1, t4, twh, 10u, force);
    if (!wide && inverse) hipLaunchKernelGGL((k_ntt20_pass<1, true, false, 8>), grid, block, 0, st, (const u64*)d_tmp, d_data, ta1, t4, twh, 10u, force);
    p2mt::prof_end(slot);
    P2MT_LAUNCH_CHECK();
  } else {
    const dim3 grid((unsigned)n_polys, 64);
    const size_t lds = (size_t)16 * 8 << m;
    if (inverse) hipLaunchKernelGGL((k_ntt_rows_small<1>), grid, dim3(kBlock), lds, st, (const u64*)d_tmp, d_data, twm, m);
    else hipLaunchKernelGGL((k_ntt_rows_small<0>), grid, dim3(kBlock), lds, st, (const u64*)d_tmp, d_data, twm, m);
    P2MT_LAUNCH_CHECK();
  }
  return P2MT_OK;
}

// DIF transform of n_polys contiguous 2^log_n-point rows, in place, natural -> bit-reversed order.
int ntt_dif_dev(u64* d_data, unsigned log_n, size_t n_polys, int inverse) {
  if (log_n == 0) return P2MT_OK;
  const u64* tw;
  P2MT_TRY(get_twiddles(log_n, inverse, &tw));
  hipStream_t st = p2mt::rt().stream;
  const unsigned c = log_n < kLdsLog ? log_n : kLdsLog;
  for (unsigned s = 0; s + c < log_n; ++s) {
    hipLaunchKernelGGL(k_ntt_global_stage, bgrid(grid_for(n_polys << (log_n - 1))), dim3(kBlock), 0, st, d_data, log_n, s,
                       tw, n_polys, barg());
    P2MT_LAUNCH_CHECK();
  }
  const size_t chunks = n_polys << (log_n - c);
  hipLaunchKernelGGL(k_ntt_lds_tail, bgrid((unsigned)chunks), dim3(kBlock), (size_t)8 << c, st, d_data, log_n, c, tw, barg());
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}

}  // namespace

using p2mt::DevBuf;
using p2mt::rt;


// =================================================================== test hook for the field primitives
namespace {
__global__ __launch_bounds__(kBlock) void k_debug_field_op(int op, const u64* __restrict__ a, const u64* __restrict__ b, size_t n,
                                                           u64* __restrict__ out, uint8_t* __restrict__ flag) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const unsigned lane = threadIdx.x & 63;
  u64 sticky = 0, r = 0;
  switch (op) {
    case 0: r = poseidon_fast::exact::reduce128(a[i], b[i]); break;
    case 1: r = poseidon_fast::exact::fold96((u32)(b[i] & 0x3FF), a[i]); break;
    case 2: r = poseidon_fast::reduce128(a[i], b[i], sticky); break;
    case 3: r = poseidon_fast::exact::mul(a[i], b[i]); break;
    case 4: r = lde12::add_l(a[i], b[i], sticky); break;
    case 5: r = lde12::sub_l(a[i], b[i], sticky); break;
    case 6: r = gl::mul(a[i], b[i]); break;                                                // the generic multiply of every prover kernel
    case 7: r = gl::mul_add(a[i], b[i], ((a[i] << 17) | (a[i] >> 47)) ^ b[i]); break;      // a b + c, c = rotl(a, 17) ^ b
    case 8: r = poseidon_fast::mul(a[i], b[i], sticky); break;                             // flag form: right unless flagged
    case 9: r = poseidon_fast::sub_any(a[i], b[i]); break;                                 // a - b mod p, any operands, exact
    case 10: r = poseidon_fast::sub_flag(a[i], b[i], sticky); break;                       // ... the second wrap left to the flag
    default: break;
  }
  out[i] = gl::canon(r);
  flag[i] = (uint8_t)((sticky >> lane) & 1);
}
// ops 11..207: the transform kernels' arithmetic (ntt_arith.hip.h), one instantiation per op: no control flow at all around the
// field operations (their sticky mask lives in an SGPR pair the asm blocks update in place).
template <int OP>
__global__ __launch_bounds__(kBlock) void k_debug_ntt_op(const u64* __restrict__ a, const u64* __restrict__ b, size_t n,
                                                         u64* __restrict__ out, uint8_t* __restrict__ flag) {
  const size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t i = i0 < n ? i0 : n - 1;
  const unsigned lane = threadIdx.x & 63;
  const u64 x = a[i], y = b[i];
  u64 sticky = 0, r = 0, other;
  if constexpr (OP == 11) r = ntt::add(x, y, sticky);
  else if constexpr (OP == 12) r = ntt::sub(x, y, sticky);
  else if constexpr (OP == 13) r = ntt::mul(x, y, sticky);
  else if constexpr (OP == 14) ntt::bfly(x, y, r, other, sticky);
  else if constexpr (OP == 15) ntt::bfly(x, y, other, r, sticky);
  else r = ntt20::mul_pow2_mod192<OP - 16>(x, sticky);  // x * 2^E for E in [0, 192)  (2^96 = -1)
  if (i0 < n) {
    out[i] = gl::canon(r);
    flag[i] = (uint8_t)((sticky >> lane) & 1);
  }
}
template <int OP>
void launch_debug_ntt_one(unsigned grid, hipStream_t st, const u64* a, const u64* b, size_t n, u64* out, uint8_t* flag) {
  hipLaunchKernelGGL((k_debug_ntt_op<OP>), dim3(grid), dim3(kBlock), 0, st, a, b, n, out, flag);
}
template <int... I>
void launch_debug_ntt_op(int op, std::integer_sequence<int, I...>, unsigned grid, hipStream_t st, const u64* a, const u64* b, size_t n,
                         u64* out, uint8_t* flag) {
  ((op == 11 + I ? launch_debug_ntt_one<11 + I>(grid, st, a, b, n, out, flag) : (void)0), ...);
}
}  // namespace

extern "C" int p2mt_debug_field_op(int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out, uint8_t* flag_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n == 0) return P2MT_OK;
  if (!a || !b || !out || !flag_out || op < 0 || op > 207) return p2mt::fail(P2MT_EINVAL, "bad argument");
  DevBuf ba, bb, bo, bf;
  P2MT_TRY(ba.alloc(n * 8));
  P2MT_TRY(bb.alloc(n * 8));
  P2MT_TRY(bo.alloc(n * 8));
  P2MT_TRY(bf.alloc(n));
  hipStream_t st = rt().stream;
  P2MT_HIP(hipMemcpyAsync(ba.p, a, n * 8, hipMemcpyHostToDevice, st));
  P2MT_HIP(hipMemcpyAsync(bb.p, b, n * 8, hipMemcpyHostToDevice, st));
  if (op >= 11)
    launch_debug_ntt_op(op, std::make_integer_sequence<int, 197>{}, grid_for(n), st, (const u64*)ba.as<u64>(),
                        (const u64*)bb.as<u64>(), n, bo.as<u64>(), bf.as<uint8_t>());
  else
    hipLaunchKernelGGL(k_debug_field_op, dim3(grid_for(n)), dim3(kBlock), 0, st, op, (const u64*)ba.as<u64>(),
                       (const u64*)bb.as<u64>(), n, bo.as<u64>(), bf.as<uint8_t>());
  P2MT_LAUNCH_CHECK();
  P2MT_HIP(hipMemcpyAsync(out, bo.p, n * 8, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipMemcpyAsync(flag_out, bf.p, n, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  return P2MT_OK;
  });
}

// =================================================================== fft_with_options / ifft_with_options
extern "C" int p2mt_ntt_batch_dev(uint64_t* d_data, unsigned log_n, size_t n_polys, int inverse) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n_polys == 0) return P2MT_OK;
  if (!d_data || log_n > 32) return p2mt::fail(P2MT_EINVAL, "ntt: bad argument (2-adicity of the field is 32)");
  const size_t total = n_polys << log_n;
  if (log_n >= 13 && log_n <= 20 && n_polys < ((size_t)1 << 31) && rt().use_lde12 && p2mt::batch_B() == 1) {  // the four-step path (two launches, natural order out)
    u64* tmp20;
    P2MT_TRY(p2mt::scratch_get(p2mt::kScratchLde, total * 8, (void**)&tmp20));
    return ntt_fourstep_natural_dev(d_data, tmp20, log_n, n_polys, inverse != 0);
  }
  DevBuf tmp;
  P2MT_TRY(tmp.alloc(total * 8));
  P2MT_HIP(hipMemcpyAsync(tmp.p, d_data, total * 8, hipMemcpyDeviceToDevice, rt().stream));
  P2MT_TRY(ntt_dif_dev(tmp.as<u64>(), log_n, n_polys, inverse));
  const u64 scale = inverse ? h_pow(((u64)1 << log_n) % gl::P, gl::P - 2) : 1;
  hipLaunchKernelGGL(k_bitrev_scale, bgrid(grid_for(total)), dim3(kBlock), 0, rt().stream, (const u64*)tmp.as<u64>(), d_data,
                     log_n, n_polys, scale, barg());
  P2MT_LAUNCH_CHECK();
  P2MT_HIP(hipStreamSynchronize(rt().stream));  // tmp dies with this call
  return P2MT_OK;
  });
}

extern "C" int p2mt_ntt_batch(uint64_t* data, unsigned log_n, size_t n_polys, int inverse) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n_polys == 0) return P2MT_OK;
  if (!data || log_n > 32) return p2mt::fail(P2MT_EINVAL, "ntt: bad argument");
  const size_t bytes = (n_polys << log_n) * 8;
  DevBuf b;
  P2MT_TRY(b.alloc(bytes));
  P2MT_HIP(hipMemcpyAsync(b.p, data, bytes, hipMemcpyHostToDevice, rt().stream));
  P2MT_TRY(p2mt_ntt_batch_dev(b.as<u64>(), log_n, n_polys, inverse));
  P2MT_HIP(hipMemcpyAsync(data, b.p, bytes, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

// LDE into leaf order (poly-major): d_out[p][brev(i)] = f_p(shift * w_N^i)
int p2mt::coset_lde_leaf_order_dev(const u64* d_coeffs, unsigned log_n, unsigned rate_bits, u64 shift, size_t n_polys,
                                   u64* d_out) {
  if (rate_bits > 8 || log_n + rate_bits > 32) return p2mt::fail(P2MT_EINVAL, "coset_lde: bad rate_bits");
  if (log_n > kLdsLog) {
    // A transform that does not fit one workgroup's LDS: coset j of polynomial p is row p * 2^r + brev_r(j) of the output, so the
    // scaling writes c_m (s w_N^j)^m straight into its row and ONE in-place DIF over all n_polys * 2^r rows (natural -> bit-reversed,
    // which is leaf order) finishes it.  General, not fast: the stages above 2^12 are radix-2 passes over HBM (only the 2^20-point
    // natural-order transform has the two-pass kernel so far); the reference's circuits end at 2^12.
    if (p2mt::batch_B() != 1) return p2mt::fail(P2MT_EINVAL, "coset_lde: log_n > 12 inside a batched pass");
    const u64 w_big = h_root_of_unity(log_n + rate_bits);
    const size_t rows = n_polys << rate_bits;
    if (rows * grid_for((size_t)1 << log_n) >= ((size_t)1 << 31)) return p2mt::fail(P2MT_EINVAL, "coset_lde: too many rows for one launch");
    const int slot = p2mt::prof_begin();
    hipLaunchKernelGGL(k_coset_scale_rows, dim3((unsigned)(rows * grid_for((size_t)1 << log_n))), dim3(kBlock), 0, rt().stream, d_coeffs, log_n,
                       rate_bits, shift % gl::P, w_big, d_out);
    p2mt::prof_end(slot);
    P2MT_LAUNCH_CHECK();
    return ntt_dif_dev(d_out, log_n, rows, 0);
  }
  const u64 *tw, *cp;
  P2MT_TRY(get_twiddles(log_n, 0, &tw));
  P2MT_TRY(get_coset_pows(log_n, rate_bits, shift, &cp));
  const int slot = p2mt::prof_begin();  // the LDE is the HBM-streaming kernel of the commit step
  if (log_n == 12 && rt().use_lde12) {
    const u64* twf;
    P2MT_TRY(get_full_twiddles(12, &twf));
    if (rt().use_lde12 == 1) {  // the round-3 kernel, kept for the A/B (P2MT_LDE12=1)
      hipLaunchKernelGGL(k_coset_lde12, bgrid((unsigned)(n_polys << rate_bits)), dim3(kBlock), 0, rt().stream, d_coeffs,
                         rate_bits, cp, twf, tw, d_out, barg());
    } else {
      const u64 *ta, *tb;
      P2MT_TRY(get_lde12_tables(twf, &ta, &tb));
      hipLaunchKernelGGL(k_coset_lde12_v2, bgrid((unsigned)(n_polys << rate_bits)), dim3(kBlock), 0, rt().stream, d_coeffs,
                         rate_bits, (unsigned)n_polys, cp, ta, tb, tw, d_out, rt().force_fallback ? 1u : 0u, barg());
    }
  } else {
    hipLaunchKernelGGL(k_coset_lde, bgrid((unsigned)(n_polys << rate_bits)), dim3(kBlock), (size_t)8 << log_n, rt().stream,
                       d_coeffs, log_n, rate_bits, cp, tw, d_out, barg());
  }
  p2mt::prof_end(slot);
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}

extern "C" int p2mt_coset_lde_batch_dev(const uint64_t* d_coeffs, unsigned log_n, unsigned rate_bits, uint64_t shift,
                                        size_t n_polys, uint64_t* d_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n_polys == 0) return P2MT_OK;
  if (!d_coeffs || !d_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  const unsigned log_big = log_n + rate_bits;
  const size_t total = n_polys << log_big;
  DevBuf tmp;
  P2MT_TRY(tmp.alloc(total * 8));
  P2MT_TRY(p2mt::coset_lde_leaf_order_dev(d_coeffs, log_n, rate_bits, shift, n_polys, tmp.as<u64>()));
  hipLaunchKernelGGL(k_bitrev_scale, bgrid(grid_for(total)), dim3(kBlock), 0, rt().stream, (const u64*)tmp.as<u64>(), d_out,
                     log_big, n_polys, (u64)1, barg());
  P2MT_LAUNCH_CHECK();
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

extern "C" int p2mt_coset_lde_leaf_order_dev(const uint64_t* d_coeffs, unsigned log_n, unsigned rate_bits, uint64_t shift,
                                             size_t n_polys, uint64_t* d_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n_polys == 0) return P2MT_OK;
  if (!d_coeffs || !d_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  return p2mt::coset_lde_leaf_order_dev(d_coeffs, log_n, rate_bits, shift, n_polys, d_out);
  });
}

extern "C" int p2mt_coset_lde_batch(const uint64_t* coeffs, unsigned log_n, unsigned rate_bits, uint64_t shift,
                                    size_t n_polys, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n_polys == 0) return P2MT_OK;
  if (!coeffs || !out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  const size_t in_bytes = (n_polys << log_n) * 8, out_bytes = in_bytes << rate_bits;
  DevBuf bi, bo;
  P2MT_TRY(bi.alloc(in_bytes));
  P2MT_TRY(bo.alloc(out_bytes));
  P2MT_HIP(hipMemcpyAsync(bi.p, coeffs, in_bytes, hipMemcpyHostToDevice, rt().stream));
  P2MT_TRY(p2mt_coset_lde_batch_dev(bi.as<u64>(), log_n, rate_bits, shift, n_polys, bo.as<u64>()));
  P2MT_HIP(hipMemcpyAsync(out, bo.p, out_bytes, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

// =================================================================== MerkleTree::new(leaves, cap_height)
static int log2_strict(size_t n) {
  if (n == 0 || (n & (n - 1))) return -1;
  return __builtin_ctzll((unsigned long long)n);
}

// digests level 0 must already be in d_level0 (n HashOuts).  Builds levels 1.. and the cap.
static int merkle_levels_to_cap(u64* d_level0, size_t n, unsigned cap_height, u64* d_digests_out, u64* d_cap_out) {
  const int k = log2_strict(n);
  // level-major digests: level j has n >> j entries, levels 0 .. k-cap_height-1; the next row is the cap
  if ((unsigned)k == cap_height) {
    return p2mt::batch_copy(d_cap_out, d_level0, n * 32);
  }
  u64* cur = d_level0;
  size_t cur_n = n;
  u64* next_store = d_digests_out ? d_digests_out + 4 * n : nullptr;
  u64* ping = nullptr;  // ping-pong rows when the caller does not want digests
  if (!d_digests_out) P2MT_TRY(p2mt::scratch_get(p2mt::kScratchPing, n * 32, (void**)&ping));
  for (unsigned level = 0; level < (unsigned)k - cap_height; ++level) {
    const unsigned remaining = (unsigned)k - cap_height - level;
    if (remaining >= 2 && remaining <= 5 && rt().mds == 2 && rt().use_quad && !rt().throughput) {  // the top of the tree in one launch
      hipLaunchKernelGGL(k_merkle_top, bgrid(1u << cap_height), dim3(1024), 0, rt().stream, (const u64*)cur, remaining,
                         d_digests_out ? next_store : nullptr, cur_n, d_cap_out, barg(), p2mt::perm_ctx());
      P2MT_LAUNCH_CHECK();
      return P2MT_OK;
    }
    const bool last = level + 1 == (unsigned)k - cap_height;
    u64* dst = last ? d_cap_out : (d_digests_out ? next_store : ping + (level & 1 ? 0 : 4 * (n / 2)));
    P2MT_TRY(p2mt::launch_merkle_level_dev(cur, dst, cur_n / 2));
    cur = dst;
    cur_n /= 2;
    if (d_digests_out && !last) next_store += 4 * cur_n;
  }
  return P2MT_OK;
}

extern "C" int p2mt_merkle_cap_commit_dev(const uint64_t* d_leaves, size_t n, size_t width, unsigned cap_height,
                                          uint64_t* d_digests_out, uint64_t* d_cap_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  const int k = log2_strict(n);
  if (k < 0 || cap_height > (unsigned)k) return p2mt::fail(P2MT_EINVAL, "MerkleTree::new: n must be a power of two >= 2^cap_height");
  if (!d_leaves || !d_cap_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  u64* d_level0 = d_digests_out;
  if (!d_level0 || (unsigned)k == cap_height) P2MT_TRY(p2mt::scratch_get(p2mt::kScratchLevel0, n * 32, (void**)&d_level0));
  P2MT_TRY(p2mt::launch_hash_rows_dev(d_leaves, n, width, 1, d_level0));
  return merkle_levels_to_cap(d_level0, n, cap_height, (unsigned)k == cap_height ? nullptr : d_digests_out, d_cap_out);
  });
}

static size_t digests_count(size_t n, unsigned cap_height) {
  const int k = log2_strict(n);
  size_t c = 0;
  for (unsigned j = 0; j + cap_height < (unsigned)k; ++j) c += n >> j;
  return c;
}

extern "C" int p2mt_merkle_cap_commit(const uint64_t* leaves, size_t n, size_t width, unsigned cap_height,
                                      uint64_t* digests_out, uint64_t* cap_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  const int k = log2_strict(n);
  if (k < 0 || cap_height > (unsigned)k) return p2mt::fail(P2MT_EINVAL, "MerkleTree::new: n must be a power of two >= 2^cap_height");
  if (!leaves || !cap_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  const size_t nd = digests_count(n, cap_height), ncap = (size_t)1 << cap_height;
  DevBuf bl, bd, bc;
  P2MT_TRY(bl.alloc(n * width * 8));
  P2MT_TRY(bd.alloc((nd ? nd : 1) * 32));
  P2MT_TRY(bc.alloc(ncap * 32));
  P2MT_HIP(hipMemcpyAsync(bl.p, leaves, n * width * 8, hipMemcpyHostToDevice, rt().stream));
  P2MT_TRY(p2mt_merkle_cap_commit_dev(bl.as<u64>(), n, width, cap_height, nd ? bd.as<u64>() : nullptr, bc.as<u64>()));
  if (digests_out && nd) P2MT_HIP(hipMemcpyAsync(digests_out, bd.p, nd * 32, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipMemcpyAsync(cap_out, bc.p, ncap * 32, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

// =================================================================== plonky2's MerkleTree { digests } layout
// hash/merkle_tree.rs (plonky2 @3b21b87, absent) keeps `digests` per cap subtree in the recursive order its
// fill_subtree writes -- left recursive output || left child digest || right child digest || right recursive output -- i.e. in
// units of sibling PAIRS: [layer 0, layer 1, layer 0, layer 2, layer 0, layer 1, layer 0, layer 3, ...]; the pair q of layer i
// sits at pair position (q << (i + 1)) + 2^i - 1 (the formula MerkleTree::prove indexes with).  This library's kernels write
// `digests` level-major (level 0 = leaf digests); the map below is what a patched PolynomialBatch::from_values needs to fill
// MerkleTree { leaves, digests, cap } without re-indexing.  One lane per digest, 32-byte records.
__global__ __launch_bounds__(kBlock) void k_digests_to_plonky2(const u64* __restrict__ level_major, u64* __restrict__ out,
                                                               unsigned log_n, unsigned cap_height, size_t total) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= total) return;
  // level of record t: levels have n, n/2, ... entries
  size_t g = t;
  unsigned lvl = 0;
  for (; lvl + cap_height < log_n; ++lvl) {
    const size_t cnt = (size_t)1 << (log_n - lvl);
    if (g < cnt) break;
    g -= cnt;
  }
  const unsigned sub_log = log_n - cap_height;                    // log2(leaves per cap subtree)
  const size_t per_sub = ((size_t)2 << sub_log) - 2;              // digests per cap subtree
  const size_t in_level_per_sub = (size_t)1 << (sub_log - lvl);   // nodes of this level per subtree
  const size_t sub = g / in_level_per_sub, j = g % in_level_per_sub;
  const size_t pair_pos = ((j >> 1) << (lvl + 1)) + (((size_t)1 << lvl) - 1);
  const size_t dst = sub * per_sub + 2 * pair_pos + (j & 1);
  const ulonglong4 v = reinterpret_cast<const ulonglong4*>(level_major)[t];
  reinterpret_cast<ulonglong4*>(out)[dst] = v;
}

extern "C" int p2mt_merkle_digests_to_plonky2_layout_dev(const uint64_t* d_level_major, size_t n_leaves, unsigned cap_height,
                                                         uint64_t* d_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  const int k = log2_strict(n_leaves);
  if (k < 0 || cap_height > (unsigned)k) return p2mt::fail(P2MT_EINVAL, "MerkleTree: n must be a power of two >= 2^cap_height");
  const size_t nd = digests_count(n_leaves, cap_height);
  if (nd == 0) return P2MT_OK;  // the tree is its cap: plonky2's `digests` is empty
  if (!d_level_major || !d_out || d_level_major == d_out) return p2mt::fail(P2MT_EINVAL, "null or aliasing pointer");
  hipLaunchKernelGGL(k_digests_to_plonky2, dim3(grid_for(nd)), dim3(kBlock), 0, rt().stream, d_level_major, d_out, (unsigned)k,
                     cap_height, nd);
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
  });
}

extern "C" int p2mt_merkle_digests_to_plonky2_layout(const uint64_t* level_major, size_t n_leaves, unsigned cap_height,
                                                     uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  const int k = log2_strict(n_leaves);
  if (k < 0 || cap_height > (unsigned)k) return p2mt::fail(P2MT_EINVAL, "MerkleTree: n must be a power of two >= 2^cap_height");
  const size_t nd = digests_count(n_leaves, cap_height);
  if (nd == 0) return P2MT_OK;
  if (!level_major || !out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  DevBuf bi, bo;
  P2MT_TRY(bi.alloc(nd * 32));
  P2MT_TRY(bo.alloc(nd * 32));
  P2MT_HIP(hipMemcpyAsync(bi.p, level_major, nd * 32, hipMemcpyHostToDevice, rt().stream));
  P2MT_TRY(p2mt_merkle_digests_to_plonky2_layout_dev(bi.as<u64>(), n_leaves, cap_height, bo.as<u64>()));
  P2MT_HIP(hipMemcpyAsync(out, bo.p, nd * 32, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

// =================================================================== PolynomialBatch::from_values / from_coeffs
int p2mt::commit_batch_dev(const uint64_t* d_polys, int is_values, size_t n_polys, unsigned log_n, unsigned rate_bits,
                           unsigned cap_height, uint64_t* d_coeffs_out, uint64_t* d_lde_out, uint64_t* d_leaves_out,
                           uint64_t* d_digests_out, uint64_t* d_cap_out) {
  if (!d_polys || !d_cap_out || n_polys == 0) return p2mt::fail(P2MT_EINVAL, "bad argument");
  const unsigned log_big = log_n + rate_bits;
  if (cap_height > log_big) return p2mt::fail(P2MT_EINVAL, "cap_height exceeds tree height");
  const size_t n = (size_t)1 << log_n, big = (size_t)1 << log_big;
  hipStream_t st = rt().stream;
  const u64* d_coeffs = d_polys;
  bool lde_done = false;
  u64* lde = d_lde_out;
  if (!lde) P2MT_TRY(p2mt::scratch_get(p2mt::kScratchLde, n_polys * big * 8, (void**)&lde));
  if (is_values && log_n <= 9 && rate_bits <= 8) {  // IFFT + x2^rate_bits coset LDE in one launch
    u64* coeffs = d_coeffs_out;
    if (!coeffs) P2MT_TRY(p2mt::scratch_get(p2mt::kScratchCoeffs, n_polys * n * 8, (void**)&coeffs));
    const u64 *tw, *twi, *cp;
    P2MT_TRY(get_twiddles(log_n, 0, &tw));
    P2MT_TRY(get_twiddles(log_n, 1, &twi));
    P2MT_TRY(get_coset_pows(log_n, rate_bits, 7, &cp));
    const int slot = p2mt::prof_begin();
    static const bool wave64_knob = [] { const char* e = getenv("P2MT_LDE_WAVE64"); return e ? atoi(e) != 0 : true; }();
    // one wavefront per polynomial for the proofs of a batch (env P2MT_LDE_WAVE64=0: the workgroup-per-coset kernel, A/B); a single
    // proof keeps the workgroup-per-coset kernel: 171 polynomials do not fill the chip either way and its 18 barrier stages take
    // 6 us where the wavefront's 54 dependent lane-exchange stages take 11
    if (log_n == 6 && wave64_knob && p2mt::batch_B() > 1)
      hipLaunchKernelGGL(k_ifft_coset_lde_wave64, bgrid((unsigned)((n_polys + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, st, d_polys,
                         rate_bits, n_polys, cp, tw, twi, h_pow((u64)n % gl::P, gl::P - 2), coeffs, lde, barg());
    else
      hipLaunchKernelGGL(k_ifft_coset_lde, bgrid((unsigned)(n_polys << rate_bits)), dim3(kBlock), (size_t)16 << log_n, st, d_polys,
                         log_n, rate_bits, cp, tw, twi, h_pow((u64)n % gl::P, gl::P - 2), coeffs, lde, barg());
    p2mt::prof_end(slot);
    P2MT_LAUNCH_CHECK();
    d_coeffs = coeffs;
    lde_done = true;
  } else if (is_values && log_n <= kLdsLog) {  // IFFT in one launch
    u64* coeffs = d_coeffs_out;
    if (!coeffs) P2MT_TRY(p2mt::scratch_get(p2mt::kScratchCoeffs, n_polys * n * 8, (void**)&coeffs));
    const u64* twi;
    P2MT_TRY(get_twiddles(log_n, 1, &twi));
    hipLaunchKernelGGL(k_ifft_small, bgrid((unsigned)n_polys), dim3(kBlock), (size_t)8 << log_n, st, d_polys, coeffs, log_n, twi,
                       h_pow((u64)n % gl::P, gl::P - 2), (u64)1, barg());
    P2MT_LAUNCH_CHECK();
    d_coeffs = coeffs;
  } else if (is_values) {  // IFFT: DIF with inverse roots, then bit-reversal + 1/n
    u64* buf;
    P2MT_TRY(p2mt::scratch_get(p2mt::kScratchCoeffs, n_polys * n * 8 * 2, (void**)&buf));
    u64* work = buf + n_polys * n;
    u64* coeffs = d_coeffs_out ? d_coeffs_out : buf;
    P2MT_TRY(p2mt::batch_copy(work, d_polys, n_polys * n * 8));
    P2MT_TRY(ntt_dif_dev(work, log_n, n_polys, 1));
    const u64 n_inv = h_pow((u64)n % gl::P, gl::P - 2);
    hipLaunchKernelGGL(k_bitrev_scale, bgrid(grid_for(n_polys * n)), dim3(kBlock), 0, st, (const u64*)work, coeffs, log_n, n_polys,
                       n_inv, barg());
    P2MT_LAUNCH_CHECK();
    d_coeffs = coeffs;
  } else if (d_coeffs_out && d_coeffs_out != d_polys) {
    P2MT_TRY(p2mt::batch_copy(d_coeffs_out, d_polys, n_polys * n * 8));
  }
  if (!lde_done) P2MT_TRY(p2mt::coset_lde_leaf_order_dev(d_coeffs, log_n, rate_bits, 7, n_polys, lde));
  const size_t big_all = big * p2mt::batch_B();  // (inside a batch the layout is chosen for all the leaves of the launch)
  const bool wave_sponge = n_polys > 4 && big_all <= ((size_t)1 << 12) && rt().mds == 2 && rt().use_quad && !rt().throughput;
  if (d_leaves_out && !wave_sponge) {
    hipLaunchKernelGGL(k_transpose, bgrid((unsigned)((big + 31) / 32), (unsigned)((n_polys + 31) / 32)), dim3(kBlock), 0, st,
                       (const u64*)lde, d_leaves_out, n_polys, big, barg());
    P2MT_LAUNCH_CHECK();
  }
  const bool cap_is_leaves = cap_height == log_big;
  u64* d_level0 = (d_digests_out && !cap_is_leaves) ? d_digests_out : nullptr;
  if (!d_level0) P2MT_TRY(p2mt::scratch_get(p2mt::kScratchLevel0, big * 32, (void**)&d_level0));
  if (wave_sponge) {
    hipLaunchKernelGGL(k_hash_columns_wave, bgrid((unsigned)((big + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, st,
                       (const u64*)lde, n_polys, big, d_level0, d_leaves_out, barg(), p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
  } else if (n_polys > 4 && big_all <= ((size_t)1 << 16) && rt().mds == 2 && rt().use_quad) {
    hipLaunchKernelGGL(k_hash_columns_quad, bgrid(grid_for(4 * big)), dim3(kBlock), 0, st, (const u64*)lde, n_polys, big,
                       d_level0, barg(), p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
  } else {
    if (rt().mds == 2 && rt().partial == 0) {  // default: dense MDS layers on the matrix pipe
      hipLaunchKernelGGL((k_hash_columns<2, 5>), bgrid(grid_for(big)), dim3(kBlock), 0, st, (const u64*)lde, n_polys, big, d_level0, barg(),
                         p2mt::perm_ctx());
      P2MT_LAUNCH_CHECK();
    } else {
      P2MT_DISPATCH(k_hash_columns, bgrid(grid_for(big)), kBlock, (const u64*)lde, n_polys, big, d_level0, barg());
    }
  }
  return merkle_levels_to_cap(d_level0, big, cap_height, cap_is_leaves ? nullptr : d_digests_out, d_cap_out);
}

extern "C" int p2mt_polynomial_batch_commit_dev(const uint64_t* d_polys, int is_values, size_t n_polys, unsigned log_n,
                                                unsigned rate_bits, unsigned cap_height, uint64_t* d_leaves_out,
                                                uint64_t* d_digests_out, uint64_t* d_cap_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  return p2mt::commit_batch_dev(d_polys, is_values, n_polys, log_n, rate_bits, cap_height, nullptr, nullptr, d_leaves_out,
                                d_digests_out, d_cap_out);
  });
}

// PolynomialValues::coset_ifft(shift): plain IFFT gives c_k shift^k; the bit-reversal pass also divides by n shift^k.
namespace {
__global__ __launch_bounds__(kBlock) void k_bitrev_coset_scale(const u64* __restrict__ in, u64* __restrict__ out, unsigned log_n,
                                                               size_t n_polys, u64 n_inv, u64 shift_inv, BatchArg ba) {
  in = bp(in, ba);
  out = bp(out, ba);
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= (n_polys << log_n)) return;
  const size_t poly = t >> log_n, q = t & (((size_t)1 << log_n) - 1);
  const size_t r = log_n ? (__brevll(q) >> (64 - log_n)) : 0;
  out[(poly << log_n) + r] = cmul(cmul(in[t], n_inv), gl::pow(shift_inv, r));
}
}  // namespace

int p2mt::coset_ifft_dev(uint64_t* d_vals, unsigned log_n, size_t n_polys, uint64_t shift, uint64_t* d_coeffs_out) {
  if (!d_vals || !d_coeffs_out || n_polys == 0 || log_n > 32) return p2mt::fail(P2MT_EINVAL, "coset_ifft: bad argument");
  const u64 n_inv = h_pow(((u64)1 << log_n) % gl::P, gl::P - 2), shift_inv = h_pow(shift % gl::P, gl::P - 2);
  if (log_n <= kLdsLog) {
    const u64* twi;
    P2MT_TRY(get_twiddles(log_n, 1, &twi));
    hipLaunchKernelGGL(k_ifft_small, bgrid((unsigned)n_polys), dim3(kBlock), (size_t)8 << log_n, rt().stream, (const u64*)d_vals,
                       d_coeffs_out, log_n, twi, n_inv, shift_inv, barg());
    P2MT_LAUNCH_CHECK();
    return P2MT_OK;
  }
  P2MT_TRY(ntt_dif_dev(d_vals, log_n, n_polys, 1));
  hipLaunchKernelGGL(k_bitrev_coset_scale, bgrid(grid_for(n_polys << log_n)), dim3(kBlock), 0, rt().stream, (const u64*)d_vals,
                     d_coeffs_out, log_n, n_polys, n_inv, shift_inv, barg());
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}

extern "C" int p2mt_polynomial_batch_commit(const uint64_t* polys, int is_values, size_t n_polys, unsigned log_n,
                                            unsigned rate_bits, unsigned cap_height, uint64_t* leaves_out,
                                            uint64_t* digests_out, uint64_t* cap_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!polys || !cap_out || n_polys == 0) return p2mt::fail(P2MT_EINVAL, "bad argument");
  const unsigned log_big = log_n + rate_bits;
  if (cap_height > log_big) return p2mt::fail(P2MT_EINVAL, "cap_height exceeds tree height");
  const size_t n = (size_t)1 << log_n, big = (size_t)1 << log_big;
  const size_t nd = digests_count(big, cap_height), ncap = (size_t)1 << cap_height;
  DevBuf bp, bl, bd, bc;
  P2MT_TRY(bp.alloc(n_polys * n * 8));
  if (leaves_out) P2MT_TRY(bl.alloc(n_polys * big * 8));
  P2MT_TRY(bd.alloc((nd ? nd : 1) * 32));
  P2MT_TRY(bc.alloc(ncap * 32));
  hipStream_t st = rt().stream;
  P2MT_HIP(hipMemcpyAsync(bp.p, polys, n_polys * n * 8, hipMemcpyHostToDevice, st));
  P2MT_TRY(p2mt_polynomial_batch_commit_dev(bp.as<u64>(), is_values, n_polys, log_n, rate_bits, cap_height,
                                            leaves_out ? bl.as<u64>() : nullptr, (digests_out && nd) ? bd.as<u64>() : nullptr,
                                            bc.as<u64>()));
  if (leaves_out) P2MT_HIP(hipMemcpyAsync(leaves_out, bl.p, n_polys * big * 8, hipMemcpyDeviceToHost, st));
  if (digests_out && nd) P2MT_HIP(hipMemcpyAsync(digests_out, bd.p, nd * 32, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipMemcpyAsync(cap_out, bc.p, ncap * 32, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  return P2MT_OK;
  });
}
