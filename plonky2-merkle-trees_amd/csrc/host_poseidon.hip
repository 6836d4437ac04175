// host_poseidon.hip -- definitions of host_poseidon.h: the Poseidon permutation on a host core.  Host-only translation unit (no kernels):
// it reads the generated tables as plain host constants, which a unit that includes poseidon.hip.h (device __constant__ tables) cannot.
#include "host_poseidon.h"

#if !defined(__HIP_DEVICE_COMPILE__)  // host pass only: CPU feature builtins and target attributes mean nothing to the device pass

#include <stdlib.h>
#include <string.h>
#include <time.h>

#define P2MT_QUAL static const
#include "poseidon_constants.h"

#include "../../include/p2mt.h"

namespace host_poseidon {
namespace {

typedef uint32_t u32;
typedef unsigned __int128 u128;
constexpr u64 P = 0xFFFFFFFF00000001ull;
constexpr u64 EPS = 0xFFFFFFFFull;  // 2^64 mod p

#define HP_INLINE static inline __attribute__((always_inline))

HP_INLINE u64 canon(u64 x) { return x >= P ? x - P : x; }

// The arithmetic in two spellings of the same corrections -- BF: branch-free (carry / borrow masks), else compare-and-branch.  The
// corrections are data-dependent; which spelling a core likes depends on its predictor and its flag handling, so both are built and
// the dispatcher below times them once (a few microseconds) on the machine it runs on.
template <bool BF>
struct Ops {
  // a (any u64) + b (canonical) mod p; the result is any u64 congruent to the sum
  HP_INLINE u64 add_canon(u64 a, u64 b) {
    if (BF) {
      u64 s;
      const u64 carry = __builtin_add_overflow(a, b, &s);
      return s + (EPS & (0 - carry));  // wrapped: + 2^64 = + EPS; no second wrap since b < p
    }
    u64 s = a + b;
    if (s < a) s += EPS;
    return s;
  }
  // 128 -> 64: lo + hi * 2^64 mod p with hi = hh * 2^32 + hl:  2^64 = 2^32 - 1, 2^96 = -1
  HP_INLINE u64 reduce128(u64 lo, u64 hi) {
    const u64 hh = hi >> 32, hl = hi & EPS;
    const u64 m = (hl << 32) - hl;
    if (BF) {
      u64 t, r;
      const u64 borrow = __builtin_sub_overflow(lo, hh, &t);
      t -= EPS & (0 - borrow);  // borrow: - 2^64 = - EPS
      const u64 carry = __builtin_add_overflow(t, m, &r);
      return r + (EPS & (0 - carry));
    }
    u64 t = lo - hh;
    if (__builtin_expect(lo < hh, 0)) t -= EPS;
    u64 r = t + m;
    if (r < m) r += EPS;
    return r;
  }
  HP_INLINE u64 mul(u64 a, u64 b) {
    const u128 x = (u128)a * b;
    return reduce128((u64)x, (u64)(x >> 64));
  }
  HP_INLINE u64 pow7(u64 x) {
    const u64 x2 = mul(x, x), x3 = mul(x2, x), x4 = mul(x2, x2);
    return mul(x3, x4);
  }
  // value < 2^96 given as (lo, top 32 bits)
  HP_INLINE u64 reduce96(u64 lo, u32 top) {
    const u64 m = ((u64)top << 32) - top;
    if (BF) {
      u64 r;
      const u64 carry = __builtin_add_overflow(lo, m, &r);
      return r + (EPS & (0 - carry));
    }
    u64 r = lo + m;
    if (r < m) r += EPS;
    return r;
  }

  // MDS layer: out[r] = sum_c MDS[r][c] s[c], MDS[r][c] = CIRC[(c - r) mod 12] + (r == c == 0 ? 8 : 0): twelve 128-bit dot products
  // (mulx / adc chains), one 96-bit fold each
  HP_INLINE void mds(u64 (&s)[12]) {
    u64 o[12];
    for (int r = 0; r < 12; ++r) {
      u128 acc = (u128)s[r] * (POSEIDON_MDS_CIRC[0] + (r == 0 ? POSEIDON_MDS_DIAG[0] : 0));
      for (int k = 1; k < 12; ++k) acc += (u128)s[(r + k) % 12] * POSEIDON_MDS_CIRC[k];
      o[r] = reduce96((u64)acc, (u32)(acc >> 64));  // < 2^74
    }
    for (int r = 0; r < 12; ++r) s[r] = o[r];
  }

  // sum of up to 2^32 128-bit products: value = v + top 2^128
  struct Acc192 {
    u128 v = 0;
    u32 top = 0;
    inline __attribute__((always_inline)) void mac(u64 x, u64 y) {
      const u128 p = (u128)x * y;
      v += p;
      top += (u32)(v < p);
    }
    inline __attribute__((always_inline)) u64 reduce() const {  // 2^128 = -2^32 (mod p)
      const u64 r = reduce128((u64)v, (u64)(v >> 64));
      const u64 sub = (u64)top << 32;  // top small: canonical
      u64 t;
      const u64 borrow = __builtin_sub_overflow(r, sub, &t);
      return t - (EPS & (0 - borrow));
    }
  };

  // The 22 partial rounds in the sparse form (plonky2's partial_first_constant_layer / mds_partial_layer_init /
  // mds_partial_layer_fast; the tables of poseidon_constants.h, derived in tools/poseidon_spec.py): one S-box, one 12-term dot product
  // and eleven multiply-adds per round instead of a dense MDS layer -- on a CPU the dense layer is the cost of a round.
  HP_INLINE void partial_rounds_fast(u64 (&s)[12]) {
    for (int i = 0; i < 12; ++i) s[i] = add_canon(s[i], POSEIDON_FAST_FIRST[i]);
    {
      u64 t[11];
      for (int r = 0; r < 11; ++r) {
        Acc192 a;
        for (int c = 0; c < 11; ++c) a.mac(s[c + 1], POSEIDON_FAST_INIT[r * 11 + c]);
        t[r] = a.reduce();
      }
      for (int r = 0; r < 11; ++r) s[r + 1] = t[r];
    }
    for (int i = 0; i < POSEIDON_PARTIAL_ROUNDS; ++i) {
      const u64 s0 = add_canon(pow7(s[0]), POSEIDON_FAST_K[i]);
      Acc192 d;
      d.mac(s0, (u64)POSEIDON_M00);
      const u64* wh = POSEIDON_FAST_W_HAT + 11 * i;
      const u64* v = POSEIDON_FAST_V + 11 * i;
      for (int j = 0; j < 11; ++j) d.mac(s[j + 1], wh[j]);
      for (int j = 0; j < 11; ++j) {
        const u128 p = (u128)s0 * v[j] + s[j + 1];  // < 2^128
        s[j + 1] = reduce128((u64)p, (u64)(p >> 64));
      }
      s[0] = d.reduce();
    }
  }

  HP_INLINE void permute(u64 (&s)[12]) {
    int r = 0;
    for (; r < POSEIDON_HALF_FULL_ROUNDS; ++r) {
      for (int i = 0; i < 12; ++i) s[i] = pow7(add_canon(s[i], POSEIDON_RC[12 * r + i]));
      mds(s);
    }
    partial_rounds_fast(s);
    for (r = POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS; r < POSEIDON_ROUNDS; ++r) {
      for (int i = 0; i < 12; ++i) s[i] = pow7(add_canon(s[i], POSEIDON_RC[12 * r + i]));
      mds(s);
    }
    for (int i = 0; i < 12; ++i) s[i] = canon(s[i]);
  }
};

void permute_br(u64 (&s)[12]) { Ops<false>::permute(s); }
void permute_bf(u64 (&s)[12]) { Ops<true>::permute(s); }
__attribute__((target("bmi2"))) void permute_br_bmi2(u64 (&s)[12]) { Ops<false>::permute(s); }
__attribute__((target("bmi2"))) void permute_bf_bmi2(u64 (&s)[12]) { Ops<true>::permute(s); }

// ---- the eight FULL rounds on AVX-512, one permutation, the state across the lanes: words 0..7 in one zmm register, 8..11 in the low
// half of another (upper lanes stay zero).  A transcript is a chain of dependent permutations, so what counts is the latency of ONE;
// in the scalar code the full rounds are three quarters of it (48 field multiplications for the S-boxes and a 144-term MDS layer per
// round).  Here a round is: + constants, x^7 as four lane-wise 64 x 64 -> 128 multiplications (four vpmuludq each) with the usual
// two-correction reduction, and the circulant MDS layer as twelve lane rotations (vpermt2q) times small constants on 32-bit halves,
// folded once.  Values between steps are any u64 representative, as in the scalar code.  The 22 partial rounds stay scalar (one
// S-box and a dot product per round: nothing for eight lanes).  Used when the CPU has it and the timing below says it is faster.
#if defined(__x86_64__)
}  // namespace
}  // namespace host_poseidon
#include <immintrin.h>
namespace host_poseidon {
namespace {
#define HP_V512 __attribute__((target("avx512f,avx512dq,avx512vl,bmi2")))
#define HP_VINLINE static inline __attribute__((always_inline)) HP_V512
namespace v512 {
typedef __m512i V;
HP_VINLINE V eps() { return _mm512_set1_epi64(0xFFFFFFFFll); }
HP_VINLINE V shr32(V a) { return _mm512_srli_epi64(a, 32); }
HP_VINLINE V shl32(V a) { return _mm512_slli_epi64(a, 32); }
HP_VINLINE V add_canon(V a, V c) {  // a any u64, c canonical
  const V t = _mm512_add_epi64(a, c);
  return _mm512_mask_add_epi64(t, _mm512_cmplt_epu64_mask(t, a), t, eps());
}
HP_VINLINE V red128(V lo, V hi) {  // lo + hi 2^64 -> any u64 congruent (2^64 = 2^32 - 1, 2^96 = -1)
  const V hh = shr32(hi), hl = _mm512_and_si512(hi, eps());
  V t0 = _mm512_sub_epi64(lo, hh);
  t0 = _mm512_mask_sub_epi64(t0, _mm512_cmplt_epu64_mask(lo, hh), t0, eps());
  const V t1 = _mm512_sub_epi64(shl32(hl), hl);
  const V t2 = _mm512_add_epi64(t0, t1);
  return _mm512_mask_add_epi64(t2, _mm512_cmplt_epu64_mask(t2, t1), t2, eps());
}
HP_VINLINE V mul(V a, V b) {
  const V ah = shr32(a), bh = shr32(b);
  const V ll = _mm512_mul_epu32(a, b), lh = _mm512_mul_epu32(a, bh), hl = _mm512_mul_epu32(ah, b), hh = _mm512_mul_epu32(ah, bh);
  const V mid = _mm512_add_epi64(lh, shr32(ll));
  const V mid2 = _mm512_add_epi64(hl, _mm512_and_si512(mid, eps()));
  const V lo = _mm512_add_epi64(ll, shl32(_mm512_add_epi64(lh, hl)));
  const V hi = _mm512_add_epi64(hh, _mm512_add_epi64(shr32(mid), shr32(mid2)));
  return red128(lo, hi);
}
HP_VINLINE V pow7(V x) {
  const V x2 = mul(x, x), x3 = mul(x2, x), x4 = mul(x2, x2);
  return mul(x3, x4);
}
// lane i of output register A is word i, of B word 8 + i; source index for rotation j: word (w + j) mod 12, which is lane w of the
// (A, B) pair as vpermt2q numbers them (B's lane w - 8 has index w); B's upper output lanes read a zero lane (index 12)
struct Idx {
  alignas(64) long long a[12][8], b[12][8];
  constexpr Idx() : a(), b() {
    for (int j = 0; j < 12; ++j)
      for (int i = 0; i < 8; ++i) {
        a[j][i] = (i + j) % 12;
        b[j][i] = i < 4 ? (8 + i + j) % 12 : 12;
      }
  }
};
static const Idx kIdx;
// sum_j c_j * rot_j(state) on 32-bit halves (every term < 2^38, twelve of them and the diagonal < 2^42), then
// L + H 2^32 = L + hL 2^32 + hH (2^32 - 1) with H = hH 2^32 + hL
HP_VINLINE V fold(V L, V H) {
  const V hL = _mm512_and_si512(H, eps()), hH = shr32(H);
  const V x = shl32(hL), y = _mm512_add_epi64(L, _mm512_sub_epi64(shl32(hH), hH));
  const V t = _mm512_add_epi64(x, y);
  return _mm512_mask_add_epi64(t, _mm512_cmplt_epu64_mask(t, x), t, eps());
}
HP_VINLINE void mds(V& a, V& b) {
  V la = _mm512_setzero_si512(), ha = la, lb = la, hb = la;
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const V pa = j ? _mm512_permutex2var_epi64(a, _mm512_load_si512(kIdx.a[j]), b) : a;
    const V pb = j ? _mm512_permutex2var_epi64(a, _mm512_load_si512(kIdx.b[j]), b) : b;
    const V c = _mm512_set1_epi64((long long)POSEIDON_MDS_CIRC[j]);
    la = _mm512_add_epi64(la, _mm512_mul_epu32(pa, c));
    ha = _mm512_add_epi64(ha, _mm512_mul_epu32(shr32(pa), c));
    lb = _mm512_add_epi64(lb, _mm512_mul_epu32(pb, c));
    hb = _mm512_add_epi64(hb, _mm512_mul_epu32(shr32(pb), c));
  }
  const V d = _mm512_maskz_set1_epi64(1, (long long)POSEIDON_MDS_DIAG[0]);  // the diagonal is (8, 0, ..., 0)
  la = _mm512_add_epi64(la, _mm512_mul_epu32(a, d));
  ha = _mm512_add_epi64(ha, _mm512_mul_epu32(shr32(a), d));
  a = fold(la, ha);
  b = fold(lb, hb);
}
HP_VINLINE void full_rounds(u64 (&s)[12], int r0) {
  V a = _mm512_loadu_si512(s), b = _mm512_maskz_loadu_epi64(0x0F, s + 8);
  for (int r = r0; r < r0 + POSEIDON_HALF_FULL_ROUNDS; ++r) {
    a = pow7(add_canon(a, _mm512_loadu_si512(POSEIDON_RC + 12 * r)));
    b = pow7(add_canon(b, _mm512_maskz_loadu_epi64(0x0F, POSEIDON_RC + 12 * r + 8)));
    mds(a, b);
  }
  _mm512_storeu_si512(s, a);
  _mm512_mask_storeu_epi64(s + 8, 0x0F, b);
}
}  // namespace v512
template <bool BF>
HP_V512 void permute_v512(u64 (&s)[12]) {
  v512::full_rounds(s, 0);
  Ops<BF>::partial_rounds_fast(s);
  v512::full_rounds(s, POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS);
  for (int i = 0; i < 12; ++i) s[i] = canon(s[i]);
}
#endif

typedef void (*PermuteFn)(u64 (&)[12]);
double time_chain(PermuteFn fn) {  // seconds per permutation of a dependent chain (what a transcript is)
  u64 s[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};
  double best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int i = 0; i < 16; ++i) fn(s);
    clock_gettime(CLOCK_MONOTONIC, &b);
    const double t = ((b.tv_sec - a.tv_sec) * 1e9 + (b.tv_nsec - a.tv_nsec)) * 1e-9 / 16;
    best = t < best ? t : best;
  }
  return best;
}
PermuteFn pick() {
  __builtin_cpu_init();
  const bool bmi2 = __builtin_cpu_supports("bmi2");
  PermuteFn cand[4] = {bmi2 ? permute_br_bmi2 : permute_br, bmi2 ? permute_bf_bmi2 : permute_bf, nullptr, nullptr};
#if defined(__x86_64__)
  if (bmi2 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl")) {
    cand[2] = permute_v512<false>;
    cand[3] = permute_v512<true>;
  }
#endif
  if (const char* e = getenv("P2MT_HOST_POSEIDON")) {  // "br" / "bf" / "v512br" / "v512bf": pin a variant (A/B, tests)
    if (!strcmp(e, "br")) return cand[0];
    if (!strcmp(e, "bf")) return cand[1];
    if (!strcmp(e, "v512br") && cand[2]) return cand[2];
    if (!strcmp(e, "v512bf") && cand[3]) return cand[3];
  }
  PermuteFn best = cand[0];
  double t_best = time_chain(cand[0]);
  for (int k = 1; k < 4; ++k) {
    if (!cand[k]) continue;
    const double t = time_chain(cand[k]);
    if (t < t_best) t_best = t, best = cand[k];
  }
  return best;
}

}  // namespace

void permute(u64 (&s)[12]) {
  static const PermuteFn fn = pick();
  fn(s);
}

void hash_no_pad(const u64* in, size_t n, u64 (&out)[4]) {
  u64 s[12] = {0};
  for (size_t off = 0; off < n; off += 8) {
    for (size_t k = 0; k < 8 && off + k < n; ++k) s[k] = canon(in[off + k]);
    permute(s);
  }
  for (int k = 0; k < 4; ++k) out[k] = s[k];
}

void Challenger::duplex() {
  for (unsigned k = 0; k < n_in; ++k) state[k] = in[k];
  n_in = 0;
  permute(state);
  for (int k = 0; k < 8; ++k) out[k] = state[k];
  n_out = 8;
}
void Challenger::observe(u64 x) {
  n_out = 0;  // buffered outputs are stale
  in[n_in++] = canon(x);
  if (n_in == 8) duplex();
}
u64 Challenger::squeeze() {
  if (n_in != 0 || n_out == 0) duplex();
  return out[--n_out];
}

}  // namespace host_poseidon

// The host permutation through the C ABI: n states of 12 words, in -> out (may alias).  No GPU involved; what a caller that keeps its
// own transcript on the host (as plonky2 does) links instead of a per-hash device round trip.
extern "C" int p2mt_host_poseidon_permute(const uint64_t* in, uint64_t* out, size_t n) {
  if (n && (!in || !out)) return P2MT_EINVAL;
  for (size_t i = 0; i < n; ++i) {
    uint64_t s[12];
    memcpy(s, in + 12 * i, 96);
    host_poseidon::permute(s);
    memcpy(out + 12 * i, s, 96);
  }
  return P2MT_OK;
}

// Test hook: the host Challenger through the C ABI -- phase k observes n_obs[k] elements (consecutive in `elements`), then squeezes
// n_sq[k] challenges (consecutive in `out`).  tests/test_host_transcript.py compares it with the oracle's transcript.
extern "C" int p2mt_debug_host_challenger(const uint64_t* elements, const uint32_t* n_obs, const uint32_t* n_sq, size_t n_phases,
                                          uint64_t* out) {
  if (n_phases && (!n_obs || !n_sq)) return P2MT_EINVAL;
  host_poseidon::Challenger ch;
  for (size_t k = 0; k < n_phases; ++k) {
    if ((n_obs[k] && !elements) || (n_sq[k] && !out)) return P2MT_EINVAL;
    ch.observe(elements, n_obs[k]);
    elements += n_obs[k];
    ch.squeeze(out, n_sq[k]);
    out += n_sq[k];
  }
  return P2MT_OK;
}
#endif  // !__HIP_DEVICE_COMPILE__
