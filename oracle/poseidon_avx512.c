/* oracle/poseidon_avx512.c -- TEST INFRASTRUCTURE: the AVX-512 CPU port used as bench.py's `cpu_baseline.port_fast` when the
 * host has AVX-512 (the GPU boxes' EPYC 9575F does).  Never runs in the product.
 *
 * Same function as oracle/poseidon.c (plonky2 @3b21b87 hash/poseidon.rs Poseidon::poseidon, absent from /root/reference; call
 * sites /root/reference/src/mmr/merkle_mountain_ranges.rs:91,96,111 and simple_merkle_tree.rs:23,33,45), eight independent
 * permutations per call, one per 64-bit lane of a zmm register (structure of arrays: V s[12], s[w] = word w of 8 states):
 *   - 64x64 -> 128 products from four vpmuludq (32x32 -> 64) partial products; reduction mod p = 2^64 - 2^32 + 1 with the
 *     usual two conditional corrections as masked add/sub (no branches); state words are arbitrary u64 between steps;
 *   - full rounds: S-box x^7 = 2 squarings + 2 products, MDS layer on 32-bit halves (13 small-constant products per half and
 *     output word, one 96-bit fold per output word) exactly like oracle/poseidon_fast.c;
 *   - partial rounds in the sparse form (constants of poseidon_fast_constants.h): the 12-term dot product accumulates the four
 *     32x32 partial products of every term in separate lanes-wide accumulators and is recombined and reduced once.
 * The MMR build below is level order (the reference's add_leaf loop is one hash at a time and cannot feed eight lanes); values
 * are bit-identical to oracle/poseidon.c (tests/test_oracle_golden.py::test_avx512_port_equals_spec_form). */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "goldilocks.h"
#include "poseidon_constants.h"
#include "poseidon_fast_constants.h"

#if defined(__AVX512F__) && defined(__AVX512DQ__)
#include <immintrin.h>

typedef __m512i V;
#define EPSV _mm512_set1_epi64(0xFFFFFFFFLL)
#define C64(x) _mm512_set1_epi64((long long)(x))

static inline V v_and(V a, V b) { return _mm512_and_si512(a, b); }
static inline V v_add(V a, V b) { return _mm512_add_epi64(a, b); }
static inline V v_sub(V a, V b) { return _mm512_sub_epi64(a, b); }
static inline V v_shr32(V a) { return _mm512_srli_epi64(a, 32); }
static inline V v_shl32(V a) { return _mm512_slli_epi64(a, 32); }
static inline V v_mul32(V a, V b) { return _mm512_mul_epu32(a, b); } /* low dwords */

/* (lo, hi) any 128-bit value -> u64 congruent mod p (not canonical) */
static inline V v_red128(V lo, V hi) {
  const V eps = EPSV;
  const V hh = v_shr32(hi), hl = v_and(hi, eps);
  V t0 = v_sub(lo, hh);
  t0 = _mm512_mask_sub_epi64(t0, _mm512_cmplt_epu64_mask(lo, hh), t0, eps);
  const V t1 = v_sub(v_shl32(hl), hl);
  V t2 = v_add(t0, t1);
  t2 = _mm512_mask_add_epi64(t2, _mm512_cmplt_epu64_mask(t2, t1), t2, eps);
  return t2;
}
/* a * b as (lo, hi); bh = b >> 32 supplied by the caller */
static inline void v_mul128(V a, V b, V bh, V *lo, V *hi) {
  const V eps = EPSV;
  const V ah = v_shr32(a);
  const V ll = v_mul32(a, b), lh = v_mul32(a, bh), hl = v_mul32(ah, b), hh = v_mul32(ah, bh);
  const V mid = v_add(lh, v_shr32(ll));
  const V mid2 = v_add(hl, v_and(mid, eps));
  *lo = v_add(ll, v_shl32(v_add(lh, hl)));
  *hi = v_add(hh, v_add(v_shr32(mid), v_shr32(mid2)));
}
static inline V v_mulr2(V a, V b, V bh) {
  V lo, hi;
  v_mul128(a, b, bh, &lo, &hi);
  return v_red128(lo, hi);
}
/* a * b + c mod p, c any u64: the addend joins the 128-bit product before the one reduction (a b + c < 2^128) */
static inline V v_mul_add_r(V a, V b, V bh, V c) {
  V lo, hi;
  v_mul128(a, b, bh, &lo, &hi);
  const V l = v_add(lo, c);
  const V h = _mm512_mask_add_epi64(hi, _mm512_cmplt_epu64_mask(l, c), hi, C64(1));
  return v_red128(l, h);
}
static inline V v_mulr(V a, V b) { return v_mulr2(a, b, v_shr32(b)); }
static inline V v_sqr(V a) {
  const V eps = EPSV;
  const V ah = v_shr32(a);
  const V ll = v_mul32(a, a), lh = v_mul32(a, ah), hh = v_mul32(ah, ah);
  const V mid = v_add(lh, v_shr32(ll));
  const V mid2 = v_add(lh, v_and(mid, eps));
  const V lo = v_add(ll, v_shl32(v_add(lh, lh)));
  const V hi = v_add(hh, v_add(v_shr32(mid), v_shr32(mid2)));
  return v_red128(lo, hi);
}
/* a + c for arbitrary u64 a and canonical constant c */
static inline V v_addc(V a, uint64_t c) {
  const V cv = C64(c);
  const V s = v_add(a, cv);
  return _mm512_mask_add_epi64(s, _mm512_cmplt_epu64_mask(s, cv), s, EPSV);
}
static inline V v_sbox7(V x) {
  const V x2 = v_sqr(x), x3 = v_mulr(x2, x), x4 = v_sqr(x2);
  return v_mulr(x3, x4);
}
static inline V v_canon(V a) {
  const V p = C64(0xFFFFFFFF00000001ULL);
  return _mm512_mask_sub_epi64(a, _mm512_cmpge_epu64_mask(a, p), a, p);
}

static inline void mds_layer(V s[12]) {
  V lo[12], hi[12], out[12];
  const V eps = EPSV;
  for (int i = 0; i < 12; ++i) lo[i] = v_and(s[i], eps), hi[i] = v_shr32(s[i]);
  for (int r = 0; r < 12; ++r) {
    const V d = C64(POSEIDON_MDS_DIAG[r]);
    V al = v_mul32(lo[r], d), ah = v_mul32(hi[r], d);
    for (int i = 0; i < 12; ++i) {
      const int c = i + r >= 12 ? i + r - 12 : i + r;
      const V k = C64(POSEIDON_MDS_CIRC[i]);
      al = v_add(al, v_mul32(lo[c], k));
      ah = v_add(ah, v_mul32(hi[c], k));
    }
    /* al + (ah << 32) as a 128-bit value: al, ah < 2^42 */
    const V sh = v_shl32(ah);
    const V l = v_add(al, sh);
    const V h = _mm512_mask_add_epi64(v_shr32(ah), _mm512_cmplt_epu64_mask(l, sh), v_shr32(ah), C64(1));
    out[r] = v_red128(l, h);
  }
  for (int i = 0; i < 12; ++i) s[i] = out[i];
}

/* sum_j a_j * k_j for n terms as one reduced word: the four 32x32 partial products of every term go to their own accumulators
 * (n <= 12: every accumulator stays below 2^68 -- so the high halves of the partial products are accumulated separately) */
typedef struct {
  V ll_lo, ll_hi, lh_lo, lh_hi, hl_lo, hl_hi, hh_lo, hh_hi;
} Dot;
static inline void dot_init(Dot *d) {
  const V z = _mm512_setzero_si512();
  d->ll_lo = d->ll_hi = d->lh_lo = d->lh_hi = d->hl_lo = d->hl_hi = d->hh_lo = d->hh_hi = z;
}
static inline void dot_acc(Dot *d, V a, uint64_t k) {
  const V eps = EPSV;
  const V kl = C64(k & 0xFFFFFFFFULL), kh = C64(k >> 32);
  const V ah = v_shr32(a);
  const V ll = v_mul32(a, kl), lh = v_mul32(a, kh), hl = v_mul32(ah, kl), hh = v_mul32(ah, kh);
  d->ll_lo = v_add(d->ll_lo, v_and(ll, eps)), d->ll_hi = v_add(d->ll_hi, v_shr32(ll));
  d->lh_lo = v_add(d->lh_lo, v_and(lh, eps)), d->lh_hi = v_add(d->lh_hi, v_shr32(lh));
  d->hl_lo = v_add(d->hl_lo, v_and(hl, eps)), d->hl_hi = v_add(d->hl_hi, v_shr32(hl));
  d->hh_lo = v_add(d->hh_lo, v_and(hh, eps)), d->hh_hi = v_add(d->hh_hi, v_shr32(hh));
}
/* value = ll + (lh + hl) 2^32 + hh 2^64 with xx = xx_lo + xx_hi 2^32, every part < 2^36:
 *   = w0 + w1 2^32 + w2 2^64 + w3 2^96,  w0 = ll_lo, w1 = ll_hi + lh_lo + hl_lo, w2 = lh_hi + hl_hi + hh_lo, w3 = hh_hi
 * 2^64 = 2^32 - 1 and 2^96 = -1 (mod p): value = w0 + w1 2^32 + w2 (2^32 - 1) - w3, every w < 2^38. */
static inline V dot_reduce(const Dot *d) {
  const V w0 = d->ll_lo, w1 = v_add(d->ll_hi, v_add(d->lh_lo, d->hl_lo)), w2 = v_add(d->lh_hi, v_add(d->hl_hi, d->hh_lo)), w3 = d->hh_hi;
  /* t = w0 + (w1 + w2) 2^32 as 128 bits, then subtract (w2 + w3) (< 2^39) */
  const V m = v_add(w1, w2);           /* < 2^39 */
  const V sh = v_shl32(m);
  const V l = v_add(w0, sh);
  const V h = _mm512_mask_add_epi64(v_shr32(m), _mm512_cmplt_epu64_mask(l, sh), v_shr32(m), C64(1));
  const V r = v_red128(l, h);          /* any u64 congruent */
  const V sub = v_add(w2, w3);         /* canonical (< 2^39 < p) */
  /* r - sub mod p on a non-canonical r: borrow => add p (= subtract EPS after the wrap) */
  const V t = v_sub(r, sub);
  return _mm512_mask_sub_epi64(t, _mm512_cmplt_epu64_mask(r, sub), t, EPSV);
}

static void permute_x8(V s[12]) {
  for (int r = 0; r < 4; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = v_sbox7(v_addc(s[i], POSEIDON_RC[12 * r + i]));
    mds_layer(s);
  }
  for (int i = 0; i < 12; ++i) s[i] = v_addc(s[i], POSEIDON_FAST_FIRST[i]);
  {
    V t[11];
    for (int rr = 0; rr < 11; ++rr) {
      Dot d;
      dot_init(&d);
      for (int c = 0; c < 11; ++c) dot_acc(&d, s[c + 1], POSEIDON_FAST_INIT[11 * rr + c]);
      t[rr] = dot_reduce(&d);
    }
    for (int rr = 0; rr < 11; ++rr) s[rr + 1] = t[rr];
  }
  for (int pr = 0; pr < 22; ++pr) {
    const uint64_t *wh = POSEIDON_FAST_W_HAT + 11 * pr, *v = POSEIDON_FAST_V + 11 * pr;
    const V s0 = v_addc(v_sbox7(s[0]), POSEIDON_FAST_K[pr]);
    Dot d;
    dot_init(&d);
    dot_acc(&d, s0, POSEIDON_M00);
    for (int j = 0; j < 11; ++j) dot_acc(&d, s[j + 1], wh[j]);
    const V s0h = v_shr32(s0);
    for (int j = 0; j < 11; ++j) s[j + 1] = v_mul_add_r(C64(v[j]), s0, s0h, s[j + 1]);
    s[0] = dot_reduce(&d);
  }
  for (int r = 26; r < 30; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = v_sbox7(v_addc(s[i], POSEIDON_RC[12 * r + i]));
    mds_layer(s);
  }
  for (int i = 0; i < 12; ++i) s[i] = v_canon(s[i]);
}

int oracle_avx512_available(void) { return 1; }

/* eight two_to_one at once: pointers to the 8 left and 8 right children (4 words each), results to out[k] */
static inline void two_to_one_x8(const uint64_t *const l[8], const uint64_t *const r[8], uint64_t *const out[8]) {
  V s[12];
  uint64_t tmp[8] __attribute__((aligned(64)));
  for (int w = 0; w < 4; ++w) {
    for (int k = 0; k < 8; ++k) tmp[k] = l[k][w];
    s[w] = _mm512_load_si512((const void *)tmp);
    for (int k = 0; k < 8; ++k) tmp[k] = r[k][w];
    s[w + 4] = _mm512_load_si512((const void *)tmp);
    s[w + 8] = _mm512_setzero_si512();
  }
  permute_x8(s);
  for (int w = 0; w < 4; ++w) {
    _mm512_store_si512((void *)tmp, s[w]);
    for (int k = 0; k < 8; ++k) out[k][w] = tmp[k];
  }
}

void oracle_avx512_permute_batch(const uint64_t *in, uint64_t *out, size_t n) {
  uint64_t tmp[8] __attribute__((aligned(64)));
  for (size_t j = 0; j < n; j += 8) {
    V s[12];
    for (int w = 0; w < 12; ++w) {
      for (int k = 0; k < 8; ++k) tmp[k] = in[12 * (j + k < n ? j + k : n - 1) + w];
      s[w] = _mm512_load_si512((const void *)tmp);
    }
    permute_x8(s);
    for (int w = 0; w < 12; ++w) {
      _mm512_store_si512((void *)tmp, s[w]);
      for (int k = 0; k < 8 && j + k < n; ++k) out[12 * (j + k) + w] = tmp[k];
    }
  }
}

/* Level-order build of the perfect 2^k-leaf MMR into the post-order array el[(2n - 1)][4] (oracle_mmr_build_pow2's layout),
 * eight hashes per permutation call, `threads` OpenMP threads (1 = the single-core rate). */
int oracle_avx512_mmr_build_pow2(const uint64_t *leaves, size_t n, uint64_t *el, int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
  threads = omp_get_max_threads();
#else
  threads = 1;
#endif
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)n; ++i) {
    uint64_t *d = el + 4 * (2 * (size_t)i - (size_t)__builtin_popcountll((unsigned long long)i));
    d[0] = gl_canon(leaves[i]), d[1] = d[2] = d[3] = 0;
  }
  for (unsigned h = 1; ((size_t)1 << h) <= n; ++h) {
    const long long cnt = (long long)(n >> h);
#pragma omp parallel for schedule(static)
    for (long long j0 = 0; j0 < cnt; j0 += 8) {
      const uint64_t *l[8], *r[8];
      uint64_t *o[8];
      uint64_t spill[8][4];
      for (int k = 0; k < 8; ++k) {
        const long long j = j0 + k < cnt ? j0 + k : cnt - 1;
        const size_t last = (((size_t)j + 1) << h) - 1;
        const size_t pos = 2 * last - (size_t)__builtin_popcountll((unsigned long long)last) + h;
        l[k] = el + 4 * (pos - ((size_t)1 << h)), r[k] = el + 4 * (pos - 1);
        o[k] = j0 + k < cnt ? el + 4 * pos : spill[k];
      }
      two_to_one_x8(l, r, o);
    }
  }
  return threads;
}

#else /* no AVX-512 on this build host */
int oracle_avx512_available(void) { return 0; }
void oracle_avx512_permute_batch(const uint64_t *in, uint64_t *out, size_t n) { (void)in, (void)out, (void)n; }
int oracle_avx512_mmr_build_pow2(const uint64_t *leaves, size_t n, uint64_t *el, int threads) {
  (void)leaves, (void)n, (void)el, (void)threads;
  return -1;
}
#endif
