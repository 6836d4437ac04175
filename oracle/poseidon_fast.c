/* oracle/poseidon_fast.c -- TEST INFRASTRUCTURE: the tuned scalar CPU port used as bench.py's `cpu_baseline.port_fast`.
 *
 * Same function as oracle/poseidon.c (plonky2 @3b21b87 hash/poseidon.rs Poseidon::poseidon, absent from /root/reference;
 * call sites /root/reference/src/mmr/merkle_mountain_ranges.rs:91,96,111 and simple_merkle_tree.rs:23,33,45), written
 * the way a CPU implementer would write it instead of the way the specification reads:
 *   - sparse ("fast") partial rounds: 1 S-box + an 11-term dot product + 11 multiply-adds instead of a 12x12 MDS layer
 *     (constants derived in tools/poseidon_spec.py::fast_partial_constants, plonky2's poseidon_goldilocks.rs tables);
 *   - lazy reduction: state words are arbitrary u64 (not canonical) between steps, the MDS layer accumulates 32-bit halves
 *     in u64 (13 terms x 2^6 x 2^32 < 2^42) and reduces once per output word, the dot product accumulates 128-bit
 *     products in two u128 halves and reduces once, `s + s0 * v` reduces once per word;
 *   - W independent permutations interleaved per call (the partial rounds are a latency chain on one hash);
 *   - built with -O3 -march=native on the machine that runs it (bench.py compiles this file into a temp dir).
 * It is bit-identical to oracle/poseidon.c (tests/test_oracle_golden.py::test_fast_port_equals_spec_form); it never runs in
 * the product.  The add_leaf loop below is the reference's algorithm (merkle_mountain_ranges.rs:89-120) on this permutation.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "goldilocks.h"
#include "poseidon_constants.h"
#include "poseidon_fast_constants.h"

#define EPS 0xFFFFFFFFULL
#ifndef PF_W
#define PF_W 2 /* permutations interleaved by the batch entry points (2 measured best on x86-64: 16 GPRs) */
#endif

/* any u128 -> u64 congruent mod p (not canonical) */
static inline uint64_t red128(u128 x) {
  const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  const uint64_t hh = hi >> 32, hl = hi & EPS;
  /* branch-free: both corrections are data-dependent coin flips, a mispredicted branch costs more than the whole reduce */
  uint64_t t0, t2;
  const uint64_t borrow = __builtin_sub_overflow(lo, hh, &t0);
  t0 -= EPS & (0 - borrow);
  const uint64_t t1 = (hl << 32) - hl;
  const uint64_t carry = __builtin_add_overflow(t0, t1, &t2);
  return t2 + (EPS & (0 - carry));
}
static inline uint64_t mulr(uint64_t a, uint64_t b) { return red128((u128)a * b); }
/* a + b for arbitrary u64 a and canonical b: congruent u64 */
static inline uint64_t addc(uint64_t a, uint64_t b) {
  uint64_t s;
  const uint64_t carry = __builtin_add_overflow(a, b, &s);
  return s + (EPS & (0 - carry)); /* lost 2^64 == 2^32 - 1; cannot wrap again because b < p */
}
static inline uint64_t sbox7(uint64_t x) {
  const uint64_t x2 = mulr(x, x), x3 = mulr(x2, x), x4 = mulr(x2, x2);
  return mulr(x3, x4);
}

/* MDS layer on W interleaved states s[word][k]; 32-bit halves accumulated in u64, one 96-bit fold per output word */
#define MDS_LAYER(W, s)                                                                  \
  do {                                                                                   \
    uint64_t lo_[12][W], hi_[12][W];                                                     \
    for (int i_ = 0; i_ < 12; ++i_)                                                      \
      for (int k_ = 0; k_ < W; ++k_) lo_[i_][k_] = s[i_][k_] & EPS, hi_[i_][k_] = s[i_][k_] >> 32; \
    for (int r_ = 0; r_ < 12; ++r_)                                                      \
      for (int k_ = 0; k_ < W; ++k_) {                                                   \
        uint64_t al_ = lo_[r_][k_] * POSEIDON_MDS_DIAG[r_], ah_ = hi_[r_][k_] * POSEIDON_MDS_DIAG[r_]; \
        for (int i_ = 0; i_ < 12; ++i_) {                                                \
          const int c_ = i_ + r_ >= 12 ? i_ + r_ - 12 : i_ + r_;                         \
          al_ += lo_[c_][k_] * POSEIDON_MDS_CIRC[i_];                                    \
          ah_ += hi_[c_][k_] * POSEIDON_MDS_CIRC[i_];                                    \
        }                                                                                \
        s[r_][k_] = red128((u128)al_ + ((u128)ah_ << 32));                               \
      }                                                                                  \
  } while (0)

#define DEFINE_PERMUTE(W)                                                                                       \
  static void permute_w##W(uint64_t s[12][W]) {                                                                 \
    int r = 0;                                                                                                  \
    for (; r < 4; ++r) {                                                                                        \
      for (int i = 0; i < 12; ++i)                                                                              \
        for (int k = 0; k < W; ++k) s[i][k] = sbox7(addc(s[i][k], POSEIDON_RC[12 * r + i]));                    \
      MDS_LAYER(W, s);                                                                                          \
    }                                                                                                           \
    /* partial rounds, sparse form: s += first; s[1..] = init * s[1..] */                                       \
    for (int i = 0; i < 12; ++i)                                                                                \
      for (int k = 0; k < W; ++k) s[i][k] = addc(s[i][k], POSEIDON_FAST_FIRST[i]);                              \
    {                                                                                                           \
      uint64_t t[11][W];                                                                                        \
      for (int rr = 0; rr < 11; ++rr)                                                                           \
        for (int k = 0; k < W; ++k) {                                                                           \
          u128 lo = 0, hi = 0;                                                                                  \
          for (int c = 0; c < 11; ++c) {                                                                        \
            const u128 p = (u128)POSEIDON_FAST_INIT[11 * rr + c] * s[c + 1][k];                                 \
            lo += (uint64_t)p;                                                                                  \
            hi += (uint64_t)(p >> 64);                                                                          \
          }                                                                                                     \
          t[rr][k] = red128(lo + (u128)red128(hi) * EPS);                                                       \
        }                                                                                                       \
      for (int rr = 0; rr < 11; ++rr)                                                                           \
        for (int k = 0; k < W; ++k) s[rr + 1][k] = t[rr][k];                                                    \
    }                                                                                                           \
    for (int pr = 0; pr < 22; ++pr) {                                                                           \
      const uint64_t *wh = POSEIDON_FAST_W_HAT + 11 * pr, *v = POSEIDON_FAST_V + 11 * pr;                       \
      for (int k = 0; k < W; ++k) {                                                                             \
        const uint64_t s0 = addc(sbox7(s[0][k]), POSEIDON_FAST_K[pr]);                                          \
        u128 lo = (u128)s0 * POSEIDON_M00, hi = 0;                                                              \
        for (int j = 0; j < 11; ++j) {                                                                          \
          const u128 p = (u128)wh[j] * s[j + 1][k];                                                             \
          lo += (uint64_t)p;                                                                                    \
          hi += (uint64_t)(p >> 64);                                                                            \
        }                                                                                                       \
        for (int j = 0; j < 11; ++j) s[j + 1][k] = red128((u128)s0 * v[j] + s[j + 1][k]);                       \
        s[0][k] = red128(lo + (u128)red128(hi) * EPS);                                                          \
      }                                                                                                         \
    }                                                                                                           \
    for (r = 26; r < 30; ++r) {                                                                                 \
      for (int i = 0; i < 12; ++i)                                                                              \
        for (int k = 0; k < W; ++k) s[i][k] = sbox7(addc(s[i][k], POSEIDON_RC[12 * r + i]));                    \
      MDS_LAYER(W, s);                                                                                          \
    }                                                                                                           \
    for (int i = 0; i < 12; ++i)                                                                                \
      for (int k = 0; k < W; ++k) s[i][k] = gl_canon(s[i][k]);                                                  \
  }

#define DEFINE_PERMUTE_X(W) DEFINE_PERMUTE(W)
DEFINE_PERMUTE(1)
#if PF_W != 1
DEFINE_PERMUTE_X(PF_W)
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define permute_batch CAT(permute_w, PF_W)

void oracle_fast_poseidon_permute(uint64_t st[12]) {
  uint64_t s[12][1];
  for (int i = 0; i < 12; ++i) s[i][0] = st[i];
  permute_w1(s);
  for (int i = 0; i < 12; ++i) st[i] = s[i][0];
}

static inline void two_to_one_1(const uint64_t *l, const uint64_t *r, uint64_t *out) {
  uint64_t s[12][1];
  for (int i = 0; i < 4; ++i) s[i][0] = l[i], s[i + 4][0] = r[i], s[i + 8][0] = 0;
  permute_w1(s);
  for (int i = 0; i < 4; ++i) out[i] = s[i][0];
}

/* n independent two_to_one: in[n][8] = left | right, out[n][4]; PF_W at a time */
void oracle_fast_two_to_one_batch(const uint64_t *in, uint64_t *out, size_t n) {
  size_t j = 0;
  for (; j + PF_W <= n; j += PF_W) {
    uint64_t s[12][PF_W];
    for (int k = 0; k < PF_W; ++k)
      for (int i = 0; i < 8; ++i) s[i][k] = in[8 * (j + k) + i];
    for (int k = 0; k < PF_W; ++k)
      for (int i = 8; i < 12; ++i) s[i][k] = 0;
    permute_batch(s);
    for (int k = 0; k < PF_W; ++k)
      for (int i = 0; i < 4; ++i) out[4 * (j + k) + i] = s[i][k];
  }
  for (; j < n; ++j) two_to_one_1(in + 8 * j, in + 8 * j + 4, out + 4 * j);
}

/* B1': the reference's own loop -- `for leaf { mmr.add_leaf(leaf) }` (merkle_mountain_ranges.rs:89-120) -- on this
 * permutation, single thread, into the post-order array el[(2n - popcount n)][4].  One leaf at a time: the carry chain of an
 * add_leaf is sequential, so nothing is interleaved here. */
void oracle_fast_mmr_add_leaf_loop(const uint64_t *leaves, size_t n, uint64_t *el) {
  size_t len = 0;
  for (size_t i = 0; i < n; ++i) {
    uint64_t *cur = el + 4 * len;
    cur[0] = gl_canon(leaves[i]), cur[1] = cur[2] = cur[3] = 0; /* hash_or_noop(&[leaf]) */
    ++len;
    /* merge while the new count has trailing carries: leaf i completes a height-h node for every trailing 1 bit of i */
    unsigned h = 0;
    for (size_t m = i; m & 1; m >>= 1, ++h) {
      const uint64_t *right = el + 4 * (len - 1), *left = el + 4 * (len - ((size_t)2 << h));
      two_to_one_1(left, right, el + 4 * len);
      ++len;
    }
  }
}

/* Level-order build of the perfect 2^k-leaf MMR into the same post-order array, PF_W hashes interleaved, `threads` OpenMP
 * threads (1 = the best single-core rate this port reaches; > 1 = the "generous" all-core baseline).  Not the reference's
 * algorithm (it is single-threaded and leaf-at-a-time), same values. */
int oracle_fast_mmr_build_pow2(const uint64_t *leaves, size_t n, uint64_t *el, int threads) {
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
    for (long long j0 = 0; j0 < cnt; j0 += PF_W) {
      uint64_t s[12][PF_W];
      size_t pos[PF_W];
      for (int k = 0; k < PF_W; ++k) {
        const long long j = j0 + k < cnt ? j0 + k : cnt - 1;
        const size_t last = (((size_t)j + 1) << h) - 1;
        pos[k] = 2 * last - (size_t)__builtin_popcountll((unsigned long long)last) + h;
        const uint64_t *l = el + 4 * (pos[k] - ((size_t)1 << h)), *r = el + 4 * (pos[k] - 1);
        for (int i = 0; i < 4; ++i) s[i][k] = l[i], s[i + 4][k] = r[i], s[i + 8][k] = 0;
      }
      permute_batch(s);
      for (int k = 0; k < PF_W && j0 + k < cnt; ++k)
        for (int i = 0; i < 4; ++i) el[4 * pos[k] + i] = s[i][k];
    }
  }
  return threads;
}
