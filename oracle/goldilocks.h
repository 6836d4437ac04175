/* oracle/goldilocks.h -- TEST INFRASTRUCTURE. Canonical-form Goldilocks arithmetic, p = 2^64 - 2^32 + 1.
 * Restates plonky2_field 0.1.0 goldilocks_field.rs semantics (values always reduced to [0,p) here,
 * so equality is plain u64 equality). Field order: /root/reference/src/mmr/common.rs:3. */
#ifndef ORACLE_GOLDILOCKS_H
#define ORACLE_GOLDILOCKS_H
#include <stdint.h>

#define GL_P 0xFFFFFFFF00000001ULL
typedef unsigned __int128 u128;

static inline uint64_t gl_canon(uint64_t a) { return a >= GL_P ? a - GL_P : a; }

static inline uint64_t gl_add(uint64_t a, uint64_t b) { /* a,b < p */
  uint64_t s = a + b;
  if (s < a || s >= GL_P) s -= GL_P;
  return s;
}

static inline uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }

/* 2^64 = 2^32 - 1 and 2^96 = -1 (mod p): x = lo + hl*2^64 + hh*2^96 = lo - hh + hl*(2^32-1). */
static inline uint64_t gl_reduce128(u128 x) {
  uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
  uint64_t t0 = lo - hh;
  if (lo < hh) t0 -= 0xFFFFFFFFULL; /* wrapped: +2^64 too much == +(2^32-1) */
  uint64_t t1 = hl * 0xFFFFFFFFULL;
  uint64_t t2 = t0 + t1;
  if (t2 < t1) t2 += 0xFFFFFFFFULL; /* wrapped: lost 2^64 == 2^32-1 */
  return gl_canon(t2);
}

static inline uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_reduce128((u128)a * b); }

static inline uint64_t gl_pow(uint64_t a, uint64_t e) {
  uint64_t r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, a);
    a = gl_mul(a, a);
    e >>= 1;
  }
  return r;
}

static inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }

/* plonky2_field: MULTIPLICATIVE_GROUP_GENERATOR = 7, TWO_ADICITY = 32,
 * POWER_OF_TWO_GENERATOR = 7^((p-1)/2^32) = 1753635133440165772. */
static inline uint64_t gl_primitive_root_of_unity(unsigned log_n) {
  uint64_t g = gl_pow(7, (GL_P - 1) >> 32);
  for (unsigned i = log_n; i < 32; ++i) g = gl_mul(g, g);
  return g;
}
#endif
