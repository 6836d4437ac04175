/* oracle/plonk.c -- TEST INFRASTRUCTURE.  [parity unpinned]
 *
 * CPU restatement of the permutation-argument stage of CircuitData::prove (reference call sites
 * /root/reference/src/mmr/mmr_plonky2_verifier.rs:148, mmr_plonky2_verifier_1_recursion.rs:192,218).  The code lives in
 * plonky2 (git rev 3b21b87d, NOT in /root/reference): plonk/prover.rs all_wires_permutation_partial_products ->
 * wires_permutation_partial_products_and_zs, plonk/permutation_argument / plonk_common.rs quotient_chunk_products and
 * partial_products_and_z_gx, restated from their published algorithm ("partial products + Z: per
 * challenge 1 Z + 9 partial products (80 routed wires in chunks of 8) => 20 polys").
 * The reference holds no vector for it; the tests check it through the permutation argument itself: with sigma a
 * permutation that only moves positions holding equal wire values, the grand product closes (Z(g x_last) = 1).
 */
#include <stdlib.h>
#include <string.h>

#include "goldilocks.h"
#include "oracle.h"

/* out: [num_challenges] Z columns, then [num_challenges][num_prods] partial-product columns, each of n values
 * (the order PolynomialBatch::from_values receives them in: "Z is expected at the front of our batch"). */
int oracle_permutation_partial_products(const uint64_t *wires, const uint64_t *sigmas, const uint64_t *k_is,
                                        const uint64_t *betas, const uint64_t *gammas, size_t num_challenges,
                                        size_t num_routed, unsigned degree_bits, unsigned chunk, uint64_t *out) {
  if (chunk < 2 || num_routed == 0 || degree_bits > 30) return -1;
  const size_t n = (size_t)1 << degree_bits;
  const size_t num_chunks = (num_routed + chunk - 1) / chunk, num_prods = num_chunks - 1;
  const uint64_t w = gl_primitive_root_of_unity(degree_bits);
  uint64_t *q = (uint64_t *)malloc(num_chunks * sizeof(uint64_t));
  for (size_t c = 0; c < num_challenges; ++c) {
    const uint64_t beta = gl_canon(betas[c]), gamma = gl_canon(gammas[c]);
    uint64_t *z_col = out + c * n;
    uint64_t *pp = out + num_challenges * n + c * num_prods * n;
    uint64_t z_x = 1, x = 1; /* subgroup[i] = w^i */
    for (size_t i = 0; i < n; ++i) {
      /* quotient_values[j] = (wire + beta k_j x + gamma) / (wire + beta sigma_j(x) + gamma); products over chunks */
      for (size_t k = 0; k < num_chunks; ++k) {
        uint64_t num = 1, den = 1;
        for (size_t j = k * chunk; j < num_routed && j < (k + 1) * chunk; ++j) {
          const uint64_t wire = gl_canon(wires[j * n + i]);
          const uint64_t s_id = gl_mul(gl_canon(k_is[j]), x);
          num = gl_mul(num, gl_add(gl_add(wire, gl_mul(beta, s_id)), gamma));
          den = gl_mul(den, gl_add(gl_add(wire, gl_mul(beta, gl_canon(sigmas[j * n + i]))), gamma));
        }
        if (den == 0) { free(q); return -2; } /* plonky2 panics on the division (probability ~ 2^-57 per proof) */
        q[k] = gl_mul(num, gl_inv(den));
      }
      /* partial_products_and_z_gx(z_x, q): running products; the last one is Z(g x) and is swapped for Z(x) */
      uint64_t acc = z_x;
      z_col[i] = z_x;
      for (size_t k = 0; k < num_chunks; ++k) {
        acc = gl_mul(acc, q[k]);
        if (k < num_prods) pp[k * n + i] = acc;
      }
      z_x = acc;
      x = gl_mul(x, w);
    }
  }
  free(q);
  return 0;
}

/* ================================================================== gate constraints, quotient, opening check */
#include "poseidon_constants.h"

/* ---- base-field instantiation */
#define FE uint64_t
#define FN(name) name##_b
#define FE_ADD(a, b) gl_add(a, b)
#define FE_SUB(a, b) gl_sub(a, b)
#define FE_MUL(a, b) gl_mul(a, b)
#define FE_MULC(a, c) gl_mul(a, c)
#define FE_ADDC(a, c) gl_add(a, c)
#define FE_SUBC(a, c) gl_sub(a, c)
#define FE_FROMC(c) ((uint64_t)(c))
#include "plonk_eval.inc.h"
#undef FE
#undef FN
#undef FE_ADD
#undef FE_SUB
#undef FE_MUL
#undef FE_MULC
#undef FE_ADDC
#undef FE_SUBC
#undef FE_FROMC

/* ---- quadratic extension F[X]/(X^2 - 7) */
typedef struct { uint64_t a, b; } fe2;
static inline fe2 e_add(fe2 x, fe2 y) { return (fe2){gl_add(x.a, y.a), gl_add(x.b, y.b)}; }
static inline fe2 e_sub(fe2 x, fe2 y) { return (fe2){gl_sub(x.a, y.a), gl_sub(x.b, y.b)}; }
static inline fe2 e_mul(fe2 x, fe2 y) {
  return (fe2){gl_add(gl_mul(x.a, y.a), gl_mul(7, gl_mul(x.b, y.b))), gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a))};
}
static inline fe2 e_mulc(fe2 x, uint64_t c) { return (fe2){gl_mul(x.a, c), gl_mul(x.b, c)}; }
static inline fe2 e_addc(fe2 x, uint64_t c) { return (fe2){gl_add(x.a, c), x.b}; }
static inline fe2 e_subc(fe2 x, uint64_t c) { return (fe2){gl_sub(x.a, c), x.b}; }
static inline fe2 e_fromc(uint64_t c) { return (fe2){c, 0}; }
#define FE fe2
#define FN(name) name##_e
#define FE_ADD(a, b) e_add(a, b)
#define FE_SUB(a, b) e_sub(a, b)
#define FE_MUL(a, b) e_mul(a, b)
#define FE_MULC(a, c) e_mulc(a, c)
#define FE_ADDC(a, c) e_addc(a, c)
#define FE_SUBC(a, c) e_subc(a, c)
#define FE_FROMC(c) e_fromc(c)
#include "plonk_eval.inc.h"

int oracle_gate_constraints_row(unsigned kind, const uint64_t *wires, const uint64_t *consts, const uint64_t pi_hash[4],
                                uint64_t *out) {
  if (kind >= ORACLE_GATE_KINDS) return -1;
  return gate_eval_unfiltered_b(kind, 2, 80, consts, wires, pi_hash, out);
}

static size_t brev(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}

static int desc_ok(const oracle_plonk_desc *d) {
  if (!d || d->num_gates == 0 || d->num_gates > ORACLE_PLONK_MAX_GATES || d->num_challenges == 0 || d->num_challenges > 16) return 0;
  if (d->num_wires != 135 || d->num_routed == 0 || d->num_routed > d->num_wires || d->num_routed % 4) return 0;
  if (d->quotient_degree_factor < 2 || (d->quotient_degree_factor & (d->quotient_degree_factor - 1))) return 0;
  if ((d->num_routed + d->quotient_degree_factor - 1) / d->quotient_degree_factor > 64 || d->num_constants > 4) return 0;
  for (unsigned g = 0; g < d->num_gates; ++g)
    if (d->gate_kind[g] >= ORACLE_GATE_KINDS || d->gate_selector[g] >= d->num_selectors || d->group_start[g] > g || d->group_end[g] <= g)
      return 0;
  return 1;
}

int oracle_plonk_quotient_polys(const oracle_plonk_desc *d, const uint64_t *k_is, const uint64_t *cs_leaves,
                                const uint64_t *wires_leaves, const uint64_t *zs_leaves, const uint64_t pi_hash[4],
                                const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas, uint64_t *out) {
  if (!desc_ok(d)) return -1;
  const unsigned qbits = (unsigned)__builtin_ctz(d->quotient_degree_factor), log_big = d->degree_bits + qbits;
  const size_t n = (size_t)1 << d->degree_bits, big = (size_t)1 << log_big, nch = d->num_challenges;
  const size_t n_cs = d->num_selectors + d->num_constants + d->num_routed;
  const size_t num_prods = (d->num_routed + d->quotient_degree_factor - 1) / d->quotient_degree_factor - 1;
  const size_t n_zs = nch * (1 + num_prods);
  const uint64_t w = gl_primitive_root_of_unity(log_big), shift = 7;
  const uint64_t shift_n = gl_pow(shift, n), w_q = gl_primitive_root_of_unity(qbits); /* x^n = 7^n w_q^(i mod q) */
  const uint64_t n_inv = gl_inv((uint64_t)n % GL_P);
  uint64_t *vals = (uint64_t *)malloc(nch * big * sizeof(uint64_t));
  uint64_t x = shift; /* x_i = 7 w^i */
  uint64_t res[16];
  for (size_t i = 0; i < big; ++i, x = gl_mul(x, w)) {
    const size_t r = brev(i, log_big), r_next = brev((i + d->quotient_degree_factor) % big, log_big);
    const uint64_t *cs = cs_leaves + r * n_cs, *zs = zs_leaves + r * n_zs, *zs_next = zs_leaves + r_next * n_zs;
    const uint64_t zh = gl_sub(gl_mul(shift_n, gl_pow(w_q, i % d->quotient_degree_factor)), 1); /* Z_H(x), never 0 on the coset */
    const uint64_t l0 = gl_mul(gl_mul(zh, n_inv), gl_inv(gl_sub(x, 1)));
    eval_vanishing_b(d, k_is, x, l0, cs, cs + d->num_selectors + d->num_constants, wires_leaves + r * d->num_wires, zs, zs_next,
                     zs + nch, pi_hash, betas, gammas, alphas, res);
    const uint64_t zh_inv = gl_inv(zh);
    for (size_t c = 0; c < nch; ++c) vals[c * big + i] = gl_mul(res[c], zh_inv);
  }
  /* coset_ifft(7): plain IFFT gives c_k 7^k */
  const uint64_t shift_inv = gl_inv(shift);
  for (size_t c = 0; c < nch; ++c) {
    oracle_ifft(vals + c * big, log_big);
    uint64_t s = 1;
    for (size_t k = 0; k < big; ++k, s = gl_mul(s, shift_inv)) out[c * big + k] = gl_mul(vals[c * big + k], s);
  }
  free(vals);
  return 0;
}

int oracle_plonk_check_openings(const oracle_plonk_desc *d, const uint64_t *k_is, const uint64_t zeta[2], const uint64_t *constants,
                                const uint64_t *sigmas, const uint64_t *wires, const uint64_t *zs, const uint64_t *next_zs,
                                const uint64_t *pps, const uint64_t *quotient, const uint64_t pi_hash[4], const uint64_t *betas,
                                const uint64_t *gammas, const uint64_t *alphas) {
  if (!desc_ok(d)) return 0;
  const size_t nch = d->num_challenges, qf = d->quotient_degree_factor;
  const fe2 z = {gl_canon(zeta[0]), gl_canon(zeta[1])};
  fe2 zn = z; /* zeta^n */
  for (unsigned i = 0; i < d->degree_bits; ++i) zn = e_mul(zn, zn);
  const fe2 zh = e_subc(zn, 1);
  /* L_0(zeta) = (zeta^n - 1) / (n (zeta - 1)) */
  uint64_t den[2] = {0, 0}, den_inv[2];
  {
    const fe2 t = e_mulc(e_subc(z, 1), ((uint64_t)1 << d->degree_bits) % GL_P);
    den[0] = t.a;
    den[1] = t.b;
  }
  if (den[0] == 0 && den[1] == 0) return 0;
  oracle_ext_inv(den, den_inv);
  const fe2 l0 = e_mul(zh, (fe2){den_inv[0], den_inv[1]});
  fe2 van[16];
  /* extension openings arrive as consecutive (a, b) pairs == fe2 */
  eval_vanishing_e(d, k_is, z, l0, (const fe2 *)constants, (const fe2 *)sigmas, (const fe2 *)wires, (const fe2 *)zs,
                   (const fe2 *)next_zs, (const fe2 *)pps, pi_hash, betas, gammas, alphas, van);
  for (size_t c = 0; c < nch; ++c) {
    fe2 acc = {0, 0};
    for (size_t k = qf; k-- > 0;) acc = e_add(e_mul(acc, zn), ((const fe2 *)quotient)[c * qf + k]);
    const fe2 rhs = e_mul(zh, acc);
    if (rhs.a != van[c].a || rhs.b != van[c].b) return 0;
  }
  return 1;
}
