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
