/* oracle/fft.c -- TEST INFRASTRUCTURE.  *** PARITY UNPINNED ***
 * Host restatement of the commit step that CircuitData::prove drives
 * (/root/reference/src/mmr/mmr_plonky2_verifier.rs:148, mmr_plonky2_verifier_1_recursion.rs:192,218):
 * PolynomialBatch::from_values/from_coeffs = IFFT -> x8 coset-LDE -> transpose -> bit-reverse -> MerkleTree(cap).
 * The algorithm lives in absent third-party crates (plonky2_field 0.1.0 fft.rs / polynomial/mod.rs,
 * plonky2 @3b21b87 fri/oracle.rs, hash/merkle_tree.rs); conventions are restated from SURVEY.md
 * Appendix B.3/B.4 (recall, no vector in the reference pins them). Since results are exact field
 * values, any correct evaluation at the same points is bit-identical; what is unpinned is only the
 * choice of points/ordering. Written as the textbook O(n log n) transform, deliberately different in
 * structure from the device kernels, and cross-checked against direct O(n^2) evaluation in tests. */
#include <stdlib.h>
#include <string.h>
#include "goldilocks.h"
#include "oracle.h"

static size_t brev(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}

/* out[i] = sum_j a[j] * w^(ij), w = primitive 2^log_n-th root; natural order in and out. */
static void fft_with_root(uint64_t *a, unsigned log_n, uint64_t w) {
  size_t n = (size_t)1 << log_n;
  for (size_t i = 0; i < n; ++i) { /* bit-reverse, then decimation-in-time butterflies */
    size_t j = brev(i, log_n);
    if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  for (unsigned s = 1; s <= log_n; ++s) {
    size_t m = (size_t)1 << s, half = m >> 1;
    uint64_t wm = w;
    for (unsigned i = s; i < log_n; ++i) wm = gl_mul(wm, wm); /* w^(n/m) */
    for (size_t k = 0; k < n; k += m) {
      uint64_t tw = 1;
      for (size_t j = 0; j < half; ++j) {
        uint64_t u = a[k + j], t = gl_mul(tw, a[k + j + half]);
        a[k + j] = gl_add(u, t);
        a[k + j + half] = gl_sub(u, t);
        tw = gl_mul(tw, wm);
      }
    }
  }
}

void oracle_fft(uint64_t *a, unsigned log_n) {
  size_t n = (size_t)1 << log_n;
  for (size_t i = 0; i < n; ++i) a[i] = gl_canon(a[i]);
  fft_with_root(a, log_n, gl_primitive_root_of_unity(log_n));
}

/* ifft: fft, then out[i] = buf[(n - i) mod n] / n   */
void oracle_ifft(uint64_t *a, unsigned log_n) {
  size_t n = (size_t)1 << log_n;
  oracle_fft(a, log_n);
  uint64_t n_inv = gl_inv((uint64_t)n % GL_P);
  uint64_t *tmp = (uint64_t *)malloc(n * 8);
  for (size_t i = 0; i < n; ++i) tmp[i] = gl_mul(a[(n - i) % n], n_inv);
  memcpy(a, tmp, n * 8);
  free(tmp);
}

/* coset LDE: scale coeff i by shift^i, zero-pad to n << rate_bits, fft */
void oracle_coset_lde(const uint64_t *coeffs, unsigned log_n, unsigned rate_bits, uint64_t shift, uint64_t *out) {
  size_t n = (size_t)1 << log_n, big = n << rate_bits;
  uint64_t pw = 1;
  shift = gl_canon(shift);
  for (size_t i = 0; i < big; ++i) {
    if (i < n) {
      out[i] = gl_mul(gl_canon(coeffs[i]), pw);
      pw = gl_mul(pw, shift);
    } else {
      out[i] = 0;
    }
  }
  oracle_fft(out, log_n + rate_bits);
}

/* MerkleTree::new(leaves, cap_height) (plonky2 hash/merkle_tree.rs):
 * leaf digest = hash_or_noop(leaf), inner = two_to_one, cap = the 2^cap_height subtree roots.
 * digests are stored level-major here (our own layout; plonky2's internal interleaved layout is
 * not observable through proofs/caps). */
int oracle_merkle_cap_commit(const uint64_t *leaves, size_t n, size_t width, unsigned cap_height,
                             uint64_t *digests_out, uint64_t *cap_out) {
  if (n == 0 || (n & (n - 1))) return -1;
  unsigned k = 0;
  while (((size_t)1 << k) < n) ++k;
  if (cap_height > k) return -1;
  uint64_t *lvl = (uint64_t *)malloc(n * 32), *nxt = (uint64_t *)malloc(n * 32);
  for (size_t i = 0; i < n; ++i) oracle_hash_or_noop(&leaves[i * width], width, &lvl[4 * i]);
  size_t cur_n = n, off = 0;
  for (unsigned level = 0; level < k - cap_height; ++level) {
    if (digests_out) memcpy(&digests_out[4 * off], lvl, cur_n * 32);
    off += cur_n;
    for (size_t j = 0; j < cur_n / 2; ++j) oracle_two_to_one(&lvl[8 * j], &lvl[8 * j + 4], &nxt[4 * j]);
    uint64_t *t = lvl; lvl = nxt; nxt = t;
    cur_n /= 2;
  }
  memcpy(cap_out, lvl, cur_n * 32);
  free(lvl);
  free(nxt);
  return 0;
}

/* PolynomialBatch::from_values / from_coeffs with blinding = false, shift = 7 */
int oracle_polynomial_batch_commit(const uint64_t *polys, int is_values, size_t n_polys, unsigned log_n,
                                   unsigned rate_bits, unsigned cap_height, uint64_t *leaves_out,
                                   uint64_t *digests_out, uint64_t *cap_out) {
  size_t n = (size_t)1 << log_n, big = n << rate_bits;
  unsigned log_big = log_n + rate_bits;
  uint64_t *coeffs = (uint64_t *)malloc(n * 8), *lde = (uint64_t *)malloc(big * 8);
  for (size_t j = 0; j < n_polys; ++j) {
    memcpy(coeffs, &polys[j * n], n * 8);
    if (is_values) oracle_ifft(coeffs, log_n);
    oracle_coset_lde(coeffs, log_n, rate_bits, 7, lde);
    /* transpose + reverse_index_bits: leaf brev(i) holds all polys at point i */
    for (size_t i = 0; i < big; ++i) leaves_out[brev(i, log_big) * n_polys + j] = lde[i];
  }
  free(coeffs);
  free(lde);
  return oracle_merkle_cap_commit(leaves_out, big, n_polys, cap_height, digests_out, cap_out);
}

/* B4: the same commit on all host cores -- what bench.py times beside the GPU's commit phase.  Same arithmetic as
 * above (textbook transforms, the tuned scalar Poseidon port of poseidon_fast.c for the sponge and the tree), parallel over the
 * polynomials, then over the leaves, then over each tree level.  Generous: the reference's prover is the reference calls plonky2 with its default features; no claim is made beyond "this C restatement, this many threads".  threads = 0: all cores. */
#include <omp.h>
static void fast_hash_or_noop(const uint64_t *in, size_t n, uint64_t out[4]) {
  if (n <= 4) {
    for (size_t i = 0; i < 4; ++i) out[i] = i < n ? gl_canon(in[i]) : 0;
    return;
  }
  uint64_t s[12] = {0};
  for (size_t off = 0; off < n; off += 8) {
    size_t len = n - off < 8 ? n - off : 8;
    for (size_t i = 0; i < len; ++i) s[i] = gl_canon(in[off + i]);
    oracle_fast_poseidon_permute(s);
  }
  memcpy(out, s, 32);
}
int oracle_polynomial_batch_commit_parallel(const uint64_t *polys, int is_values, size_t n_polys, unsigned log_n,
                                            unsigned rate_bits, unsigned cap_height, uint64_t *leaves_out,
                                            uint64_t *cap_out, int threads) {
  size_t n = (size_t)1 << log_n, big = n << rate_bits;
  unsigned log_big = log_n + rate_bits;
  if (cap_height > log_big) return -1;
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
  {
    uint64_t *coeffs = (uint64_t *)malloc(n * 8), *lde = (uint64_t *)malloc(big * 8);
#pragma omp for schedule(dynamic, 1)
    for (size_t j = 0; j < n_polys; ++j) {
      memcpy(coeffs, &polys[j * n], n * 8);
      if (is_values) oracle_ifft(coeffs, log_n);
      oracle_coset_lde(coeffs, log_n, rate_bits, 7, lde);
      for (size_t i = 0; i < big; ++i) leaves_out[brev(i, log_big) * n_polys + j] = lde[i];
    }
    free(coeffs);
    free(lde);
  }
  uint64_t *lvl = (uint64_t *)malloc(big * 32), *nxt = (uint64_t *)malloc(big * 32);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (size_t i = 0; i < big; ++i) fast_hash_or_noop(&leaves_out[i * n_polys], n_polys, &lvl[4 * i]);
  size_t cur_n = big;
  for (unsigned level = 0; level < log_big - cap_height; ++level) {
#pragma omp parallel for num_threads(threads) schedule(static) if (cur_n >= 256)
    for (size_t j = 0; j < cur_n / 2; ++j) {
      uint64_t in[8];
      memcpy(in, &lvl[8 * j], 64);
      oracle_fast_two_to_one_batch(in, &nxt[4 * j], 1);
    }
    uint64_t *t = lvl; lvl = nxt; nxt = t;
    cur_n /= 2;
  }
  memcpy(cap_out, lvl, cur_n * 32);
  free(lvl);
  free(nxt);
  return threads;
}
