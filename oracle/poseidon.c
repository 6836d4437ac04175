/* oracle/poseidon.c -- TEST INFRASTRUCTURE. Spec-form Poseidon permutation and plonky2's hash modes.
 *
 * Third-party algorithm (absent from /root/reference): plonky2 @3b21b87 hash/poseidon.rs
 * (Poseidon::poseidon: 4 full, 22 partial, 4 full rounds; x^7; circulant MDS + diagonal) and
 * hash/hashing.rs (hash_n_to_m_no_pad: overwrite-mode sponge, rate 8; two_to_one; hash_or_noop).
 * Restated from SURVEY.md Appendix A.2/A.3. The naive (non-"fast") partial round is used on
 * purpose: it is the specification, the HIP kernels use the derived fast form.
 *
 * Reference call sites: simple_merkle_tree.rs:23,33,45,93,100,102;
 *                       merkle_mountain_ranges.rs:91,96,111,125,233,238,240,249.
 */
#include <string.h>
#include "goldilocks.h"
#include "oracle.h"
#include "poseidon_constants.h"

static inline uint64_t sbox7(uint64_t x) {
  uint64_t x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x3 = gl_mul(x2, x);
  return gl_mul(x4, x3);
}

/* out[r] = sum_c MDS[r][c] * s[c], MDS[r][c] = CIRC[(c - r) mod 12] + (r == c ? DIAG[r] : 0) */
static void mds_layer(uint64_t s[12]) {
  uint64_t out[12];
  for (int r = 0; r < 12; ++r) {
    u128 acc = 0; /* 12 terms < 2^6 * 2^64 each: fits easily */
    for (int i = 0; i < 12; ++i) acc += (u128)s[(i + r) % 12] * POSEIDON_MDS_CIRC[i];
    acc += (u128)s[r] * POSEIDON_MDS_DIAG[r];
    out[r] = gl_reduce128(acc);
  }
  memcpy(s, out, sizeof out);
}

void oracle_poseidon_permute(uint64_t s[12]) {
  for (int i = 0; i < 12; ++i) s[i] = gl_canon(s[i]);
  for (int r = 0; r < POSEIDON_ROUNDS; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = gl_add(s[i], POSEIDON_RC[12 * r + i]);
    if (r < POSEIDON_HALF_FULL_ROUNDS || r >= POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS) {
      for (int i = 0; i < 12; ++i) s[i] = sbox7(s[i]);
    } else {
      s[0] = sbox7(s[0]);
    }
    mds_layer(s);
  }
}

/* Witness of one PoseidonGate row (plonky2 @3b21b87 gates/poseidon.rs, absent; layout from recall -- PARITY
 * UNPINNED).  This is what the PoseidonGenerator fills for every hash_n_to_hash_no_pad / two_to_one the circuits
 * of /root/reference/src/mmr/mmr_plonky2_verifier.rs:46-54,81 and mmr_plonky2_verifier_1_recursion.rs:44-52 add.
 * The partial-round wires hold lane 0's S-box input, which is identical in the spec form used here and in the
 * sparse form plonky2 evaluates (checked in tools/poseidon_spec.py). */
void oracle_poseidon_gate_witness(const uint64_t in[12], int swap, uint64_t out[135]) {
  uint64_t s[12];
  for (int i = 0; i < 12; ++i) out[i] = s[i] = gl_canon(in[i]);
  out[24] = swap ? 1 : 0;
  for (int i = 0; i < 4; ++i) {
    uint64_t delta = swap ? gl_sub(s[i + 4], s[i]) : 0;
    out[25 + i] = delta;
    uint64_t l = gl_add(s[i], delta), r = gl_sub(s[i + 4], delta);
    s[i] = l;
    s[i + 4] = r;
  }
  for (int r = 0; r < POSEIDON_ROUNDS; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = gl_add(s[i], POSEIDON_RC[12 * r + i]);
    if (r < 4) {
      if (r >= 1) for (int i = 0; i < 12; ++i) out[29 + 12 * (r - 1) + i] = s[i];
      for (int i = 0; i < 12; ++i) s[i] = sbox7(s[i]);
    } else if (r < 26) {
      out[65 + (r - 4)] = s[0];
      s[0] = sbox7(s[0]);
    } else {
      for (int i = 0; i < 12; ++i) out[87 + 12 * (r - 26) + i] = s[i];
      for (int i = 0; i < 12; ++i) s[i] = sbox7(s[i]);
    }
    mds_layer(s);
  }
  for (int i = 0; i < 12; ++i) out[12 + i] = s[i];
}

/* Hasher::two_to_one: perm([l, r, 0,0,0,0])[0..4] */
void oracle_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
  uint64_t s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
  oracle_poseidon_permute(s);
  memcpy(out, s, 4 * sizeof(uint64_t));
}

/* hash_n_to_hash_no_pad: state = 0; per chunk of <=8 inputs OVERWRITE state[0..len) then permute;
 * output state[0..4] (no chunk at all => zeros, no permutation). */
void oracle_hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4]) {
  uint64_t s[12] = {0};
  for (size_t off = 0; off < n; off += 8) {
    size_t len = n - off < 8 ? n - off : 8;
    for (size_t i = 0; i < len; ++i) s[i] = gl_canon(in[off + i]);
    oracle_poseidon_permute(s);
  }
  memcpy(out, s, 4 * sizeof(uint64_t));
}

/* Hasher::hash_or_noop: <= 4 elements => zero-padded copy, no permutation (Quirk Q1) */
void oracle_hash_or_noop(const uint64_t *in, size_t n, uint64_t out[4]) {
  if (n <= 4) {
    for (size_t i = 0; i < 4; ++i) out[i] = i < n ? gl_canon(in[i]) : 0;
  } else {
    oracle_hash_no_pad(in, n, out);
  }
}

void oracle_poseidon_round_constants(uint64_t out[360]) { memcpy(out, POSEIDON_RC, sizeof(POSEIDON_RC)); }

uint64_t oracle_gl_add(uint64_t a, uint64_t b) { return gl_add(gl_canon(a), gl_canon(b)); }
uint64_t oracle_gl_sub(uint64_t a, uint64_t b) { return gl_sub(gl_canon(a), gl_canon(b)); }
uint64_t oracle_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a, b); }
uint64_t oracle_gl_pow(uint64_t a, uint64_t e) { return gl_pow(gl_canon(a), e); }
uint64_t oracle_gl_inv(uint64_t a) { return gl_inv(gl_canon(a)); }
uint64_t oracle_gl_primitive_root_of_unity(unsigned log_n) { return gl_primitive_root_of_unity(log_n); }
