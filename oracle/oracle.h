/*
 * oracle/ -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (plonky2-merkle-trees_amd/, include/) never links, imports or calls it.
 *
 * What it restates (reference = /root/reference, hashcloak/plonky2-merkle-trees):
 *   - src/simple_merkle_tree/simple_merkle_tree.rs:18-109  (MerkleTree, verify_merkle_proof)
 *   - src/mmr/merkle_mountain_ranges.rs:39-270             (MMR, MMR_proof, index maths)
 * and the third-party arithmetic those call (NOT in /root/reference, un-vendored deps):
 *   - plonky2 git rev 3b21b87d0ab3f8ef4b9ff0b9dd70f8e32f5573f4: PoseidonHash
 *     (hash/poseidon.rs, hash/hashing.rs), MerkleTree/MerkleCap (hash/merkle_tree.rs),
 *     PolynomialBatch (fri/oracle.rs)
 *   - plonky2_field 0.1.0: GoldilocksField, fft.rs
 *   restated from their published algorithm.
 *
 * Parity pinning:
 *   PINNED   Poseidon permutation, two_to_one, no-op leaf hash, tree build, Merkle path:
 *            every golden value the reference holds (simple_merkle_tree.rs:136-140,
 *            :181-190, :210-211) and the 32 index-table rows (merkle_mountain_ranges.rs:280-327)
 *            are checked in tests/test_oracle_golden.py.
 *   UNPINNED ("parity unpinned") sponge with >8 inputs, FFT/LDE/Merkle-cap conventions:
 *            the reference holds no vector for them; they follow plonky2's published
 *            conventions from recall and are checked only
 *            through mathematical identities (IFFT.FFT = id, LDE == direct evaluation).
 *
 * All values crossing this API are canonical Goldilocks elements (< p) as little-endian u64;
 * a HashOut is 4 consecutive u64.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_GOLDILOCKS_P 0xFFFFFFFF00000001ULL /* src/mmr/common.rs:3 */

/* ---- field (plonky2_field::goldilocks_field) ---- */
uint64_t oracle_gl_add(uint64_t a, uint64_t b);
uint64_t oracle_gl_sub(uint64_t a, uint64_t b);
uint64_t oracle_gl_mul(uint64_t a, uint64_t b);
uint64_t oracle_gl_pow(uint64_t a, uint64_t e);
uint64_t oracle_gl_inv(uint64_t a);
uint64_t oracle_gl_primitive_root_of_unity(unsigned log_n);

/* ---- Poseidon (plonky2::hash::poseidon, hashing) ---- */
void oracle_poseidon_permute(uint64_t state[12]);
void oracle_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);
void oracle_hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4]);
void oracle_hash_or_noop(const uint64_t *in, size_t n, uint64_t out[4]);

/* PoseidonGate witness row [parity unpinned: plonky2 gates/poseidon.rs wire layout from recall, SURVEY.md B.2]:
 * out[0..12) inputs, [12..24) outputs, [24] swap, [25..29) delta_i = swap*(in[i+4]-in[i]),
 * [29..65) S-box inputs of first-half full rounds 1..3, [65..87) lane-0 S-box inputs of the 22 partial rounds,
 * [87..135) S-box inputs of the second-half full rounds 0..3.  The permutation runs on the swapped state. */
void oracle_poseidon_gate_witness(const uint64_t in[12], int swap, uint64_t out[135]);

/* ---- simple_merkle_tree.rs ---- */
/* levels_out: level-major, level i has n>>i HashOuts, levels 0..k-1 => (2n-2) HashOuts.
 * Returns count_levels (= log2 n), or -1 where the reference panics (n not a power of two, n < 2). */
int oracle_merkle_build(const uint64_t *leaves, size_t n, uint64_t *levels_out, uint64_t root[4]);
int oracle_merkle_get_proof(const uint64_t *levels, size_t n, size_t leaf_index, uint64_t *proof_out);
int oracle_merkle_get_in_between_hashes(const uint64_t *levels, const uint64_t root[4], size_t n,
                                        size_t leaf_index, uint64_t *out);
int oracle_verify_merkle_proof(uint64_t leaf, size_t leaf_index, const uint64_t root[4],
                               const uint64_t *hashes, size_t n_hashes);

/* ---- merkle_mountain_ranges.rs ---- */
typedef struct oracle_mmr oracle_mmr;
uint64_t oracle_get_heights_bitmap_for_mmr_size(size_t mmr_size, size_t *remainder);
size_t oracle_get_mmr_index(size_t leaf_normal_index);
oracle_mmr *oracle_mmr_new(void);
void oracle_mmr_free(oracle_mmr *m);
void oracle_mmr_add_leaf(oracle_mmr *m, uint64_t leaf);
void oracle_mmr_add_leaves(oracle_mmr *m, const uint64_t *leaves, size_t n); /* for leaf { add_leaf } */
/* BASELINE.md B2 ("generous" CPU baseline, NOT the reference's algorithm): the same post-order array for n = 2^k
 * leaves built level by level with every host core (OpenMP).  Returns the number of threads used. */
int oracle_mmr_build_pow2_parallel(const uint64_t *leaves, size_t n, uint64_t *elements_out, int threads /* 0 = all */);
/* Tuned scalar port (poseidon_fast.c: sparse partial rounds, lazy reduction; bench.py's cpu_baseline.port_fast).  Same values. */
void oracle_poseidon_round_constants(uint64_t out[360]); /* ALL_ROUND_CONSTANTS, round r = [12r, 12r + 12) */
void oracle_fast_poseidon_permute(uint64_t state[12]);
void oracle_fast_two_to_one_batch(const uint64_t *in /*[n][8]*/, uint64_t *out /*[n][4]*/, size_t n);
void oracle_fast_mmr_add_leaf_loop(const uint64_t *leaves, size_t n, uint64_t *elements_out); /* for leaf { add_leaf }, 1 thread */
int oracle_fast_mmr_build_pow2(const uint64_t *leaves, size_t n, uint64_t *elements_out, int threads /* 0 = all */);
size_t oracle_mmr_len(const oracle_mmr *m);
const uint64_t *oracle_mmr_elements(const oracle_mmr *m);
/* returns number of peaks, -1 where the reference panics (empty MMR / len >= 2^32) */
int oracle_mmr_get_peaks(const oracle_mmr *m, uint64_t *peaks_out);
int oracle_mmr_bagging_the_peaks(const oracle_mmr *m, uint64_t root[4]);
/* siblings_out: up to 64 HashOuts, lefts_out: up to 64 bytes. Returns n siblings or -1 (index OOB). */
int oracle_mmr_get_subtree_proof_elm(const oracle_mmr *m, size_t mmr_index, uint64_t *siblings_out,
                                     uint8_t *lefts_out);
int oracle_mmr_get_proof(const oracle_mmr *m, size_t mmr_index, uint64_t *siblings_out,
                         uint8_t *lefts_out, int *n_siblings, uint64_t *peaks_out, int *n_peaks,
                         size_t *mmr_size);
/* 1 = true, 0 = false, -1 = the reference's assert!(peaks.contains(..)) panics (:245) */
int oracle_mmr_proof_verify(const uint64_t *siblings, const uint8_t *lefts, int n_siblings,
                            const uint64_t *peaks, int n_peaks, uint64_t leaf, const uint64_t root[4]);

/* ---- plonky2_field fft.rs / plonky2 fri/oracle.rs + hash/merkle_tree.rs  [parity unpinned] ---- */
void oracle_fft(uint64_t *a, unsigned log_n);  /* natural order in/out: out[i] = f(w^i) */
void oracle_ifft(uint64_t *a, unsigned log_n); /* inverse of the above */
/* out[i] = f(shift * w_{n*2^rate_bits}^i), i in natural order; out has n << rate_bits entries */
void oracle_coset_lde(const uint64_t *coeffs, unsigned log_n, unsigned rate_bits, uint64_t shift,
                      uint64_t *out);
/* Merkle tree over n leaves of `width` elements (row-major n x width), n = 2^k, cap_height <= k.
 * digests_out: level-major (level 0 = leaf digests, n HashOuts; level j has n>>j), levels 0 .. k-cap_height-1;
 * cap_out: 2^cap_height HashOuts (the level k-cap_height row). */
int oracle_merkle_cap_commit(const uint64_t *leaves, size_t n, size_t width, unsigned cap_height,
                             uint64_t *digests_out, uint64_t *cap_out);
/* PolynomialBatch::from_coeffs / from_values: polys row-major n_polys x 2^log_n.
 * leaves_out: (n<<rate_bits) x n_polys row-major, leaf index bit-reversed (leaf[brev(i)] = all polys at point i). */
int oracle_polynomial_batch_commit(const uint64_t *polys, int is_values, size_t n_polys, unsigned log_n,
                                   unsigned rate_bits, unsigned cap_height, uint64_t *leaves_out,
                                   uint64_t *digests_out, uint64_t *cap_out);
/* the same on `threads` host cores (0 = all), tuned scalar Poseidon port: bench.py's B4.  Returns the thread count (< 0: error). */
int oracle_polynomial_batch_commit_parallel(const uint64_t *polys, int is_values, size_t n_polys, unsigned log_n,
                                            unsigned rate_bits, unsigned cap_height, uint64_t *leaves_out,
                                            uint64_t *cap_out, int threads);

/* ---- plonky2 iop/challenger.rs, fri/{oracle,prover,verifier}.rs  [parity unpinned] (oracle/fri.c) ---- */
typedef struct oracle_challenger {
  uint64_t state[12];
  uint64_t in[8];
  uint64_t out[8];
  uint32_t n_in, n_out;
} oracle_challenger;
void oracle_challenger_init(oracle_challenger *c);
void oracle_challenger_observe(oracle_challenger *c, const uint64_t *e, size_t n);
uint64_t oracle_challenger_get(oracle_challenger *c);

typedef struct oracle_fri_params {
  uint32_t degree_bits, rate_bits, cap_height, proof_of_work_bits, num_query_rounds, num_reductions;
  uint32_t reduction_arity_bits[8];
} oracle_fri_params;
/* one committed PolynomialBatch: coeffs poly-major n_polys x 2^degree_bits; leaves/digests as produced by
 * oracle_polynomial_batch_commit (leaves row-major N x n_polys in bit-reversed point order, digests level-major) */
typedef struct oracle_fri_oracle {
  const uint64_t *coeffs, *leaves, *digests;
  uint64_t n_polys;
} oracle_fri_oracle;
/* FriBatchInfo: opening point (extension element) + (oracle_index, polynomial_index) pairs */
typedef struct oracle_fri_batch {
  uint64_t point[2];
  const uint32_t *polys;
  uint64_t n_polys;
} oracle_fri_batch;
void oracle_ext_mul(const uint64_t x[2], const uint64_t y[2], uint64_t out[2]);
void oracle_ext_inv(const uint64_t x[2], uint64_t out[2]);
void oracle_fri_params_standard(unsigned degree_bits, oracle_fri_params *p);
/* proof words: commit-phase caps | per query: per oracle (leaf row | siblings), per layer (evals | siblings) | final poly | pow witness */
size_t oracle_fri_proof_len(const oracle_fri_params *p, size_t n_oracles, const uint64_t *n_polys);
void oracle_eval_polys_ext(const uint64_t *coeffs, size_t n_polys, unsigned log_n, const uint64_t point[2], uint64_t *out);
int oracle_fri_prove(const oracle_fri_oracle *oracles, size_t n_oracles, const oracle_fri_batch *batches, size_t n_batches,
                     const oracle_fri_params *p, oracle_challenger *ch, uint64_t *proof_out);
/* caps: n_oracles x 2^cap_height x 4; openings: per batch, n_polys extension values (2 words each), batches concatenated */
int oracle_fri_verify(const uint64_t *n_polys, size_t n_oracles, const uint64_t *caps, const oracle_fri_batch *batches,
                      size_t n_batches, const uint64_t *openings, const oracle_fri_params *p, oracle_challenger *ch,
                      const uint64_t *proof, int *reason);

/* ---- plonky2 plonk/prover.rs wires_permutation_partial_products_and_zs  [parity unpinned] (oracle/plonk.c) ----
 * wires, sigmas: [num_routed][n] columns of values on the subgroup (natural order); k_is: the coset shifts of the identity
 * permutation; chunk = quotient_degree_factor.  out: num_challenges Z columns, then num_challenges x num_prods
 * partial-product columns (num_prods = ceil(num_routed / chunk) - 1).  Returns 0, -1 bad shape, -2 zero denominator. */
int oracle_permutation_partial_products(const uint64_t *wires, const uint64_t *sigmas, const uint64_t *k_is,
                                        const uint64_t *betas, const uint64_t *gammas, size_t num_challenges,
                                        size_t num_routed, unsigned degree_bits, unsigned chunk, uint64_t *out);

/* ---- plonky2 gates + plonk/vanishing_poly.rs + prover.rs compute_quotient_polys + verifier.rs  [parity unpinned] (oracle/plonk.c) ----
 * Circuit shape the verifier circuits of /root/reference/src/mmr/mmr_plonky2_verifier.rs:13-91 build to under
 * CircuitConfig::standard_recursion_config(): the gate types in plonky2's sorted order (degree, id), each with its selector
 * polynomial and selector group (gates/selectors.rs). */
enum { ORACLE_GATE_NOOP = 0, ORACLE_GATE_CONSTANT = 1, ORACLE_GATE_PUBLIC_INPUT = 2, ORACLE_GATE_ARITHMETIC = 3, ORACLE_GATE_POSEIDON = 4,
       /* the gates plonky2's in-circuit verifier (builder.verify_proof, mmr_plonky2_verifier_1_recursion.rs:101-104) adds under
        * standard_recursion_config; parameters fixed by that config (D = 2): */
       ORACLE_GATE_BASE_SUM = 5,            /* BaseSumGate<2> { num_limbs: 63 } */
       ORACLE_GATE_ARITHMETIC_EXT = 6,      /* ArithmeticExtensionGate { num_ops: 10 } */
       ORACLE_GATE_MUL_EXT = 7,             /* MulExtensionGate { num_ops: 13 } */
       ORACLE_GATE_REDUCING = 8,            /* ReducingGate { num_coeffs: 43 } */
       ORACLE_GATE_REDUCING_EXT = 9,        /* ReducingExtensionGate { num_coeffs: 32 } */
       ORACLE_GATE_RANDOM_ACCESS = 10,      /* RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2 } */
       ORACLE_GATE_COSET_INTERPOLATION = 11,/* CosetInterpolationGate { subgroup_bits: 4, degree: 6 } */
       ORACLE_GATE_POSEIDON_MDS = 12,       /* PoseidonMdsGate */
       ORACLE_GATE_KINDS = 13 };
#define ORACLE_PLONK_NUM_GATE_CONSTRAINTS 123 /* PoseidonGate: 1 + 4 + 36 + 22 + 48 + 12 (the maximum over all gate types) */
#define ORACLE_PLONK_MAX_GATES 16
typedef struct oracle_plonk_desc {
  uint32_t degree_bits, num_wires, num_routed, num_constants, num_selectors, num_challenges, quotient_degree_factor, num_gates;
  uint32_t gate_kind[ORACLE_PLONK_MAX_GATES], gate_selector[ORACLE_PLONK_MAX_GATES], group_start[ORACLE_PLONK_MAX_GATES],
      group_end[ORACLE_PLONK_MAX_GATES];
} oracle_plonk_desc;
/* compute_quotient_polys: cs/wires/zs leaves are the LDE matrices of the three committed batches as
 * oracle_polynomial_batch_commit leaves them ([8n][n_polys], leaf index bit-reversed; cs = selectors | constants | sigmas,
 * zs = Z per challenge | partial products).  out: num_challenges x quotient_degree_factor coefficient chunks of n. Returns 0. */
int oracle_plonk_quotient_polys(const oracle_plonk_desc *d, const uint64_t *k_is, const uint64_t *cs_leaves,
                                const uint64_t *wires_leaves, const uint64_t *zs_leaves, const uint64_t pi_hash[4],
                                const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas, uint64_t *out);
/* The verifier's check of the opened values (verifier.rs verify_with_challenges, before verify_fri_proof):
 * vanishing(zeta) == Z_H(zeta) * sum_k t_k(zeta) zeta^(n k) per challenge.  Every opening is an extension element (2 words):
 * constants [num_selectors + num_constants], sigmas [num_routed], wires [num_wires], zs / next_zs [num_challenges],
 * pps [num_challenges * num_prods], quotient [num_challenges * quotient_degree_factor].  Returns 1 accept, 0 reject. */
/* Unfiltered constraints of ONE gate type on ONE row of base-field wires (tests: every row of a generated witness must satisfy
 * the constraints of its own gate).  consts: the row's gate constants.  Returns the number of constraints written to out. */
int oracle_gate_constraints_row(unsigned kind, const uint64_t *wires /*[135]*/, const uint64_t *consts /*[2]*/,
                                const uint64_t pi_hash[4], uint64_t *out /*[123]*/);
int oracle_plonk_check_openings(const oracle_plonk_desc *d, const uint64_t *k_is, const uint64_t zeta[2], const uint64_t *constants,
                                const uint64_t *sigmas, const uint64_t *wires, const uint64_t *zs, const uint64_t *next_zs,
                                const uint64_t *pps, const uint64_t *quotient, const uint64_t pi_hash[4], const uint64_t *betas,
                                const uint64_t *gammas, const uint64_t *alphas);

#ifdef __cplusplus
}
#endif
#endif
