/*
 * p2mt.h -- C ABI of the MI355X-native Poseidon/Goldilocks Merkle + MMR + commit library
 *           (libp2mt_hip.so, built from plonky2-merkle-trees_amd/csrc/).
 *
 * This is the drop-in boundary for the hot path of hashcloak/plonky2-merkle-trees.
 * The reference has no FFI of its own: its hot path is reached through its public Rust API and,
 * below that, plonky2's per-hash `Hasher` trait, which is useless as a GPU boundary (one hash per
 * call).  The boundary is therefore one level up and batch-shaped; every entry point names the
 * reference interface (file:line under /root/reference) it replaces.  INTEGRATION.md shows the Rust
 * `extern "C"` block + shim a maintainer would add.
 *
 * Conventions
 *   - GoldilocksField = u64 (little-endian host order).  Inputs may be non-canonical (>= p);
 *     every output is canonical (< p = 0xFFFFFFFF00000001).
 *   - HashOut = 4 consecutive u64 (32-byte record); Vec<HashOut> = contiguous records.
 *   - Every function returns 0 on success or a negative p2mt_status; the reference's convention is
 *     panic (assert!/unwrap/log2_strict), the shim maps non-zero to panic!.  No exceptions cross.
 *   - Host-pointer entry points copy in/out synchronously.  `_dev` entry points take pointers to
 *     device memory (hipMalloc / torch tensors), enqueue on the library stream and do not
 *     synchronise unless they return host-visible results.
 *   - Caller owns every buffer; device-resident state lives only behind opaque handles.
 *   - A handle is not thread-safe; distinct handles are independent and may be driven from distinct host threads
 *     concurrently (the library stream, scratch buffers and p2mt_last_error() are per thread).
 *   - All hashing runs on the GPU.  There is no CPU fallback: without a HIP device every compute
 *     entry point returns P2MT_EHIP.
 */
#ifndef P2MT_H
#define P2MT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum p2mt_status {
  P2MT_OK = 0,
  P2MT_EINVAL = -1,   /* bad size / index: where the reference panics (log2_strict, assert!, index OOB) */
  P2MT_ENOMEM = -2,   /* host or device allocation failed */
  P2MT_EHIP = -3,     /* HIP runtime error or no device; see p2mt_last_error() */
  P2MT_ERANGE = -4,   /* size limits of the reference: len >= 2^32 (merkle_mountain_ranges.rs:184), n >= 2^30 (:264) */
  P2MT_ENOTPEAK = -5  /* MMR_proof::verify: assert!(self.peaks.contains(&next_hash)) (merkle_mountain_ranges.rs:245) */
} p2mt_status;

#define P2MT_GOLDILOCKS_FIELD_ORDER 0xFFFFFFFF00000001ULL /* src/mmr/common.rs:3 */
#define P2MT_MAX_PROOF_LEN 64

/* ------------------------------------------------------------------ device / runtime control */
int p2mt_init(int device);              /* select device, upload Poseidon tables; idempotent */
int p2mt_device_count(void);            /* number of visible HIP devices (0 => nothing can run) */
int p2mt_set_stream(void *hip_stream);  /* stream for all subsequent launches of the CALLING THREAD (NULL = default stream) */
int p2mt_get_stream(void **hip_stream_out); /* the calling thread's current library stream (to restore after a scoped p2mt_set_stream) */
/* Give the calling host thread its own non-blocking stream (and, implicitly, its own scratch buffers): from then on its
 * calls enqueue there.  This is how several provers run concurrently on one GPU -- one handle (MMR, circuit data,
 * challenger) per thread; a 64-row prove occupies a few CUs for ~2.6 ms, so independent proofs overlap almost freely. */
int p2mt_thread_stream_create(void);
int p2mt_thread_stream_destroy(void); /* before such a thread ends: frees its stream and its scratch buffers */
/* Process-wide kernel-selection policy.  0 (default) = latency: small batches of hashes run on the 12-lanes-per-permutation
 * layout (one proof as fast as possible; a wavefront then uses 12 of its 64 lanes).  1 = throughput: leaf sponges, Merkle
 * levels and the proof-of-work grind use the 4-lanes / 1-lane-per-hash layouts instead -- a single proof gets slower, but
 * concurrent provers stop competing for SIMD issue slots.  Results are bit-identical in both modes.  (Also: env
 * P2MT_THROUGHPUT=1 at p2mt_init.) */
int p2mt_set_throughput_mode(int on);
int p2mt_sync(void);                    /* hipStreamSynchronize on the library stream */
const char *p2mt_last_error(void);
/* Kernel variant for the Poseidon permutation.  mds 2 (default) = issue-optimised path (constants folded into the
 * MDS mad chains, carry-mask reduction, sticky rare-event flag + exact fallback; with partial 0, the default, the dense
 * MDS layers of the tree-build kernels run on the matrix pipe and four partial rounds share one MDS application);
 * mds 0 / 1 = exact reference variants (v_mad_u64_u32 / v_dot2_u32_u16 MDS) with partial 0 = spec-form, 1 = sparse
 * partial rounds.  With mds 2, partial 1..8 select older forms of the stage-1 MMR kernel for A/B measurements
 * (1 sparse partial rounds, 2 / 3 MDS as 4x4x4 MFMAs in all / the partial rounds, 4 one MDS layer per partial round,
 * 5 VALU MDS everywhere, 6 = 5 with the previous field multiply, 7 = the default with three partial rounds per MDS
 * application instead of four, 8 = the default with the flag-form folds in its MDS layers).  All variants are bit-identical. */
int p2mt_set_variant(int mds, int partial);
int p2mt_get_variant(int *mds, int *partial);
/* Which stage-1 kernel an MMR build uses with the current variant / environment: subtree_levels = 4|5 -> k_mmr_subtree (each lane
 * builds levels 1..subtree_levels of its own leaves), 0 -> k_mmr_tile over 2^tile_log-leaf tiles.  (bench.py labels its roofline
 * with it.) */
int p2mt_get_build_config(int *subtree_levels, int *tile_log, int *subtree_block);
/* Levels the stage-1 launch of a build of n_leaves (from an empty MMR) fuses: with per-lane subtrees (the default) the subtree size
 * adapts to the build -- 2^4 leaves per lane from 2^24 leaves up, 2^3 from 2^23, 2^2 for smaller builds: the smallest subtree that
 * still leaves 2^20 lanes (a 2^21-leaf shard of an 8-GPU strong-scaling run at 2^4 per lane is two wavefronts per SIMD running 15
 * dependent hashes each) -- unless the environment pins it (P2MT_SUBTREE=2|3|4|5).  > 0: levels; a negative value is a status
 * code (the library could not initialise), never a configuration. */
int p2mt_mmr_stage1_levels(size_t n_leaves);
/* Debug/test knob: make every wave of the mds=2 path take its exact fallback (results must not change). */
int p2mt_debug_force_fallback(int on);
/* Test hook: the next `n` device allocations made while growing an MMR handle report P2MT_ENOMEM (fault injection: a failed
 * flush / extend must leave len(), num_leaves() and every later root or proof consistent, and must be retryable). */
int p2mt_debug_fail_allocs(int n);
/* debug: per-generator completion ticks (100 MHz) of the dataflow witness interpreter; see csrc/p2mt_circuit.hip */
struct p2mt_circuit_data;
int p2mt_debug_witness_trace(struct p2mt_circuit_data *c, int enable, uint64_t *out, size_t cap, size_t *n_out);
/* Test hook: run one Goldilocks primitive of the device code on caller-supplied operands (host pointers).
 * op 0: exact reduce of the 128-bit value a[i] + b[i]*2^64;  op 1: exact fold of (b[i] & 0x3FF)*2^64 + a[i];
 * op 2: flag-form reduce (poseidon_fast::reduce128): out = value, flag_out[i] = 1 if the lane raised the sticky flag
 *       (then the value is unspecified);  op 3: a[i] * b[i] with the exact multiply;
 * op 4 / 5: loose add / sub of the LDE kernel (flag_out as in op 2);  op 6: a[i] * b[i] with gl::mul (the generic multiply of the
 * prover kernels);  op 7: gl::mul_add(a, b, c) with c = rotl(a, 17) ^ b;  op 8: the flag-form multiply of the one-hash-per-lane
 * permutation (flag_out as in op 2);  op 9: a[i] - b[i] mod p for any operands, exact;  op 10: the same with its second wrap left
 * to the flag (flag_out as in op 2);  op 11 / 12 / 13: add / sub / multiply of the transform kernels (ntt_arith.hip.h; flag_out as in
 * op 2);  op 14 / 15: the sum / the difference out of their fused butterfly;  op 16 + E, E in [0, 192): a[i] * 2^E by shifts
 * (2^96 = -1; flag_out as in op 2).  Results are canonicalised. */
int p2mt_debug_field_op(int op, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out, uint8_t *flag_out);
/* Test hook for the batched partial rounds of the one-hash-per-lane permutation (poseidon_fast::partial_rounds_g): applies group
 * `group` of the shipped schedule to n arbitrary 12-word states (n x 12 in, n x 12 canonical out).  group 0: the MDS layers of rounds
 * 3..6 with the word-0 S-boxes of rounds 4..6 between them (its input is the state BEHIND round 3's full S-box layer); 1..4: four
 * rounds starting with the word-0 S-box of round 7 / 11 / 15 / 19; 5: the three rounds 23..25.  Every group ends with the constants of
 * the following round added.  flag_out[i] = 1 if the lane raised the sticky flag (a chain of a four-round group carried out of 64
 * bits, or one of the rare borrows of the flag-form arithmetic): its value is then unspecified -- the kernels redo such a hash with
 * the exact permutation. */
int p2mt_debug_partial_group(int group, const uint64_t *states, size_t n, uint64_t *out, uint8_t *flag_out);
/* The Poseidon permutation on a HOST core (csrc/host_poseidon.hip; product code: single proves and verifications keep their
 * transcript -- a chain of dependent permutations -- there, as plonky2 keeps its Challenger): n states of 12 words, in -> out (may
 * alias), canonical output.  No device involved. */
int p2mt_host_poseidon_permute(const uint64_t *in, uint64_t *out, size_t n);
/* Test hook: the host Challenger (plonky2 iop/challenger.rs: duplex sponge, inputs absorbed 8 at a time, outputs popped from the back).
 * Phase k observes n_obs[k] elements (consecutive in `elements`), then squeezes n_sq[k] challenges (consecutive in `out`). */
int p2mt_debug_host_challenger(const uint64_t *elements, const uint32_t *n_obs, const uint32_t *n_sq, size_t n_phases, uint64_t *out);
/* Where single proves / verifications run their transcript: 1 = host core (default; env P2MT_HOST_TRANSCRIPT), 0 = device (what the
 * batched passes always do).  Results are identical; the knob exists for A/B measurements and tests. */
int p2mt_debug_host_transcript(int on);
/* Single proves with the transcript on the host also evaluate their long dependency chains of PoseidonGate rows there before the
 * launch (the recursion's outer circuit: the inner proof's 105-row transcript; csrc/p2mt_circuit.hip select_host_chain) and hand the
 * outputs down with the witness inputs: 1 = on (default; env P2MT_HOST_CHAIN), 0 = every generator on the device.  Proofs are identical. */
int p2mt_debug_host_chain(int on);
/* The generator schedule of the last prove: dependency levels on the device, PoseidonGate rows evaluated on the host. */
int p2mt_circuit_schedule_info(const struct p2mt_circuit_data *c, uint32_t *n_levels, uint32_t *host_rows);
/* One-launch tree builds (csrc/p2mt_plan.hip: stage 1 and every level above it as dependency-ordered workgroups of one grid).
 * Knobs for measurement (a negative argument keeps the current value; the environment sets the defaults: P2MT_PLAN, P2MT_PLAN_MIN_LOG,
 * P2MT_PLAN_ORDER, P2MT_PLAN_TQ, P2MT_PLAN_TW): enabled 0/1; min_log: smallest subtree (log2 leaves) that takes the one-launch path;
 * order: 0 = stage 1 first, then level by level, 1 = upper levels interleaved behind the stage-1 items they depend on; tq / tw: a level
 * with more than 2^tq nodes runs one hash per lane, down to 2^tw four lanes per hash, below one wavefront per hash. */
int p2mt_debug_plan_knobs(int enabled, int min_log, int order, int tq, int tw);
/* Record device-clock timestamps per work item in the one-launch builds this thread enqueues from now on (on != 0), and read the rows
 * of the last such launch: rows_out[item][8] = kind (0 S, 1 U, 2 Q, 3 W), level, first node, start, inputs ready, end (ticks of 10 ns
 * since the first item started), XCC id, HW_ID.  Returns the number of items of that launch (at most max_items rows are written), or a
 * negative status. */
int p2mt_debug_plan_profile(int on);
int64_t p2mt_debug_plan_profile_read(uint64_t *rows_out, size_t max_items);
/* Per-launch HIP-event timing of the dominant kernels (the fused MMR tile stage; the LDE / leaf-sponge kernels of
 * the commit step): enable, run, then read the summed duration and the number of launches recorded. */
int p2mt_profile_enable(int on);
int p2mt_profile_read(float *total_ms, int *launches);
/* HIP-event timer on the library stream (what bench.py uses for per-launch durations). */
int p2mt_timer_start(void);
int p2mt_timer_stop(float *elapsed_ms); /* records + synchronises the stop event */

/* ------------------------------------------------------------------ plonky2 Hasher, batch-shaped
 * Replaces PoseidonHash::{two_to_one, hash_or_noop, hash_no_pad} + Poseidon::poseidon
 * (plonky2 @3b21b87, absent) as called at simple_merkle_tree.rs:23,33,45,93,100,102 and
 * merkle_mountain_ranges.rs:91,96,111,125,233,238,240,249. */
int p2mt_poseidon_permute_batch(const uint64_t *in /*[n][12]*/, uint64_t *out /*[n][12]*/, size_t n);
int p2mt_poseidon_permute_batch_dev(const uint64_t *d_in, uint64_t *d_out, size_t n); /* d_in != d_out */
int p2mt_two_to_one_batch(const uint64_t *in /*[n][8] = left|right*/, uint64_t *out /*[n][4]*/, size_t n);
int p2mt_two_to_one_batch_dev(const uint64_t *d_in, uint64_t *d_out, size_t n);
/* hash_or_noop of n rows of `len` elements each (len <= 4: zero-padded copy, no permutation; else sponge) */
int p2mt_hash_or_noop_batch(const uint64_t *in /*[n][len]*/, size_t n, size_t len, uint64_t *out /*[n][4]*/);
int p2mt_hash_or_noop_batch_dev(const uint64_t *d_in, size_t n, size_t len, uint64_t *d_out);
/* hash_n_to_hash_no_pad (always the sponge, even for len <= 4) */
int p2mt_hash_no_pad_batch(const uint64_t *in /*[n][len]*/, size_t n, size_t len, uint64_t *out /*[n][4]*/);
int p2mt_hash_no_pad_batch_dev(const uint64_t *d_in, size_t n, size_t len, uint64_t *d_out);

/* Witness fill for PoseidonGate rows (what plonky2's PoseidonGenerator computes for every hash the verifier
 * circuits add: mmr_plonky2_verifier.rs:46-54,81; mmr_plonky2_verifier_1_recursion.rs:44-52), batched over rows.
 * inputs [n][12], swaps [n] (0/1); wires_out is WIRE-MAJOR [135][n] -- the column layout of the prover's wire
 * polynomials.  Wire map (plonky2 gates/poseidon.rs, from recall -- parity unpinned): 0-11 inputs, 12-23 outputs,
 * 24 swap, 25-28 delta, 29-64 first-half full-round S-box inputs (rounds 1-3), 65-86 partial-round S-box inputs,
 * 87-134 second-half full-round S-box inputs. */
int p2mt_poseidon_gate_witness_batch(const uint64_t *inputs, const uint8_t *swaps, size_t n, uint64_t *wires_out);
int p2mt_poseidon_gate_witness_batch_dev(const uint64_t *d_inputs, const uint8_t *d_swaps, size_t n,
                                         uint64_t *d_wires_out);

/* ------------------------------------------------------------------ simple_merkle_tree.rs
 * MerkleTree::build (simple_merkle_tree.rs:28-51).  n must be a power of two >= 2 (else P2MT_EINVAL:
 * log2_strict panic :30 / usize underflow :38).  levels_out is level-major and matches
 * `MerkleTree.tree`: level i has n>>i HashOuts, levels 0..log2(n)-1, (2n-2) HashOuts in total;
 * root_out gets `MerkleTree.root`. */
int p2mt_merkle_build_pow2(const uint64_t *leaves, size_t n, uint64_t *levels_out /*[(2n-2)][4]*/,
                           uint64_t *root_out /*[4]*/);
int p2mt_merkle_build_pow2_dev(const uint64_t *d_leaves, size_t n, uint64_t *d_levels_out, uint64_t *d_root_out);
/* get_merkle_proof (:55-74) / get_in_between_hashes (:76-86) on a level-major tree held by the caller
 * (pure index arithmetic + copies; host memory). */
int p2mt_merkle_get_proof(const uint64_t *levels, size_t n, size_t leaf_index, uint64_t *proof_out /*[log2 n][4]*/);
int p2mt_merkle_get_in_between_hashes(const uint64_t *levels, const uint64_t *root, size_t n, size_t leaf_index,
                                      uint64_t *out /*[log2 n][4]*/);
/* verify_merkle_proof (:91-109), batched: result_out[i] = 1/0.  All proofs have n_hashes siblings. */
int p2mt_verify_merkle_proof_batch(const uint64_t *leaves /*[m]*/, const uint64_t *leaf_indices /*[m]*/,
                                   const uint64_t *roots /*[m][4]*/, const uint64_t *hashes /*[m][n_hashes][4]*/,
                                   size_t n_hashes, size_t m, uint8_t *result_out /*[m]*/);

/* ------------------------------------------------------------------ merkle_mountain_ranges.rs
 * Index maths (pure host functions). */
uint64_t p2mt_get_heights_bitmap_for_mmr_size(size_t mmr_size, size_t *remainder_out); /* :39-81 */
int64_t p2mt_get_mmr_index(size_t leaf_normal_index); /* :257-270; P2MT_ERANGE where the i32 maths overflows */

/* Device-resident MMR: `elements` is the reference's post-order Vec<HashOut> (:8-12), kept in HBM. */
typedef struct p2mt_mmr p2mt_mmr;
int p2mt_mmr_create(p2mt_mmr **out);                    /* MMR::new (:84-86) */
int p2mt_mmr_destroy(p2mt_mmr *m);
int p2mt_mmr_reserve(p2mt_mmr *m, size_t n_leaves);     /* pre-size HBM for n_leaves (no reallocation while extending) */
int p2mt_mmr_reset(p2mt_mmr *m);                        /* back to the empty MMR, keeps the allocation */
/* MMR::add_leaf (:89-120), one leaf.  Write-combining: the leaf is queued on the host and the queue is flushed as ONE
 * bulk extend before anything observes the MMR (len/copy/peaks/root/proof/save/extend/elements_dev) or when 2^20
 * leaves are pending, so `for leaf { add_leaf }` -- how every caller in the reference builds an MMR
 * (mmr_plonky2_verifier.rs:109-112) -- runs at bulk speed with identical observable state. */
int p2mt_mmr_add_leaf(p2mt_mmr *m, uint64_t leaf);
int p2mt_mmr_flush(p2mt_mmr *m);
/* MMR::add_leaf (:89-120) for k leaves at once: identical `elements` to k successive add_leaf calls. */
int p2mt_mmr_extend(p2mt_mmr *m, const uint64_t *leaves, size_t k);
int p2mt_mmr_extend_dev(p2mt_mmr *m, const uint64_t *d_leaves, size_t k);
size_t p2mt_mmr_num_leaves(const p2mt_mmr *m);
size_t p2mt_mmr_len(const p2mt_mmr *m);                 /* elements.len() = 2N - popcount(N) */
const uint64_t *p2mt_mmr_elements_dev(const p2mt_mmr *m); /* device pointer to elements[0] */
int p2mt_mmr_copy_elements(const p2mt_mmr *m, size_t first, size_t count, uint64_t *out /*[count][4]*/);
/* Getting `elements` (1.07 GB at 2^24 leaves) to the host at the link's rate instead of a pageable copy's: page-locked host memory from
 * the library (what a Rust caller wraps in a slice), an enqueue-only copy into it (complete after p2mt_sync()), and an extend that
 * streams the elements it appends -- MMR.elements is append-only in post-order, so an extend of leaves [n0, n1) creates exactly
 * elements [len(n0), len(n1)) -- to out[0 .. 4 * (len after - len before)) chunk by chunk (2^chunk_log leaves) on the copy engines
 * while later chunks are still being hashed.  Enqueue only: the buffer is complete after p2mt_sync().  The caller keeps `out` alive
 * and does not free it before that. */
int p2mt_host_alloc_pinned(size_t bytes, void **out);
int p2mt_host_free_pinned(void *p);
int p2mt_mmr_copy_elements_async(const p2mt_mmr *m, size_t first, size_t count, uint64_t *out_pinned /*[count][4]*/);
int p2mt_mmr_extend_dev_to_host(p2mt_mmr *m, const uint64_t *d_leaves, size_t k, unsigned chunk_log, uint64_t *out_pinned);
/* MMR::get_peaks (:179-200).  P2MT_EINVAL on the empty MMR, P2MT_ERANGE for len >= 2^32 (Quirk Q6). */
int p2mt_mmr_peaks(const p2mt_mmr *m, uint64_t *peaks_out /*[<=64][4]*/, int *n_peaks);
/* MMR::bagging_the_peaks (:122-127): hash_or_noop over all peak elements (one peak => the peak). */
int p2mt_mmr_root(const p2mt_mmr *m, uint64_t *root_out /*[4]*/);
/* The same with the result left in device memory: enqueued on the library stream, nothing synchronised (what the sharded
 * build hands to the all-gather without a host round trip). */
int p2mt_mmr_root_dev(const p2mt_mmr *m, uint64_t *d_root_out /*[4], device*/);
/* MMR::get_proof (:209-223) = get_subtree_proof_elm (:147-176) + get_peaks. */
int p2mt_mmr_proof(const p2mt_mmr *m, size_t mmr_index, uint64_t *siblings_out /*[<=64][4]*/,
                   uint8_t *lefts_out /*[<=64]*/, int *n_siblings, uint64_t *peaks_out /*[<=64][4]*/, int *n_peaks,
                   size_t *mmr_size);
/* Many proofs per call: siblings_out is [m][max_siblings][4], n_siblings_out [m]. */
int p2mt_mmr_proof_batch(const p2mt_mmr *m, const uint64_t *mmr_indices, size_t count, size_t max_siblings,
                         uint64_t *siblings_out, uint8_t *lefts_out, int32_t *n_siblings_out);
/* Device-resident forms of the batched proof service: every pointer is device memory, nothing is copied or
 * synchronised.  n_siblings_out[i] = -1 marks an out-of-range index (the host form returns P2MT_EINVAL for it). */
int p2mt_mmr_proof_batch_dev(const p2mt_mmr *m, const uint64_t *d_mmr_indices, size_t count, size_t max_siblings,
                             uint64_t *d_siblings_out, uint8_t *d_lefts_out, int32_t *d_n_siblings_out);
int p2mt_mmr_proof_verify_batch_dev(const uint64_t *d_siblings, const uint8_t *d_lefts, const int32_t *d_n_siblings,
                                    size_t max_siblings, const uint64_t *d_peaks, int n_peaks, const uint64_t *d_leaves,
                                    const uint64_t *d_root, size_t m, int8_t *d_status_out);
/* MMR_proof::verify (:232-252).  *result_out = 1/0; returns P2MT_ENOTPEAK where the reference panics (:245). */
int p2mt_mmr_proof_verify(const uint64_t *siblings, const uint8_t *lefts, int n_siblings, const uint64_t *peaks,
                          int n_peaks, uint64_t leaf, const uint64_t *root, int *result_out);
/* Batched verify against one peak set/root: status_out[i] = 1 ok, 0 root mismatch, P2MT_ENOTPEAK (-5). */
int p2mt_mmr_proof_verify_batch(const uint64_t *siblings /*[m][max_siblings][4]*/, const uint8_t *lefts,
                                const int32_t *n_siblings, size_t max_siblings, const uint64_t *peaks, int n_peaks,
                                const uint64_t *leaves /*[m]*/, const uint64_t *root, size_t m,
                                int8_t *status_out /*[m]*/);

/* On-disk checkpoint of the MMR state (the reference keeps `elements` only in memory and has no serde).
 * File = 32-byte header {magic "P2MTMMR1", u64 n_leaves, u64 n_elements, u64 xor-fold checksum of the payload}
 * followed by `elements` as little-endian u64 x 4 records in post-order -- byte-identical to the reference's
 * Vec<HashOut<GoldilocksField>> contents.  load replaces the handle's state; extend continues from it. */
int p2mt_mmr_save(const p2mt_mmr *m, const char *path);
int p2mt_mmr_load(p2mt_mmr *m, const char *path);

/* ------------------------------------------------------------------ multi-GPU sharded build
 * Rank r of `world` (both powers of two) owns leaves [r*n_local, (r+1)*n_local) of a 2^k-leaf MMR and
 * builds that perfect subtree locally with p2mt_mmr_extend*.  After an all-gather of the `world`
 * 32-byte subtree roots (the only exchange; done by the host side with RCCL), this hashes the top
 * log2(world) levels.  top_nodes_out: (world-1) HashOuts, level-major bottom-up; root_out: the peak. */
int p2mt_mmr_combine_shard_roots(const uint64_t *shard_roots /*[world][4]*/, size_t world,
                                 uint64_t *top_nodes_out /*[world-1][4] or NULL*/, uint64_t *root_out /*[4]*/);
/* The same on device pointers: ONE launch for all log2(world) levels on the library stream, nothing synchronised, so a step of
 * the sharded build is build -> p2mt_mmr_root_dev -> all-gather (RCCL) -> this -> one 32-byte read-back.  world <= 1024. */
int p2mt_mmr_combine_shard_roots_dev(const uint64_t *d_shard_roots /*[world][4]*/, size_t world,
                                     uint64_t *d_top_nodes_out /*[world-1][4] or NULL*/, uint64_t *d_root_out /*[4]*/);
/* The whole sharded build behind the ABI (csrc/p2mt_sharded.hip; north_star: "a single RCCL all-gather over xGMI of the per-shard
 * subtree roots", replacing the serial loop merkle_mountain_ranges.rs:89-120 on N GPUs): one process per GPU, rank r of `world` (a
 * power of two <= 1024) owns leaves [r * n_local, (r + 1) * n_local), n_local a power of two.  A build = reset + extend of the local
 * shard + p2mt_mmr_root_dev -> ONE ncclAllGather of world x 32 bytes -> the log2(world) top levels (one launch), all enqueued on the
 * library stream; nothing visits the host until p2mt_sharded_mmr_root.  RCCL is loaded on first use (dlopen "librccl.so.1": the copy
 * the process already holds, if any); machines without it get P2MT_EHIP here and lose nothing else.
 * COMMUNICATOR OWNERSHIP.  p2mt_sharded_mmr_create takes an ncclComm_t the CALLER made (as void*; its size and rank must equal
 * world / rank; NULL allowed when world == 1): the library uses it for that one collective per build, on its own stream, and never
 * destroys or aborts it; the caller keeps it alive until p2mt_sharded_mmr_destroy and does not run other collectives on it
 * concurrently from another thread.  p2mt_sharded_mmr_create_with_id makes a communicator of the library's own (ncclCommInitRank --
 * collective: every rank calls it with the 128-byte id rank 0 got from p2mt_nccl_unique_id) and destroys it with the handle.
 * p2mt_sharded_mmr_set_exchange replaces RCCL by a host callback (an all-gather of 32 bytes over MPI, gloo, a socket: fill
 * all[world][4] from every rank's mine[4], return 0); the roots then make one host round trip per build. */
typedef struct p2mt_sharded_mmr p2mt_sharded_mmr;
typedef int (*p2mt_allgather32_fn)(void *user, const uint64_t *mine /*[4]*/, uint64_t *all /*[world][4]*/);
int p2mt_sharded_mmr_create(p2mt_sharded_mmr **out, size_t n_local, int rank, int world, void *nccl_comm);
int p2mt_nccl_unique_id(void *id_out /*128 bytes*/);
int p2mt_sharded_mmr_create_with_id(p2mt_sharded_mmr **out, size_t n_local, int rank, int world, const void *nccl_unique_id_128);
int p2mt_sharded_mmr_create_exchange(p2mt_sharded_mmr **out, size_t n_local, int rank, int world, p2mt_allgather32_fn fn, void *user);
int p2mt_sharded_mmr_set_exchange(p2mt_sharded_mmr *s, p2mt_allgather32_fn fn, void *user);
int p2mt_sharded_mmr_destroy(p2mt_sharded_mmr *s);
p2mt_mmr *p2mt_sharded_mmr_local(p2mt_sharded_mmr *s); /* this rank's shard, borrowed: proofs inside it, elements, checkpoints */
int p2mt_sharded_mmr_build_dev(p2mt_sharded_mmr *s, const uint64_t *d_local_leaves /*[n_local], device*/);
int p2mt_sharded_mmr_build(p2mt_sharded_mmr *s, const uint64_t *local_leaves /*[n_local], host*/);
int p2mt_sharded_mmr_finish(p2mt_sharded_mmr *s); /* the exchange + top levels alone, behind a shard the caller extended itself */
/* one read-back: the root of the whole MMR; optionally every shard's root and the world - 1 top nodes (level-major, bottom-up) */
int p2mt_sharded_mmr_root(p2mt_sharded_mmr *s, uint64_t *root_out /*[4]*/, uint64_t *shard_roots_out /*[world][4] or NULL*/,
                          uint64_t *top_nodes_out /*[world-1][4] or NULL*/);
/* MMR::get_proof (:209-223) for a leaf (global index) THIS rank owns: log2(n_local) siblings from the shard + log2(world) from the
 * gathered roots; the one peak is root_out.  P2MT_EINVAL for a leaf of another rank (the owner makes the < 2 KB proof). */
int p2mt_sharded_mmr_proof(p2mt_sharded_mmr *s, size_t global_leaf, uint64_t *siblings_out /*[<=64][4]*/, uint8_t *lefts_out /*[<=64]*/,
                           int *n_siblings, uint64_t *root_out /*[4]*/);
/* post-order position of the first element of shard `rank` and of top node (height h above shard roots, index j) */
size_t p2mt_mmr_shard_first_pos(size_t n_local, size_t rank);
size_t p2mt_mmr_node_pos(size_t last_leaf, unsigned height); /* 2L - popcount(L) + h */

/* ------------------------------------------------------------------ Plonky2 commit step
 * Replaces, inside CircuitData::prove (called at mmr_plonky2_verifier.rs:148 and
 * mmr_plonky2_verifier_1_recursion.rs:192,218), plonky2's PolynomialBatch::from_values/from_coeffs:
 * IFFT -> x2^rate_bits coset LDE (shift 7) -> transpose -> bit-reverse -> MerkleTree::new(cap_height),
 * and its pieces fft_with_options / ifft_with_options / MerkleTree::new. */
int p2mt_ntt_batch(uint64_t *data /*[n_polys][2^log_n], in place*/, unsigned log_n, size_t n_polys, int inverse);
int p2mt_ntt_batch_dev(uint64_t *d_data, unsigned log_n, size_t n_polys, int inverse);
/* out[j][i] = f_j(shift * w^i), w = primitive 2^(log_n+rate_bits)-th root, natural order */
int p2mt_coset_lde_batch(const uint64_t *coeffs /*[n_polys][2^log_n]*/, unsigned log_n, unsigned rate_bits,
                         uint64_t shift, size_t n_polys, uint64_t *out /*[n_polys][2^(log_n+rate_bits)]*/);
int p2mt_coset_lde_batch_dev(const uint64_t *d_coeffs, unsigned log_n, unsigned rate_bits, uint64_t shift,
                             size_t n_polys, uint64_t *d_out);
/* The same LDE in the order PolynomialBatch keeps it (and the one the kernel produces in a single pass over HBM, 8 B read +
 * 8 * 2^rate_bits B written per coefficient): d_out[j][brev(i)] = f_j(shift * w^i), i.e. the point index bit-reversed over
 * log_n + rate_bits bits, polynomial-major.  Enqueues on the library stream and returns (no synchronisation). */
int p2mt_coset_lde_leaf_order_dev(const uint64_t *d_coeffs, unsigned log_n, unsigned rate_bits, uint64_t shift,
                                  size_t n_polys, uint64_t *d_out);
/* MerkleTree::new(leaves, cap_height): n = 2^k leaves of `width` elements (row-major n x width).
 * digests_out: level-major, level 0 = leaf digests, levels 0..k-cap_height-1; may be NULL.  cap_out: 2^cap_height. */
int p2mt_merkle_cap_commit(const uint64_t *leaves, size_t n, size_t width, unsigned cap_height,
                           uint64_t *digests_out, uint64_t *cap_out);
int p2mt_merkle_cap_commit_dev(const uint64_t *d_leaves, size_t n, size_t width, unsigned cap_height,
                               uint64_t *d_digests_out, uint64_t *d_cap_out);
/* plonky2's own `MerkleTree.digests` order from the level-major digests the calls above and below return: per cap subtree the
 * recursive "left subtree || left child || right child || right subtree" order of hash/merkle_tree.rs fill_subtree (SURVEY.md
 * App. B.4) -- pair q of layer i at pair position (q << (i+1)) + 2^i - 1, which is how MerkleTree::prove indexes it -- so that a
 * patched PolynomialBatch::from_values / MerkleTree::new can fill MerkleTree { leaves, digests, cap } with one copy and no
 * re-indexing.  level_major / out: 2 * (n_leaves - 2^cap_height) HashOuts; out must not alias the input. */
int p2mt_merkle_digests_to_plonky2_layout(const uint64_t *level_major, size_t n_leaves, unsigned cap_height, uint64_t *out);
int p2mt_merkle_digests_to_plonky2_layout_dev(const uint64_t *d_level_major, size_t n_leaves, unsigned cap_height,
                                              uint64_t *d_out);
/* PolynomialBatch::from_values (is_values = 1) / from_coeffs (0).  leaves_out: [2^(log_n+rate_bits)][n_polys],
 * leaf index bit-reversed; digests_out/cap_out as above.  Any output pointer may be NULL except cap_out. */
int p2mt_polynomial_batch_commit(const uint64_t *polys /*[n_polys][2^log_n]*/, int is_values, size_t n_polys,
                                 unsigned log_n, unsigned rate_bits, unsigned cap_height, uint64_t *leaves_out,
                                 uint64_t *digests_out, uint64_t *cap_out);
int p2mt_polynomial_batch_commit_dev(const uint64_t *d_polys, int is_values, size_t n_polys, unsigned log_n,
                                     unsigned rate_bits, unsigned cap_height, uint64_t *d_leaves_out,
                                     uint64_t *d_digests_out, uint64_t *d_cap_out);

/* ------------------------------------------------------------------ Plonky2 permutation argument
 * plonky2 plonk/prover.rs all_wires_permutation_partial_products (inside CircuitData::prove, mmr_plonky2_verifier.rs:148):
 * for every challenge pair (beta, gamma) the Z polynomial and the partial products of
 *   (w_j + beta k_j x + gamma) / (w_j + beta sigma_j(x) + gamma)   over the routed wires, in chunks of `chunk`
 * (= quotient_degree_factor, 8 in standard_recursion_config: 80 routed wires -> 1 Z + 9 partial products per challenge).
 * wires, sigmas: [num_routed][2^degree_bits] columns of values on the subgroup, natural order; k_is: coset shifts of the
 * identity permutation (7^j).  out: num_challenges Z columns, then num_challenges x num_prods partial-product columns
 * (num_prods = ceil(num_routed / chunk) - 1) -- the value matrix PolynomialBatch::from_values commits next ("Z is
 * expected at the front").  P2MT_EINVAL where plonky2 panics: chunk < 2, or a zero denominator.  [parity unpinned] */
int p2mt_permutation_partial_products(const uint64_t *wires, const uint64_t *sigmas, const uint64_t *k_is,
                                      const uint64_t *betas, const uint64_t *gammas, size_t num_challenges,
                                      size_t num_routed, unsigned degree_bits, unsigned chunk, uint64_t *out);
/* d_wires, d_sigmas, d_out: device memory; k_is, betas, gammas: host arrays.  Synchronises (division check). */
int p2mt_permutation_partial_products_dev(const uint64_t *d_wires, const uint64_t *d_sigmas, const uint64_t *k_is,
                                          const uint64_t *betas, const uint64_t *gammas, size_t num_challenges,
                                          size_t num_routed, unsigned degree_bits, unsigned chunk, uint64_t *d_out);

/* ------------------------------------------------------------------ Plonky2 opening proof (challenger + FRI)
 * The last stage of CircuitData::prove (mmr_plonky2_verifier.rs:148, mmr_plonky2_verifier_1_recursion.rs:192,218):
 * plonky2's iop/challenger.rs Challenger, OpeningSet evaluation (plonk/proof.rs eval_all) and
 * PolynomialBatch::prove_openings -> fri_proof (fri/oracle.rs, fri/prover.rs): alpha-composition of the opened
 * polynomials, division by (X - z), commit phase (fold by 2^arity_bits, coset FFT, Merkle cap per layer), proof-of-work
 * grind, query openings.  Extension elements (F[X]/(X^2 - 7)) are 2 consecutive words (a + bX).  Everything runs on the
 * device; the challenger state lives in device memory so that challenges feed the next kernel without a host round trip.
 * [parity unpinned: plonky2's source is not part of the reference tree; the tests check against a CPU restatement.] */
typedef struct p2mt_challenger p2mt_challenger;
int p2mt_challenger_create(p2mt_challenger **out);  /* Challenger::new(): zero sponge, empty buffers */
int p2mt_challenger_destroy(p2mt_challenger *c);
int p2mt_challenger_clone(const p2mt_challenger *src, p2mt_challenger **out);
/* observe_elements / observe_hash / observe_cap / observe_extension_elements: all are element streams */
int p2mt_challenger_observe(p2mt_challenger *c, const uint64_t *elements, size_t n);
int p2mt_challenger_observe_dev(p2mt_challenger *c, const uint64_t *d_elements, size_t n); /* enqueues, returns */
/* get_n_challenges (an extension challenge = 2 of them, in this order) */
int p2mt_challenger_get_challenges(p2mt_challenger *c, size_t n, uint64_t *out);
int p2mt_challenger_get_challenges_dev(p2mt_challenger *c, size_t n, uint64_t *d_out); /* enqueues, returns */
/* observe + squeeze in one launch (the duplex the prover performs after every commitment), and a reset to Challenger::new() */
int p2mt_challenger_reset(p2mt_challenger *c);
int p2mt_challenger_duplex_dev(p2mt_challenger *c, const uint64_t *d_elements, size_t n_obs, uint64_t *d_out, size_t n_out);
/* the same from a fresh transcript (Challenger::new() + observe + squeeze in one launch) */
int p2mt_challenger_restart_duplex_dev(p2mt_challenger *c, const uint64_t *d_elements, size_t n_obs, uint64_t *d_out,
                                       size_t n_out);
/* state words for tests/checkpoints: sponge_state[12] | input_buffer[8] | output_buffer[8] | n_in | n_out */
#define P2MT_CHALLENGER_STATE_WORDS 30
int p2mt_challenger_get_state(const p2mt_challenger *c, uint64_t *out);
int p2mt_challenger_set_state(p2mt_challenger *c, const uint64_t *in);

/* PolynomialCoeffs::eval of base-field polynomials at one extension point: out[j] = f_j(point), 2 words each */
int p2mt_eval_polys_ext(const uint64_t *coeffs /*[n_polys][2^log_n]*/, size_t n_polys, unsigned log_n,
                        const uint64_t point[2], uint64_t *out /*[n_polys][2]*/);
int p2mt_eval_polys_ext_dev(const uint64_t *d_coeffs, size_t n_polys, unsigned log_n, const uint64_t point[2],
                            uint64_t *d_out);

/* FriParams (fri/mod.rs): FriConfig + degree_bits + reduction_arity_bits; hiding = false (zero_knowledge off in
 * standard_recursion_config, mmr_plonky2_verifier.rs:30) */
typedef struct p2mt_fri_params {
  uint32_t degree_bits, rate_bits, cap_height, proof_of_work_bits, num_query_rounds, num_reductions;
  uint32_t reduction_arity_bits[8]; /* each in 1..4 */
} p2mt_fri_params;
/* standard_recursion_config: rate_bits 3, cap_height 4, PoW 16 bits, 28 queries, ConstantArityBits(4, 5) */
int p2mt_fri_params_standard(unsigned degree_bits, p2mt_fri_params *out);
/* One committed PolynomialBatch, as p2mt_polynomial_batch_commit* leaves it: coefficients poly-major
 * [n_polys][2^degree_bits]; leaves [N][n_polys] (N = 2^(degree_bits+rate_bits), bit-reversed point order); digests
 * level-major.  Host pointers for p2mt_fri_prove_openings, device pointers for the _dev form. */
typedef struct p2mt_fri_oracle {
  const uint64_t *coeffs, *leaves, *digests;
  uint64_t n_polys;
} p2mt_fri_oracle;
/* FriBatchInfo: one opening point and the polynomials opened there as (oracle index, polynomial index) pairs */
typedef struct p2mt_fri_batch {
  uint64_t point[2];
  const uint32_t *polys; /* host memory, 2 * n_polys entries */
  uint64_t n_polys;
} p2mt_fri_batch;
/* FriOpenings (OpeningSet::to_fri_openings): the value of every batch's polynomials at the batch's point, in batch
 * order, 2 words each -- what the challenger observes before prove_openings and what the verifier is given.
 * Only `coeffs` and `n_polys` of the oracles are read.  One launch for the whole set. */
int p2mt_fri_openings(const p2mt_fri_oracle *oracles, size_t n_oracles, const p2mt_fri_batch *batches,
                      size_t n_batches, unsigned degree_bits, uint64_t *out /*[sum n_polys][2]*/);
int p2mt_fri_openings_dev(const p2mt_fri_oracle *d_oracles, size_t n_oracles, const p2mt_fri_batch *batches,
                          size_t n_batches, unsigned degree_bits, uint64_t *d_out);
/* FriProof as words, plonky2's serialisation order:
 *   commit_phase_merkle_caps [num_reductions][2^cap_height][4]
 *   query_round_proofs [num_query_rounds] x { per oracle: leaf row [n_polys] | siblings [log N - cap_height][4];
 *                                             per layer: evals [2^arity_bits][2] | siblings [..][4] }
 *   final_poly [2^(degree_bits - sum arity_bits)][2]
 *   pow_witness [1]  -- the SMALLEST valid witness (plonky2 takes any: parallel find_any, non-deterministic) */
size_t p2mt_fri_proof_len(const p2mt_fri_params *params, size_t n_oracles, const uint64_t *n_polys);
/* PolynomialBatch::prove_openings(instance, oracles, challenger, fri_params).  The challenger must have observed
 * everything up to and including the openings; it is advanced exactly as plonky2's is (alpha, per-layer cap/beta,
 * final poly, PoW witness/response, query indices).  Synchronises the library stream before returning. */
int p2mt_fri_prove_openings(const p2mt_fri_oracle *oracles, size_t n_oracles, const p2mt_fri_batch *batches,
                            size_t n_batches, const p2mt_fri_params *params, p2mt_challenger *challenger,
                            uint64_t *proof_out);
int p2mt_fri_prove_openings_dev(const p2mt_fri_oracle *d_oracles, size_t n_oracles, const p2mt_fri_batch *batches,
                                size_t n_batches, const p2mt_fri_params *params, p2mt_challenger *challenger,
                                uint64_t *d_proof_out);

/* ------------------------------------------------------------------ Plonky2 CircuitBuilder / CircuitData::prove
 * What the reference does through plonky2 at mmr_plonky2_verifier.rs:13-91 (verify_mmr_proof_circuit: CircuitBuilder calls,
 * with the gadgets of src/mmr/common.rs:5-58 on top), :122-146 (PartialWitness) and :148 (circuit_data.prove(pw)), under
 * CircuitConfig::standard_recursion_config() (:30): 135 wires, 80 routed, 2 constants, 2 challenges, quotient degree factor 8,
 * FRI rate 1/8, cap height 4, 16-bit proof of work, 28 queries, zero-knowledge off.  Supported gate set: NoopGate,
 * ConstantGate, PublicInputGate, ArithmeticGate, PoseidonGate -- everything those circuits instantiate.  The builder and the
 * generator schedule are host code; witness fill, the three commitments, the permutation argument, the quotient
 * polynomials, the openings and the FRI proof run on the device, chained on the library stream.
 * [parity unpinned: plonky2's source is not part of the reference tree; checked bit for bit against the tests' CPU
 *  restatement, whose verifier must accept the proof.]
 * A Target is an opaque 64-bit handle (plonky2 Target::{Wire, VirtualTarget}); BoolTarget / HashOutTarget are 1 / 4 of them. */
typedef uint64_t p2mt_target;
typedef struct p2mt_circuit_builder p2mt_circuit_builder; /* CircuitBuilder<GoldilocksField, 2> */
typedef struct p2mt_circuit_data p2mt_circuit_data;       /* CircuitData<GoldilocksField, PoseidonGoldilocksConfig, 2> */
typedef struct p2mt_partial_witness p2mt_partial_witness; /* PartialWitness<GoldilocksField> */
int p2mt_cb_create(p2mt_circuit_builder **out);           /* CircuitBuilder::new(standard_recursion_config()) (:31) */
int p2mt_cb_destroy(p2mt_circuit_builder *b);
int p2mt_cb_add_virtual_target(p2mt_circuit_builder *b, p2mt_target *out);            /* :33; add_virtual_hash = 4 of them (:41,:66) */
int p2mt_cb_add_virtual_bool_target_safe(p2mt_circuit_builder *b, p2mt_target *out);  /* :42 (virtual target + assert_bool) */
int p2mt_cb_constant(p2mt_circuit_builder *b, uint64_t c, p2mt_target *out);          /* zero() / one() (:76) */
int p2mt_cb_connect(p2mt_circuit_builder *b, p2mt_target x, p2mt_target y);           /* :77; P2MT_EINVAL for an unroutable wire */
/* const_0 * multiplicand_0 * multiplicand_1 + const_1 * addend (gadgets/arithmetic.rs), with plonky2's constant folding,
 * operation cache and ArithmeticGate slot packing (20 operations per row, one row per constant pair) */
int p2mt_cb_arithmetic(p2mt_circuit_builder *b, uint64_t const_0, uint64_t const_1, p2mt_target multiplicand_0,
                       p2mt_target multiplicand_1, p2mt_target addend, p2mt_target *out);
int p2mt_cb_add(p2mt_circuit_builder *b, p2mt_target x, p2mt_target y, p2mt_target *out);
int p2mt_cb_sub(p2mt_circuit_builder *b, p2mt_target x, p2mt_target y, p2mt_target *out);
int p2mt_cb_mul(p2mt_circuit_builder *b, p2mt_target x, p2mt_target y, p2mt_target *out);                     /* common.rs:48-51 */
int p2mt_cb_mul_add(p2mt_circuit_builder *b, p2mt_target x, p2mt_target y, p2mt_target z, p2mt_target *out);  /* common.rs:52-55 */
int p2mt_cb_mul_sub(p2mt_circuit_builder *b, p2mt_target x, p2mt_target y, p2mt_target z, p2mt_target *out);
int p2mt_cb_not(p2mt_circuit_builder *b, p2mt_target x, p2mt_target *out);                                    /* common.rs:46 */
int p2mt_cb_or(p2mt_circuit_builder *b, p2mt_target b1, p2mt_target b2, p2mt_target *out);                    /* common.rs:13-15,25 */
int p2mt_cb_assert_bool(p2mt_circuit_builder *b, p2mt_target x);
int p2mt_cb_is_equal(p2mt_circuit_builder *b, p2mt_target x, p2mt_target y, p2mt_target *out);                /* common.rs:9-12 */
/* hash_n_to_hash_no_pad::<PoseidonHash> (:81): one PoseidonGate row per 8 inputs; hash_or_noop (:34,:46-54): <= 4 inputs
 * are padded with zero() and no gate is added.  out: 4 targets. */
int p2mt_cb_hash_n_to_hash_no_pad(p2mt_circuit_builder *b, const p2mt_target *inputs, size_t n, p2mt_target *out);
int p2mt_cb_hash_or_noop(p2mt_circuit_builder *b, const p2mt_target *inputs, size_t n, p2mt_target *out);
int p2mt_cb_register_public_inputs(p2mt_circuit_builder *b, const p2mt_target *targets, size_t n);           /* :83,:86 */
size_t p2mt_cb_num_gates(const p2mt_circuit_builder *b);
/* builder.build::<PoseidonGoldilocksConfig>() (:89): public-input hash + PublicInputGate, ConstantGates, padding to a power
 * of two with NoopGates, selector / constant / sigma polynomials and their commitment on the device, circuit digest.
 * The builder must not be used afterwards (plonky2's build consumes it); destroy it.  P2MT_EINVAL for more than 2^12 rows. */
/* Limits (both are what the reference's circuits need, not a property of the method): at most 2^12 rows after padding
 * (P2MT_EINVAL beyond; the outer circuit of mmr_plonky2_verifier_1_recursion is 2^12), and ONE level of recursion -- see
 * p2mt_cb_verify_proof. */
int p2mt_cb_build(p2mt_circuit_builder *b, p2mt_circuit_data **out);
int p2mt_circuit_destroy(p2mt_circuit_data *c);
typedef struct p2mt_circuit_info {
  uint32_t degree_bits, num_gate_types, num_selectors, num_constants_sigmas, num_public_inputs, num_partial_products;
  /* rows per gate type, indexed by gate kind: 0 Noop, 1 Constant, 2 PublicInput, 3 Arithmetic, 4 Poseidon, and the gate types of
   * the in-circuit verifier: 5 BaseSum, 6 ArithmeticExtension, 7 MulExtension, 8 Reducing, 9 ReducingExtension, 10 RandomAccess,
   * 11 CosetInterpolation, 12 PoseidonMds */
  uint32_t gate_counts[16];
  uint32_t gate_kinds[16], gate_selector[16], group_start[16], group_end[16]; /* sorted gate types and their selector groups */
  uint64_t proof_len, fri_proof_len; /* words */
} p2mt_circuit_info;
int p2mt_circuit_get_info(const p2mt_circuit_data *c, p2mt_circuit_info *info);
int p2mt_circuit_public_inputs(const p2mt_circuit_data *c, p2mt_target *out); /* circuit_data.prover_only.public_inputs (:137) */
/* constants_sigmas values [num_constants_sigmas][2^degree_bits] (selectors | constants | sigmas), the commitment's cap [16][4]
 * and circuit_digest [4]; any pointer may be NULL */
int p2mt_circuit_constants_sigmas(const p2mt_circuit_data *c, uint64_t *values_out, uint64_t *cap_out, uint64_t *digest_out);
int p2mt_pw_create(p2mt_partial_witness **out);  /* PartialWitness::new() (:123) */
int p2mt_pw_destroy(p2mt_partial_witness *pw);
int p2mt_pw_clear(p2mt_partial_witness *pw);
int p2mt_pw_set_target(p2mt_partial_witness *pw, p2mt_target t, uint64_t value); /* set_target / set_bool_target; set_hash_target = 4 calls (:126-146) */
/* generate_partial_witness + full_witness only: wires_out [135][2^degree_bits].  P2MT_EINVAL where plonky2 panics: a target
 * set twice with different values (a witness that contradicts the circuit), generators that cannot run. */
int p2mt_circuit_generate_witness(p2mt_circuit_data *c, const p2mt_partial_witness *pw, uint64_t *wires_out);
/* circuit_data.prove(pw) (:148).  proof_out (p2mt_circuit_info.proof_len words), plonky2's serialisation order:
 *   wires_cap [16][4] | plonk_zs_partial_products_cap [16][4] | quotient_polys_cap [16][4]
 *   OpeningSet, 2 words each: constants [num_selectors + 2] | plonk_sigmas [80] | wires [135] | plonk_zs [2] | plonk_zs_next [2]
 *                             | partial_products [18] | quotient_polys [16]
 *   FriProof (layout at p2mt_fri_proof_len) | public_inputs [num_public_inputs]
 * Deterministic (the PublicInputGate's unused wires stay zero where plonky2 randomises them; smallest proof-of-work witness). */
int p2mt_circuit_prove(p2mt_circuit_data *c, const p2mt_partial_witness *pw, uint64_t *proof_out, size_t proof_cap);
/* n proves spread over n_handles worker threads (one per handle, each on its own stream): handles must be distinct builds of
 * the same circuit; witnesses[i] -> proofs_out + i * proof_stride (proof_stride >= proof_len words).  status_out[i] (may be
 * NULL) = status of prove i; returns the first non-zero status.  For the best rate set GPU_MAX_HW_QUEUES=32 and
 * hipSetDeviceFlags(hipDeviceScheduleBlockingSync) before the first HIP call of the process. */
int p2mt_circuit_prove_many(p2mt_circuit_data *const *circuits, size_t n_handles, const p2mt_partial_witness *const *witnesses,
                            size_t n, uint64_t *proofs_out, size_t proof_stride, int *status_out);
/* ProofWithPublicInputs::to_bytes() / from_bytes() in plonky2's Buffer order (util/serialization.rs @3b21b87d; SURVEY.md
 * App. B.5; recalled -- parity unpinned): field elements as 8 little-endian bytes, extension elements as two of them, caps /
 * openings / evaluations / final polynomial without length prefixes, every MerkleProof as one length byte + its sibling hashes.
 * = the words of p2mt_circuit_prove plus one byte in front of each of the 28 x (4 + layers) Merkle paths.  Host only.
 * from_bytes rejects a wrong total length, a path length that does not match the circuit and non-canonical elements. */
size_t p2mt_proof_bytes_len(const p2mt_circuit_data *c);
int p2mt_proof_to_bytes(const p2mt_circuit_data *c, const uint64_t *proof, size_t proof_len, uint8_t *bytes_out, size_t bytes_cap);
int p2mt_proof_from_bytes(const p2mt_circuit_data *c, const uint8_t *bytes, size_t n_bytes, uint64_t *proof_out, size_t proof_cap);
/* Batched prover: up to `batch` proofs of ONE circuit per pass of the pipeline, the proof index riding in a grid dimension of
 * every launch (a pass costs the ~44 dispatch packets of one proof; the per-proof path above is bound by the device's packet
 * rate when many provers run).  Any circuit p2mt_cb_build accepts: the reference's MMR-verifier circuits
 * (mmr_plonky2_verifier.rs:148), and both the inner and the outer prove of mmr_plonky2_verifier_1_recursion.rs:192,218.
 * Proofs are bit-identical to p2mt_circuit_prove's.  The prover borrows the circuit handle: one thread at a time, and the
 * circuit must outlive it.  Device memory, allocated on the first prove: per proof of the batch ~2 MB for a 64-row circuit,
 * ~115 MB for the 2^12-row outer recursion circuit.
 * witnesses[i] -> proofs_out + i * proof_stride for i < n (any n: passes of up to `batch`); every witness must set the same
 * targets in the same order.  status_out[i] (may be NULL) = status of proof i; returns the first non-zero status.  A statement
 * whose status is non-zero (a witness that contradicts the circuit, a zero denominator, zeta in the subgroup: where plonky2 panics or
 * returns Err) leaves ALL-ZERO words in its slot of proofs_out, never a well-formed-looking proof; the same holds for
 * p2mt_circuit_prove's proof_out. */
typedef struct p2mt_batch_prover p2mt_batch_prover;
int p2mt_batch_prover_create(p2mt_circuit_data *c, size_t batch, p2mt_batch_prover **out);
int p2mt_batch_prover_destroy(p2mt_batch_prover *b);
int p2mt_batch_prover_prove(p2mt_batch_prover *b, const p2mt_partial_witness *const *witnesses, size_t n, uint64_t *proofs_out,
                            size_t proof_stride, int *status_out);
size_t p2mt_batch_prover_batch(const p2mt_batch_prover *b);
/* circuit_data.verify(proof) (:150).  The transcript (Challenger) and every Merkle path (28 queries x (4 oracle rows + one
 * coset per FRI layer), one wavefront each) run on the device; the field arithmetic (vanishing polynomial at zeta, FRI
 * folding) is a few thousand extension-field multiplications on the host.  Returns 0 with *accepted = 1/0 and *reason (may
 * be NULL): 0 ok, 10 malformed (length / non-canonical word), 11 vanishing polynomial != Z_H * quotient at zeta, 1 proof of
 * work, 2 Merkle proof of an oracle row, 4 Merkle proof of a FRI layer, 3 inconsistent layer value, 5 final polynomial. */
int p2mt_circuit_verify(p2mt_circuit_data *c, const uint64_t *proof, size_t proof_len, int *accepted, int *reason);
/* circuit_data.verify for n proofs of this circuit (proofs[i] at proofs + i * proof_stride words): passes of up to 256 proofs with
 * the proof index in a grid dimension of every launch: one transcript replay, the row sponges and the vanishing-polynomial check
 * beside it, the FRI arithmetic and the path folds behind it, one small copy back per pass (csrc/p2mt_verify_dev.hip; nothing of a
 * verification runs on the host).  accepted[i] / reason[i] as p2mt_circuit_verify. */
int p2mt_circuit_verify_batch(p2mt_circuit_data *c, const uint64_t *proofs, size_t n, size_t proof_stride, int *accepted, int *reason);
/* ---- the outer circuit of the recursion (mmr_plonky2_verifier_1_recursion.rs:84-140): plonky2's in-circuit verifier.
 * `inner` stands for `inner_circuit_data.common` (and `.verifier_only` for the witness): the built inner circuit.
 * A ProofWithPublicInputsTarget is a flat array of proof_len targets (the inner circuit's p2mt_circuit_info), one per word of the proof
 * p2mt_circuit_prove writes (same order); a VerifierCircuitTarget is 68 targets: constants_sigmas_cap [16][4], circuit_digest [4].
 *   builder.add_virtual_proof_with_pis(&common) (:95), builder.add_virtual_verifier_data(cap_height) (:98),
 *   builder.verify_proof::<PoseidonGoldilocksConfig>(&proof, &verifier_data, &common) (:101-104): public-input hash, the whole
 *   transcript (RecursiveChallenger), the vanishing polynomial at zeta (the inner gates' constraints evaluated with extension
 *   arithmetic gates and one PoseidonMdsGate per round), quotient recombination, FRI: proof-of-work range check, 28 query rounds
 *   of Merkle paths (PoseidonGate with swap + RandomAccessGate cap lookup), reduced openings (ReducingGates), coset interpolation.
 * The inner circuit may contain the gate types Noop / Constant / PublicInput / Arithmetic / Poseidon (what the reference's inner
 * circuits use); P2MT_EINVAL otherwise.  public_inputs of the proof target = its last num_public_inputs targets. */
int p2mt_cb_add_virtual_proof_with_pis(p2mt_circuit_builder *b, const p2mt_circuit_data *inner, p2mt_target *out, size_t out_len);
int p2mt_cb_add_virtual_verifier_data(p2mt_circuit_builder *b, unsigned cap_height, p2mt_target *out /*[68]*/);
int p2mt_cb_verify_proof(p2mt_circuit_builder *b, const p2mt_target *proof_with_pis, size_t proof_len,
                         const p2mt_target *verifier_data /*[68]*/, const p2mt_circuit_data *inner);
/* pw.set_proof_with_pis_target(&target, &proof) (:201) and pw.set_verifier_data_target(&target, &inner.verifier_only) (:202) */
int p2mt_pw_set_proof_with_pis_target(p2mt_partial_witness *pw, const p2mt_target *proof_target, const uint64_t *proof_words,
                                      size_t proof_len);
int p2mt_pw_set_verifier_data_target(p2mt_partial_witness *pw, const p2mt_target *verifier_data_target,
                                     const p2mt_circuit_data *inner);
/* intermediates of the last prove (parity tests): 0 wires [135][n], 1 Z | partial products [20][n] (values), 2 quotient chunks
 * [16][n] (coefficients), 3 challenges {betas[2], gammas[2], alphas[2], zeta[2]}, 4 public_inputs_hash [4] */
int p2mt_circuit_prove_trace(const p2mt_circuit_data *c, int what, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif /* P2MT_H */
