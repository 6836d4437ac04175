// runtime.h -- host-side runtime state shared by the translation units of libp2mt_hip.so.
// One process drives one GPU (one process per GPU); state is process-global.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include <exception>
#include <new>

#include "../../include/p2mt.h"

namespace host_poseidon {
struct Challenger;
}

namespace p2mt {

// What every hashing kernel needs besides its data: the round-constant table in global memory (wide scalar
// loads) and a debug knob that forces the fast path's exact fallback (tests exercise both paths with it).
struct PermCtx {
  const uint64_t* rc;
  uint64_t force_fallback;
  // tables of the 12-lane layout's batched partial rounds (permute_wave_impl): [7][14] u64 addends, then [14][14] u32 per-lane rows
  // (poseidon_fast.hip.h kP3K, kP3W); global memory, or the workgroup's LDS copy behind stage_round_constants()
  const uint64_t* w3;
};

// The library stream is per host thread: a thread that called p2mt_thread_stream_create() (or p2mt_set_stream) enqueues on
// its own stream, every other thread on the default stream.  Together with per-thread scratch this makes distinct handles
// usable from distinct threads concurrently (one prover per thread, SURVEY.md 8e "replicas").
struct StreamRef {
  operator hipStream_t() const;
  StreamRef& operator=(hipStream_t s);
};

struct Runtime {
  bool initialised = false;
  int device = 0;
  StreamRef stream;
  int mds = 2;      // 0 = v_mad_u64_u32 MDS, 1 = v_dot2_u32_u16 MDS (exact variants), 2 = issue-optimised fast path
  int partial = 0;  // exact variants only: 0 = spec-form partial rounds, 1 = sparse form
  uint64_t* d_rc = nullptr;  // 360 round constants, device global memory
  int force_fallback = 0;
  int use_quad = 1;          // four-lanes-per-hash kernels for 2^12 < items <= 2^16 (env P2MT_QUAD=0 disables)
  unsigned subtree_block = 256; // workgroup size of the per-lane subtree kernel (env P2MT_SUBTREE_BLOCK=64|128|256)
  unsigned subtree_levels = 4;  // stage 1 as per-lane subtrees of 2^L leaves (env P2MT_SUBTREE=2|3|4|5 pins L); 0 = fused tiles
  bool subtree_auto = true;     // no P2MT_SUBTREE in the environment: L adapts to the size of the build (subtree_levels_for)
  int throughput = 0;        // p2mt_set_throughput_mode: prefer lane-efficient layouts over the latency-optimised ones
  int use_lde12 = 2;         // register-blocked 2^12 LDE kernel: 2 = round-4 form (shift twiddles), 1 = round-3 form, 0 = generic radix-2 (env P2MT_LDE12)
  unsigned tile_log = 10;    // fused MMR stage: 2^tile_log inputs per workgroup (env P2MT_TILE_LOG = 9|10|11)
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  // per-kernel HIP-event profiling of the dominant launches (p2mt_profile_*): pairs recorded around each launch
  bool profile = false;
  static constexpr int kMaxProf = 256;
  hipEvent_t prof_ev[2 * kMaxProf] = {};
  int prof_n = 0;
};
char* err_buf();  // 512 bytes, per thread
constexpr size_t kErrLen = 512;

Runtime& rt();
int fail_hip(hipError_t e, const char* what, const char* file, int line);
int fail(int code, const char* msg);
int ensure_init();
// per-lane subtree size (log2 leaves) of the stage-1 launch over n_leaves leaves: the largest L in 2..4 that still gives every SIMD
// of the chip ~4 wavefronts (n_leaves / 2^L lanes >= 4 x 1024 SIMDs x 64), the pinned value when the environment sets one
unsigned subtree_levels_for(size_t n_leaves);
// No exception crosses the C ABI: every `extern "C" int` body runs inside this guard.  std::bad_alloc (a std::vector
// growing inside the builder, the batch prover's staging, a worker pool) becomes P2MT_ENOMEM, anything else P2MT_EINVAL with its
// what() as the message.
template <typename F>
inline int abi_guard(F&& f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(P2MT_ENOMEM, "out of host memory (std::bad_alloc inside the library)");
  } catch (const std::exception& e) {
    return fail(P2MT_EINVAL, e.what());
  } catch (...) {
    return fail(P2MT_EINVAL, "unexpected C++ exception inside the library");
  }
}
// record an event on the library stream if profiling is on (slot = 2*i for start, 2*i+1 for stop)
int prof_begin();
void prof_end(int slot);
PermCtx perm_ctx();  // (runtime.hip: needs the table layout of poseidon_fast.hip.h)

// Grow-only device scratch slots for the commit pipeline: no hipMalloc/hipFree (and so no implicit device
// synchronisation) on the steady-state path; all users run on the one library stream, in order.
enum ScratchSlot { kScratchCoeffs = 0, kScratchLde, kScratchLevel0, kScratchPing, kScratchTmp, kScratchFri, kScratchPlonk, kScratchCircuit, kScratchTables, kScratchPtrs, kScratchPoints, kScratchPlan, kScratchCount };
int scratch_get(int slot, size_t bytes, void** out);
// the same, but never from a batch's per-proof arena: tables the host uploads once for all the proofs of a batch
int scratch_get_shared(int slot, size_t bytes, void** out);
// largest request per slot since the last scratch_track_reset() of this thread (sizes the batched prover's per-proof arena)
void scratch_track_reset();
size_t scratch_track_max(int slot);

// Batched pipelines (p2mt_batch_prover): B independent proofs ride in grid dimension z of every launch.  Each proof owns one
// block of `stride` bytes; the host enqueues the pipeline ONCE with the pointers of block 0, and a kernel moves every pointer that
// lies inside block 0 (base <= p < base + span) to its own block (device helper bp() in tree_common.hip.h).  Pointers outside
// block 0 (circuit constants, twiddles, round constants) are shared by all proofs.  B = 1 / span = 0 outside a batch: bp() is
// the identity.  Per host thread, like the stream and the scratch.
struct BatchArg {
  uint64_t base, span, stride;  // bytes
};
struct BatchCtx {
  unsigned B = 1;
  BatchArg arg{0, 0, 0};
  // scratch_get() inside a batch is served from block 0's arena
  char* arena = nullptr;
  size_t slot_off[kScratchCount] = {}, slot_cap[kScratchCount] = {};
};
BatchCtx& batch();
inline unsigned batch_B() { return batch().B; }
// copy [p, p + bytes) of block 0 into every other block (tables the host uploaded once); no-op outside a batch
int batch_broadcast(void* p, size_t bytes);
// device-to-device copy / fill on the library stream that follows the batch: every proof's block when dst / src are per-proof
// buffers (bytes a multiple of 8), one plain hipMemcpyAsync / hipMemsetAsync outside a batch
int batch_copy(void* dst, const void* src, size_t bytes);
int batch_fill(void* dst, int byte_value, size_t bytes);
uint64_t scratch_epoch();  // changes whenever this thread's scratch buffers were (re)allocated or freed: content caches key on it
void scratch_release_thread();  // frees the calling thread's scratch (threads that end must call it)

// RAII: run a scope on another stream, restore the library stream afterwards
struct StreamScope {
  hipStream_t saved;
  explicit StreamScope(hipStream_t s) : saved(rt().stream) { rt().stream = s; }
  ~StreamScope() { rt().stream = saved; }
};

// RAII device scratch buffer
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    if (bytes == 0) bytes = 8;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      p = nullptr;
      snprintf(err_buf(), kErrLen, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      return P2MT_ENOMEM;
    }
    return P2MT_OK;
  }
  template <typename T>
  T* as() {
    return static_cast<T*>(p);
  }
};

// Host link (runtime.hip): what a single prove's host-side transcript (host_poseidon.h) talks to the device through.  fetch: a small
// kernel behind the producer copies `n` words into mapped pinned host memory and raises a sequence word at system scope; the host spins
// on that word (no hipStreamSynchronize: ~2-3 us from the producer's last store instead of ~20).  put: up to 32 words travel as kernel
// arguments and a one-wave kernel stores them where the consumers read (no staging copy, ~5 us from enqueue to visible).
struct HostLink;
int hostlink_create(HostLink** out, size_t max_words);
void hostlink_destroy(HostLink* l);
// enqueue the copy of d_src[0..n) on rt().stream and wait for it; *h_out points into the link's buffer (valid until the next fetch)
int hostlink_fetch(HostLink* l, const uint64_t* d_src, size_t n, const uint64_t** h_out);
// two sources in one go (n0 + n1 <= max_words); *h_out = [src0 words | src1 words]
int hostlink_fetch2(HostLink* l, const uint64_t* d_src0, size_t n0, const uint64_t* d_src1, size_t n1, const uint64_t** h_out);
int hostlink_put(const uint64_t* vals, size_t n, uint64_t* d_dst);  // n <= 32

// p2mt_plan.hip: a whole perfect subtree of 2^H leaves in ONE launch (stage 1 and every level above it as dependency-ordered
// workgroups of one grid).  Where the nodes go: kind 0 = MMR.elements (post-order; `base` = element 0, node indices are global),
// kind 1 = MerkleTree.tree of an n-leaf tree (level-major; `base` = leaf digest 0, the root goes to `root`).
struct TreeLayout {
  int kind;
  uint64_t* base;
  uint64_t* root;
  uint64_t n;
};
bool tree_plan_wanted(unsigned H);  // knobs (env P2MT_PLAN*, p2mt_debug_plan_knobs) and the Poseidon variant allow it for 2^H leaves
// enqueue on rt().stream; d_leaves[0] is leaf `first_leaf` (a multiple of 2^H).  d_state: tree_plan_state_bytes(H) bytes of device
// memory owned by the caller, or null for the calling thread's scratch; after the launch word 1 is non-zero if a hand-off poll gave up
// (cannot happen with a valid plan; a caller that keeps the buffer reports P2MT_EHIP when the tree is next read)
int tree_plan_state_bytes(unsigned H, size_t* bytes_out);
int tree_plan_launch(const TreeLayout& lay, const uint64_t* d_leaves, size_t first_leaf, unsigned H, uint32_t* d_state);

// launchers exported by p2mt_hash.hip to the other translation units (enqueue on rt().stream, device pointers)
int launch_hash_rows_dev(const uint64_t* d_in, size_t n, size_t len, int noop_short, uint64_t* d_out);
int launch_merkle_level_dev(const uint64_t* d_in, uint64_t* d_out, size_t n_out);
// exported by p2mt_commit.hip: x2^rate_bits coset LDE (log_n <= 12) into leaf order, poly-major:
// d_out[p][brev(i)] = f_p(shift * w_N^i)
int coset_lde_leaf_order_dev(const uint64_t* d_coeffs, unsigned log_n, unsigned rate_bits, uint64_t shift, size_t n_polys,
                             uint64_t* d_out);

// exported by p2mt_commit.hip: PolynomialBatch::from_values / from_coeffs keeping every intermediate the prover needs later.
// d_coeffs_out [n_polys][n] (coefficients; with is_values = 0 the input IS the coefficient array and this may be null),
// d_lde_out [n_polys][8n] poly-major in leaf order (the quotient kernel reads it), d_leaves_out [8n][n_polys] (FRI
// queries), d_digests_out level-major, d_cap_out.  Null outputs go to grow-only scratch (or are skipped).
int commit_batch_dev(const uint64_t* d_polys, int is_values, size_t n_polys, unsigned log_n, unsigned rate_bits,
                     unsigned cap_height, uint64_t* d_coeffs_out, uint64_t* d_lde_out, uint64_t* d_leaves_out,
                     uint64_t* d_digests_out, uint64_t* d_cap_out);
// exported by p2mt_commit.hip: PolynomialValues::coset_ifft(shift) of n_polys rows of 2^log_n values given in natural
// order (d_vals is used as workspace); coefficients in natural order to d_coeffs_out.
int coset_ifft_dev(uint64_t* d_vals, unsigned log_n, size_t n_polys, uint64_t shift, uint64_t* d_coeffs_out);
// exported by p2mt_plonk.hip: Z and partial products with every operand in device memory, nothing synchronised;
// *d_zero_den is set to 1 on a zero denominator (where plonky2 panics).  d_k_is: [num_routed] canonical.
int partial_products_async_dev(const uint64_t* d_wires, const uint64_t* d_sigmas, const uint64_t* d_k_is,
                               const uint64_t* d_betas, const uint64_t* d_gammas, size_t num_challenges, size_t num_routed,
                               unsigned degree_bits, unsigned chunk, uint64_t* d_q_scratch, uint64_t* d_out, int* d_zero_den);

// exported by p2mt_fri.hip: the opening points of the FRI batches as device values, point b = d_point[b][0..2) * scale[b]
// (p2mt_fri_batch::point is ignored by the functions that take this)
constexpr int kMaxFriBatches = 4;
struct FriPointsDev {
  const uint64_t* d_point[kMaxFriBatches];
  uint64_t scale[kMaxFriBatches];
};
int fri_openings_points_dev(const p2mt_fri_oracle* oracles, size_t n_oracles, const p2mt_fri_batch* batches, size_t n_batches,
                            const FriPointsDev& pts, unsigned degree_bits, uint64_t* d_out);
// p2mt_fri_prove_openings_dev + one device-to-host copy that rides on its final synchronisation.  Inside a batch (runtime.h
// BatchCtx) epi_dst / epi_src are per-proof with the pitches epi_dpitch (host) and the batch stride (device).
// hch / link (both or neither; single proofs only): the transcript runs on the host (host_poseidon.h) -- caps, the final polynomial
// and the proof-of-work witness come down through the link, challenges go up as kernel arguments; `ch` then only lends its device
// state to the grind kernels.
int fri_prove_openings_epilogue_dev(const p2mt_fri_oracle* oracles, size_t n_oracles, const p2mt_fri_batch* batches,
                                    size_t n_batches, const FriPointsDev& pts, const p2mt_fri_params* p, p2mt_challenger* ch,
                                    uint64_t* d_proof, void* epi_dst, const void* epi_src, size_t epi_bytes, size_t epi_dpitch,
                                    host_poseidon::Challenger* hch = nullptr, HostLink* link = nullptr);
// a challenger over caller-owned device state (the batched prover keeps one state per proof block)
int challenger_wrap(void* d_state, p2mt_challenger** out);
void challenger_unwrap(p2mt_challenger* c);
constexpr size_t kChallengerStateBytes = 8 * (12 + 8 + 8) + 8;
// p2mt_verify_dev.hip: the field arithmetic of CircuitData::verify, on the device like the hashing (p2mt_circuit.hip)
struct VerifyDesc {
  uint32_t degree_bits, num_wires, num_routed, num_constants, num_selectors, num_challenges, quotient_degree_factor, n_kinds;
  uint32_t kind[16], sel[16], gs[16], ge[16];
};
// the same arithmetic on the device (p2mt_verify_dev.hip): word offsets are relative to the per-proof block `dv` of verify_pass
struct VerifyDevArgs {
  VerifyDesc d;
  p2mt_fri_params fri;
  uint64_t n_polys[4];
  uint64_t w_big, w_n, w16;  // primitive roots of unity of order 2^(degree_bits + rate_bits), 2^degree_bits, 16
  uint32_t o_proof, o_fo, o_out, o_cscap, off_open, off_fri, off_final, final_len, query_words;
};
// staged behind the transcript (see p2mt_verify_dev.hip): begin after the proof upload, after_zeta once zeta is squeezed, finish at the end
int verify_streams_create(void** out);
void verify_streams_destroy(void* vs);
void verify_streams_join(void* vs);  // host wait for both side streams (error paths)
int verify_dev_begin(void* vs, const uint64_t* dv, uint64_t* d_digests, const VerifyDevArgs& a, bool early_gates = false);
int verify_dev_after_zeta(void* vs, const uint64_t* dv, int* d_res, const uint64_t* d_k_is, const VerifyDevArgs& a);
int verify_dev_finish(void* vs, const uint64_t* dv, void* d_items, const uint64_t* d_digests, int* d_flag, int* d_res,
                      const VerifyDevArgs& a);
// the transcript was computed on the host and its challenges are on the device: everything behind them in one go
int verify_dev_with_challenges(void* vs, const uint64_t* dv, const uint64_t* d_digests, int* d_flag, int* d_res,
                               const uint64_t* d_k_is, const VerifyDevArgs& a);
}  // namespace p2mt

#define P2MT_HIP(x)                                                           \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) return p2mt::fail_hip(e_, #x, __FILE__, __LINE__);  \
  } while (0)
#define P2MT_TRY(x)          \
  do {                       \
    int rc_ = (x);           \
    if (rc_ != P2MT_OK) return rc_; \
  } while (0)
#define P2MT_LAUNCH_CHECK() P2MT_HIP(hipGetLastError())

// Launch KERNEL<mds, partial> for the runtime-selected Poseidon variant on the library stream; the PermCtx is
// appended as the last kernel argument.  mds 2 = issue-optimised path (default), 0/1 = exact reference variants.
// (mds 2, partial 1 -- the fast path's sparse partial rounds, an A/B that lost -- exists for the stage-1 MMR kernel only.)
#define P2MT_DISPATCH(KERNEL, GRID, BLOCK, ...)                                                              \
  do {                                                                                                       \
    hipStream_t st_ = p2mt::rt().stream;                                                                     \
    const p2mt::PermCtx ctx_ = p2mt::perm_ctx();                                                             \
    switch (p2mt::rt().mds * 2 + (p2mt::rt().mds == 2 ? 0 : p2mt::rt().partial)) {                          \
      case 0: hipLaunchKernelGGL((KERNEL<0, 0>), dim3(GRID), dim3(BLOCK), 0, st_, __VA_ARGS__, ctx_); break; \
      case 1: hipLaunchKernelGGL((KERNEL<0, 1>), dim3(GRID), dim3(BLOCK), 0, st_, __VA_ARGS__, ctx_); break; \
      case 2: hipLaunchKernelGGL((KERNEL<1, 0>), dim3(GRID), dim3(BLOCK), 0, st_, __VA_ARGS__, ctx_); break; \
      case 3: hipLaunchKernelGGL((KERNEL<1, 1>), dim3(GRID), dim3(BLOCK), 0, st_, __VA_ARGS__, ctx_); break; \
      default: hipLaunchKernelGGL((KERNEL<2, 0>), dim3(GRID), dim3(BLOCK), 0, st_, __VA_ARGS__, ctx_); break; \
    }                                                                                                        \
    P2MT_LAUNCH_CHECK();                                                                                     \
  } while (0)
