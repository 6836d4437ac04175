// p2mt_fri.hip -- Plonky2's opening proof on the device: Challenger, opening evaluation and FRI.
//
// Replaces, inside CircuitData::prove (reference call sites /root/reference/src/mmr/mmr_plonky2_verifier.rs:148 and
// mmr_plonky2_verifier_1_recursion.rs:192,218), plonky2's iop/challenger.rs, plonk/proof.rs eval_all and
// fri/oracle.rs prove_openings -> fri/prover.rs fri_proof [plonky2 source is not in the reference tree: parity unpinned,
// the tests compare against a CPU restatement of the same algorithm, whose verifier is the acceptance check].
//
// Shape of the computation: a chain of small dependent steps (challenge -> compose -> divide -> LDE -> Merkle cap ->
// challenge -> fold -> ...), so the design goal is latency, not bandwidth:
//   * the challenger state lives in device memory and is advanced by a one-wavefront kernel (12 lanes share one
//     permutation, permute_wave); alpha/beta/query indices are read by the next kernel straight from device memory, so
//     the whole proof is enqueued without a host round trip except for the proof-of-work result;
//   * the alpha-composition is a dot product against a table of alpha powers split into real/imaginary parts (two
//     base-field multiply-adds per coefficient and polynomial instead of an extension multiplication);
//   * (F(X) - F(z)) / (X - z) is a suffix scan of affine maps in LDS; folding and layer LDEs reuse the commit step's
//     coset-LDE kernels on the two components of the extension polynomial (base-field twiddles act componentwise);
//   * the proof-of-work grind is the one throughput kernel: one candidate per lane on the issue-optimised permutation,
//     atomicMin keeps the smallest witness (deterministic proofs).
// HBM layout: extension polynomials are two base-field arrays (component-major) while they are transformed, and
// (a, b) pairs once they become Merkle leaves / proof words.
#include "tree_common.hip.h"
#include "host_poseidon.h"

#include <string.h>

#include <algorithm>

#include <new>
#include <vector>

using namespace p2mt_dev;
using p2mt::DevBuf;
using p2mt::rt;

namespace {

// ---------------------------------------------------------------- F[X]/(X^2 - 7), loose u64 components
struct Ext {
  u64 a, b;
};
GL_DEV Ext ext_add(Ext x, Ext y) { return Ext{gl::add(x.a, y.a), gl::add(x.b, y.b)}; }
GL_DEV Ext ext_mul(Ext x, Ext y) {
  const u64 bb7 = gl::mul(gl::mul(x.b, y.b), 7);
  return Ext{gl::mul_add(x.a, y.a, bb7), gl::mul_add(x.a, y.b, gl::mul(x.b, y.a))};
}
// x * z + c, c in the base field
GL_DEV Ext ext_mul_add_base(Ext x, Ext z, u64 c) {
  Ext r = ext_mul(x, z);
  r.a = gl::add(r.a, c);
  return r;
}
GL_DEV Ext ext_pow(Ext x, u64 e) {
  Ext r{1, 0};
  while (e) {
    if (e & 1) r = ext_mul(r, x);
    x = ext_mul(x, x);
    e >>= 1;
  }
  return r;
}

// ---------------------------------------------------------------- iop/challenger.rs
struct ChState {
  u64 state[12];
  u64 in[8];
  u64 out[8];
  u32 n_in, n_out;
};
static_assert(sizeof(ChState) == 8 * (12 + 8 + 8) + 8, "layout shared with p2mt_challenger_get_state");

// Observe obs[0..n_obs), then squeeze n_sq challenges into sq.  One wavefront; lane w < 12 owns sponge word w.
__global__ __launch_bounds__(64) void k_challenger(ChState* __restrict__ st, const u64* __restrict__ obs, u32 n_obs,
                                                   u64* __restrict__ sq, u32 n_sq, u32 fresh, ChState* __restrict__ save_to,
                                                   unsigned long long* __restrict__ init_wit, BatchArg ba, PermCtx ctx) {
  st = bp(st, ba);
  obs = bp(obs, ba);
  sq = bp(sq, ba);
  save_to = bp(save_to, ba);
  init_wit = bp(init_wit, ba);
  __shared__ u64 s_in[8], s_out[8];
  __shared__ u64 rc_lds[kWaveRcWords];
  ctx = stage_round_constants(rc_lds, ctx);
  const unsigned lane = threadIdx.x;
  // fresh: start from Challenger::new() (zero sponge, empty buffers) without a separate reset launch
  u64 x = (lane < 12 && !fresh) ? st->state[lane] : 0;
  u32 n_in = fresh ? 0 : st->n_in, n_out = fresh ? 0 : st->n_out;  // wave-uniform
  if (lane < 8) {
    s_in[lane] = fresh ? 0 : st->in[lane];
    s_out[lane] = fresh ? 0 : st->out[lane];
  }
  __syncthreads();
  // duplexing(): overwrite the first n_in words with the buffered inputs, permute, refill the output buffer
  auto duplex = [&]() {
    if (lane < n_in) x = s_in[lane];
    n_in = 0;
    x = permute_wave(x, ctx);
    __syncthreads();
    if (lane < 8) s_out[lane] = gl::canon(x);
    __syncthreads();
    n_out = 8;
  };
  u32 i = 0;
#pragma unroll 1
  for (; i < n_obs && n_in != 0; ++i) {  // top up a partly filled input buffer element by element
    n_out = 0;  // observe_element: buffered outputs are stale
    if (lane == 0) s_in[n_in] = gl::canon(obs[i]);
    n_in += 1;
    __syncthreads();
    if (n_in == 8) duplex();
  }
  if (i < n_obs) {
    // whole 8-element chunks straight from memory into the state (lane k < 8 = word k), the next chunk prefetched under
    // the permutation: one coalesced load per duplexing instead of eight dependent ones
    u64 nxt = (lane < 8 && i + lane < n_obs) ? obs[i + lane] : 0;
    bool any = false;
#pragma unroll 1
    while (i + 8 <= n_obs) {
      const u64 cur = nxt;
      if (lane < 8 && i + 8 + lane < n_obs) nxt = obs[i + 8 + lane];
      if (lane < 8) x = gl::canon(cur);
      x = permute_wave(x, ctx);
      i += 8;
      any = true;
    }
    __syncthreads();
    if (any) {
      if (lane < 8) s_out[lane] = gl::canon(x);
      n_out = 8;
    }
    const u32 rem = n_obs - i;  // < 8 trailing elements stay buffered
    if (rem) {
      if (lane < rem) s_in[lane] = gl::canon(nxt);
      n_in = rem;
      n_out = 0;
    }
    __syncthreads();
  }
#pragma unroll 1
  for (u32 k = 0; k < n_sq; ++k) {
    if (n_in != 0 || n_out == 0) duplex();
    n_out -= 1;  // Vec::pop: from the back
    if (lane == 0) sq[k] = s_out[n_out];
  }
  __syncthreads();
  if (lane < 12) st->state[lane] = gl::canon(x);
  if (lane < 8) {
    st->in[lane] = s_in[lane];
    st->out[lane] = s_out[lane];
  }
  if (lane == 0) {
    st->n_in = n_in;
    st->n_out = n_out;
  }
  // optional extras of the FRI prover, folded in to save two launches: a copy of the updated transcript (rolled back to if
  // a proof-of-work chunk finds no witness) and the "no witness yet" marker
  if (save_to) {
    if (lane < 12) save_to->state[lane] = gl::canon(x);
    if (lane < 8) {
      save_to->in[lane] = s_in[lane];
      save_to->out[lane] = s_out[lane];
    }
    if (lane == 0) {
      save_to->n_in = n_in;
      save_to->n_out = n_out;
    }
  }
  if (init_wit && lane == 0) *init_wit = ~0ull;
}

// fri_proof_of_work: candidate = base + gid goes where the next observed element would; the response is the first
// challenge after it (word 7 of the permuted state).  *result = smallest passing candidate (init ~0).
template <int M, int PR>
__global__ __launch_bounds__(kBlock) void k_fri_pow(const ChState* __restrict__ st, u32 pow_bits, u64 base, u64 count,
                                                    unsigned long long* __restrict__ result, BatchArg ba, PermCtx ctx) {
  st = bp(st, ba);
  result = bp(result, ba);
  const u64 gid = (u64)blockIdx.x * kBlock + threadIdx.x;
  if (gid >= count) return;
  const u64 cand = base + gid;
  const u32 n_in = st->n_in;  // < 8: a full buffer is duplexed at once
  u64 s[12];
  permute_reloadable<M, PR>(s, ctx, [&](u64 (&t)[12]) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      u64 v = st->state[k];
      if (k < 8) {
        if ((u32)k < n_in) v = st->in[k];
        else if ((u32)k == n_in) v = cand;
      }
      t[k] = v;
    }
  });
  const u64 resp = gl::canon(s[7]);
  if (pow_bits == 0 || (resp >> (64 - pow_bits)) == 0) atomicMin(result, (unsigned long long)cand);
}

// The grind of a batch as a work queue.  Proofs need very different numbers of candidates (geometric, mean 2^pow_bits), so a
// fixed chunk per proof would idle most of the chip while the slowest proofs finish.  Here a resident grid of workgroups
// serves all the proofs: a workgroup looks for the next proof (round robin from where it last worked) that still has candidate
// blocks below its best known witness, takes that proof's next block of 256 candidates from the proof's counter, evaluates it,
// and moves on; it exits when no proof has work left, so the grid always drains.  Blocks are handed out in increasing order and
// a taken block is always evaluated unless a smaller witness is already known, so the result is the smallest witness, as in
// the single-proof search.  counter: one u32 per proof (zeroed by the host), blocks of [base, base + max_blocks * 256).
// Round 0 of the grind's permutation, the part every candidate of a proof shares: twelve lanes (threadIdx.x < 12; all threads of the
// workgroup must call it) form base[r] = sum_{k != n_in} M[r][k] (t[k] + rc[k])^7 + rc[12 + r] in `s_base`.
GL_DEV void pow_shared_first_round(const ChState* st, const PermCtx& ctx, u64* s_y /*[12]*/, u64* s_base /*[12]*/) {
  const u32 n_in = st->n_in;
  if (threadIdx.x < 12) {
    const u32 k = threadIdx.x;
    u64 v = st->state[k];
    if (k < 8 && k < n_in) v = st->in[k];
    s_y[k] = k == n_in ? 0 : gl::canon(gl::pow7(gl::add_c(v, ctx.rc[k])));
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const u32 r = threadIdx.x;
    u64 acc = ctx.rc[12 + r];
#pragma unroll
    for (u32 k = 0; k < 12; ++k) {
      const u32 idx = k >= r ? k - r : k + 12 - r;  // MDS[r][k] = CIRC[(k - r) mod 12] (+ 8 at [0][0])
      const u64 word = idx < 8 ? 0x0D0D1C0210290F11ull : 0x14221227ull;
      const u64 m = ((word >> (8 * (idx & 7))) & 0xFF) + ((r == 0 && k == 0) ? 8u : 0u);
      acc = gl::canon(gl::mul_add(s_y[k], m, acc));
    }
    s_base[r] = acc;
  }
  __syncthreads();
}
// ... once per proof of a batch, ahead of the queue kernel (whose workgroups would otherwise redo it for every block of 256 candidates:
// ~450 instructions on one wave and two barriers in front of each ~9 400-instruction candidate hash)
__global__ __launch_bounds__(64) void k_fri_pow_prep(const ChState* __restrict__ st, u64* __restrict__ base_out, BatchArg ba, PermCtx ctx) {
  __shared__ u64 s_y[12], s_base[12];
  pow_shared_first_round(bp(st, ba), ctx, s_y, s_base);
  if (threadIdx.x < 12) bp(base_out, ba)[threadIdx.x] = s_base[threadIdx.x];
}

// pow_base0: the per-proof shared first round from k_fri_pow_prep, or nullptr (each block computes it itself)
// (four waves per SIMD asked for, as for the stage-1 tree kernel: the allocator settles for 137 VGPRs = three on its own)
template <int M, int PR>
__global__ __launch_bounds__(kBlock, 4) void k_fri_pow_queue(const ChState* __restrict__ st0, u32 pow_bits, u64 base, u32 max_blocks,
                                                          unsigned long long* __restrict__ result0, u32* __restrict__ counter0,
                                                          u32 B, BatchArg ba, PermCtx ctx, const u64* __restrict__ pow_base0 = nullptr) {
  __shared__ u32 s_dist, s_blk, s_go;
  unsigned p = blockIdx.x % B;
  poseidon_fast::MfmaCtx mc;  // PR == 5: the permutation below runs with every lane of the workgroup active (its guard is workgroup-uniform)
  if constexpr (PR == 5) poseidon_fast::mfma32_ctx_init(mc);
  // counter0 == nullptr: one proof, one block per workgroup (block = blockIdx.x), no queue -- a single proof's chunk needs no counter
  const bool oneshot = counter0 == nullptr;
  bool first = true;
#pragma unroll 1
  for (;;) {
    u32 dist = 0;
    if (oneshot) {
      if (!first) return;
      first = false;
    } else {
      if (threadIdx.x == 0) s_dist = ~0u;
      __syncthreads();
      for (u32 k = threadIdx.x; k < B; k += kBlock) {  // distance to the next proof with unassigned candidates below its witness
        const u32 q = (p + k) % B;
        const u64 r = __hip_atomic_load(bp_at(result0, ba, q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u32 nb = __hip_atomic_load(bp_at(counter0, ba, q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nb < max_blocks && base + (u64)nb * kBlock < r) {
          atomicMin(&s_dist, k);
          break;
        }
      }
      __syncthreads();
      dist = s_dist;
      if (dist == ~0u) return;  // workgroup-uniform
    }
    const u32 q = oneshot ? 0 : (p + dist) % B;
    unsigned long long* result = bp_at(result0, ba, q);
    if (threadIdx.x == 0) {  // ONE lane decides for the workgroup: other workgroups lower *result concurrently, and the block below
                             // holds barriers and MFMAs (every lane must take the same side)
      const u32 b = oneshot ? blockIdx.x : atomicAdd(bp_at(counter0, ba, q), 1u);
      s_blk = b;
      s_go = b < max_blocks && base + (u64)b * kBlock < __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const u32 blk = s_blk;
    if (s_go) {
      const ChState* st = bp_at(st0, ba, q);
      const u64 cand = base + (u64)blk * kBlock + threadIdx.x;
      const u32 n_in = st->n_in;
      auto load = [&](u64 (&t)[12]) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          u64 v = st->state[k];
          if (k < 8) {
            if ((u32)k < n_in) v = st->in[k];
            else if ((u32)k == n_in) v = cand;
          }
          t[k] = v;
        }
      };
      u64 s[12];
      if constexpr (PR == 5) {
        // All 256 candidates of the block share eleven of the twelve input words, hence eleven of round 0's S-boxes and their
        // share of its MDS layer: twelve lanes form base[r] = sum_{k != n_in} M[r][k] (t[k] + rc[k])^7 + rc[12 + r] once per block;
        // a candidate then costs ONE first-round S-box and twelve two-mad rows, and of the last layer only row 7.
        __shared__ u64 s_y[12], s_base[12];
        if (pow_base0) {  // (workgroup-uniform)
          if (threadIdx.x < 12) s_base[threadIdx.x] = bp_at(pow_base0, ba, q)[threadIdx.x];
          __syncthreads();
        } else {
          pow_shared_first_round(st, ctx, s_y, s_base);
        }
        u64 sticky = ctx.force_fallback;
        const u64 y = poseidon_fast::pow7(gl::add_c(cand, ctx.rc[n_in]), sticky);
        const u32 yl = (u32)y, yh = (u32)(y >> 32);
#pragma unroll
        for (u32 r = 0; r < 12; ++r) {
          const u32 idx = n_in >= r ? n_in - r : n_in + 12 - r;
          const u64 word = idx < 8 ? 0x0D0D1C0210290F11ull : 0x14221227ull;
          const u32 m = (u32)((word >> (8 * (idx & 7))) & 0xFF) + ((r == 0 && n_in == 0) ? 8u : 0u);  // wave-uniform
          const u64 b = s_base[r];
          const u64 al = (u64)yl * m + (u32)b;
          u64 ah = (u64)yh * m + (u32)(b >> 32);
          ah = poseidon_fast::add32((u32)(al >> 32), ah);
          u64 cm;
          s[r] = poseidon_fast::mad_eps_carry((u32)(ah >> 32), ((u64)(u32)ah << 32) | (u32)al, cm);
          sticky |= cm;
        }
        sticky |= poseidon_fast::permute<false, 12, false, false, false, 3, 2, 0, true, 7>(s, ctx.rc, &mc);
        if (__builtin_expect(sticky != 0, 0)) {
          // `sticky` is the mask of the LANES whose rare carry fired (each lane is its own hash): redo those candidates only, one
          // at a time on the 12-lane layout (permute_wave: ~7 us each) -- the exact permutation for all 64 lanes took ~70 us on a
          // lone wavefront, and about one launch in five waited for such a wavefront (tools/grind_probe.py: 58 -> 115-140 us).
          const unsigned lane = threadIdx.x & 63;
          u64 todo = ctx.force_fallback ? ~0ull : sticky;  // (the tests' knob: every lane)
          while (todo) {                                    // wave-uniform
            const unsigned fl = (unsigned)__builtin_ctzll(todo);
            todo &= todo - 1;
            const u64 cf = base + (u64)blk * kBlock + (threadIdx.x & ~63u) + fl;
            u64 x = 0;
            if (lane < 12) {
              x = st->state[lane];
              if (lane < 8) {
                if (lane < n_in) x = st->in[lane];
                else if (lane == n_in) x = cf;
              }
            }
            x = permute_wave(x, ctx);
            const u64 r7 = __shfl((unsigned long long)x, 7);
            if (lane == fl) s[7] = r7;
          }
        }
      } else {
        permute_reloadable<M, PR>(s, ctx, load, &mc);
      }
      const u64 resp = gl::canon(s[7]);
      if (pow_bits == 0 || (resp >> (64 - pow_bits)) == 0) atomicMin(result, (unsigned long long)cand);
    }
    p = (q + 1) % B;
  }
}

// The same grind with four lanes per candidate (poseidon_quad.hip.h): a third of the single-hash latency, which is what
// bounds a 2^17-candidate launch.
__global__ __launch_bounds__(kBlock) void k_fri_pow_quad(const ChState* __restrict__ st, u32 pow_bits, u64 base, u64 count,
                                                         unsigned long long* __restrict__ result, BatchArg ba, PermCtx ctx) {
  st = bp(st, ba);
  result = bp(result, ba);
  const u64 gid = ((u64)blockIdx.x * kBlock + threadIdx.x) >> 2;
  if (gid >= count) return;  // quad-uniform
  if (*result < base) return;  // (batched grind) an earlier chunk already found this proof's witness; workgroup-uniform enough:
                               // values found by this launch are >= base
  poseidon_quad::Lane ln;
  poseidon_quad::lane_init(ln, ctx.rc);
  const u64 cand = base + gid;
  const u32 n_in = st->n_in;
  u64 x[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const u32 w = 3 * ln.q + i;
    u64 v = st->state[w];
    if (w < n_in) v = st->in[w];
    else if (w == n_in) v = cand;  // n_in < 8
    x[i] = v;
  }
  poseidon_quad::permute(x, ln);
  if (ln.q == 2) {  // word 7
    const u64 resp = gl::canon(x[1]);
    if (pow_bits == 0 || (resp >> (64 - pow_bits)) == 0) atomicMin(result, (unsigned long long)cand);
  }
}

// ---------------------------------------------------------------- alpha-composition
// table[j] = base^j, j < count (canonical pairs)
__global__ __launch_bounds__(kBlock) void k_ext_powers(const u64* __restrict__ base, u32 count, u64* __restrict__ table, BatchArg ba) {
  base = bp(base, ba);
  table = bp(table, ba);
  const u32 j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= count) return;
  const Ext r = ext_pow(Ext{base[0], base[1]}, j);
  table[2 * j] = gl::canon(r.a);
  table[2 * j + 1] = gl::canon(r.b);
}

constexpr u32 kComposeGroup = 32;  // polynomials per partial sum

// partial[g][c][i] = sum_{j in group g} (alpha^j)_c * f_j[i]   (ReducingFactor::reduce_polys_base as a dot product)
__global__ __launch_bounds__(kBlock) void k_fri_compose(const u64* const* __restrict__ ptrs, u32 cnt, u32 n,
                                                        const u64* __restrict__ apow, u64* __restrict__ partial, BatchArg ba) {
  ptrs = bp(ptrs, ba);
  apow = bp(apow, ba);
  partial = bp(partial, ba);
  const u32 i = blockIdx.x * kBlock + threadIdx.x, g = blockIdx.y;
  if (i >= n) return;
  const u32 j0 = g * kComposeGroup, j1 = min(cnt, j0 + kComposeGroup);
  u64 re = 0, im = 0;
#pragma unroll 4
  for (u32 j = j0; j < j1; ++j) {
    const u64 c = bp(ptrs[j], ba)[i];
    re = gl::mul_add(apow[2 * j], c, re);
    im = gl::mul_add(apow[2 * j + 1], c, im);
  }
  partial[((size_t)g * 2) * n + i] = re;
  partial[((size_t)g * 2 + 1) * n + i] = im;
}

// divide_by_linear + ReducingFactor::shift_poly + `final_poly += quotient`, one workgroup:
//   b_n = 0, b_m = b_{m+1} z + c_m, quotient coefficient m = b_{m+1};  fin[m] = fin[m] * alpha^cnt + b_{m+1}.
// The recurrence is a suffix scan: each thread reduces its chunk to one value, a log-step scan over the chunk values
// gives every chunk its carry-in, then the chunk is replayed.
constexpr int kQBlock = 1024, kQMaxLen = 4;  // n <= 2^12 coefficients: at most 4 per thread, kept in registers
// The opening point z = point * scale is read from device memory (the transcript's zeta; scale = 1 or the subgroup generator).
__global__ __launch_bounds__(kQBlock) void k_fri_quotient(const u64* __restrict__ partial, u32 n_groups, u32 n,
                                                          const u64* __restrict__ point, u64 scale,
                                                          const u64* __restrict__ apow_cnt, int first,
                                                          u64* __restrict__ fin, u64* __restrict__ shifted, BatchArg ba) {
  partial = bp(partial, ba);
  point = bp(point, ba);
  apow_cnt = bp(apow_cnt, ba);
  fin = bp(fin, ba);
  shifted = bp(shifted, ba);
  __shared__ Ext S[kQBlock];
  const u32 len = n >= (u32)kQBlock ? n / kQBlock : 1, T = n / len, t = threadIdx.x;
  const Ext z{gl::mul(point[0], scale), gl::mul(point[1], scale)};
  const u32 s = t * len;
  Ext c[kQMaxLen];  // this thread's composition coefficients (partial sums added up)
#pragma unroll
  for (int i = 0; i < kQMaxLen; ++i) {
    c[i] = Ext{0, 0};
    if (t < T && (u32)i < len)
      for (u32 g = 0; g < n_groups; ++g) {
        c[i].a = gl::add(c[i].a, partial[((size_t)g * 2) * n + s + i]);
        c[i].b = gl::add(c[i].b, partial[((size_t)g * 2 + 1) * n + s + i]);
      }
  }
  Ext loc{0, 0};
#pragma unroll
  for (int i = kQMaxLen - 1; i >= 0; --i)
    if ((u32)i < len) loc = ext_add(ext_mul(loc, z), c[i]);
  S[t] = loc;
  __syncthreads();
  Ext zp = ext_pow(z, len);  // z^(len 2^k) in step k
  for (u32 d = 1; d < T; d *= 2) {
    Ext v = S[t];
    if (t + d < T) v = ext_add(v, ext_mul(zp, S[t + d]));
    __syncthreads();
    S[t] = v;
    __syncthreads();
    zp = ext_mul(zp, zp);
  }
  if (t >= T) return;
  Ext acc = t + 1 < T ? S[t + 1] : Ext{0, 0};  // b at the end of this chunk
  const Ext sh = first ? Ext{0, 0} : Ext{apow_cnt[0], apow_cnt[1]};  // alpha^cnt (ReducingFactor::shift_poly)
#pragma unroll
  for (int i = kQMaxLen - 1; i >= 0; --i) {
    if ((u32)i >= len) continue;
    const u32 m = s + i;
    Ext q = acc;
    if (!first) q = ext_add(q, ext_mul(Ext{fin[m], fin[n + m]}, sh));
    q.a = gl::canon(q.a);
    q.b = gl::canon(q.b);
    if (shifted) {  // last batch: coeffs.insert(0, ZERO) (plonky2 PR 436); coefficient n-1 of the sum is zero by construction
      if (m + 1 < n) {
        shifted[m + 1] = q.a;
        shifted[n + m + 1] = q.b;
      }
      if (m == 0) shifted[0] = shifted[n] = 0;
    } else {
      fin[m] = q.a;
      fin[n + m] = q.b;
    }
    acc = ext_add(ext_mul(acc, z), c[i]);
  }
}

// (a, b) pairs from component-major arrays: the layer's Merkle leaves (2^arity_bits consecutive pairs per row) and
// the final polynomial's proof words
__global__ __launch_bounds__(kBlock) void k_fri_pairs(const u64* __restrict__ comp, u32 n, u64* __restrict__ out, BatchArg ba) {
  comp = bp(comp, ba);
  out = bp(out, ba);
  const u32 i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  reinterpret_cast<ulonglong2*>(out)[i] = make_ulonglong2(comp[i], comp[n + i]);
}

// P(x) = sum_{i<r} x^i P_i(x^r)  ->  sum_{i<r} beta^i P_i(x): out[k] = sum_i beta^i in[k r + i]
__global__ __launch_bounds__(kBlock) void k_fri_fold(const u64* __restrict__ in, u32 n_in, u32 arity_bits,
                                                     const u64* __restrict__ beta, u64* __restrict__ out, BatchArg ba) {
  in = bp(in, ba);
  beta = bp(beta, ba);
  out = bp(out, ba);
  const u32 n_out = n_in >> arity_bits, k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n_out) return;
  const Ext b{beta[0], beta[1]};
  const u32 r = 1u << arity_bits;
  Ext acc{0, 0};
  for (u32 i = r; i-- > 0;) {
    acc = ext_mul(acc, b);
    acc.a = gl::add(acc.a, in[k * r + i]);
    acc.b = gl::add(acc.b, in[n_in + k * r + i]);
  }
  out[k] = gl::canon(acc.a);
  out[n_out + k] = gl::canon(acc.b);
}

// ---------------------------------------------------------------- fri_prover_query_rounds
struct QTree {
  const u64* leaves;   // row-major rows x width
  const u64* digests;  // level-major
  u32 width, log_rows, idx_shift, n_sib;
};
constexpr int kMaxQTrees = 16;
struct QArgs {
  QTree t[kMaxQTrees];
  u32 n_trees, log_big;
  u64 per_query;
};

// One workgroup per query round: x_index = challenge mod N; per tree the opened row and its Merkle path.
__global__ __launch_bounds__(kBlock) void k_fri_queries(QArgs a, const u64* __restrict__ qch, u64* __restrict__ out, BatchArg ba) {
  qch = bp(qch, ba);
  out = bp(out, ba);
  const u64 x_index = qch[blockIdx.x] & (((u64)1 << a.log_big) - 1);
  u64* w = out + (u64)blockIdx.x * a.per_query;
  for (u32 ti = 0; ti < a.n_trees; ++ti) {
    const QTree& tr = a.t[ti];
    const u64 *tr_leaves = bp(tr.leaves, ba), *tr_digests = bp(tr.digests, ba);
    const u64 row = x_index >> tr.idx_shift;
    for (u32 i = threadIdx.x; i < tr.width; i += kBlock) w[i] = tr_leaves[row * tr.width + i];
    w += tr.width;
    const u64 rows = (u64)1 << tr.log_rows;
    for (u32 i = threadIdx.x; i < 4 * tr.n_sib; i += kBlock) {
      const u32 s = i >> 2;
      const u64 off = 2 * rows - 2 * (rows >> s);  // rows + rows/2 + ... (s terms)
      w[i] = tr_digests[4 * (off + ((row >> s) ^ 1)) + (i & 3)];
    }
    w += 4 * tr.n_sib;
  }
}

// ---------------------------------------------------------------- PolynomialCoeffs::eval at an extension point
// One workgroup per polynomial: per-thread Horner over a chunk, then a tree reduction with z^(len 2^s).
__global__ __launch_bounds__(kBlock) void k_eval_ext(const u64* __restrict__ coeffs, u32 log_n, u64 za, u64 zb,
                                                     u64* __restrict__ out) {
  __shared__ Ext S[kBlock];
  const u32 n = 1u << log_n, len = n >= (u32)kBlock ? n / kBlock : 1, T = n / len, t = threadIdx.x;
  const u64* c = coeffs + ((size_t)blockIdx.x << log_n);
  const Ext z{za, zb};
  Ext loc{0, 0};
  if (t < T)
    for (u32 m = (t + 1) * len; m-- > t * len;) loc = ext_mul_add_base(loc, z, c[m]);
  S[t] = loc;
  __syncthreads();
  Ext zp = ext_pow(z, len);
  for (u32 d = 1; d < T; d *= 2) {
    if ((t & (2 * d - 1)) == 0 && t + d < T) S[t] = ext_add(S[t], ext_mul(zp, S[t + d]));
    __syncthreads();
    zp = ext_mul(zp, zp);
  }
  if (t == 0) {
    out[2 * blockIdx.x] = gl::canon(S[0].a);
    out[2 * blockIdx.x + 1] = gl::canon(S[0].b);
  }
}

// The same for a table of polynomials with a point each (FriOpenings in one launch).  The point is point[0..2) * scale, read
// from device memory: the prover's zeta never visits the host.  The table itself is shared by the proofs of a batch (it lives
// outside the per-proof blocks); its pointers are rebased per proof.
struct EvalEntry {
  const u64* coeffs;
  const u64* point;
  u64 scale, pad;
};
__global__ __launch_bounds__(kBlock) void k_eval_ext_table(const EvalEntry* __restrict__ tab, u32 log_n, u64* __restrict__ out, BatchArg ba) {
  out = bp(out, ba);
  __shared__ Ext S[kBlock];
  const u32 n = 1u << log_n, len = n >= (u32)kBlock ? n / kBlock : 1, T = n / len, t = threadIdx.x;
  const u64* c = bp(tab[blockIdx.x].coeffs, ba);
  const u64* pt = bp(tab[blockIdx.x].point, ba);
  const Ext z{gl::mul(pt[0], tab[blockIdx.x].scale), gl::mul(pt[1], tab[blockIdx.x].scale)};
  Ext loc{0, 0};
  if (t < T)
    for (u32 m = (t + 1) * len; m-- > t * len;) loc = ext_mul_add_base(loc, z, c[m]);
  S[t] = loc;
  __syncthreads();
  Ext zp = ext_pow(z, len);
  for (u32 d = 1; d < T; d *= 2) {
    if ((t & (2 * d - 1)) == 0 && t + d < T) S[t] = ext_add(S[t], ext_mul(zp, S[t + d]));
    __syncthreads();
    zp = ext_mul(zp, zp);
  }
  if (t == 0) {
    out[2 * blockIdx.x] = gl::canon(S[0].a);
    out[2 * blockIdx.x + 1] = gl::canon(S[0].b);
  }
}

// The same for 64 coefficients per polynomial (the 64-row circuits: every proof of the batched prover) with EIGHT LANES per
// polynomial: a lane evaluates its 8 coefficients by Horner, three lane-exchange steps add the eight partial sums with z^8, z^16,
// z^32.  No LDS, no barrier; a wavefront serves 8 polynomials.  (k_eval_ext_table gives a 64-coefficient polynomial a workgroup of 256
// threads, 64 of them busy for one coefficient each, and a 6-step tree with a barrier per step: 331 us for the 65 792 openings of a
// 256-proof pass.)
__global__ __launch_bounds__(kBlock) void k_eval_ext_table_oct64(const EvalEntry* __restrict__ tab, u32 n_tab, u64* __restrict__ out, BatchArg ba) {
  out = bp(out, ba);
  const u32 gid = blockIdx.x * kBlock + threadIdx.x, e = gid >> 3, q = gid & 7, lane = threadIdx.x & 63;
  if (e >= n_tab) return;  // whole groups of eight lanes
  const u64* c = bp(tab[e].coeffs, ba) + 8 * q;
  const u64* pt = bp(tab[e].point, ba);
  const Ext z{gl::mul(pt[0], tab[e].scale), gl::mul(pt[1], tab[e].scale)};
  Ext S{0, 0};
#pragma unroll
  for (int m = 7; m >= 0; --m) S = ext_mul_add_base(S, z, c[m]);
  Ext zp = ext_mul(z, z);
  zp = ext_mul(zp, zp);
  zp = ext_mul(zp, zp);  // z^8
#pragma unroll
  for (u32 d = 1; d < 8; d *= 2) {
    const int addr = (int)((lane + d) << 2);  // (the lanes that use the value have their partner inside the group)
    Ext o;
    o.a = ((u64)(u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)(S.a >> 32)) << 32) | (u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)S.a);
    o.b = ((u64)(u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)(S.b >> 32)) << 32) | (u32)__builtin_amdgcn_ds_bpermute(addr, (int)(u32)S.b);
    const Ext sum = ext_add(S, ext_mul(zp, o));
    if ((q & (2 * d - 1)) == 0) S = sum;
    zp = ext_mul(zp, zp);
  }
  if (q == 0) {
    out[2 * e] = gl::canon(S.a);
    out[2 * e + 1] = gl::canon(S.b);
  }
}

// ---------------------------------------------------------------- host helpers
inline u64 h_mul(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % gl::P); }
inline u64 h_pow(u64 a, u64 e) {
  u64 r = 1;
  for (; e; e >>= 1, a = h_mul(a, a))
    if (e & 1) r = h_mul(r, a);
  return r;
}
inline u64 h_add(u64 a, u64 b) { return (u64)(((unsigned __int128)a + b) % gl::P); }
inline void h_ext_mul(const u64 x[2], const u64 y[2], u64 out[2]) {
  const u64 a = h_add(h_mul(x[0], y[0]), h_mul(7, h_mul(x[1], y[1])));
  const u64 b = h_add(h_mul(x[0], y[1]), h_mul(x[1], y[0]));
  out[0] = a;
  out[1] = b;
}

unsigned total_arity_bits(const p2mt_fri_params* p) {
  unsigned t = 0;
  for (uint32_t l = 0; l < p->num_reductions; ++l) t += p->reduction_arity_bits[l];
  return t;
}

bool params_ok(const p2mt_fri_params* p) {
  if (!p || p->num_reductions > 8 || p->degree_bits > 12 || p->rate_bits > 8 || p->proof_of_work_bits > 40) return false;
  if (p->cap_height > p->degree_bits + p->rate_bits) return false;
  unsigned log_sz = p->degree_bits + p->rate_bits, d = p->degree_bits;
  for (uint32_t l = 0; l < p->num_reductions; ++l) {
    const unsigned ab = p->reduction_arity_bits[l];
    if (ab == 0 || ab > 4 || ab > d || log_sz < ab + p->cap_height) return false;
    log_sz -= ab;
    d -= ab;
  }
  return true;
}

size_t digests_count(size_t rows, unsigned cap_height) {
  size_t c = 0;
  for (size_t r = rows; r > ((size_t)1 << cap_height); r >>= 1) c += r;
  return c;
}

int launch_challenger(ChState* st, const u64* d_obs, size_t n_obs, u64* d_sq, size_t n_sq, bool fresh = false,
                      ChState* save_to = nullptr, unsigned long long* init_wit = nullptr) {
  if (n_obs > 0xFFFFFFFFull || n_sq > 0xFFFFFFFFull) return p2mt::fail(P2MT_EINVAL, "challenger: too many elements");
  hipLaunchKernelGGL(k_challenger, bgrid(1), dim3(64), 0, rt().stream, st, d_obs, (u32)n_obs, d_sq, (u32)n_sq, fresh ? 1u : 0u,
                     save_to, init_wit, barg(), p2mt::perm_ctx());
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}

}  // namespace

struct p2mt_challenger {
  ChState* d = nullptr;
  bool owned = true;
};
static_assert(sizeof(ChState) == p2mt::kChallengerStateBytes, "runtime.h");
int p2mt::challenger_wrap(void* d_state, p2mt_challenger** out) {
  p2mt_challenger* c = new (std::nothrow) p2mt_challenger;
  if (!c) return p2mt::fail(P2MT_ENOMEM, "out of host memory");
  c->d = static_cast<ChState*>(d_state);
  c->owned = false;
  *out = c;
  return P2MT_OK;
}
void p2mt::challenger_unwrap(p2mt_challenger* c) { delete c; }

// =================================================================== Challenger
extern "C" int p2mt_challenger_create(p2mt_challenger** out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  p2mt_challenger* c = new (std::nothrow) p2mt_challenger;
  if (!c) return p2mt::fail(P2MT_ENOMEM, "out of host memory");
  if (hipMalloc((void**)&c->d, sizeof(ChState)) != hipSuccess) {
    delete c;
    return p2mt::fail(P2MT_ENOMEM, "hipMalloc(challenger) failed");
  }
  P2MT_HIP(hipMemsetAsync(c->d, 0, sizeof(ChState), rt().stream));
  *out = c;
  return P2MT_OK;
  });
}

extern "C" int p2mt_challenger_destroy(p2mt_challenger* c) {
  return p2mt::abi_guard([&]() -> int {
  if (!c) return P2MT_OK;
  if (c->d && c->owned) {
    (void)hipStreamSynchronize(rt().stream);
    (void)hipFree(c->d);
  }
  delete c;
  return P2MT_OK;
  });
}

extern "C" int p2mt_challenger_clone(const p2mt_challenger* src, p2mt_challenger** out) {
  return p2mt::abi_guard([&]() -> int {
  if (!src || !out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  P2MT_TRY(p2mt_challenger_create(out));
  P2MT_HIP(hipMemcpyAsync((*out)->d, src->d, sizeof(ChState), hipMemcpyDeviceToDevice, rt().stream));
  return P2MT_OK;
  });
}

extern "C" int p2mt_challenger_observe_dev(p2mt_challenger* c, const uint64_t* d_elements, size_t n) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || (n && !d_elements)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (n == 0) return P2MT_OK;
  return launch_challenger(c->d, d_elements, n, nullptr, 0);
  });
}

extern "C" int p2mt_challenger_observe(p2mt_challenger* c, const uint64_t* elements, size_t n) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || (n && !elements)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (n == 0) return P2MT_OK;
  DevBuf b;
  P2MT_TRY(b.alloc(n * 8));
  P2MT_HIP(hipMemcpyAsync(b.p, elements, n * 8, hipMemcpyHostToDevice, rt().stream));
  P2MT_TRY(launch_challenger(c->d, b.as<u64>(), n, nullptr, 0));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

extern "C" int p2mt_challenger_reset(p2mt_challenger* c) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c) return p2mt::fail(P2MT_EINVAL, "null pointer");
  P2MT_HIP(hipMemsetAsync(c->d, 0, sizeof(ChState), rt().stream));
  return P2MT_OK;
  });
}

extern "C" int p2mt_challenger_duplex_dev(p2mt_challenger* c, const uint64_t* d_elements, size_t n_obs, uint64_t* d_out,
                                          size_t n_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || (n_obs && !d_elements) || (n_out && !d_out)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (n_obs == 0 && n_out == 0) return P2MT_OK;
  return launch_challenger(c->d, d_elements, n_obs, d_out, n_out);
  });
}

extern "C" int p2mt_challenger_restart_duplex_dev(p2mt_challenger* c, const uint64_t* d_elements, size_t n_obs, uint64_t* d_out,
                                                  size_t n_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || (n_obs && !d_elements) || (n_out && !d_out)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  return launch_challenger(c->d, d_elements, n_obs, d_out, n_out, true);
  });
}

extern "C" int p2mt_challenger_get_challenges_dev(p2mt_challenger* c, size_t n, uint64_t* d_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || (n && !d_out)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (n == 0) return P2MT_OK;
  return launch_challenger(c->d, nullptr, 0, d_out, n);
  });
}

extern "C" int p2mt_challenger_get_challenges(p2mt_challenger* c, size_t n, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || (n && !out)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (n == 0) return P2MT_OK;
  DevBuf b;
  P2MT_TRY(b.alloc(n * 8));
  P2MT_TRY(launch_challenger(c->d, nullptr, 0, b.as<u64>(), n));
  P2MT_HIP(hipMemcpyAsync(out, b.p, n * 8, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

extern "C" int p2mt_challenger_get_state(const p2mt_challenger* c, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  if (!c || !out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  ChState h;
  P2MT_HIP(hipMemcpyAsync(&h, c->d, sizeof h, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  memcpy(out, h.state, 96);
  memcpy(out + 12, h.in, 64);
  memcpy(out + 20, h.out, 64);
  out[28] = h.n_in;
  out[29] = h.n_out;
  return P2MT_OK;
  });
}

extern "C" int p2mt_challenger_set_state(p2mt_challenger* c, const uint64_t* in) {
  return p2mt::abi_guard([&]() -> int {
  if (!c || !in) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (in[28] > 7 || in[29] > 8) return p2mt::fail(P2MT_EINVAL, "challenger state: buffer counts out of range");
  ChState h;
  memcpy(h.state, in, 96);
  memcpy(h.in, in + 12, 64);
  memcpy(h.out, in + 20, 64);
  h.n_in = (u32)in[28];
  h.n_out = (u32)in[29];
  P2MT_HIP(hipMemcpyAsync(c->d, &h, sizeof h, hipMemcpyHostToDevice, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

// =================================================================== OpeningSet evaluation
extern "C" int p2mt_eval_polys_ext_dev(const uint64_t* d_coeffs, size_t n_polys, unsigned log_n, const uint64_t point[2],
                                       uint64_t* d_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n_polys == 0) return P2MT_OK;
  if (!d_coeffs || !point || !d_out || log_n > 24 || n_polys > 0x7FFFFFFF) return p2mt::fail(P2MT_EINVAL, "eval_polys_ext: bad argument");
  hipLaunchKernelGGL(k_eval_ext, dim3((unsigned)n_polys), dim3(kBlock), 0, rt().stream, d_coeffs, log_n, point[0] % gl::P,
                     point[1] % gl::P, d_out);
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
  });
}

extern "C" int p2mt_eval_polys_ext(const uint64_t* coeffs, size_t n_polys, unsigned log_n, const uint64_t point[2],
                                   uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n_polys == 0) return P2MT_OK;
  if (!coeffs || !point || !out || log_n > 24) return p2mt::fail(P2MT_EINVAL, "eval_polys_ext: bad argument");
  DevBuf bi, bo;
  P2MT_TRY(bi.alloc((n_polys << log_n) * 8));
  P2MT_TRY(bo.alloc(n_polys * 16));
  P2MT_HIP(hipMemcpyAsync(bi.p, coeffs, (n_polys << log_n) * 8, hipMemcpyHostToDevice, rt().stream));
  P2MT_TRY(p2mt_eval_polys_ext_dev(bi.as<u64>(), n_polys, log_n, point, bo.as<u64>()));
  P2MT_HIP(hipMemcpyAsync(out, bo.p, n_polys * 16, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

// FriOpenings: every batch's polynomials at the batch's point, batches concatenated, one launch
namespace {
// upload the host points of the public entry points (canonical) to a small device array
int upload_points(const p2mt_fri_batch* batches, size_t n_batches, p2mt::FriPointsDev* pts) {
  if (n_batches > (size_t)p2mt::kMaxFriBatches) return p2mt::fail(P2MT_EINVAL, "fri: too many batches");
  u64* d_pts;
  P2MT_TRY(p2mt::scratch_get_shared(p2mt::kScratchPoints, 2 * p2mt::kMaxFriBatches * 8, (void**)&d_pts));
  u64 h[2 * p2mt::kMaxFriBatches] = {};
  for (size_t b = 0; b < n_batches; ++b) {
    h[2 * b] = batches[b].point[0] % gl::P;
    h[2 * b + 1] = batches[b].point[1] % gl::P;
    pts->d_point[b] = d_pts + 2 * b;
    pts->scale[b] = 1;
  }
  P2MT_HIP(hipMemcpyAsync(d_pts, h, sizeof h, hipMemcpyHostToDevice, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));  // `h` is on the stack
  return P2MT_OK;
}
}  // namespace

int p2mt::fri_openings_points_dev(const p2mt_fri_oracle* oracles, size_t n_oracles, const p2mt_fri_batch* batches,
                                  size_t n_batches, const FriPointsDev& pts, unsigned degree_bits, uint64_t* d_out) {
  if (!oracles || !batches || !d_out || n_oracles == 0 || degree_bits > 24 || n_batches > (size_t)kMaxFriBatches)
    return p2mt::fail(P2MT_EINVAL, "fri_openings: bad argument");
  std::vector<EvalEntry> tab;
  for (size_t b = 0; b < n_batches; ++b) {
    if (batches[b].n_polys && !batches[b].polys) return p2mt::fail(P2MT_EINVAL, "fri_openings: null batch");
    for (size_t j = 0; j < batches[b].n_polys; ++j) {
      const uint32_t o = batches[b].polys[2 * j], pi = batches[b].polys[2 * j + 1];
      if (o >= n_oracles || !oracles[o].coeffs || pi >= oracles[o].n_polys)
        return p2mt::fail(P2MT_EINVAL, "fri_openings: batch polynomial index out of range");
      tab.push_back(EvalEntry{oracles[o].coeffs + ((size_t)pi << degree_bits), pts.d_point[b], pts.scale[b] % gl::P, 0});
    }
  }
  if (tab.empty()) return P2MT_OK;
  EvalEntry* d_tab;
  P2MT_TRY(p2mt::scratch_get_shared(p2mt::kScratchTables, tab.size() * sizeof(EvalEntry), (void**)&d_tab));
  hipStream_t st = rt().stream;
  {  // the table is the same for every proof of a circuit (the points are device values): upload it only when it changed
    thread_local std::vector<u64> last;
    thread_local const EvalEntry* last_dst = nullptr;
    thread_local uint64_t last_epoch = ~0ull;
    const u64* words = reinterpret_cast<const u64*>(tab.data());
    const size_t n_words = tab.size() * sizeof(EvalEntry) / 8;
    if (last_dst != d_tab || last_epoch != p2mt::scratch_epoch() || last.size() != n_words ||
        memcmp(last.data(), words, n_words * 8) != 0) {
      P2MT_HIP(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(EvalEntry), hipMemcpyHostToDevice, st));
      P2MT_HIP(hipStreamSynchronize(st));  // `tab` is pageable host memory; rare
      last.assign(words, words + n_words);
      last_dst = d_tab;
      last_epoch = p2mt::scratch_epoch();
    }
  }
  static const bool oct_knob = [] { const char* e = getenv("P2MT_EVAL_OCT"); return e ? atoi(e) != 0 : true; }();
  if (degree_bits == 6 && oct_knob)  // eight lanes per polynomial (env P2MT_EVAL_OCT=0: the workgroup-per-polynomial kernel, A/B)
    hipLaunchKernelGGL(k_eval_ext_table_oct64, bgrid((unsigned)((tab.size() * 8 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       (const EvalEntry*)d_tab, (u32)tab.size(), d_out, barg());
  else
    hipLaunchKernelGGL(k_eval_ext_table, bgrid((unsigned)tab.size()), dim3(kBlock), 0, st, (const EvalEntry*)d_tab, degree_bits, d_out,
                       barg());
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}

extern "C" int p2mt_fri_openings_dev(const p2mt_fri_oracle* oracles, size_t n_oracles, const p2mt_fri_batch* batches,
                                     size_t n_batches, unsigned degree_bits, uint64_t* d_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!oracles || !batches || !d_out || n_oracles == 0 || degree_bits > 24) return p2mt::fail(P2MT_EINVAL, "fri_openings: bad argument");
  p2mt::FriPointsDev pts{};
  P2MT_TRY(upload_points(batches, n_batches, &pts));
  P2MT_TRY(p2mt::fri_openings_points_dev(oracles, n_oracles, batches, n_batches, pts, degree_bits, d_out));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

extern "C" int p2mt_fri_openings(const p2mt_fri_oracle* oracles, size_t n_oracles, const p2mt_fri_batch* batches,
                                 size_t n_batches, unsigned degree_bits, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!oracles || !batches || !out || n_oracles == 0 || degree_bits > 24) return p2mt::fail(P2MT_EINVAL, "fri_openings: bad argument");
  std::vector<DevBuf> bufs(n_oracles + 1);
  std::vector<p2mt_fri_oracle> dev(n_oracles);
  size_t total = 0;
  for (size_t b = 0; b < n_batches; ++b) total += batches[b].n_polys;
  hipStream_t st = rt().stream;
  for (size_t o = 0; o < n_oracles; ++o) {
    if (!oracles[o].coeffs || oracles[o].n_polys == 0) return p2mt::fail(P2MT_EINVAL, "fri_openings: bad oracle");
    const size_t bytes = (oracles[o].n_polys << degree_bits) * 8;
    P2MT_TRY(bufs[o].alloc(bytes));
    P2MT_HIP(hipMemcpyAsync(bufs[o].p, oracles[o].coeffs, bytes, hipMemcpyHostToDevice, st));
    dev[o] = p2mt_fri_oracle{bufs[o].as<u64>(), nullptr, nullptr, oracles[o].n_polys};
  }
  P2MT_TRY(bufs[n_oracles].alloc(total * 16));
  P2MT_TRY(p2mt_fri_openings_dev(dev.data(), n_oracles, batches, n_batches, degree_bits, bufs[n_oracles].as<u64>()));
  P2MT_HIP(hipMemcpyAsync(out, bufs[n_oracles].p, total * 16, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  return P2MT_OK;
  });
}

// =================================================================== FRI
extern "C" int p2mt_fri_params_standard(unsigned degree_bits, p2mt_fri_params* out) {
  return p2mt::abi_guard([&]() -> int {
  if (!out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  memset(out, 0, sizeof *out);
  out->degree_bits = degree_bits;
  out->rate_bits = 3;
  out->cap_height = 4;
  out->proof_of_work_bits = 16;
  out->num_query_rounds = 28;
  // ConstantArityBits(4, 5) (fri/reduction_strategies.rs)
  const unsigned arity_bits = 4, final_poly_bits = 5;
  unsigned d = degree_bits;
  while (d > final_poly_bits && d + out->rate_bits - arity_bits >= out->cap_height && out->num_reductions < 8) {
    out->reduction_arity_bits[out->num_reductions++] = arity_bits;
    d -= arity_bits;
  }
  return P2MT_OK;
  });
}

extern "C" size_t p2mt_fri_proof_len(const p2mt_fri_params* p, size_t n_oracles, const uint64_t* n_polys) {
  // (pure integer arithmetic on caller data: nothing here can throw, so no abi_guard; 0 = bad parameters)
  if (!params_ok(p) || (n_oracles && !n_polys)) return 0;
  const unsigned log_big = p->degree_bits + p->rate_bits;
  size_t len = (size_t)p->num_reductions * ((size_t)4 << p->cap_height);
  size_t per_query = 0;
  for (size_t o = 0; o < n_oracles; ++o) per_query += n_polys[o] + 4 * (size_t)(log_big - p->cap_height);
  unsigned log_sz = log_big;
  for (uint32_t l = 0; l < p->num_reductions; ++l) {
    const unsigned ab = p->reduction_arity_bits[l];
    per_query += ((size_t)2 << ab) + 4 * (size_t)(log_sz - ab - p->cap_height);
    log_sz -= ab;
  }
  len += per_query * p->num_query_rounds;
  len += (size_t)2 << (p->degree_bits - total_arity_bits(p));
  return len + 1;
}

extern "C" int p2mt_fri_prove_openings_dev(const p2mt_fri_oracle* oracles, size_t n_oracles, const p2mt_fri_batch* batches,
                                           size_t n_batches, const p2mt_fri_params* p, p2mt_challenger* ch,
                                           uint64_t* d_proof) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!batches || n_batches == 0) return p2mt::fail(P2MT_EINVAL, "fri: null argument");
  p2mt::FriPointsDev pts{};
  P2MT_TRY(upload_points(batches, n_batches, &pts));
  return p2mt::fri_prove_openings_epilogue_dev(oracles, n_oracles, batches, n_batches, pts, p, ch, d_proof, nullptr, nullptr, 0, 0);
  });
}

// The same with the opening points as device values and a device-to-host copy enqueued behind the speculative tail of the
// proof, in front of the synchronisation that reads the proof-of-work result: the caller's proof comes back with that one wait
// (dst is valid when this returns 0).
int p2mt::fri_prove_openings_epilogue_dev(const p2mt_fri_oracle* oracles, size_t n_oracles, const p2mt_fri_batch* batches,
                                          size_t n_batches, const FriPointsDev& pts, const p2mt_fri_params* p,
                                          p2mt_challenger* ch, uint64_t* d_proof, void* epi_dst, const void* epi_src,
                                          size_t epi_bytes, size_t epi_dpitch, host_poseidon::Challenger* hch, HostLink* link) {
  P2MT_TRY(p2mt::ensure_init());
  if (!oracles || !batches || !ch || !d_proof || n_oracles == 0 || n_batches == 0) return p2mt::fail(P2MT_EINVAL, "fri: null argument");
  if (n_batches > (size_t)kMaxFriBatches) return p2mt::fail(P2MT_EINVAL, "fri: too many batches");
  if (!params_ok(p)) return p2mt::fail(P2MT_EINVAL, "fri: unsupported FriParams (degree_bits <= 12, arity_bits in 1..4, layer trees >= cap)");
  if (n_oracles + p->num_reductions > (size_t)kMaxQTrees) return p2mt::fail(P2MT_EINVAL, "fri: too many oracles");
  const unsigned log_n = p->degree_bits, log_big = log_n + p->rate_bits;
  const u32 n = 1u << log_n;
  const size_t cap_words = (size_t)4 << p->cap_height;
  size_t max_cnt = 0, total_cnt = 0;
  std::vector<uint64_t> n_polys(n_oracles);
  for (size_t o = 0; o < n_oracles; ++o) {
    if (!oracles[o].coeffs || !oracles[o].leaves || oracles[o].n_polys == 0) return p2mt::fail(P2MT_EINVAL, "fri: bad oracle");
    if (!oracles[o].digests && log_big > p->cap_height) return p2mt::fail(P2MT_EINVAL, "fri: oracle without digests");
    n_polys[o] = oracles[o].n_polys;
  }
  for (size_t b = 0; b < n_batches; ++b) {
    if (!batches[b].polys || batches[b].n_polys == 0) return p2mt::fail(P2MT_EINVAL, "fri: empty batch");
    for (size_t j = 0; j < batches[b].n_polys; ++j)
      if (batches[b].polys[2 * j] >= n_oracles || batches[b].polys[2 * j + 1] >= oracles[batches[b].polys[2 * j]].n_polys)
        return p2mt::fail(P2MT_EINVAL, "fri: batch polynomial index out of range");
    max_cnt = batches[b].n_polys > max_cnt ? batches[b].n_polys : max_cnt;
    total_cnt += batches[b].n_polys;
  }
  const size_t total = p2mt_fri_proof_len(p, n_oracles, n_polys.data());
  const size_t final_len = (size_t)1 << (log_n - total_arity_bits(p));
  const size_t off_final = total - 1 - 2 * final_len;
  const size_t max_groups = (max_cnt + kComposeGroup - 1) / kComposeGroup;

  // ---- workspace (u64 words), one grow-only slot
  size_t wsz = 0;
  auto carve = [&](size_t words) {
    const size_t at = wsz;
    wsz += (words + 3) & ~(size_t)3;  // 32-byte granules
    return at;
  };
  const size_t o_alpha = carve(2), o_betas = carve(16), o_qch = carve(1 + p->num_query_rounds), o_chsave = carve(32), o_cnt = carve(1);
  const size_t o_powbase = carve(12);
  const size_t o_apow = carve(2 * (max_cnt + 1)), o_partial = carve(max_groups * 2 * n);
  const size_t o_fin = carve(2 * (size_t)n), o_c0 = carve(2 * (size_t)n), o_c1 = carve(2 * (size_t)n);
  size_t o_vals[8], o_leaves[8], o_dig[8];
  {
    unsigned log_sz = log_big;
    for (uint32_t l = 0; l < p->num_reductions; ++l) {
      const unsigned ab = p->reduction_arity_bits[l];
      o_vals[l] = carve((size_t)2 << log_sz);
      o_leaves[l] = carve((size_t)2 << log_sz);
      o_dig[l] = carve(4 * digests_count((size_t)1 << (log_sz - ab), p->cap_height));
      log_sz -= ab;
    }
  }
  u64* ws;
  P2MT_TRY(p2mt::scratch_get(p2mt::kScratchFri, wsz * 8, (void**)&ws));
  u64* d_ptrs;  // the table of coefficient pointers is shared by the proofs of a batch (k_fri_compose rebases its entries)
  P2MT_TRY(p2mt::scratch_get_shared(p2mt::kScratchPtrs, total_cnt * 8, (void**)&d_ptrs));
  hipStream_t st = rt().stream;
  const unsigned B = p2mt::batch_B();

  // ---- alpha, composition, quotients
  const bool host_tr = hch != nullptr && link != nullptr;
  if (host_tr && B != 1) return p2mt::fail(P2MT_EINVAL, "fri: the host transcript serves single proofs only");
  if (host_tr) {
    u64 a[2];
    hch->squeeze(a, 2);
    P2MT_TRY(p2mt::hostlink_put(a, 2, ws + o_alpha));
  } else {
    P2MT_TRY(launch_challenger(ch->d, nullptr, 0, ws + o_alpha, 2));
  }
  hipLaunchKernelGGL(k_ext_powers, bgrid(grid_for(max_cnt + 1)), dim3(kBlock), 0, st, (const u64*)(ws + o_alpha),
                     (u32)max_cnt + 1, ws + o_apow, barg());
  P2MT_LAUNCH_CHECK();
  std::vector<const u64*> h_ptrs(total_cnt);
  {
    size_t k = 0;
    for (size_t b = 0; b < n_batches; ++b)
      for (size_t j = 0; j < batches[b].n_polys; ++j)
        h_ptrs[k++] = oracles[batches[b].polys[2 * j]].coeffs + (size_t)batches[b].polys[2 * j + 1] * n;
  }
  {  // the table is the same for every proof of a circuit: upload it only when it (or its place in the scratch) changed
    thread_local std::vector<const u64*> last_ptrs;
    thread_local const u64* last_dst = nullptr;
    thread_local uint64_t last_epoch = ~0ull;
    if (last_dst != d_ptrs || last_epoch != p2mt::scratch_epoch() || last_ptrs != h_ptrs) {
      P2MT_HIP(hipMemcpyAsync(d_ptrs, h_ptrs.data(), total_cnt * 8, hipMemcpyHostToDevice, st));
      P2MT_HIP(hipStreamSynchronize(st));  // pageable source; rare
      last_ptrs = h_ptrs;
      last_dst = d_ptrs;
      last_epoch = p2mt::scratch_epoch();
    }
  }
  {
    size_t k = 0;
    for (size_t b = 0; b < n_batches; ++b) {
      const u32 cnt = (u32)batches[b].n_polys, groups = (cnt + kComposeGroup - 1) / kComposeGroup;
      hipLaunchKernelGGL(k_fri_compose, bgrid(grid_for(n), groups), dim3(kBlock), 0, st,
                         reinterpret_cast<const u64* const*>(d_ptrs + k), cnt, n, (const u64*)(ws + o_apow), ws + o_partial, barg());
      P2MT_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_fri_quotient, bgrid(1), dim3(kQBlock), 0, st, (const u64*)(ws + o_partial), groups, n, pts.d_point[b],
                         pts.scale[b] % gl::P, (const u64*)(ws + o_apow + 2 * cnt), b == 0 ? 1 : 0, ws + o_fin,
                         b + 1 == n_batches ? ws + o_c0 : (u64*)nullptr, barg());
      P2MT_LAUNCH_CHECK();
      k += cnt;
    }
  }

  // ---- fri_committed_trees
  QArgs qa;
  memset(&qa, 0, sizeof qa);
  for (size_t o = 0; o < n_oracles; ++o)
    qa.t[o] = QTree{oracles[o].leaves, oracles[o].digests, (u32)oracles[o].n_polys, log_big, 0, log_big - p->cap_height};
  qa.n_trees = (u32)n_oracles;
  qa.log_big = log_big;
  u64 *cur = ws + o_c0, *nxt = ws + o_c1;
  u64 shift = 7;
  unsigned log_sz = log_big, log_deg = log_n, idx_shift = 0;
  for (uint32_t l = 0; l < p->num_reductions; ++l) {
    const unsigned ab = p->reduction_arity_bits[l];
    const size_t sz = (size_t)1 << log_sz, rows = sz >> ab;
    u64* vals = ws + o_vals[l];
    u64* leaves = ws + o_leaves[l];
    const size_t nd = digests_count(rows, p->cap_height);
    u64* dig = nd ? ws + o_dig[l] : nullptr;
    // values on the coset shift * <w>, already in the bit-reversed order the leaves are chunked in
    P2MT_TRY(p2mt::coset_lde_leaf_order_dev(cur, log_deg, p->rate_bits, shift, 2, vals));
    hipLaunchKernelGGL(k_fri_pairs, bgrid(grid_for(sz)), dim3(kBlock), 0, st, (const u64*)vals, (u32)sz, leaves, barg());
    P2MT_LAUNCH_CHECK();
    P2MT_TRY(p2mt_merkle_cap_commit_dev(leaves, rows, (size_t)2 << ab, p->cap_height, dig, d_proof + l * cap_words));
    if (host_tr) {
      const uint64_t* h_cap;
      P2MT_TRY(p2mt::hostlink_fetch(link, d_proof + l * cap_words, cap_words, &h_cap));
      hch->observe(h_cap, cap_words);
      u64 b2[2];
      hch->squeeze(b2, 2);
      P2MT_TRY(p2mt::hostlink_put(b2, 2, ws + o_betas + 2 * l));
    } else {
      P2MT_TRY(launch_challenger(ch->d, d_proof + l * cap_words, cap_words, ws + o_betas + 2 * l, 2));
    }
    hipLaunchKernelGGL(k_fri_fold, bgrid(grid_for((size_t)1 << (log_deg - ab))), dim3(kBlock), 0, st, (const u64*)cur, 1u << log_deg,
                       ab, (const u64*)(ws + o_betas + 2 * l), nxt, barg());
    P2MT_LAUNCH_CHECK();
    idx_shift += ab;
    qa.t[qa.n_trees++] = QTree{leaves, dig, (u32)(2u << ab), log_sz - ab, idx_shift, log_sz - ab - p->cap_height};
    u64* t = cur; cur = nxt; nxt = t;
    shift = h_pow(shift, (u64)1 << ab);
    log_sz -= ab;
    log_deg -= ab;
  }
  // final polynomial -> proof words, observe
  hipLaunchKernelGGL(k_fri_pairs, bgrid(grid_for(final_len)), dim3(kBlock), 0, st, (const u64*)cur, (u32)final_len,
                     d_proof + off_final, barg());
  P2MT_LAUNCH_CHECK();
  // (the same launch keeps a copy of the transcript for the proof-of-work rollback and marks "no witness yet")
  if (host_tr) {
    const uint64_t* h_fin;
    P2MT_TRY(p2mt::hostlink_fetch(link, d_proof + off_final, 2 * final_len, &h_fin));
    hch->observe(h_fin, 2 * final_len);
    // the grind kernels read the transcript's state from the device: send it up (29 words), and mark "no witness yet"
    static_assert(sizeof(host_poseidon::Challenger) == sizeof(ChState), "host and device transcript states share one layout");
    u64 stw[30];
    memcpy(stw, hch, sizeof(ChState));
    stw[29] = ~0ull;
    P2MT_TRY(p2mt::hostlink_put(stw, 29, reinterpret_cast<u64*>(ch->d)));
    P2MT_TRY(p2mt::hostlink_put(stw + 29, 1, d_proof + total - 1));
  } else {
  P2MT_TRY(launch_challenger(ch->d, d_proof + off_final, 2 * final_len, nullptr, 0, false,
                             reinterpret_cast<ChState*>(ws + o_chsave), reinterpret_cast<unsigned long long*>(d_proof + total - 1)));
  }

  // ---- fri_proof_of_work (smallest witness, searched in chunks) + fri_prover_query_rounds.
  // The rest of the proof is enqueued behind each chunk's grind on the assumption that it finds a witness (it does with
  // probability 1 - e^-2); if not, the challenger is rolled back and the next chunk is tried.
  {
    unsigned long long* d_wit = reinterpret_cast<unsigned long long*>(d_proof + total - 1);
    ChState* d_saved = reinterpret_cast<ChState*>(ws + o_chsave);
    const u64 chunk = (u64)1 << (p->proof_of_work_bits + 1 < 17 ? 17 : p->proof_of_work_bits + 1);
    qa.per_query = p->num_query_rounds ? (off_final - p->num_reductions * cap_words) / p->num_query_rounds : 0;
    const bool quad_grind = rt().mds == 2 && rt().use_quad && !rt().force_fallback && !rt().throughput && B == 1;
    const u64 batch_chunk = std::max<u64>(chunk, (u64)1 << (p->proof_of_work_bits + 4 < 32 ? p->proof_of_work_bits + 4 : 32));
    // A single proof's chunk of 2^17 candidates on the same queue kernel: 512 workgroups, one block each = two wavefronts per SIMD of the
    // one-hash-per-lane permutation (shared first round, matrix-pipe MDS, one word of the last layer) instead of eight of the
    // four-lanes-per-hash one.  (env P2MT_GRIND_QUEUE=0: the four-lane kernel, for A/B.)
    static const bool queue_knob = [] { const char* e = getenv("P2MT_GRIND_QUEUE"); return e ? atoi(e) != 0 : true; }();
    const bool single_queue = B == 1 && queue_knob && rt().mds == 2 && rt().partial == 0 && !rt().throughput;
    auto grind = [&](u64 base) -> int {
      if (single_queue) {
        const u32 max_blocks = (u32)(chunk / kBlock);
        // (Measured with tools/grind_probe.py under rocprofv3: this launch takes 56-58 us in an inner prove and ~70 us in an outer one --
        // the same code at the clock the heavier prove leaves the chip at.  Until round 5 about one launch in five took 115-140 us: a
        // wavefront whose rare-carry flag was set redid all 64 candidates with the exact permutation, ~70 us on a lone wavefront; it
        // redoes the flagged lanes only now, 67-76 us.  Which launches those are is a function of the transcript.)
        hipLaunchKernelGGL((k_fri_pow_queue<2, 5>), dim3(max_blocks), dim3(kBlock), 0, st, (const ChState*)ch->d,
                           (u32)p->proof_of_work_bits, base, max_blocks, d_wit, (u32*)nullptr, 1u, barg(), p2mt::perm_ctx());
        P2MT_LAUNCH_CHECK();
        return P2MT_OK;
      }
      if (B > 1) {  // one resident grid serves every proof of the batch (k_fri_pow_queue)
        P2MT_TRY(p2mt::batch_fill(ws + o_cnt, 0, 8));
        const u32 max_blocks = (u32)(batch_chunk / kBlock);
        const u64 wgs = std::min<u64>((u64)B * max_blocks, 2048);
        if (rt().mds == 2 && rt().partial == 0) {  // default: dense MDS layers on the matrix pipe
          hipLaunchKernelGGL(k_fri_pow_prep, bgrid(1), dim3(64), 0, st, (const ChState*)ch->d, ws + o_powbase, barg(), p2mt::perm_ctx());
          P2MT_LAUNCH_CHECK();
          hipLaunchKernelGGL((k_fri_pow_queue<2, 5>), dim3((unsigned)wgs), dim3(kBlock), 0, st, (const ChState*)ch->d,
                             (u32)p->proof_of_work_bits, base, max_blocks, d_wit, reinterpret_cast<u32*>(ws + o_cnt), B, barg(),
                             p2mt::perm_ctx(), (const u64*)(ws + o_powbase));
          P2MT_LAUNCH_CHECK();
          return P2MT_OK;
        }
        P2MT_DISPATCH(k_fri_pow_queue, dim3((unsigned)wgs), kBlock, (const ChState*)ch->d, (u32)p->proof_of_work_bits, base, max_blocks,
                      d_wit, reinterpret_cast<u32*>(ws + o_cnt), B, barg());
        return P2MT_OK;
      }
      if (quad_grind) {
        hipLaunchKernelGGL(k_fri_pow_quad, bgrid(grid_for(4 * chunk)), dim3(kBlock), 0, st, (const ChState*)ch->d,
                           (u32)p->proof_of_work_bits, base, chunk, d_wit, barg(), p2mt::perm_ctx());
        P2MT_LAUNCH_CHECK();
      } else {
        P2MT_DISPATCH(k_fri_pow, bgrid(grid_for(chunk)), kBlock, (const ChState*)ch->d, (u32)p->proof_of_work_bits, base, chunk, d_wit,
                      barg());
      }
      return P2MT_OK;
    };
    auto tail = [&]() -> int {  // observe the witness; pow_response + one challenge per query round; the query rounds
      P2MT_TRY(launch_challenger(ch->d, d_proof + total - 1, 1, ws + o_qch, 1 + p->num_query_rounds));
      if (p->num_query_rounds) {
        hipLaunchKernelGGL(k_fri_queries, bgrid(p->num_query_rounds), dim3(kBlock), 0, st, qa, (const u64*)(ws + o_qch + 1),
                           d_proof + p->num_reductions * cap_words, barg());
        P2MT_LAUNCH_CHECK();
      }
      return P2MT_OK;
    };
    if (B > 1) {
      // A batch grinds chunk after chunk (2^(pow_bits + 4) candidates: one chunk almost always) until every proof has its witness
      // -- the work queue of k_fri_pow_queue keeps the smallest one, like the single-proof search --, then finishes all the
      // proofs with one tail.
      std::vector<unsigned long long> found(B);
      for (u64 base = 0;; base += batch_chunk) {
        if (base >= ((u64)1 << 48)) return p2mt::fail(P2MT_EHIP, "fri: proof-of-work search exhausted");
        P2MT_TRY(grind(base));
        P2MT_HIP(hipMemcpy2DAsync(found.data(), 8, d_wit, p2mt::batch().arg.stride, 8, B, hipMemcpyDeviceToHost, st));
        P2MT_HIP(hipStreamSynchronize(st));
        bool all = true;
        for (unsigned i = 0; i < B; ++i) all = all && found[i] != ~0ull;
        if (all) break;
      }
      P2MT_TRY(tail());
      if (epi_dst && epi_bytes)
        P2MT_HIP(hipMemcpy2DAsync(epi_dst, epi_dpitch, epi_src, p2mt::batch().arg.stride, epi_bytes, B, hipMemcpyDeviceToHost, st));
      P2MT_HIP(hipStreamSynchronize(st));
      return P2MT_OK;
    }
    if (host_tr) {
      // the host waits for the witness (one word through the link), observes it, derives the proof-of-work response and the query
      // indices and sends them up in front of the query rounds; a chunk without a witness (probability e^-2) just grinds the next one
      for (u64 base = 0;; base += chunk) {
        if (base >= ((u64)1 << 48)) return p2mt::fail(P2MT_EHIP, "fri: proof-of-work search exhausted");
        P2MT_TRY(grind(base));
        const uint64_t* h_wit;
        P2MT_TRY(p2mt::hostlink_fetch(link, d_proof + total - 1, 1, &h_wit));
        if (h_wit[0] == ~0ull) continue;
        hch->observe(h_wit[0]);
        u64 q[32];
        const size_t nq1 = 1 + p->num_query_rounds;
        if (nq1 > 32) return p2mt::fail(P2MT_EINVAL, "fri: more than 31 query rounds");
        hch->squeeze(q, nq1);
        P2MT_TRY(p2mt::hostlink_put(q, nq1, ws + o_qch));
        if (p->num_query_rounds) {
          hipLaunchKernelGGL(k_fri_queries, bgrid(p->num_query_rounds), dim3(kBlock), 0, st, qa, (const u64*)(ws + o_qch + 1),
                             d_proof + p->num_reductions * cap_words, barg());
          P2MT_LAUNCH_CHECK();
        }
        break;
      }
      if (epi_dst && epi_bytes) P2MT_HIP(hipMemcpyAsync(epi_dst, epi_src, epi_bytes, hipMemcpyDeviceToHost, st));
      P2MT_HIP(hipStreamSynchronize(st));
      return P2MT_OK;
    }
    unsigned long long found = ~0ull;
    for (u64 base = 0;; base += chunk) {
      if (base >= ((u64)1 << 48)) return p2mt::fail(P2MT_EHIP, "fri: proof-of-work search exhausted");
      if (base != 0) P2MT_HIP(hipMemsetAsync(d_wit, 0xFF, 8, st));  // first chunk: marked by the launch above
      P2MT_TRY(grind(base));
      P2MT_TRY(tail());
      P2MT_HIP(hipMemcpyAsync(&found, d_wit, 8, hipMemcpyDeviceToHost, st));
      if (epi_dst && epi_bytes) P2MT_HIP(hipMemcpyAsync(epi_dst, epi_src, epi_bytes, hipMemcpyDeviceToHost, st));
      P2MT_HIP(hipStreamSynchronize(st));
      if (found != ~0ull) break;
      P2MT_HIP(hipMemcpyAsync(ch->d, d_saved, sizeof(ChState), hipMemcpyDeviceToDevice, st));
    }
  }
  P2MT_HIP(hipStreamSynchronize(st));  // h_ptrs and the caller's view of d_proof
  return P2MT_OK;
}

extern "C" int p2mt_fri_prove_openings(const p2mt_fri_oracle* oracles, size_t n_oracles, const p2mt_fri_batch* batches,
                                       size_t n_batches, const p2mt_fri_params* p, p2mt_challenger* ch, uint64_t* proof_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!oracles || !proof_out || n_oracles == 0) return p2mt::fail(P2MT_EINVAL, "fri: null argument");
  if (!params_ok(p)) return p2mt::fail(P2MT_EINVAL, "fri: unsupported FriParams");
  const unsigned log_big = p->degree_bits + p->rate_bits;
  const size_t n = (size_t)1 << p->degree_bits, big = (size_t)1 << log_big;
  const size_t nd = digests_count(big, p->cap_height);
  std::vector<DevBuf> bufs(3 * n_oracles + 1);
  std::vector<p2mt_fri_oracle> dev(n_oracles);
  std::vector<uint64_t> n_polys(n_oracles);
  hipStream_t st = rt().stream;
  for (size_t o = 0; o < n_oracles; ++o) {
    const size_t k = oracles[o].n_polys;
    if (!oracles[o].coeffs || !oracles[o].leaves || k == 0 || (nd && !oracles[o].digests)) return p2mt::fail(P2MT_EINVAL, "fri: bad oracle");
    P2MT_TRY(bufs[3 * o].alloc(k * n * 8));
    P2MT_TRY(bufs[3 * o + 1].alloc(k * big * 8));
    P2MT_TRY(bufs[3 * o + 2].alloc((nd ? nd : 1) * 32));
    P2MT_HIP(hipMemcpyAsync(bufs[3 * o].p, oracles[o].coeffs, k * n * 8, hipMemcpyHostToDevice, st));
    P2MT_HIP(hipMemcpyAsync(bufs[3 * o + 1].p, oracles[o].leaves, k * big * 8, hipMemcpyHostToDevice, st));
    if (nd) P2MT_HIP(hipMemcpyAsync(bufs[3 * o + 2].p, oracles[o].digests, nd * 32, hipMemcpyHostToDevice, st));
    dev[o] = p2mt_fri_oracle{bufs[3 * o].as<u64>(), bufs[3 * o + 1].as<u64>(), bufs[3 * o + 2].as<u64>(), k};
    n_polys[o] = k;
  }
  const size_t total = p2mt_fri_proof_len(p, n_oracles, n_polys.data());
  DevBuf& bp = bufs[3 * n_oracles];
  P2MT_TRY(bp.alloc(total * 8));
  P2MT_TRY(p2mt_fri_prove_openings_dev(dev.data(), n_oracles, batches, n_batches, p, ch, bp.as<u64>()));
  P2MT_HIP(hipMemcpyAsync(proof_out, bp.p, total * 8, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  return P2MT_OK;
  });
}
