// p2mt_hash.hip -- stateless Poseidon batch kernels, PoseidonGate witness rows and the power-of-two Merkle tree.
//
// Replaces the hashing loops of
//   /root/reference/src/simple_merkle_tree/simple_merkle_tree.rs:18-109   (MerkleTree, verify_merkle_proof)
// and the per-hash plonky2 Hasher calls under them.  HBM layout: a HashOut is a 32-byte record (4 x u64, AoS)
// exactly as the reference's Vec<HashOut>, so `MerkleTree.tree` (level-major) can be copied out verbatim.
// A level uses one lane, four lanes (DPP quad) or one wavefront per node depending on how many nodes it has
// (DESIGN.md 4.3).  The MMR lives in p2mt_mmr.hip; device helpers shared by both are in tree_common.hip.h.
#include "tree_common.hip.h"

#include <string.h>

#include <vector>

using namespace p2mt_dev;

namespace {

// ---------------------------------------------------------------- stateless batch kernels
template <int M, int PR>
__global__ __launch_bounds__(kBlock, 4) void k_permute_batch(const u64* __restrict__ in, u64* __restrict__ out, size_t n,
                                                          PermCtx ctx) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  u64 s[12];
  permute_reloadable<M, PR>(s, ctx, [&](u64 (&st)[12]) {
#pragma unroll
    for (int k = 0; k < 12; ++k) st[k] = in[12 * i + k];
  });
#pragma unroll
  for (int k = 0; k < 12; ++k) out[12 * i + k] = gl::canon(s[k]);
}

template <int M, int PR>
__global__ __launch_bounds__(kBlock, 4) void k_two_to_one_batch(const u64* __restrict__ in, u64* __restrict__ out, size_t n,
                                                             PermCtx ctx) {
  poseidon_fast::MfmaCtx mc;  // PR == 5: matrix-pipe MDS; every lane stays in the permutation, lanes past the end redo the last pair
  if constexpr (PR == 5) poseidon_fast::mfma32_ctx_init(mc);
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = i < n;
  if constexpr (PR == 5) i = live ? i : n - 1;
  else if (!live) return;
  u64 o[4];
  two_to_one_r<M, PR>(ctx, o, [&](u64 (&l)[4], u64 (&r)[4]) {
    load_hash(in + 8 * i, l);
    load_hash(in + 8 * i + 4, r);
  }, &mc);
  if (live) store_hash(out + 4 * i, o);
}

template <int M, int PR>
__global__ __launch_bounds__(kBlock, 4) void k_hash_rows(const u64* __restrict__ in, size_t n, size_t len, int noop_short,
                                                      u64* __restrict__ out, BatchArg ba, PermCtx ctx) {
  in = bp(in, ba);
  out = bp(out, ba);
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  u64 o[4];
  if (noop_short && len <= 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (size_t)k < len ? gl::canon(in[i * len + k]) : 0;
  } else {
    const u64* row = in + i * len;
    sponge<M, PR>(len, ctx, o, [&](size_t k) { return row[k]; });
  }
  store_hash(out + 4 * i, o);
}

// Same rows, one wavefront per row (latency path for small batches, e.g. the 2^7..2^11 leaves of a FRI layer tree):
// the sponge's permutations are sequential, so a row costs len/8 wave-permutation latencies (~8 us each) instead of
// len/8 single-lane ones (~60 us each).
__global__ __launch_bounds__(kBlock) void k_hash_rows_wave(const u64* __restrict__ in, size_t n, size_t len, int noop_short,
                                                           u64* __restrict__ out, BatchArg ba, PermCtx ctx) {
  in = bp(in, ba);
  out = bp(out, ba);
  __shared__ u64 rc_lds[kWaveRcWords];
  ctx = stage_round_constants(rc_lds, ctx);
  // four rows per wavefront, one per 16-lane row of it (permute_wave4); `len` is the same for all, so the loop is wave-uniform
  const size_t row0 = 4 * ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
  if (row0 >= n) return;  // wave-uniform
  const unsigned lane = threadIdx.x & 15;
  const size_t row = row0 + ((threadIdx.x >> 4) & 3);
  const bool live = row < n;
  const u64* r = in + (live ? row : row0) * len;
  if (noop_short && len <= 4) {
    if (live && lane < 4) out[4 * row + lane] = lane < len ? gl::canon(r[lane]) : 0;
    return;
  }
  u64 x = 0;
  u64 nx = (lane < 8 && lane < len) ? r[lane] : 0;  // the next chunk's word is fetched under the current permutation
#pragma unroll 1
  for (size_t off = 0; off < len; off += 8) {
    if (lane < 8 && off + lane < len) x = nx;  // overwrite mode: the other words keep the previous state
    if (lane < 8 && off + 8 + lane < len) nx = r[off + 8 + lane];
    x = permute_wave4(x, ctx);
  }
  if (live && lane < 4) out[4 * row + lane] = gl::canon(x);
}

// ---------------------------------------------------------------- PoseidonGate witness rows
// One row per lane, exact spec-form arithmetic (this is a throughput kernel for batches of proofs; a single proof's
// rows form a dependent chain and belong on the host).  Stores are wire-major, so consecutive lanes write
// consecutive addresses of each of the 135 wire columns.
__global__ __launch_bounds__(kBlock) void k_poseidon_gate_witness(const u64* __restrict__ in, const uint8_t* __restrict__ swaps,
                                                                  size_t n, u64* __restrict__ wires) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    s[k] = gl::canon(in[12 * i + k]);
    wires[(size_t)k * n + i] = s[k];
  }
  const bool swap = swaps[i] != 0;
  wires[(size_t)24 * n + i] = swap ? 1 : 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u64 delta = swap ? gl::canon(gl::sub_c(s[k + 4], s[k])) : 0;
    wires[(size_t)(25 + k) * n + i] = delta;
    const u64 l = gl::canon(gl::add_c(s[k], delta)), r = gl::canon(gl::sub_c(s[k + 4], delta));
    s[k] = l;
    s[k + 4] = r;
  }
#pragma unroll 1
  for (int r = 0; r < POSEIDON_ROUNDS; ++r) {
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = gl::canon(gl::add_c(s[k], POSEIDON_RC[12 * r + k]));
    if (r < 4 || r >= 26) {
      if (r >= 1) {
        const int base = r < 4 ? 29 + 12 * (r - 1) : 87 + 12 * (r - 26);
#pragma unroll
        for (int k = 0; k < 12; ++k) wires[(size_t)(base + k) * n + i] = s[k];
      }
#pragma unroll
      for (int k = 0; k < 12; ++k) s[k] = gl::pow7(s[k]);
    } else {
      wires[(size_t)(65 + (r - 4)) * n + i] = s[0];
      s[0] = gl::pow7(s[0]);
    }
    poseidon::mds_mad64(s);
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) wires[(size_t)(12 + k) * n + i] = gl::canon(s[k]);
}

// ---------------------------------------------------------------- simple_merkle_tree.rs
// level0[i] = hash_or_noop([leaf]) = [leaf, 0, 0, 0]  (:33; no permutation, Quirk Q1)
__global__ __launch_bounds__(kBlock) void k_leaf_digests(const u64* __restrict__ leaves, u64* __restrict__ level0, size_t n) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const u64 o[4] = {gl::canon(leaves[i]), 0, 0, 0};
  store_hash(level0 + 4 * i, o);
}

// next_level_hashes (:21-25): out[j] = two_to_one(in[2j], in[2j+1])
template <int M, int PR>
__global__ __launch_bounds__(kBlock, 4) void k_merkle_level(const u64* __restrict__ in, u64* __restrict__ out, size_t n_out,
                                                         BatchArg ba, PermCtx ctx) {
  in = bp(in, ba);
  out = bp(out, ba);
  // PR == 5 (matrix-pipe MDS, the default of the fast path): an MFMA ignores EXEC and every lane's A operand serves the whole wave,
  // so no lane leaves before the permutation -- lanes past the end redo the last node and skip the store
  poseidon_fast::MfmaCtx mc;
  if constexpr (PR == 5) poseidon_fast::mfma32_ctx_init(mc);
  size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = j < n_out;
  if constexpr (PR == 5) j = live ? j : n_out - 1;
  else if (!live) return;
  u64 o[4];
  two_to_one_r<M, PR, false, true>(ctx, o, [&](u64 (&l)[4], u64 (&r)[4]) {  // (exact folds: see k_mmr_level)
    load_hash(in + 8 * j, l);
    load_hash(in + 8 * j + 4, r);
  }, &mc);
  if (live) store_hash(out + 4 * j, o);
}

// Stage 1 of a large MerkleTree::build, the level-major twin of k_mmr_subtree (p2mt_mmr.hip): each LANE builds the perfect subtree
// over its own 2^LV consecutive leaves depth-first -- 2^LV - 1 chained permutations, no barrier, pending left siblings in a per-lane LDS
// stack -- and writes the leaf digests and every node to their level-major slots (level l of an n-leaf tree starts 2n - (2n >> l)
// digests into `levels`; a lane's nodes of one level are contiguous).  Matrix-pipe MDS: lanes past the end rebuild the last subtree
// and store nothing.
template <unsigned LV>
__global__ __launch_bounds__(256, 4) void k_merkle_subtree(const u64* __restrict__ leaves, u64* __restrict__ levels, size_t n, PermCtx ctx) {
  __shared__ __attribute__((aligned(16))) u64 stack[LV - 1][256 * 4];
  // (k_mmr_subtree, p2mt_mmr.hip, has the measurements behind the two things below: a whole 64-byte sector of leaves per fetch with the
  // other pairs parked in LDS, and the lane's indices worked out afresh in every step instead of living -- spilled -- across the hashes)
  constexpr unsigned kG = LV >= 3 ? 4 : 2;
  __shared__ __attribute__((aligned(16))) u64 lcache[kG - 1][256 * 2];
  poseidon_fast::MfmaCtx mc;
  poseidon_fast::mfma32_ctx_init(mc);
  const size_t n_blocks = n >> LV;
  const unsigned wave_base = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.x) & ~63u;
  auto thread_index = [&]() -> unsigned {
    unsigned t = wave_base + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(t));
    return t;
  };
  auto level_at = [&](unsigned h) -> u64* { return levels + 4 * (2 * n - ((2 * n) >> h)); };
  u64 cur[4] = {0, 0, 0, 0};
  unsigned pairs_done = 0, h = 0, merges = 0;
#pragma unroll 1
  for (unsigned step = 0; step < (1u << LV) - 1; ++step) {
    const unsigned tid = thread_index();
    size_t blk = (size_t)blockIdx.x * 256 + tid;
    const bool live = blk < n_blocks;
    blk = live ? blk : n_blocks - 1;
    const size_t first_leaf = blk << LV;
    const u64* lp = leaves + first_leaf;
    u64 o[4];
    if (merges == 0) {  // hash the next leaf pair
      const unsigned slot = pairs_done % kG;  // wave-uniform
      u64 a, b;
      if (slot == 0) {
        const ulonglong2* q = reinterpret_cast<const ulonglong2*>(lp + 2 * pairs_done);
        ulonglong2 pr[kG];
#pragma unroll
        for (unsigned k = 0; k < kG; ++k) pr[k] = q[k];
        a = pr[0].x, b = pr[0].y;
#pragma unroll
        for (unsigned k = 1; k < kG; ++k) reinterpret_cast<ulonglong2*>(&lcache[k - 1][tid * 2])[0] = pr[k];
      } else {
        const ulonglong2 pr = reinterpret_cast<const ulonglong2*>(&lcache[slot - 1][tid * 2])[0];
        a = pr.x, b = pr.y;
      }
      a = gl::canon(a), b = gl::canon(b);
      const size_t leaf = first_leaf + 2 * pairs_done;
      if (live) {
        const u64 la[4] = {a, 0, 0, 0}, lb[4] = {b, 0, 0, 0};
        store_hash(levels + 4 * leaf, la);  // hash_or_noop([leaf]) = [leaf, 0, 0, 0]
        store_hash(levels + 4 * (leaf + 1), lb);
      }
      two_to_one_r<IMPL_FAST, 5, true>(ctx, o, [&](u64 (&ll)[4], u64 (&rr)[4]) {
        ll[0] = a; ll[1] = ll[2] = ll[3] = 0;
        rr[0] = b; rr[1] = rr[2] = rr[3] = 0;
      }, &mc);
      merges = (unsigned)__builtin_ctz(~pairs_done);
      pairs_done += 1;
      h = 1;
    } else {  // merge the pending left sibling of height h with cur
      two_to_one_r<IMPL_FAST, 5>(ctx, o, [&](u64 (&ll)[4], u64 (&rr)[4]) {
        load_hash(&stack[h - 1][thread_index() * 4], ll);
#pragma unroll
        for (int k = 0; k < 4; ++k) rr[k] = cur[k];
      }, &mc);
      merges -= 1;
      h += 1;
    }
    {
      const unsigned tid2 = thread_index();
      size_t blk2 = (size_t)blockIdx.x * 256 + tid2;
      const bool live2 = blk2 < n_blocks;
      blk2 = live2 ? blk2 : n_blocks - 1;
      // the node just made: height h, index ((first leaf + 2 * pairs_done) >> h) - 1 of its level
      if (live2) store_hash(level_at(h) + 4 * ((((blk2 << LV) + 2 * pairs_done) >> h) - 1), o);
#pragma unroll
      for (int k = 0; k < 4; ++k) cur[k] = o[k];
      if (merges == 0 && h < LV) store_hash(&stack[h - 1][tid2 * 4], cur);
    }
  }
}

// verify_merkle_proof (:91-109), one proof per lane
template <int M, int PR>
__global__ __launch_bounds__(kBlock, 4) void k_verify_merkle_proof(const u64* __restrict__ leaves, const u64* __restrict__ idx,
                                                                const u64* __restrict__ roots,
                                                                const u64* __restrict__ hashes, size_t n_hashes, size_t m,
                                                                uint8_t* __restrict__ result, PermCtx ctx) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  u64 cur[4] = {gl::canon(leaves[i]), 0, 0, 0};
  u64 index = idx[i];
#pragma unroll 1
  for (size_t k = 0; k < n_hashes; ++k) {
    const bool even = (index & 1) == 0;
    u64 nxt[4];
    two_to_one_r<M, PR>(ctx, nxt, [&](u64 (&l)[4], u64 (&r)[4]) {
      u64 sib[4];
      load_hash(hashes + 4 * (i * n_hashes + k), sib);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        l[t] = even ? cur[t] : sib[t];
        r[t] = even ? sib[t] : cur[t];
      }
    });
#pragma unroll
    for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
    index >>= 1;
  }
  u64 root[4];
  load_hash(roots + 4 * i, root);
  bool ok = true;
#pragma unroll
  for (int t = 0; t < 4; ++t) ok = ok && (cur[t] == gl::canon(root[t]));
  result[i] = ok ? 1 : 0;
}

// level-major tree level: out[j] = two_to_one(in[2j], in[2j+1])
__global__ __launch_bounds__(kBlock) void k_merkle_level_wave(const u64* __restrict__ in, u64* __restrict__ out, size_t n_out,
                                                              BatchArg ba, PermCtx ctx) {
  in = bp(in, ba);
  out = bp(out, ba);
  __shared__ u64 rc_lds[kWaveRcWords];
  ctx = stage_round_constants(rc_lds, ctx);
  const size_t j0 = 4 * ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));  // four nodes per wavefront, one per 16-lane row
  if (j0 >= n_out) return;  // wave-uniform
  two_to_one_wave4(in, out, j0, n_out, ctx);
}

__global__ __launch_bounds__(kBlock) void k_merkle_level_quad(const u64* __restrict__ in, u64* __restrict__ out, size_t n_out,
                                                              BatchArg ba, PermCtx ctx) {
  in = bp(in, ba);
  out = bp(out, ba);
  const size_t j = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 2;
  if (j >= n_out) return;  // quad-uniform
  poseidon_quad::Lane ln;
  poseidon_quad::lane_init(ln, ctx.rc);
  two_to_one_quad(in + 8 * j, in + 8 * j + 4, out + 4 * j, ln);
}

}  // namespace

// Launch KERNEL<mds, partial> for the runtime-selected variant; the PermCtx is appended as the last argument.

namespace p2mt {

// exported to the other translation units
int launch_hash_rows_dev(const u64* d_in, size_t n, size_t len, int noop_short, u64* d_out) {
  if (n == 0) return P2MT_OK;
  // (inside a batch of B proofs the layout is chosen for all n B rows of the launch)
  if (n * p2mt::batch_B() <= ((size_t)1 << 12) && rt().mds == 2) {  // small batch: one wavefront per row (latency path)
    const unsigned per_block = 4 * (kBlock / 64);
    hipLaunchKernelGGL(k_hash_rows_wave, bgrid((unsigned)((n + per_block - 1) / per_block)), dim3(kBlock), 0, rt().stream,
                       d_in, n, len, noop_short, d_out, barg(), p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
    return P2MT_OK;
  }
  P2MT_DISPATCH(k_hash_rows, bgrid(grid_for(n)), kBlock, d_in, n, len, noop_short, d_out, barg());
  return P2MT_OK;
}
int launch_merkle_level_dev(const u64* d_in, u64* d_out, size_t n_out) {
  if (n_out == 0) return P2MT_OK;
  const size_t n_all = n_out * p2mt::batch_B();
  if (n_all <= ((size_t)1 << 13) && rt().mds == 2 && !(rt().throughput && n_all > 16)) {  // small level: four nodes per wavefront on the 12-lane layout (latency path)
    const unsigned per_block = 4 * (kBlock / 64);
    hipLaunchKernelGGL(k_merkle_level_wave, bgrid((unsigned)((n_out + per_block - 1) / per_block)), dim3(kBlock), 0,
                       rt().stream, d_in, d_out, n_out, barg(), p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
    return P2MT_OK;
  }
  if (n_all <= ((size_t)1 << 16) && rt().mds == 2 && rt().use_quad) {
    hipLaunchKernelGGL(k_merkle_level_quad, bgrid(grid_for(4 * n_out)), dim3(kBlock), 0, rt().stream, d_in, d_out, n_out,
                       barg(), p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
    return P2MT_OK;
  }
  if (rt().mds == 2 && rt().partial == 0) {  // default: dense MDS layers on the matrix pipe
    hipLaunchKernelGGL((k_merkle_level<2, 5>), bgrid(grid_for(n_out)), dim3(kBlock), 0, rt().stream, d_in, d_out, n_out, barg(),
                       p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
    return P2MT_OK;
  }
  P2MT_DISPATCH(k_merkle_level, bgrid(grid_for(n_out)), kBlock, d_in, d_out, n_out, barg());
  return P2MT_OK;
}

}  // namespace p2mt

using p2mt::DevBuf;
using p2mt::rt;

// =================================================================== stateless batch entry points
// =================================================================== test hook: one group of partial rounds on arbitrary states
// (the carry-out of a four-round group's last link needs all twelve halves of a state within 4 % of 2^32: no hash input gets there,
// so the tests hand the states in directly)
namespace {
template <int G, bool LEAD>
__global__ __launch_bounds__(kBlock) void k_debug_partial_group(const u64* __restrict__ in, size_t n, unsigned table_off,
                                                                u64* __restrict__ out, uint8_t* __restrict__ flag, PermCtx ctx) {
  const size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const size_t i = i0 < n ? i0 : n - 1;
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) s[k] = in[12 * i + k];
  u64 sticky = 0;
  const poseidon_fast::ctab rc = poseidon_fast::as_const_table(ctx.rc);
  auto sbox = [&](u64 x) -> u64 { return poseidon_fast::pow7<0>(x, sticky); };
  poseidon_fast::partial_rounds_g<G, LEAD>(s, rc + table_off, sbox, sticky);
  if (i0 < n) {
#pragma unroll
    for (int k = 0; k < 12; ++k) out[12 * i + k] = gl::canon(s[k]);
    flag[i] = (uint8_t)((sticky >> (threadIdx.x & 63)) & 1);
  }
}
}  // namespace

extern "C" int p2mt_debug_partial_group(int group, const uint64_t* states, size_t n, uint64_t* out, uint8_t* flag_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n == 0) return P2MT_OK;
  using namespace poseidon_fast;
  if (!states || !out || !flag_out || group < 0 || group >= kPG4Groups + kPG3Groups) return p2mt::fail(P2MT_EINVAL, "bad argument");
  p2mt::DevBuf bi, bo, bf;
  P2MT_TRY(bi.alloc(n * 96));
  P2MT_TRY(bo.alloc(n * 96));
  P2MT_TRY(bf.alloc(n));
  hipStream_t st = p2mt::rt().stream;
  P2MT_HIP(hipMemcpyAsync(bi.p, states, n * 96, hipMemcpyHostToDevice, st));
  const unsigned off = group < kPG4Groups ? kPGTab + pg_words(4) * group : kPGTab + pg_words(4) * kPG4Groups + pg_words(3) * (group - kPG4Groups);
  const dim3 grid(grid_for(n)), block(kBlock);
  if (group == 0) hipLaunchKernelGGL((k_debug_partial_group<4, false>), grid, block, 0, st, (const u64*)bi.as<u64>(), n, off, bo.as<u64>(), bf.as<uint8_t>(), p2mt::perm_ctx());
  else if (group < kPG4Groups) hipLaunchKernelGGL((k_debug_partial_group<4, true>), grid, block, 0, st, (const u64*)bi.as<u64>(), n, off, bo.as<u64>(), bf.as<uint8_t>(), p2mt::perm_ctx());
  else hipLaunchKernelGGL((k_debug_partial_group<3, true>), grid, block, 0, st, (const u64*)bi.as<u64>(), n, off, bo.as<u64>(), bf.as<uint8_t>(), p2mt::perm_ctx());
  P2MT_LAUNCH_CHECK();
  P2MT_HIP(hipMemcpyAsync(out, bo.p, n * 96, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipMemcpyAsync(flag_out, bf.p, n, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  return P2MT_OK;
  });
}

extern "C" int p2mt_poseidon_permute_batch_dev(const uint64_t* d_in, uint64_t* d_out, size_t n) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n == 0) return P2MT_OK;
  if (!d_in || !d_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  P2MT_DISPATCH(k_permute_batch, grid_for(n), kBlock, d_in, d_out, n);
  return P2MT_OK;
  });
}

extern "C" int p2mt_poseidon_permute_batch(const uint64_t* in, uint64_t* out, size_t n) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n == 0) return P2MT_OK;
  if (!in || !out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  DevBuf bi, bo;  // distinct buffers: the kernel's pointers are __restrict__
  P2MT_TRY(bi.alloc(n * 96));
  P2MT_TRY(bo.alloc(n * 96));
  P2MT_HIP(hipMemcpyAsync(bi.p, in, n * 96, hipMemcpyHostToDevice, rt().stream));
  P2MT_TRY(p2mt_poseidon_permute_batch_dev(bi.as<u64>(), bo.as<u64>(), n));
  P2MT_HIP(hipMemcpyAsync(out, bo.p, n * 96, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

extern "C" int p2mt_two_to_one_batch_dev(const uint64_t* d_in, uint64_t* d_out, size_t n) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n == 0) return P2MT_OK;
  if (!d_in || !d_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (rt().mds == 2 && rt().partial == 0 && n >= 4096) {  // default, once a launch fills whole waves: dense MDS layers on the matrix pipe
    hipLaunchKernelGGL((k_two_to_one_batch<2, 5>), dim3(grid_for(n)), dim3(kBlock), 0, rt().stream, d_in, d_out, n, p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
    return P2MT_OK;
  }
  P2MT_DISPATCH(k_two_to_one_batch, grid_for(n), kBlock, d_in, d_out, n);
  return P2MT_OK;
  });
}

extern "C" int p2mt_two_to_one_batch(const uint64_t* in, uint64_t* out, size_t n) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n == 0) return P2MT_OK;
  if (!in || !out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  DevBuf bi, bo;
  P2MT_TRY(bi.alloc(n * 64));
  P2MT_TRY(bo.alloc(n * 32));
  P2MT_HIP(hipMemcpyAsync(bi.p, in, n * 64, hipMemcpyHostToDevice, rt().stream));
  P2MT_TRY(p2mt_two_to_one_batch_dev(bi.as<u64>(), bo.as<u64>(), n));
  P2MT_HIP(hipMemcpyAsync(out, bo.p, n * 32, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

static int hash_rows_host(const uint64_t* in, size_t n, size_t len, int noop_short, uint64_t* out) {
  P2MT_TRY(p2mt::ensure_init());
  if (n == 0) return P2MT_OK;
  if (!out || (!in && len)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  DevBuf bi, bo;
  P2MT_TRY(bi.alloc(n * len * 8));
  P2MT_TRY(bo.alloc(n * 32));
  if (len) P2MT_HIP(hipMemcpyAsync(bi.p, in, n * len * 8, hipMemcpyHostToDevice, rt().stream));
  P2MT_TRY(p2mt::launch_hash_rows_dev(bi.as<u64>(), n, len, noop_short, bo.as<u64>()));
  P2MT_HIP(hipMemcpyAsync(out, bo.p, n * 32, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
}

extern "C" int p2mt_hash_or_noop_batch(const uint64_t* in, size_t n, size_t len, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  return hash_rows_host(in, n, len, 1, out);
  });
}
extern "C" int p2mt_hash_no_pad_batch(const uint64_t* in, size_t n, size_t len, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  return hash_rows_host(in, n, len, 0, out);
  });
}
extern "C" int p2mt_hash_or_noop_batch_dev(const uint64_t* d_in, size_t n, size_t len, uint64_t* d_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  return p2mt::launch_hash_rows_dev(d_in, n, len, 1, d_out);
  });
}
extern "C" int p2mt_hash_no_pad_batch_dev(const uint64_t* d_in, size_t n, size_t len, uint64_t* d_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  return p2mt::launch_hash_rows_dev(d_in, n, len, 0, d_out);
  });
}

extern "C" int p2mt_poseidon_gate_witness_batch_dev(const uint64_t* d_inputs, const uint8_t* d_swaps, size_t n,
                                                    uint64_t* d_wires_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n == 0) return P2MT_OK;
  if (!d_inputs || !d_swaps || !d_wires_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  hipLaunchKernelGGL(k_poseidon_gate_witness, dim3(grid_for(n)), dim3(kBlock), 0, rt().stream, d_inputs, d_swaps, n,
                     d_wires_out);
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
  });
}

extern "C" int p2mt_poseidon_gate_witness_batch(const uint64_t* inputs, const uint8_t* swaps, size_t n, uint64_t* wires_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (n == 0) return P2MT_OK;
  if (!inputs || !swaps || !wires_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  DevBuf bi, bs, bo;
  P2MT_TRY(bi.alloc(n * 96));
  P2MT_TRY(bs.alloc(n));
  P2MT_TRY(bo.alloc(n * 135 * 8));
  hipStream_t st = rt().stream;
  P2MT_HIP(hipMemcpyAsync(bi.p, inputs, n * 96, hipMemcpyHostToDevice, st));
  P2MT_HIP(hipMemcpyAsync(bs.p, swaps, n, hipMemcpyHostToDevice, st));
  P2MT_TRY(p2mt_poseidon_gate_witness_batch_dev(bi.as<u64>(), bs.as<uint8_t>(), n, bo.as<u64>()));
  P2MT_HIP(hipMemcpyAsync(wires_out, bo.p, n * 135 * 8, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  return P2MT_OK;
  });
}

// =================================================================== simple_merkle_tree.rs
static int log2_strict(size_t n) {
  if (n == 0 || (n & (n - 1))) return -1;
  return __builtin_ctzll((unsigned long long)n);
}

extern "C" int p2mt_merkle_build_pow2_dev(const uint64_t* d_leaves, size_t n, uint64_t* d_levels, uint64_t* d_root) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  const int k = log2_strict(n);
  if (k < 1) return p2mt::fail(P2MT_EINVAL, "MerkleTree::build: leaf count must be a power of two >= 2");
  if (!d_leaves || !d_levels || !d_root) return p2mt::fail(P2MT_EINVAL, "null pointer");
  u64* cur = d_levels;
  size_t cur_n = n;
  int i0 = 0;
  // large trees (the sizes at which the MMR build uses it too): stage 1 as per-lane subtrees -- leaf digests and levels 1..lv in one
  // barrier-free launch; the level-by-level loop continues above it.  The subtree size adapts to the tree like the MMR's.
  const unsigned lv = (rt().mds == 2 && rt().partial == 0 && rt().subtree_auto && k >= 18) ? p2mt::subtree_levels_for(n) : 0;
  if (lv >= 2 && lv <= 4 && (unsigned)k > lv) {
    const unsigned grid = (unsigned)(((n >> lv) + 255) / 256);
    if (lv == 4) hipLaunchKernelGGL((k_merkle_subtree<4>), dim3(grid), dim3(256), 0, rt().stream, d_leaves, d_levels, n, p2mt::perm_ctx());
    else if (lv == 3) hipLaunchKernelGGL((k_merkle_subtree<3>), dim3(grid), dim3(256), 0, rt().stream, d_leaves, d_levels, n, p2mt::perm_ctx());
    else hipLaunchKernelGGL((k_merkle_subtree<2>), dim3(grid), dim3(256), 0, rt().stream, d_leaves, d_levels, n, p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
    for (unsigned l = 0; l < lv; ++l) {
      cur += 4 * cur_n;
      cur_n /= 2;
    }
    i0 = (int)lv;
  } else {
    hipLaunchKernelGGL(k_leaf_digests, dim3(grid_for(n)), dim3(kBlock), 0, rt().stream, d_leaves, d_levels, n);
    P2MT_LAUNCH_CHECK();
  }
  for (int i = i0; i < k - 1; ++i) {  // levels i0 + 1 .. k-1
    u64* next = cur + 4 * cur_n;
    P2MT_TRY(p2mt::launch_merkle_level_dev(cur, next, cur_n / 2));
    cur = next;
    cur_n /= 2;
  }
  return p2mt::launch_merkle_level_dev(cur, d_root, 1);  // root = two_to_one(last[0], last[1])
  });
}

extern "C" int p2mt_merkle_build_pow2(const uint64_t* leaves, size_t n, uint64_t* levels_out, uint64_t* root_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  const int k = log2_strict(n);
  if (k < 1) return p2mt::fail(P2MT_EINVAL, "MerkleTree::build: leaf count must be a power of two >= 2");
  if (!leaves || !levels_out || !root_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  DevBuf bl, bt;
  P2MT_TRY(bl.alloc(n * 8));
  P2MT_TRY(bt.alloc((2 * n - 2 + 1) * 32));
  u64* d_root = bt.as<u64>() + 4 * (2 * n - 2);
  P2MT_HIP(hipMemcpyAsync(bl.p, leaves, n * 8, hipMemcpyHostToDevice, rt().stream));
  P2MT_TRY(p2mt_merkle_build_pow2_dev(bl.as<u64>(), n, bt.as<u64>(), d_root));
  P2MT_HIP(hipMemcpyAsync(levels_out, bt.p, (2 * n - 2) * 32, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipMemcpyAsync(root_out, d_root, 32, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

static const uint64_t* level_ptr(const uint64_t* levels, size_t n, int level) {
  size_t off = 0;
  for (int i = 0; i < level; ++i) off += n >> i;
  return levels + 4 * off;
}

extern "C" int p2mt_merkle_get_proof(const uint64_t* levels, size_t n, size_t leaf_index, uint64_t* proof_out) {
  return p2mt::abi_guard([&]() -> int {
  const int k = log2_strict(n);
  if (k < 1 || !levels || !proof_out) return p2mt::fail(P2MT_EINVAL, "get_merkle_proof: bad tree");
  if (leaf_index >= n) return p2mt::fail(P2MT_EINVAL, "get_merkle_proof: assert!(leaf_index < n)");
  size_t idx = leaf_index;
  for (int i = 0; i < k; ++i, idx >>= 1) memcpy(proof_out + 4 * i, level_ptr(levels, n, i) + 4 * (idx ^ 1), 32);
  return P2MT_OK;
  });
}

extern "C" int p2mt_merkle_get_in_between_hashes(const uint64_t* levels, const uint64_t* root, size_t n,
                                                 size_t leaf_index, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  const int k = log2_strict(n);
  if (k < 1 || !levels || !root || !out) return p2mt::fail(P2MT_EINVAL, "get_in_between_hashes: bad tree");
  if (leaf_index >= n) return p2mt::fail(P2MT_EINVAL, "get_in_between_hashes: assert!(leaf_index < n)");
  size_t idx = leaf_index >> 1;
  for (int i = 1; i < k; ++i, idx >>= 1) memcpy(out + 4 * (i - 1), level_ptr(levels, n, i) + 4 * idx, 32);
  memcpy(out + 4 * (k - 1), root, 32);
  return P2MT_OK;
  });
}

extern "C" int p2mt_verify_merkle_proof_batch(const uint64_t* leaves, const uint64_t* leaf_indices, const uint64_t* roots,
                                              const uint64_t* hashes, size_t n_hashes, size_t m, uint8_t* result_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (m == 0) return P2MT_OK;
  if (!leaves || !leaf_indices || !roots || (!hashes && n_hashes) || !result_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  DevBuf bl, bi, br, bh, bo;
  P2MT_TRY(bl.alloc(m * 8));
  P2MT_TRY(bi.alloc(m * 8));
  P2MT_TRY(br.alloc(m * 32));
  P2MT_TRY(bh.alloc(m * n_hashes * 32));
  P2MT_TRY(bo.alloc(m));
  hipStream_t st = rt().stream;
  P2MT_HIP(hipMemcpyAsync(bl.p, leaves, m * 8, hipMemcpyHostToDevice, st));
  P2MT_HIP(hipMemcpyAsync(bi.p, leaf_indices, m * 8, hipMemcpyHostToDevice, st));
  P2MT_HIP(hipMemcpyAsync(br.p, roots, m * 32, hipMemcpyHostToDevice, st));
  if (n_hashes) P2MT_HIP(hipMemcpyAsync(bh.p, hashes, m * n_hashes * 32, hipMemcpyHostToDevice, st));
  P2MT_DISPATCH(k_verify_merkle_proof, grid_for(m), kBlock, bl.as<u64>(), bi.as<u64>(), br.as<u64>(), bh.as<u64>(),
                n_hashes, m, bo.as<uint8_t>());
  P2MT_HIP(hipMemcpyAsync(result_out, bo.p, m, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  return P2MT_OK;
  });
}

