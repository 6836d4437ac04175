// p2mt_mmr.hip -- the device-resident Merkle Mountain Range.
//
// Replaces the hashing loops of
//   /root/reference/src/mmr/merkle_mountain_ranges.rs:84-252              (MMR, MMR_proof)
// HBM layout: `MMR.elements` is the reference's post-order Vec<HashOut> (32-byte AoS records), copied out verbatim.
// The MMR is built level-synchronously straight into its post-order positions: the node of height h whose last
// leaf is L lives at 2L - popcount(L) + h, its right child at pos-1, its left child at pos-2^h.
// Kernels by regime: stage 1 (the dominant launch) is k_mmr_subtree -- every lane builds levels
// 1..4 of its own 16 leaves depth-first, barrier-free (k_mmr_tile, the LDS-fused predecessor, serves the exact
// variants); levels above use one lane, four lanes (DPP quad) or one wavefront per node depending on how many nodes
// the level has.
#include "tree_common.hip.h"

#include <string.h>

#include <algorithm>
#include <vector>

using namespace p2mt_dev;

namespace {

// ---------------------------------------------------------------- merkle_mountain_ranges.rs
// add_leaf's push of hash_or_noop([leaf]) (:91/:96/:104) for leaves [n0, n0+k): leaf i sits at 2i - popcount(i)
__global__ __launch_bounds__(kBlock) void k_mmr_leaves(const u64* __restrict__ leaves, u64* __restrict__ elements, size_t n0,
                                                       size_t k) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= k) return;
  const size_t i = n0 + t;
  const u64 o[4] = {gl::canon(leaves[t]), 0, 0, 0};
  store_hash(elements + 4 * node_pos(i, 0), o);
}

// the carry chain of add_leaf (:106-119), level-synchronous: all height-h nodes j in [j0, j1)
// (four waves per SIMD asked for: without the hint the scheduler of the PR == 5 form spills SGPRs to VGPR lanes -- 93 v_readlane /
// v_writelane per full round -- and settles at 131 VGPRs = three waves)
template <int M, int PR>
__global__ __launch_bounds__(kBlock, 4) void k_mmr_level(u64* __restrict__ elements, unsigned h, size_t j0, size_t j1,
                                                      PermCtx ctx) {
  // PR == 5 (matrix-pipe MDS, the default of the fast path): an MFMA ignores EXEC and every lane's A operand serves the whole wave,
  // so no lane may leave before the permutation -- lanes past the end redo the last node and skip the store
  poseidon_fast::MfmaCtx mc;
  if constexpr (PR == 5) poseidon_fast::mfma32_ctx_init(mc);
  size_t j = j0 + (size_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = j < j1;
  if constexpr (PR == 5) j = live ? j : j1 - 1;
  else if (!live) return;
  const size_t last_leaf = ((j + 1) << h) - 1;
  const size_t pos = node_pos(last_leaf, h);
  u64 o[4];
  // (exact folds: a wave that redid its hash in the last round of a level launch held the launch for ~60 us)
  two_to_one_r<M, PR, false, true>(ctx, o, [&](u64 (&l)[4], u64 (&r)[4]) {
    load_hash(elements + 4 * (pos - ((size_t)1 << h)), l);
    load_hash(elements + 4 * (pos - 1), r);
  }, &mc);
  if (live) store_hash(elements + 4 * pos, o);
}

// Fused multi-level build of aligned tiles: one workgroup takes 2^kTileLog consecutive nodes of height h0 (raw
// leaves when h0 == 0) and produces the next n_levels levels, handing digests from level to level through LDS
// (ping-pong buffers, 48 KB) and writing every node once to its post-order slot.  HBM traffic is the
// algorithmic minimum: 8 B read per leaf, 32 B written per node.  The caller stops a stage while every level
// still fills whole waves (2^(kTileLog - n_levels) >= 64) except in the tiny top stages.
template <int M, int PR, unsigned kTileLog>
__global__ __launch_bounds__(kBlock) void k_mmr_tile(const u64* __restrict__ leaves, size_t leaf_base,
                                                     u64* __restrict__ elements, unsigned h0, unsigned n_levels,
                                                     size_t tile0, PermCtx ctx) {
  __shared__ __attribute__((aligned(16))) u64 buf0[4 << (kTileLog - 1)];
  __shared__ __attribute__((aligned(16))) u64 buf1[4 << (kTileLog - 2)];
  const size_t tile = tile0 + blockIdx.x;
  {  // first level: children come from HBM (leaves: coalesced 16 B per lane; nodes: 32-B records)
    const unsigned h = h0 + 1;
    for (unsigned i = threadIdx.x; i < (1u << (kTileLog - 1)); i += kBlock) {
      const size_t j = (tile << (kTileLog - 1)) + i;
      const size_t pos = node_pos(((j + 1) << h) - 1, h);
      u64 o[4];
      if (h0 == 0) {
        const u64* lp = leaves + (2 * j - leaf_base);  // 8-byte aligned only (leaf_base may be odd)
        const u64 l[4] = {gl::canon(lp[0]), 0, 0, 0}, r[4] = {gl::canon(lp[1]), 0, 0, 0};
        store_hash(elements + 4 * (pos - 2), l);  // hash_or_noop([leaf]) = [leaf, 0, 0, 0]
        store_hash(elements + 4 * (pos - 1), r);
        two_to_one_r<M, PR>(ctx, o, [&](u64 (&ll)[4], u64 (&rr)[4]) {
          ll[0] = gl::canon(lp[0]); ll[1] = ll[2] = ll[3] = 0;
          rr[0] = gl::canon(lp[1]); rr[1] = rr[2] = rr[3] = 0;
        });
      } else {
        two_to_one_r<M, PR>(ctx, o, [&](u64 (&ll)[4], u64 (&rr)[4]) {
          load_hash(elements + 4 * (pos - ((size_t)1 << h)), ll);
          load_hash(elements + 4 * (pos - 1), rr);
        });
      }
      store_hash(elements + 4 * pos, o);
      store_hash(buf0 + 4 * i, o);
    }
  }
  __syncthreads();
  for (unsigned lvl = 2; lvl <= n_levels; ++lvl) {
    const unsigned h = h0 + lvl;
    const u64* src = (lvl & 1) ? buf1 : buf0;
    u64* dst = (lvl & 1) ? buf0 : buf1;
    for (unsigned i = threadIdx.x; i < (1u << (kTileLog - lvl)); i += kBlock) {
      const size_t j = (tile << (kTileLog - lvl)) + i;
      const size_t pos = node_pos(((j + 1) << h) - 1, h);
      u64 o[4];
      two_to_one_r<M, PR>(ctx, o, [&](u64 (&ll)[4], u64 (&rr)[4]) {
        load_hash(src + 8 * i, ll);
        load_hash(src + 8 * i + 4, rr);
      });
      store_hash(elements + 4 * pos, o);
      store_hash(dst + 4 * i, o);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- per-lane subtrees (stage 1, barrier-free)
// Each LANE builds the perfect subtree over its own 2^LV consecutive leaves depth-first (post-order), so every lane
// hashes on every step, waves never wait for each other and there is no barrier: the loop is 2^LV - 1 iterations of
// ONE inlined permutation.  Pending left siblings (at most one per height) live in a per-lane LDS stack; the control
// flow is the binary-counter carry chain and is identical in all lanes.  A lane's 2^(LV+1)-1 nodes are a contiguous
// post-order span, written node by node.  HBM traffic is the same algorithmic minimum as k_mmr_tile.
// (the shipped instantiation asks for four waves per SIMD, i.e. <= 128 VGPRs: a shard of 2^22 / 2^23 leaves is 4 / 8 waves per SIMD
// and ran as 3 + 1 / 3 + 3 + 2 at the three waves the allocator settles for on its own; at 2^24 it makes no difference)
template <unsigned LV, int BLK, int PR = 0, int OCC = 1>
__global__ __launch_bounds__(BLK, OCC) void k_mmr_subtree(const u64* __restrict__ leaves, size_t leaf_base,
                                                     u64* __restrict__ elements, size_t block0, size_t n_blocks,
                                                     PermCtx ctx) {
  __shared__ __attribute__((aligned(16))) u64 stack[LV - 1][BLK * 4];  // stack[h-1][lane]: pending left sibling of height h < LV
  // A lane's 2^LV leaves are one or two 64-byte sectors of HBM, and one 16-byte load per leaf pair fetched each sector four times over
  // (the other ~530 MB of the launch's FETCH_SIZE once the spills were gone): the lane fetches kG pairs -- a whole sector -- at the
  // first pair that needs it and parks the others here (12 KB per workgroup; four workgroups per CU still fit the 160 KB)
  constexpr unsigned kG = LV >= 3 ? 4 : 2;
  __shared__ __attribute__((aligned(16))) u64 lcache[kG - 1][BLK * 2];
  poseidon_fast::MfmaCtx mc;  // PR >= 2 (matrix-pipe MDS): per-lane A operands, made while every lane is active (MFMA ignores EXEC)
  if constexpr (PR == 2 || PR == 3) mc = poseidon_fast::mfma_ctx_init();
  if constexpr (PR == 5 || PR == 7 || PR == 8) poseidon_fast::mfma32_ctx_init(mc);
  // no lane leaves before the loop (PR == 5 / 7: an MFMA ignores EXEC and every lane's A operand serves the whole wave; a copy or spill
  // of it made under a partial EXEC would lose the idle lanes' rows): lanes past the end rebuild the last subtree and store nothing
  if constexpr (!(PR == 5 || PR == 7 || PR == 8))
    if (block0 + (size_t)blockIdx.x * BLK + threadIdx.x >= block0 + n_blocks) return;
  u64 cur[4] = {0, 0, 0, 0};
  unsigned pairs_done = 0;  // leaf pairs consumed so far
  unsigned h = 0;           // height of `cur`
  unsigned merges = 0;      // merges still owed before the next leaf pair
  // (the thread index itself: its wave's first thread in a scalar register + the lane number from v_mbcnt, so that not even v0 has to
  // stay alive across the permutations)
  const unsigned wave_base = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.x) & ~63u;
  auto thread_index = [&]() -> unsigned {
    unsigned t = wave_base + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(t));
    return t;
  };
#pragma unroll 1
  for (unsigned step = 0; step < (1u << LV) - 1; ++step) {
    // The lane's own indices -- its subtree, its first leaf, its leaf pointer, its LDS slots -- are worked out again in every step from
    // an opaque copy of the thread index (a dozen integer instructions per 10 k of a hash).  Computed once in front of the loop they
    // were eight registers alive across every permutation; the allocator spilled them (9 VGPRs, 40 bytes of scratch) and reloaded them
    // in every step, and THOSE reloads were the launch's extra traffic: ~28 bytes per lane and step = 440-600 MB of FETCH_SIZE per
    // 2^24-leaf launch against 134 MB of leaves (r04: 589 MB).  The leaf reads and the node stores were never the cause: fetching a
    // whole 64-byte sector per lane and writing a pair's three nodes together were both built in round 5 and changed no counter.
    const unsigned tid = thread_index();
    size_t blk = block0 + (size_t)blockIdx.x * BLK + tid;
    const bool live = blk < block0 + n_blocks;
    blk = live ? blk : block0 + n_blocks - 1;
    const size_t first_leaf = blk << LV;
    const u64* lp = leaves + (first_leaf - leaf_base);
    u64 o[4];
    if (merges == 0) {  // hash the next leaf pair
      const unsigned slot = pairs_done % kG;  // wave-uniform
      u64 a, b;
      if (slot == 0) {  // (8-byte aligned only: leaf_base may be odd)
        const u64* q = lp + 2 * pairs_done;
        u64 w[2 * kG];
#pragma unroll
        for (unsigned k = 0; k < 2 * kG; ++k) w[k] = q[k];
        a = w[0], b = w[1];
#pragma unroll
        for (unsigned k = 1; k < kG; ++k) reinterpret_cast<ulonglong2*>(&lcache[k - 1][tid * 2])[0] = make_ulonglong2(w[2 * k], w[2 * k + 1]);
      } else {
        const ulonglong2 pr = reinterpret_cast<const ulonglong2*>(&lcache[slot - 1][tid * 2])[0];
        a = pr.x, b = pr.y;
      }
      a = gl::canon(a), b = gl::canon(b);
      const size_t pos = node_pos(first_leaf + 2 * pairs_done + 1, 1);
      const u64 la[4] = {a, 0, 0, 0}, lb[4] = {b, 0, 0, 0};
      if (live) {
        store_hash(elements + 4 * (pos - 2), la);  // hash_or_noop([leaf]) = [leaf, 0, 0, 0]
        store_hash(elements + 4 * (pos - 1), lb);
      }
      two_to_one_r<IMPL_FAST, PR, true>(ctx, o, [&](u64 (&ll)[4], u64 (&rr)[4]) {  // leaf pair: 10 of the 12 first S-boxes are constants
        // (the first call takes the values in hand; the practically-never redo re-reads the pair where it lies, indices worked out afresh)
        ll[0] = a; ll[1] = ll[2] = ll[3] = 0;
        rr[0] = b; rr[1] = rr[2] = rr[3] = 0;
      }, &mc);
      merges = (unsigned)__builtin_ctz(~pairs_done);  // trailing ones: carries of the binary counter
      pairs_done += 1;
      h = 1;
    } else {  // merge the pending left sibling of height h with cur
      two_to_one_r<IMPL_FAST, PR>(ctx, o, [&](u64 (&ll)[4], u64 (&rr)[4]) {
        load_hash(&stack[h - 1][thread_index() * 4], ll);
#pragma unroll
        for (int k = 0; k < 4; ++k) rr[k] = cur[k];
      }, &mc);
      merges -= 1;
      h += 1;
    }
    {  // (indices again behind the permutation, for the same reason)
      const unsigned tid2 = thread_index();
      size_t blk2 = block0 + (size_t)blockIdx.x * BLK + tid2;
      const bool live2 = blk2 < block0 + n_blocks;
      blk2 = live2 ? blk2 : block0 + n_blocks - 1;
      const size_t fl2 = blk2 << LV;
      // the node just made: height h, last leaf fl2 + 2 * pairs_done - 1 (a leaf pair's parent included: pairs_done was advanced)
      if (live2) store_hash(elements + 4 * node_pos(fl2 + 2 * pairs_done - 1, h), o);
#pragma unroll
      for (int k = 0; k < 4; ++k) cur[k] = o[k];
      if (merges == 0 && h < LV) store_hash(&stack[h - 1][tid2 * 4], cur);  // becomes a pending left sibling
    }
  }
}

// MMR level (post-order, in place): node j of height h
__global__ __launch_bounds__(kBlock) void k_mmr_level_wave(u64* __restrict__ elements, unsigned h, size_t j0, size_t j1,
                                                           PermCtx ctx) {
  __shared__ u64 rc_lds[kWaveRcWords];
  ctx = stage_round_constants(rc_lds, ctx);
  // four nodes per wavefront, one per 16-lane row (tree_common.hip.h permute_wave4)
  const size_t jw = j0 + 4 * ((size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
  if (jw >= j1) return;  // wave-uniform
  const unsigned rl = threadIdx.x & 15, row = (threadIdx.x >> 4) & 3;
  const size_t j = jw + row;
  const bool live = j < j1;
  const size_t pos = node_pos((((live ? j : jw) + 1) << h) - 1, h);
  u64 x = 0;
  if (live && rl < 8) x = rl < 4 ? elements[4 * (pos - ((size_t)1 << h)) + rl] : elements[4 * (pos - 1) + (rl - 4)];
  x = permute_wave4(x, ctx);
  if (live && rl < 4) elements[4 * pos + rl] = gl::canon(x);
}

__global__ __launch_bounds__(kBlock) void k_mmr_level_quad(u64* __restrict__ elements, unsigned h, size_t j0, size_t j1,
                                                           PermCtx ctx) {
  const size_t j = j0 + (((size_t)blockIdx.x * kBlock + threadIdx.x) >> 2);
  if (j >= j1) return;  // quad-uniform
  poseidon_quad::Lane ln;
  poseidon_quad::lane_init(ln, ctx.rc);
  const size_t pos = node_pos(((j + 1) << h) - 1, h);
  two_to_one_quad(elements + 4 * (pos - ((size_t)1 << h)), elements + 4 * (pos - 1), elements + 4 * pos, ln);
}

struct PosList {
  u64 pos[P2MT_MAX_PROOF_LEN];
  int n;
};

// get_peaks (:179-200) gather + bagging_the_peaks (:122-127): one lane (<= 32 permutations)
template <int M, int PR>
__global__ void k_mmr_peaks_root(const u64* __restrict__ elements, PosList pl, u64* __restrict__ peaks_out,
                                 u64* __restrict__ root_out, const u32* __restrict__ plan_state, PermCtx ctx) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (plan_state && plan_state[1] != 0) {  // a hand-off of the one-launch build gave up (p2mt_plan.hip): no root, the host reports it
    for (int k = 0; k < 4; ++k) root_out[k] = ~0ull;
    return;
  }
  for (int i = 0; i < pl.n; ++i) {
    u64 h[4];
    load_hash(elements + 4 * pl.pos[i], h);
    store_hash(peaks_out + 4 * i, h);
  }
  u64 o[4];
  if (pl.n == 1) {  // hash_or_noop of 4 elements: the peak itself (Quirk Q2)
    load_hash(peaks_out, o);
  } else {
    sponge<M, PR>((size_t)pl.n * 4, ctx, o, [&](size_t k) { return peaks_out[k]; });
  }
  store_hash(root_out, o);
}

// bagging of caller-supplied peaks (MMR_proof::verify :248-249)
template <int M, int PR>
__global__ void k_bag_peaks(const u64* __restrict__ peaks, int n_peaks, u64* __restrict__ root_out, PermCtx ctx) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u64 o[4];
  if (n_peaks * 4 <= 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = k < n_peaks * 4 ? gl::canon(peaks[k]) : 0;
  } else {
    sponge<M, PR>((size_t)n_peaks * 4, ctx, o, [&](size_t k) { return peaks[k]; });
  }
  store_hash(root_out, o);
}

// The top log2(world) levels above the gathered shard roots in ONE launch: a single 1024-lane workgroup, one
// wavefront per node on the 12-lane layout, levels handed on through LDS.  world <= kMaxWorld.
constexpr unsigned kMaxWorld = 1024;
__global__ __launch_bounds__(1024) void k_combine_roots(const u64* __restrict__ roots, unsigned world, u64* __restrict__ top,
                                                        u64* __restrict__ root_out, PermCtx ctx) {
  __shared__ u64 rc_lds[kWaveRcWords];
  __shared__ __attribute__((aligned(16))) u64 lvl[2][kMaxWorld * 4];  // ping-pong: level l lives in lvl[l & 1]
  ctx = stage_round_constants(rc_lds, ctx);
  for (unsigned k = threadIdx.x; k < world * 4; k += blockDim.x) lvl[0][k] = gl::canon(roots[k]);
  __syncthreads();
  const unsigned wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  unsigned l = 0, off = 0;  // off: nodes of the lower top levels already written (level-major, bottom-up)
  for (unsigned cnt = world; cnt > 1; cnt >>= 1, ++l) {
    const u64* src = lvl[l & 1];
    u64* dst = lvl[(l + 1) & 1];
    for (unsigned j = wave; j < cnt / 2; j += n_waves) {  // wave-uniform
      two_to_one_wave(src + 8 * j, src + 8 * j + 4, dst + 4 * j, ctx);
      if (top && (threadIdx.x & 63) < 4) top[4 * (off + j) + (threadIdx.x & 63)] = dst[4 * j + (threadIdx.x & 63)];
    }
    off += cnt / 2;
    __syncthreads();
  }
  if (threadIdx.x < 4) root_out[threadIdx.x] = lvl[l & 1][threadIdx.x];
}

// remainder of the greedy perfect-subtree decomposition of x == height of the element at index x
// (get_heights_bitmap_for_mmr_size(x).1, :39-81)
__host__ __device__ inline unsigned mmr_remainder(size_t x) {
  if (x == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
  size_t sub = (~(size_t)0) >> __clzll((long long)x);
#else
  size_t sub = (~(size_t)0) >> __builtin_clzll((unsigned long long)x);
#endif
  while (sub) {
    if (x >= sub) x -= sub;
    sub >>= 1;
  }
  return (unsigned)x;
}

// get_subtree_proof_elm (:147-176): walk up from mmr_index; one proof per lane; siblings gathered from HBM
__global__ __launch_bounds__(kBlock) void k_mmr_proof_batch(const u64* __restrict__ elements, size_t len,
                                                            const u64* __restrict__ indices, size_t count,
                                                            size_t max_sib, u64* __restrict__ sib_out,
                                                            uint8_t* __restrict__ lefts_out, int32_t* __restrict__ n_out) {
  const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= count) return;
  size_t cur = indices[t];
  if (cur >= len) {  // the reference panics (index out of bounds)
    n_out[t] = -1;
    return;
  }
  int n = 0;
  unsigned h = 0;
  for (;;) {
    const size_t span = ((size_t)2 << h) - 1;
    size_t sib;
    bool left;
    if (cur >= span && mmr_remainder(cur - span) == h) {  // :157-164 element `span` before is at the same height
      sib = cur - span;
      left = true;
      cur += 1;
    } else {  // add_right_elm (:129-144)
      const size_t nxt = cur + span;
      if (!(nxt < len - 1)) break;
      sib = nxt;
      left = false;
      cur = nxt + 1;
    }
    if ((size_t)n < max_sib) {
      u64 hsh[4];
      load_hash(elements + 4 * sib, hsh);
      store_hash(sib_out + 4 * (t * max_sib + n), hsh);
      lefts_out[t * max_sib + n] = left ? 1 : 0;
    }
    ++n;
    ++h;
  }
  n_out[t] = n;
}

// MMR_proof::verify (:232-252), one proof per lane; `bagged` = hash_or_noop(peaks) computed once by k_bag_peaks
template <int M, int PR>
__global__ __launch_bounds__(kBlock, 4) void k_mmr_verify_batch(const u64* __restrict__ sib, const uint8_t* __restrict__ lefts,
                                                             const int32_t* __restrict__ n_sib, size_t max_sib,
                                                             const u64* __restrict__ peaks, int n_peaks,
                                                             const u64* __restrict__ leaves, const u64* __restrict__ root,
                                                             const u64* __restrict__ bagged, size_t m,
                                                             int8_t* __restrict__ status, PermCtx ctx) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  u64 cur[4] = {gl::canon(leaves[i]), 0, 0, 0};
  const int ns = n_sib[i];
#pragma unroll 1
  for (int k = 0; k < ns; ++k) {
    const bool on_left = lefts[i * max_sib + k] != 0;
    u64 nxt[4];
    two_to_one_r<M, PR>(ctx, nxt, [&](u64 (&l)[4], u64 (&r)[4]) {
      u64 s[4];
      load_hash(sib + 4 * (i * max_sib + k), s);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        l[t] = on_left ? s[t] : cur[t];
        r[t] = on_left ? cur[t] : s[t];
      }
    });
#pragma unroll
    for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
  }
  bool found = false;
  for (int p = 0; p < n_peaks; ++p) {
    bool eq = true;
#pragma unroll
    for (int t = 0; t < 4; ++t) eq = eq && (gl::canon(peaks[4 * p + t]) == cur[t]);
    found = found || eq;
  }
  if (!found) {
    status[i] = (int8_t)P2MT_ENOTPEAK;  // the reference panics here (:245)
    return;
  }
  bool ok = true;
#pragma unroll
  for (int t = 0; t < 4; ++t) ok = ok && (bagged[t] == gl::canon(root[t]));
  status[i] = ok ? 1 : 0;
}

// MMR_proof::verify for a FEW proofs (the reference's one-proof-at-a-time use): one wavefront per proof on the 12-lane layout.  A
// proof is a chain of n_sib dependent two_to_one -- ~9.5 us each here against ~36 us with one lane per proof (config 2's four
// proofs of 20 siblings: 0.7 ms of chain each in the lane kernel).
__global__ __launch_bounds__(kBlock) void k_mmr_verify_wave(const u64* __restrict__ sib, const uint8_t* __restrict__ lefts,
                                                            const int32_t* __restrict__ n_sib, size_t max_sib,
                                                            const u64* __restrict__ peaks, int n_peaks,
                                                            const u64* __restrict__ leaves, const u64* __restrict__ root,
                                                            const u64* __restrict__ bagged, size_t m,
                                                            int8_t* __restrict__ status, PermCtx ctx) {
  __shared__ u64 rc_lds[kWaveRcWords];
  ctx = stage_round_constants(rc_lds, ctx);
  const size_t i = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (i >= m) return;  // wave-uniform
  const unsigned lane = threadIdx.x & 63;
  u64 cur = lane == 0 ? gl::canon(leaves[i]) : 0;  // lanes 0..3: the running hash
  const int ns = n_sib[i];
#pragma unroll 1
  for (int k = 0; k < ns; ++k) {
    const bool on_left = lefts[i * max_sib + k] != 0;
    const u64 s = lane < 4 ? sib[4 * (i * max_sib + k) + lane] : 0;
    const u64 first = on_left ? s : cur, second = on_left ? cur : s;          // valid in lanes 0..3
    const u64 second_up = (u64)__shfl((unsigned long long)second, (int)(lane & 3));  // lanes 4..7 take lanes 0..3
    u64 x = lane < 4 ? first : (lane < 8 ? second_up : 0);
    x = permute_wave(x, ctx);
    cur = lane < 4 ? gl::canon(x) : 0;
  }
  bool found = false;
  for (int p = 0; p < n_peaks; ++p) {
    const bool eq = lane < 4 ? gl::canon(peaks[4 * p + lane]) == cur : true;
    found = found || (__ballot(eq) & 0xFull) == 0xFull;
  }
  if (lane == 0) {
    if (!found) {
      status[i] = (int8_t)P2MT_ENOTPEAK;  // the reference panics here (:245)
    } else {
      bool ok = true;
      for (int t = 0; t < 4; ++t) ok = ok && (bagged[t] == gl::canon(root[t]));
      status[i] = ok ? 1 : 0;
    }
  }
}

}  // namespace

using p2mt::DevBuf;
using p2mt::rt;

// =================================================================== merkle_mountain_ranges.rs
extern "C" uint64_t p2mt_get_heights_bitmap_for_mmr_size(size_t mmr_size, size_t* remainder_out) {
  // one bit per perfect subtree that fits, scanning sizes 2^(b+1)-1 from the largest not exceeding
  // the all-ones envelope of mmr_size down to 1 (:39-81)
  uint64_t bitmap = 0;
  size_t rest = mmr_size;
  if (mmr_size) {
    for (size_t sub = (~(size_t)0) >> __builtin_clzll((unsigned long long)mmr_size); sub; sub >>= 1) {
      bitmap <<= 1;
      if (rest >= sub) {
        bitmap |= 1;
        rest -= sub;
      }
    }
  }
  if (remainder_out) *remainder_out = rest;
  return bitmap;
}

extern "C" int64_t p2mt_get_mmr_index(size_t n) {
  // 2n - popcount(n); the reference accumulates 2^(i+1)-1 per set bit i in i32 and panics on overflow (:264)
  if (n >> 30) return P2MT_ERANGE;
  const int64_t r = 2 * (int64_t)n - __builtin_popcountll((unsigned long long)n);
  return r > INT32_MAX ? (int64_t)P2MT_ERANGE : r;
}

extern "C" size_t p2mt_mmr_node_pos(size_t last_leaf, unsigned height) {
  return 2 * last_leaf - (size_t)__builtin_popcountll((unsigned long long)last_leaf) + height;
}

extern "C" size_t p2mt_mmr_shard_first_pos(size_t n_local, size_t rank) {
  const size_t first_leaf = n_local * rank;
  return 2 * first_leaf - (size_t)__builtin_popcountll((unsigned long long)first_leaf);
}

struct p2mt_mmr {
  u64* elements = nullptr;  // device, post-order HashOut records
  size_t cap_nodes = 0;
  size_t n_leaves = 0;      // leaves already built into `elements`
  u64* scratch = nullptr;   // device: 64 peaks + root (65 HashOuts)
  std::vector<u64> pending; // add_leaf queue (host), flushed as one bulk extend before the MMR is observed
  size_t plan_state_cap = 0;
  uint32_t* plan_state = nullptr;  // state words of the last one-launch subtree build (p2mt_plan.hip); word 1 != 0: a hand-off gave up
};

static constexpr size_t kMaxPendingLeaves = (size_t)1 << 20;
static int mmr_extend_host(p2mt_mmr* m, const uint64_t* leaves, size_t k);
// The add_leaf queue is dropped only once its leaves are in `elements`: a failed flush (allocation, copy) leaves the queue
// intact, so n_leaves + pending -- what len() / num_leaves() report -- stays true and the call can simply be retried.
static int mmr_flush(const p2mt_mmr* cm) {
  p2mt_mmr* m = const_cast<p2mt_mmr*>(cm);  // the queue is an implementation detail of the logical state
  if (!m || m->pending.empty()) return P2MT_OK;
  P2MT_TRY(mmr_extend_host(m, m->pending.data(), m->pending.size()));
  m->pending.clear();
  return P2MT_OK;
}
// test hook (p2mt_debug_fail_allocs): the next `n` device allocations made while growing an MMR report out-of-memory
static int g_fail_allocs = 0;
extern "C" int p2mt_debug_fail_allocs(int n) {
  return p2mt::abi_guard([&]() -> int {
  g_fail_allocs = n < 0 ? 0 : n;
  return P2MT_OK;
  });
}

static size_t mmr_len_for(size_t n_leaves) { return 2 * n_leaves - (size_t)__builtin_popcountll((unsigned long long)n_leaves); }

extern "C" int p2mt_mmr_create(p2mt_mmr** out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  p2mt_mmr* m = new (std::nothrow) p2mt_mmr();
  if (!m) return p2mt::fail(P2MT_ENOMEM, "host allocation failed");
  hipError_t e = hipMalloc((void**)&m->scratch, 65 * 32);
  if (e != hipSuccess) {
    delete m;
    return p2mt::fail_hip(e, "hipMalloc(scratch)", __FILE__, __LINE__);
  }
  *out = m;
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_destroy(p2mt_mmr* m) {
  return p2mt::abi_guard([&]() -> int {
  if (!m) return P2MT_OK;
  if (m->elements) (void)hipFree(m->elements);
  if (m->scratch) (void)hipFree(m->scratch);
  if (m->plan_state) (void)hipFree(m->plan_state);
  delete m;
  return P2MT_OK;
  });
}

static int mmr_grow(p2mt_mmr* m, size_t need_nodes) {
  if (need_nodes <= m->cap_nodes) return P2MT_OK;
  size_t cap = m->cap_nodes ? m->cap_nodes : 1024;
  while (cap < need_nodes) cap *= 2;
  if (m->cap_nodes == 0) cap = need_nodes > 1024 ? need_nodes : 1024;  // first allocation: exact (reserve)
  if (g_fail_allocs > 0) {
    --g_fail_allocs;
    return p2mt::fail(P2MT_ENOMEM, "hipMalloc failed while growing the MMR (injected by p2mt_debug_fail_allocs)");
  }
  DevBuf fresh;  // released on every error path
  if (fresh.alloc(cap * 32) != P2MT_OK) return p2mt::fail(P2MT_ENOMEM, "hipMalloc failed while growing the MMR");
  const size_t used = mmr_len_for(m->n_leaves);
  if (used) P2MT_HIP(hipMemcpyAsync(fresh.p, m->elements, used * 32, hipMemcpyDeviceToDevice, rt().stream));
  if (m->elements) {
    P2MT_HIP(hipStreamSynchronize(rt().stream));
    (void)hipFree(m->elements);
  }
  m->elements = fresh.as<u64>();
  fresh.p = nullptr;
  m->cap_nodes = cap;
  return P2MT_OK;
}

extern "C" int p2mt_mmr_reserve(p2mt_mmr* m, size_t n_leaves) {
  return p2mt::abi_guard([&]() -> int {
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  if (n_leaves >> 40) return p2mt::fail(P2MT_ERANGE, "MMR too large");
  return mmr_grow(m, 2 * n_leaves);
  });
}

extern "C" int p2mt_mmr_reset(p2mt_mmr* m) {
  return p2mt::abi_guard([&]() -> int {
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  m->n_leaves = 0;
  m->pending.clear();
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_add_leaf(p2mt_mmr* m, uint64_t leaf) {
  return p2mt::abi_guard([&]() -> int {
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  m->pending.push_back(leaf);
  return m->pending.size() >= kMaxPendingLeaves ? mmr_flush(m) : P2MT_OK;
  });
}

extern "C" int p2mt_mmr_flush(p2mt_mmr* m) {
  return p2mt::abi_guard([&]() -> int {
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  return mmr_flush(m);
  });
}

// one level over [j0, j1) of height h: one wavefront per node while the level is small (latency-bound), one
// lane per node otherwise
constexpr size_t kWavePerNodeMax = (size_t)1 << 13;  // four nodes per wavefront (permute_wave4): 2^13 nodes are two wavefronts per SIMD
constexpr size_t kMinTilesPerStage = 2048;
// stage 1 at four waves per SIMD (<= 128 VGPRs, 28 bytes of scratch) instead of the three the allocator settles for on its own
// (136 VGPRs); env P2MT_SUBTREE_OCC=3 selects the latter, for A/B
static bool subtree_occ4() {
  static const bool v = [] {
    const char* e = getenv("P2MT_SUBTREE_OCC");
    return e ? atoi(e) == 4 : true;
  }();
  return v;
}
// four lanes per node: the latency path for 2^12 < nodes <= 2^kQuadLog.  (env P2MT_QUAD_MAX_LOG overrides, for A/B.)
static size_t quad_per_node_max() {
  static const size_t v = [] {
    const char* e = getenv("P2MT_QUAD_MAX_LOG");
    const int lg = e ? atoi(e) : 16;
    return (size_t)1 << (lg >= 12 && lg <= 24 ? lg : 16);
  }();
  return v;
}

static int launch_level(p2mt_mmr* m, unsigned h, size_t j0, size_t j1) {
  if (j1 <= j0) return P2MT_OK;
  const size_t cnt = j1 - j0;
  if (cnt <= kWavePerNodeMax && rt().mds == 2) {
    const unsigned per_block = 4 * (kBlock / 64);
    hipLaunchKernelGGL(k_mmr_level_wave, dim3((unsigned)((cnt + per_block - 1) / per_block)), dim3(kBlock), 0, rt().stream,
                       m->elements, h, j0, j1, p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
    return P2MT_OK;
  }
  if (cnt <= quad_per_node_max() && rt().mds == 2 && rt().use_quad) {
    hipLaunchKernelGGL(k_mmr_level_quad, dim3(grid_for(4 * cnt)), dim3(kBlock), 0, rt().stream, m->elements, h, j0, j1,
                       p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
    return P2MT_OK;
  }
  if (rt().mds == 2 && rt().partial == 0) {  // default: dense MDS layers on the matrix pipe (poseidon_fast::mds_layer_mfma32)
    hipLaunchKernelGGL((k_mmr_level<2, 5>), dim3(grid_for(cnt)), dim3(kBlock), 0, rt().stream, m->elements, h, j0, j1, p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
    return P2MT_OK;
  }
  P2MT_DISPATCH(k_mmr_level, grid_for(cnt), kBlock, m->elements, h, j0, j1);
  return P2MT_OK;
}

// All new nodes of height 1..cap whose leaves lie in [lo, hi), plus the leaf digests of [lo, hi), on rt().stream.
// `d_leaves[0]` is leaf `leaf_base`.  A height-h node j is new iff it ends after leaf lo and complete iff it ends by
// hi: j in [lo>>h, hi>>h).  Stages of fused tiles (2^(h0+kTileLog)-leaf aligned blocks, n_lev levels each) cover the
// bulk while a level still has more nodes than the quad/wave-per-node paths like; ragged edges and the thin top go
// level by level.
// h_from > 0: the nodes of height <= h_from exist already (a finished chunk); only heights h_from+1..cap are built.
static int build_levels(p2mt_mmr* m, const u64* d_leaves, size_t leaf_base, size_t n0, size_t n1, unsigned cap,
                        unsigned h_from = 0) {
  if (n1 <= n0) return P2MT_OK;
  hipStream_t st = rt().stream;
  const unsigned kTileLog = rt().mds == 2 ? rt().tile_log : 11;  // 2^kTileLog inputs per workgroup (12 / 24 / 48 KB of LDS)
  unsigned h0 = h_from;
  for (;;) {
    if (h0 > 0 && (h0 >= cap || (n1 >> (h0 + 1)) <= (n0 >> (h0 + 1)))) break;  // no node above h0 (or not ours)
    // (A second per-lane stage -- two levels per lane over the roots stage 1 left, 2^18 lanes at 2^24 leaves, k_mmr_upper -- was built
    // in round 3, bit-exact, and measured: 5.56 / 5.64 ms against 5.58 / 5.58 ms per build without it.  The level launches it replaces
    // do the same work at the same occupancy; what they lose is not launch overhead.  Removed.)
    // levels this stage would fuse: stage 1 stops while every level fills whole waves; later stages also stop where
    // the quad/wave-per-node kernels take over
    unsigned n_lev = 0;
    const unsigned sub_lv = (h0 == 0 && rt().mds == 2) ? p2mt::subtree_levels_for(n1 - n0) : 0;  // per-lane subtree stage 1
    if (h0 == 0) {
      n_lev = sub_lv ? sub_lv : kTileLog - 6;  // last fused level still has 64 nodes per tile
    } else {
      while (n_lev < kTileLog - 6 && ((n1 >> (h0 + n_lev + 1)) - (n0 >> (h0 + n_lev + 1))) > kWavePerNodeMax) ++n_lev;
    }
    if (h0 + n_lev > cap) n_lev = cap > h0 ? cap - h0 : 0;
    const unsigned span_log = sub_lv ? sub_lv : h0 + kTileLog;  // log2(leaves per tile / per lane subtree)
    size_t a = n1, b = 0;
    if (n_lev && span_log < 48) {
      a = ((n0 + (((size_t)1 << span_log) - 1)) >> span_log) << span_log;
      b = (n1 >> span_log) << span_log;
    }
    const bool tiles = n_lev && a < b && (h0 == 0 || ((b - a) >> span_log) >= kMinTilesPerStage);
    if (h0 == 0) {  // leaf digests outside the tiled range (tiles write their own)
      const size_t lo_end = tiles ? a : n1;
      if (lo_end > n0) {
        hipLaunchKernelGGL(k_mmr_leaves, dim3(grid_for(lo_end - n0)), dim3(kBlock), 0, st, d_leaves + (n0 - leaf_base),
                           m->elements, n0, lo_end - n0);
        P2MT_LAUNCH_CHECK();
      }
      if (tiles && n1 > b) {
        hipLaunchKernelGGL(k_mmr_leaves, dim3(grid_for(n1 - b)), dim3(kBlock), 0, st, d_leaves + (b - leaf_base), m->elements,
                           b, n1 - b);
        P2MT_LAUNCH_CHECK();
      }
    }
    if (!tiles) {  // no fused stage from here: one launch per remaining level
      for (unsigned h = h0 + 1; h <= cap && (n1 >> h) > (n0 >> h); ++h) P2MT_TRY(launch_level(m, h, n0 >> h, n1 >> h));
      break;
    }
    if (sub_lv) {
      const size_t n_blocks = (b - a) >> span_log;
      const unsigned sb = rt().subtree_block;
      const size_t block0 = a >> span_log;
      const unsigned sgrid = (unsigned)((n_blocks + sb - 1) / sb);
      const int prof_slot = p2mt::prof_begin();  // stage 1 is the dominant launch
#define P2MT_SUB(LVV, BB, PRR) \
  hipLaunchKernelGGL((k_mmr_subtree<LVV, BB, PRR>), dim3(sgrid), dim3(BB), 0, st, d_leaves, leaf_base, m->elements, block0, n_blocks, \
                     p2mt::perm_ctx())
      // every branch launches the instantiation whose subtree size IS sub_lv (a mismatch would write out of bounds); the A/B
      // variants exist for 2^4-leaf subtrees in 256-lane workgroups only, and subtree_levels_for() never asks for anything else
      // while one of them is selected
      const int variant = (sub_lv == 4 && sb == 256) ? rt().partial : 0;
      // template PR of the fast path: 5 = the default (dense MDS layers on the matrix pipe, partial rounds in groups of four), 0 = VALU
      // MDS (variant 5), 6 = VALU MDS with the previous field multiply (variant 6), 7 = 5 with the partial rounds in groups of three
      // (variant 7), 1..4 = the older A/B forms (variants 1..4)
#define P2MT_SUB4(LVV, PRR)                                                                                                          \
  hipLaunchKernelGGL((k_mmr_subtree<LVV, 256, PRR, 4>), dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, st, d_leaves, leaf_base, \
                     m->elements, block0, n_blocks, p2mt::perm_ctx())
      if (sub_lv == 3) P2MT_SUB4(3, 5);
      else if (sub_lv == 2) P2MT_SUB4(2, 5);
      else if (sub_lv == 5) P2MT_SUB4(5, 5);
      else if (sub_lv != 4) return p2mt::fail(P2MT_EINVAL, "stage 1: no kernel for this subtree size");
      else if (variant == 5) P2MT_SUB4(4, 0);      // VALU MDS in the full rounds too (A/B: p2mt_set_variant(2, 5))
      else if (variant == 6) P2MT_SUB4(4, 6);      // ... and the previous field multiply (A/B: p2mt_set_variant(2, 6))
      else if (variant == 7) P2MT_SUB4(4, 7);      // partial rounds in groups of three, round 3's default (A/B: p2mt_set_variant(2, 7))
      else if (variant == 8) P2MT_SUB4(4, 8);      // the default with flag-form folds in the MDS layers: 1.6 % fewer instructions, redo tails (p2mt_set_variant(2, 8))
#ifndef P2MT_DEV_FEWER_VARIANTS
      else if (variant == 2) P2MT_SUB(4, 256, 2);  // MDS layers as 4x4x4 MFMAs (A/B: p2mt_set_variant(2, 2))
      else if (variant == 3) P2MT_SUB(4, 256, 3);  // ... of the 22 partial rounds only (p2mt_set_variant(2, 3))
      else if (variant == 1) P2MT_SUB(4, 256, 1);  // sparse partial rounds (A/B: p2mt_set_variant(2, 1))
      else if (variant == 4) P2MT_SUB(4, 256, 4);  // one MDS layer per partial round, round 2's form (A/B: p2mt_set_variant(2, 4))
      else if (sb == 64) P2MT_SUB(4, 64, 0);
      else if (sb == 128) P2MT_SUB(4, 128, 0);
      else if (!subtree_occ4()) P2MT_SUB(4, 256, 5);
#endif
      else P2MT_SUB4(4, 5);
#undef P2MT_SUB4
#undef P2MT_SUB
      P2MT_LAUNCH_CHECK();
      p2mt::prof_end(prof_slot);
    } else {
      const unsigned grid = (unsigned)((b - a) >> span_log);
      const size_t t0 = a >> span_log;
      const p2mt::PermCtx ctx = p2mt::perm_ctx();
      const int prof_slot = h0 == 0 ? p2mt::prof_begin() : -1;  // stage 1 is the dominant launch
#define P2MT_TILE(M, PR, TL) \
  hipLaunchKernelGGL((k_mmr_tile<M, PR, TL>), dim3(grid), dim3(kBlock), 0, st, d_leaves, leaf_base, m->elements, h0, n_lev, t0, ctx)
      if (rt().mds == 2) {
        if (kTileLog == 9) P2MT_TILE(2, 0, 9);
        else if (kTileLog == 10) P2MT_TILE(2, 0, 10);
        else P2MT_TILE(2, 0, 11);
      } else if (rt().mds == 1) {
        if (rt().partial) P2MT_TILE(1, 1, 11); else P2MT_TILE(1, 0, 11);
      } else {
        if (rt().partial) P2MT_TILE(0, 1, 11); else P2MT_TILE(0, 0, 11);
      }
#undef P2MT_TILE
      P2MT_LAUNCH_CHECK();
      p2mt::prof_end(prof_slot);
    }
    for (unsigned h = h0 + 1; h <= h0 + n_lev; ++h) {  // ragged edges of the fused levels
      P2MT_TRY(launch_level(m, h, n0 >> h, a >> h));
      P2MT_TRY(launch_level(m, h, b >> h, n1 >> h));
    }
    h0 += n_lev;
  }
  return P2MT_OK;
}

static int mmr_extend_dev_noflush(p2mt_mmr* m, const uint64_t* d_leaves, size_t k) {
  if (k == 0) return P2MT_OK;
  if (!d_leaves) return p2mt::fail(P2MT_EINVAL, "null pointer");
  const size_t n0 = m->n_leaves, n1 = n0 + k;
  if (n1 < n0 || (n1 >> 40)) return p2mt::fail(P2MT_ERANGE, "MMR too large");
  P2MT_TRY(mmr_grow(m, mmr_len_for(n1)));
  // (A chunked two-stream variant -- bulk tile kernels on one stream, each chunk's latency-bound upper levels on
  // another -- was measured and removed: 8 chunk launches of 2048 workgroups lose more to partial-wave tails on
  // the 1280 resident slots than the hidden ~0.5 ms of upper-level latency gains; 10.5 ms against 8.5 ms.)
  // Round 5: every aligned perfect subtree of the range that is large enough goes to ONE launch (p2mt_plan.hip: stage 1 and all the
  // levels above it as dependency-ordered workgroups of one grid); what lies between such subtrees, and the carry chain above each
  // (the nodes of height > H that end with the subtree's last leaf), are built by the launches below as before.
  size_t plain_lo = n0, pos = n0;
  while (pos < n1) {
    unsigned H = pos ? (unsigned)__builtin_ctzll((unsigned long long)pos) : 63;
    while (H > 0 && (H >= 63 || pos + ((size_t)1 << H) > n1)) --H;
    const size_t span = (size_t)1 << H;
    if (pos + span <= n1 && p2mt::tree_plan_wanted(H)) {
      if (plain_lo < pos) P2MT_TRY(build_levels(m, d_leaves, n0, plain_lo, pos, 63));
      const p2mt::TreeLayout lay{0, m->elements, nullptr, 0};
      size_t state_bytes = 0;
      P2MT_TRY(p2mt::tree_plan_state_bytes(H, &state_bytes));
      if (state_bytes > m->plan_state_cap) {  // owned by the handle: k_mmr_peaks_root reads its error word later
        if (m->plan_state) {
          P2MT_HIP(hipStreamSynchronize(rt().stream));
          (void)hipFree(m->plan_state);
          m->plan_state = nullptr;
          m->plan_state_cap = 0;
        }
        P2MT_HIP(hipMalloc((void**)&m->plan_state, state_bytes));
        m->plan_state_cap = state_bytes;
      }
      P2MT_TRY(p2mt::tree_plan_launch(lay, d_leaves + (pos - n0), pos, H, m->plan_state));
      P2MT_TRY(build_levels(m, d_leaves, n0, pos, pos + span, 63, H));
      plain_lo = pos + span;
    }
    pos += span;
  }
  if (plain_lo < n1) P2MT_TRY(build_levels(m, d_leaves, n0, plain_lo, n1, 63));
  m->n_leaves = n1;
  return P2MT_OK;
}

// host leaves -> staging buffer -> extend; m->n_leaves moves only when everything was enqueued and the copy completed
static int mmr_extend_host(p2mt_mmr* m, const uint64_t* leaves, size_t k) {
  if (k == 0) return P2MT_OK;
  if (!leaves) return p2mt::fail(P2MT_EINVAL, "null pointer");
  DevBuf b;
  P2MT_TRY(b.alloc(k * 8));
  P2MT_HIP(hipMemcpyAsync(b.p, leaves, k * 8, hipMemcpyHostToDevice, rt().stream));
  const size_t before = m->n_leaves;
  int rc = mmr_extend_dev_noflush(m, b.as<u64>(), k);
  if (rc == P2MT_OK && hipStreamSynchronize(rt().stream) != hipSuccess) rc = p2mt::fail(P2MT_EHIP, "hipStreamSynchronize failed in mmr_extend");
  if (rc != P2MT_OK) m->n_leaves = before;  // the staging buffer dies with this call: nothing half-built stays visible
  return rc;
}

extern "C" int p2mt_mmr_extend_dev(p2mt_mmr* m, const uint64_t* d_leaves, size_t k) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  P2MT_TRY(mmr_flush(m));  // queued add_leaf calls come first
  return mmr_extend_dev_noflush(m, d_leaves, k);
  });
}

extern "C" int p2mt_mmr_extend(p2mt_mmr* m, const uint64_t* leaves, size_t k) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  P2MT_TRY(mmr_flush(m));
  return mmr_extend_host(m, leaves, k);
  });
}

// (sizes are known without flushing: the queue only adds leaves)
extern "C" size_t p2mt_mmr_num_leaves(const p2mt_mmr* m) { return m ? m->n_leaves + m->pending.size() : 0; }
extern "C" size_t p2mt_mmr_len(const p2mt_mmr* m) { return m ? mmr_len_for(m->n_leaves + m->pending.size()) : 0; }
extern "C" const uint64_t* p2mt_mmr_elements_dev(const p2mt_mmr* m) {
  const uint64_t* out = nullptr;  // NULL on any failure, an exception inside the flush included (p2mt_last_error has the reason)
  (void)p2mt::abi_guard([&]() -> int {
    if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
    P2MT_TRY(mmr_flush(m));
    out = m->elements;
    return P2MT_OK;
  });
  return out;
}

extern "C" int p2mt_mmr_copy_elements(const p2mt_mmr* m, size_t first, size_t count, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  P2MT_TRY(mmr_flush(m));
  const size_t len = mmr_len_for(m->n_leaves);
  if (first > len || count > len - first) return p2mt::fail(P2MT_EINVAL, "copy_elements: range out of bounds");
  if (count == 0) return P2MT_OK;
  if (!out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  P2MT_HIP(hipMemcpyAsync(out, m->elements + 4 * first, count * 32, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

// ---------------------------------------------------------------- getting `elements` to the host without paying 13x the build
// MMR.elements of a 2^24-leaf MMR is 1.07 GB.  Into a pageable buffer it moves at ~17 GB/s (63.7 ms, 13 builds); into pinned
// memory it moves at the link's rate, and because `elements` is append-only in post-order -- an extend of leaves [n0, n1) creates
// exactly elements [len(n0), len(n1)) -- the copy of one chunk can run on the copy engines while the next chunk hashes.
extern "C" int p2mt_host_alloc_pinned(size_t bytes, void** out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  *out = nullptr;
  if (hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    *out = nullptr;
    return p2mt::fail(P2MT_ENOMEM, "hipHostMalloc failed");
  }
  return P2MT_OK;
  });
}
extern "C" int p2mt_host_free_pinned(void* p) {
  return p2mt::abi_guard([&]() -> int {
  if (p && hipHostFree(p) != hipSuccess) return p2mt::fail(P2MT_EINVAL, "hipHostFree: not a pinned allocation of this library");
  return P2MT_OK;
  });
}

// enqueue only: the bytes are in `out` (pinned, or the copy degrades to a staged one) after p2mt_sync()
extern "C" int p2mt_mmr_copy_elements_async(const p2mt_mmr* m, size_t first, size_t count, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  P2MT_TRY(mmr_flush(m));
  const size_t len = mmr_len_for(m->n_leaves);
  if (first > len || count > len - first) return p2mt::fail(P2MT_EINVAL, "copy_elements: range out of bounds");
  if (count == 0) return P2MT_OK;
  if (!out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  P2MT_HIP(hipMemcpyAsync(out, m->elements + 4 * first, count * 32, hipMemcpyDeviceToHost, rt().stream));
  return P2MT_OK;
  });
}

namespace {
struct CopyLane {  // per host thread: the stream the chunk copies ride on, and the events that tie it to the library stream
  hipStream_t s = nullptr;
  std::vector<hipEvent_t> ev;
  ~CopyLane() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    if (s) (void)hipStreamDestroy(s);
  }
};
thread_local CopyLane tl_copy;
}  // namespace

// k x add_leaf from device-resident leaves, with the elements this extend appends -- [len before, len after) -- streamed to
// out[0 .. 4 * (len after - len before)) while later chunks are still being hashed: chunks of 2^chunk_log leaves on the library
// stream, each chunk's new elements copied by the copy engines on a second stream behind an event.  Enqueue only; after p2mt_sync()
// the host buffer is complete (the library stream waits for the last copy).  `out` should be pinned (p2mt_host_alloc_pinned).
extern "C" int p2mt_mmr_extend_dev_to_host(p2mt_mmr* m, const uint64_t* d_leaves, size_t k, unsigned chunk_log, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  if (k && (!d_leaves || !out)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (chunk_log < 10 || chunk_log > 40) return p2mt::fail(P2MT_EINVAL, "extend_dev_to_host: chunk_log out of range (10..40)");
  P2MT_TRY(mmr_flush(m));
  if (k == 0) return P2MT_OK;
  const size_t n0 = m->n_leaves, n1 = n0 + k;
  if (n1 < n0 || (n1 >> 40)) return p2mt::fail(P2MT_ERANGE, "MMR too large");
  P2MT_TRY(mmr_grow(m, mmr_len_for(n1)));  // once, up front: no chunk may move the array under a copy in flight
  CopyLane& cl = tl_copy;
  if (!cl.s) P2MT_HIP(hipStreamCreateWithFlags(&cl.s, hipStreamNonBlocking));
  const size_t chunk = (size_t)1 << chunk_log, n_chunks = (k + chunk - 1) / chunk;
  while (cl.ev.size() < n_chunks + 1) {
    hipEvent_t e;
    P2MT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    cl.ev.push_back(e);
  }
  hipStream_t st = rt().stream;
  const size_t len0 = mmr_len_for(n0);
  size_t done = 0;
  for (size_t ci = 0; ci < n_chunks; ++ci) {
    const size_t kk = std::min(chunk, k - done);
    const size_t before = mmr_len_for(m->n_leaves);
    P2MT_TRY(mmr_extend_dev_noflush(m, d_leaves + done, kk));
    const size_t after = mmr_len_for(m->n_leaves);
    P2MT_HIP(hipEventRecord(cl.ev[ci], st));
    P2MT_HIP(hipStreamWaitEvent(cl.s, cl.ev[ci], 0));
    P2MT_HIP(hipMemcpyAsync(out + 4 * (before - len0), m->elements + 4 * before, (after - before) * 32, hipMemcpyDeviceToHost, cl.s));
    done += kk;
  }
  P2MT_HIP(hipEventRecord(cl.ev[n_chunks], cl.s));
  P2MT_HIP(hipStreamWaitEvent(st, cl.ev[n_chunks], 0));
  return P2MT_OK;
  });
}

// ---------------------------------------------------------------- checkpoint
namespace {
struct CkptHeader {
  char magic[8];
  uint64_t n_leaves, n_elements, checksum;
};
uint64_t fold_checksum(const uint64_t* p, size_t n_words) {
  uint64_t acc = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < n_words; ++i) acc = (acc ^ p[i]) * 0x100000001B3ull + (acc >> 29);
  return acc;
}
}  // namespace

extern "C" int p2mt_mmr_save(const p2mt_mmr* m, const char* path) {
  return p2mt::abi_guard([&]() -> int {
  if (!m || !path) return p2mt::fail(P2MT_EINVAL, "null argument");
  P2MT_TRY(mmr_flush(m));
  const size_t len = mmr_len_for(m->n_leaves);
  std::vector<uint64_t> host;
  try {
    host.resize(4 * len);
  } catch (const std::bad_alloc&) {
    return p2mt::fail(P2MT_ENOMEM, "mmr_save: out of host memory");
  }
  if (len) P2MT_TRY(p2mt_mmr_copy_elements(m, 0, len, host.data()));
  CkptHeader h;
  memcpy(h.magic, "P2MTMMR1", 8);
  h.n_leaves = m->n_leaves;
  h.n_elements = len;
  h.checksum = fold_checksum(host.data(), host.size());
  FILE* f = fopen(path, "wb");
  if (!f) return p2mt::fail(P2MT_EINVAL, "mmr_save: cannot open file for writing");
  const bool ok = fwrite(&h, sizeof h, 1, f) == 1 && (host.empty() || fwrite(host.data(), 8, host.size(), f) == host.size());
  if (fclose(f) != 0 || !ok) return p2mt::fail(P2MT_EINVAL, "mmr_save: short write");
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_load(p2mt_mmr* m, const char* path) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!m || !path) return p2mt::fail(P2MT_EINVAL, "null argument");
  FILE* f = fopen(path, "rb");
  if (!f) return p2mt::fail(P2MT_EINVAL, "mmr_load: cannot open file");
  CkptHeader h;
  if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "P2MTMMR1", 8) != 0) {
    fclose(f);
    return p2mt::fail(P2MT_EINVAL, "mmr_load: not a P2MTMMR1 checkpoint");
  }
  if (h.n_elements != mmr_len_for(h.n_leaves) || (h.n_leaves >> 40)) {
    fclose(f);
    return p2mt::fail(P2MT_EINVAL, "mmr_load: header inconsistent (n_elements != 2N - popcount(N))");
  }
  {  // the payload must be exactly 32 * n_elements bytes: checked against the file BEFORE anything of that size is allocated
    const long at = ftell(f);
    if (at < 0 || fseek(f, 0, SEEK_END) != 0) {
      fclose(f);
      return p2mt::fail(P2MT_EINVAL, "mmr_load: cannot size the file");
    }
    const long end = ftell(f);
    if (end < at || (uint64_t)(end - at) != 32 * h.n_elements || fseek(f, at, SEEK_SET) != 0) {
      fclose(f);
      return p2mt::fail(P2MT_EINVAL, "mmr_load: truncated or oversized payload");
    }
  }
  std::vector<uint64_t> host;
  try {
    host.resize(4 * h.n_elements);
  } catch (const std::bad_alloc&) {
    fclose(f);
    return p2mt::fail(P2MT_ENOMEM, "mmr_load: out of host memory");
  }
  const bool ok = host.empty() || fread(host.data(), 8, host.size(), f) == host.size();
  fclose(f);
  if (!ok) return p2mt::fail(P2MT_EINVAL, "mmr_load: truncated or oversized payload");
  if (fold_checksum(host.data(), host.size()) != h.checksum) return p2mt::fail(P2MT_EINVAL, "mmr_load: checksum mismatch");
  for (size_t i = 0; i < host.size(); ++i)
    if (host[i] >= gl::P) return p2mt::fail(P2MT_EINVAL, "mmr_load: non-canonical field element");
  // the handle changes only once the payload is validated and on the device: upload to a fresh buffer, then swap
  DevBuf fresh;
  const size_t cap = h.n_elements > 1024 ? h.n_elements : 1024;
  if (fresh.alloc(cap * 32) != P2MT_OK) return p2mt::fail(P2MT_ENOMEM, "mmr_load: hipMalloc failed");
  if (h.n_elements) {
    P2MT_HIP(hipMemcpyAsync(fresh.p, host.data(), host.size() * 8, hipMemcpyHostToDevice, rt().stream));
  }
  P2MT_HIP(hipStreamSynchronize(rt().stream));  // also: nothing enqueued earlier still reads the old array
  if (m->elements) (void)hipFree(m->elements);
  m->elements = fresh.as<u64>();
  fresh.p = nullptr;
  m->cap_nodes = cap;
  m->pending.clear();
  m->n_leaves = h.n_leaves;
  return P2MT_OK;
  });
}

// peaks left to right = one per set bit of N, decreasing height; the peak of height b ends at the running leaf prefix
static int mmr_peak_positions(const p2mt_mmr* m, PosList* pl) {
  const size_t len = mmr_len_for(m->n_leaves);
  if (len == 0) return p2mt::fail(P2MT_EINVAL, "get_peaks on an empty MMR (the reference overflows a shift, Q6)");
  if (len > 0xFFFFFFFFull) return p2mt::fail(P2MT_ERANGE, "get_peaks: mmr_len.to_u32().unwrap() (Q6)");
  pl->n = 0;
  size_t prefix = 0;
  for (int b = 63; b >= 0; --b) {
    if ((m->n_leaves >> b) & 1) {
      prefix += (size_t)1 << b;
      pl->pos[pl->n++] = p2mt_mmr_node_pos(prefix - 1, (unsigned)b);
    }
  }
  return P2MT_OK;
}

static int mmr_peaks_root(const p2mt_mmr* m, uint64_t* peaks_out, int* n_peaks, uint64_t* root_out) {
  P2MT_TRY(p2mt::ensure_init());
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  P2MT_TRY(mmr_flush(m));
  PosList pl;
  P2MT_TRY(mmr_peak_positions(m, &pl));
  u64* d_peaks = m->scratch;
  u64* d_root = m->scratch + 4 * 64;
  P2MT_DISPATCH(k_mmr_peaks_root, 1, 64, (const u64*)m->elements, pl, d_peaks, d_root, (const u32*)m->plan_state);
  if (peaks_out) P2MT_HIP(hipMemcpyAsync(peaks_out, d_peaks, (size_t)pl.n * 32, hipMemcpyDeviceToHost, rt().stream));
  if (root_out) P2MT_HIP(hipMemcpyAsync(root_out, d_root, 32, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  if (n_peaks) *n_peaks = pl.n;
  if (m->plan_state) {  // (an all-ones word is not a field element: the kernel's mark for "a hand-off of the one-launch build gave up")
    uint32_t err = 0;
    if (root_out ? root_out[0] == ~0ull : false) err = 1;
    if (!root_out) P2MT_HIP(hipMemcpy(&err, m->plan_state + 1, 4, hipMemcpyDeviceToHost));
    if (err) return p2mt::fail(P2MT_EHIP, "MMR build: an in-launch hand-off timed out (tree plan); the node array is incomplete");
  }
  return P2MT_OK;
}

extern "C" int p2mt_mmr_root_dev(const p2mt_mmr* m, uint64_t* d_root_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!m || !d_root_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  P2MT_TRY(mmr_flush(m));
  PosList pl;
  P2MT_TRY(mmr_peak_positions(m, &pl));
  P2MT_DISPATCH(k_mmr_peaks_root, 1, 64, (const u64*)m->elements, pl, m->scratch, d_root_out, (const u32*)m->plan_state);
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_peaks(const p2mt_mmr* m, uint64_t* peaks_out, int* n_peaks) {
  return p2mt::abi_guard([&]() -> int {
  if (!peaks_out || !n_peaks) return p2mt::fail(P2MT_EINVAL, "null pointer");
  return mmr_peaks_root(m, peaks_out, n_peaks, nullptr);
  });
}

extern "C" int p2mt_mmr_root(const p2mt_mmr* m, uint64_t* root_out) {
  return p2mt::abi_guard([&]() -> int {
  if (!root_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  return mmr_peaks_root(m, nullptr, nullptr, root_out);
  });
}

extern "C" int p2mt_mmr_proof_batch_dev(const p2mt_mmr* m, const uint64_t* d_mmr_indices, size_t count, size_t max_siblings,
                                        uint64_t* d_siblings_out, uint8_t* d_lefts_out, int32_t* d_n_siblings_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  P2MT_TRY(mmr_flush(m));
  if (count == 0) return P2MT_OK;
  if (!d_mmr_indices || !d_siblings_out || !d_lefts_out || !d_n_siblings_out || max_siblings == 0)
    return p2mt::fail(P2MT_EINVAL, "null pointer");
  const size_t len = mmr_len_for(m->n_leaves);
  if (len == 0) return p2mt::fail(P2MT_EINVAL, "get_proof on an empty MMR");
  hipLaunchKernelGGL(k_mmr_proof_batch, dim3(grid_for(count)), dim3(kBlock), 0, rt().stream, (const u64*)m->elements, len,
                     d_mmr_indices, count, max_siblings, d_siblings_out, d_lefts_out, d_n_siblings_out);
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_proof_batch(const p2mt_mmr* m, const uint64_t* mmr_indices, size_t count, size_t max_siblings,
                                    uint64_t* siblings_out, uint8_t* lefts_out, int32_t* n_siblings_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!m) return p2mt::fail(P2MT_EINVAL, "null handle");
  if (count == 0) return P2MT_OK;
  if (!mmr_indices || !siblings_out || !lefts_out || !n_siblings_out || max_siblings == 0)
    return p2mt::fail(P2MT_EINVAL, "null pointer");
  // staging comes from the thread's grow-only scratch (one carve-up): a hipMalloc / hipFree pair per buffer cost more than the
  // whole call for the reference's one-proof-at-a-time use (config 2: four proofs took 5 ms)
  const size_t o_idx = 0, o_sib = (count * 8 + 255) & ~(size_t)255, o_left = o_sib + ((count * max_siblings * 32 + 255) & ~(size_t)255);
  const size_t o_n = o_left + ((count * max_siblings + 255) & ~(size_t)255), total = o_n + count * 4;
  char* base = nullptr;
  P2MT_TRY(p2mt::scratch_get(p2mt::kScratchPing, total, (void**)&base));
  u64* d_idx = reinterpret_cast<u64*>(base + o_idx);
  u64* d_sib = reinterpret_cast<u64*>(base + o_sib);
  uint8_t* d_left = reinterpret_cast<uint8_t*>(base + o_left);
  int32_t* d_n = reinterpret_cast<int32_t*>(base + o_n);
  hipStream_t st = rt().stream;
  P2MT_HIP(hipMemcpyAsync(d_idx, mmr_indices, count * 8, hipMemcpyHostToDevice, st));
  P2MT_HIP(hipMemsetAsync(base + o_sib, 0, o_n - o_sib, st));
  P2MT_TRY(p2mt_mmr_proof_batch_dev(m, d_idx, count, max_siblings, d_sib, d_left, d_n));
  P2MT_HIP(hipMemcpyAsync(siblings_out, d_sib, count * max_siblings * 32, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipMemcpyAsync(lefts_out, d_left, count * max_siblings, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipMemcpyAsync(n_siblings_out, d_n, count * 4, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  for (size_t i = 0; i < count; ++i) {
    if (n_siblings_out[i] < 0) return p2mt::fail(P2MT_EINVAL, "get_proof: mmr_index out of bounds");
    if ((size_t)n_siblings_out[i] > max_siblings) return p2mt::fail(P2MT_EINVAL, "get_proof: max_siblings too small");
  }
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_proof(const p2mt_mmr* m, size_t mmr_index, uint64_t* siblings_out, uint8_t* lefts_out,
                              int* n_siblings, uint64_t* peaks_out, int* n_peaks, size_t* mmr_size) {
  return p2mt::abi_guard([&]() -> int {
  if (!siblings_out || !lefts_out || !n_siblings || !peaks_out || !n_peaks) return p2mt::fail(P2MT_EINVAL, "null pointer");
  const uint64_t idx = mmr_index;
  int32_t ns = 0;
  P2MT_TRY(p2mt_mmr_proof_batch(m, &idx, 1, P2MT_MAX_PROOF_LEN, siblings_out, lefts_out, &ns));
  P2MT_TRY(p2mt_mmr_peaks(m, peaks_out, n_peaks));
  *n_siblings = ns;
  if (mmr_size) *mmr_size = p2mt_mmr_len(m);
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_proof_verify_batch_dev(const uint64_t* d_siblings, const uint8_t* d_lefts, const int32_t* d_n_siblings,
                                               size_t max_siblings, const uint64_t* d_peaks, int n_peaks,
                                               const uint64_t* d_leaves, const uint64_t* d_root, size_t m, int8_t* d_status_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (m == 0) return P2MT_OK;
  if (!d_n_siblings || !d_peaks || !d_leaves || !d_root || !d_status_out || n_peaks < 1 || n_peaks > 64)
    return p2mt::fail(P2MT_EINVAL, "bad argument");
  if (max_siblings && (!d_siblings || !d_lefts)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  u64* d_bagged;
  P2MT_TRY(p2mt::scratch_get(p2mt::kScratchTmp, 32, (void**)&d_bagged));
  P2MT_DISPATCH(k_bag_peaks, 1, 64, d_peaks, n_peaks, d_bagged);
  if (m <= ((size_t)1 << 12) && rt().mds == 2) {  // few proofs: the latency layout
    hipLaunchKernelGGL(k_mmr_verify_wave, dim3((unsigned)((m + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, rt().stream, d_siblings,
                       d_lefts, d_n_siblings, max_siblings ? max_siblings : 1, d_peaks, n_peaks, d_leaves, d_root,
                       (const u64*)d_bagged, m, d_status_out, p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
    return P2MT_OK;
  }
  P2MT_DISPATCH(k_mmr_verify_batch, grid_for(m), kBlock, d_siblings, d_lefts, d_n_siblings, max_siblings ? max_siblings : 1,
                d_peaks, n_peaks, d_leaves, d_root, (const u64*)d_bagged, m, d_status_out);
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_proof_verify_batch(const uint64_t* siblings, const uint8_t* lefts, const int32_t* n_siblings,
                                           size_t max_siblings, const uint64_t* peaks, int n_peaks, const uint64_t* leaves,
                                           const uint64_t* root, size_t m, int8_t* status_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (m == 0) return P2MT_OK;
  if (!n_siblings || !peaks || !leaves || !root || !status_out || n_peaks < 0 || n_peaks > 64)
    return p2mt::fail(P2MT_EINVAL, "bad argument");
  if (n_peaks == 0) {  // contains() on an empty peak list is false: the reference's assert fires for every proof
    for (size_t i = 0; i < m; ++i) status_out[i] = (int8_t)P2MT_ENOTPEAK;
    return P2MT_OK;
  }
  if (max_siblings && (!siblings || !lefts)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  for (size_t i = 0; i < m; ++i)
    if (n_siblings[i] < 0 || (size_t)n_siblings[i] > max_siblings) return p2mt::fail(P2MT_EINVAL, "n_siblings out of range");
  const size_t ms = max_siblings ? max_siblings : 1;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_sib = 0, o_left = up(m * ms * 32), o_n = o_left + up(m * ms), o_peaks = o_n + up(m * 4);
  const size_t o_leaf = o_peaks + up((size_t)n_peaks * 32), o_root = o_leaf + up(m * 8), o_out = o_root + 256, total = o_out + up(m);
  char* base = nullptr;  // the thread's grow-only scratch instead of seven hipMalloc / hipFree pairs per call
  P2MT_TRY(p2mt::scratch_get(p2mt::kScratchPing, total, (void**)&base));
  hipStream_t st = rt().stream;
  if (max_siblings) {
    P2MT_HIP(hipMemcpyAsync(base + o_sib, siblings, m * ms * 32, hipMemcpyHostToDevice, st));
    P2MT_HIP(hipMemcpyAsync(base + o_left, lefts, m * ms, hipMemcpyHostToDevice, st));
  }
  P2MT_HIP(hipMemcpyAsync(base + o_n, n_siblings, m * 4, hipMemcpyHostToDevice, st));
  P2MT_HIP(hipMemcpyAsync(base + o_peaks, peaks, (size_t)n_peaks * 32, hipMemcpyHostToDevice, st));
  P2MT_HIP(hipMemcpyAsync(base + o_leaf, leaves, m * 8, hipMemcpyHostToDevice, st));
  P2MT_HIP(hipMemcpyAsync(base + o_root, root, 32, hipMemcpyHostToDevice, st));
  P2MT_TRY(p2mt_mmr_proof_verify_batch_dev(reinterpret_cast<u64*>(base + o_sib), reinterpret_cast<uint8_t*>(base + o_left),
                                           reinterpret_cast<int32_t*>(base + o_n), max_siblings, reinterpret_cast<u64*>(base + o_peaks),
                                           n_peaks, reinterpret_cast<u64*>(base + o_leaf), reinterpret_cast<u64*>(base + o_root), m,
                                           reinterpret_cast<int8_t*>(base + o_out)));
  P2MT_HIP(hipMemcpyAsync(status_out, base + o_out, m, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_proof_verify(const uint64_t* siblings, const uint8_t* lefts, int n_siblings, const uint64_t* peaks,
                                     int n_peaks, uint64_t leaf, const uint64_t* root, int* result_out) {
  return p2mt::abi_guard([&]() -> int {
  if (!result_out || n_siblings < 0 || n_siblings > P2MT_MAX_PROOF_LEN) return p2mt::fail(P2MT_EINVAL, "bad argument");
  int8_t status = 0;
  const int32_t ns = n_siblings;
  P2MT_TRY(p2mt_mmr_proof_verify_batch(siblings, lefts, &ns, (size_t)n_siblings, peaks, n_peaks, &leaf, root, 1, &status));
  if (status == (int8_t)P2MT_ENOTPEAK) return p2mt::fail(P2MT_ENOTPEAK, "MMR_proof::verify: assert!(self.peaks.contains(&next_hash))");
  *result_out = status;
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_combine_shard_roots_dev(const uint64_t* d_shard_roots, size_t world, uint64_t* d_top_nodes_out,
                                                uint64_t* d_root_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!d_shard_roots || !d_root_out || world == 0 || (world & (world - 1)) || world > kMaxWorld)
    return p2mt::fail(P2MT_EINVAL, "world must be a power of two <= 1024");
  // (the reference hashes nothing at world == 1: the kernel then only canonicalises and copies the root)
  hipLaunchKernelGGL(k_combine_roots, dim3(1), dim3(1024), 0, rt().stream, d_shard_roots, (unsigned)world, d_top_nodes_out,
                     d_root_out, p2mt::perm_ctx());
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_combine_shard_roots(const uint64_t* shard_roots, size_t world, uint64_t* top_nodes_out,
                                            uint64_t* root_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!shard_roots || !root_out || world == 0 || (world & (world - 1)) || world > kMaxWorld)
    return p2mt::fail(P2MT_EINVAL, "world must be a power of two <= 1024");
  if (world == 1) {
    memcpy(root_out, shard_roots, 32);
    return P2MT_OK;
  }
  // level-major pairing of the shard roots: world/2 + world/4 + ... + 1 = world-1 nodes
  DevBuf b;  // [world roots | world-1 top nodes | root]
  P2MT_TRY(b.alloc(2 * world * 32));
  hipStream_t st = rt().stream;
  u64* d = b.as<u64>();
  P2MT_HIP(hipMemcpyAsync(d, shard_roots, world * 32, hipMemcpyHostToDevice, st));
  P2MT_TRY(p2mt_mmr_combine_shard_roots_dev(d, world, d + 4 * world, d + 4 * (2 * world - 1)));
  if (top_nodes_out) P2MT_HIP(hipMemcpyAsync(top_nodes_out, d + 4 * world, (world - 1) * 32, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipMemcpyAsync(root_out, d + 4 * (2 * world - 1), 32, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  return P2MT_OK;
  });
}
