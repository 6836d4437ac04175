// p2mt_plan.hip -- a whole perfect subtree in ONE launch: the hashing loops of
//   /root/reference/src/mmr/merkle_mountain_ranges.rs:89-120        (MMR::add_leaf's carry chain, all levels)
//   /root/reference/src/simple_merkle_tree/simple_merkle_tree.rs:28-51 (MerkleTree::build, all levels)
// as dependency-ordered workgroups of one grid.
//
// The launch executes a PLAN: a list of work items in ticket order.  A workgroup that becomes resident takes the next ticket (one
// agent-scope atomic add -- HIP promises nothing about dispatch order, so the ticket, not blockIdx, decides what a workgroup does), waits
// until the nodes its item reads are published, hashes, publishes.  Every item's inputs come from items with SMALLER tickets (checked on
// the host when the plan is built), and a workgroup that holds a ticket is resident and runs to completion, so the item with the smallest
// unfinished ticket can always proceed: no deadlock under any dispatch order or placement.  Item kinds, by how many lanes share a hash
// (the three layouts of tree_common.hip.h):
//   S  stage 1: each lane builds the 2^lv-leaf subtree of its own leaves depth-first (k_mmr_subtree's loop) -- 256 subtree roots
//   U  one two_to_one per lane, 256 nodes of one level (the same inlined permutation as S's merge step: children from HBM instead of LDS)
//   Q  four lanes per node, 64 nodes   (latency: 0.35x)
//   W  one wavefront per node, 4 nodes (latency: ~7 us per permutation)
// Hand-off between workgroups (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", the form with
// sc1 loads in place of the acquire): every handed-off node is stored write-through (agent-scope relaxed atomic stores = `sc1`), each
// storing wave drains its stores (`s_waitcnt vmcnt(0)`) and then adds the number of nodes it stored to the counter of their chunk (one
// agent-scope atomic add per wave); a consumer's first wave polls the chunk counters of its children with `sc1` loads, the workgroup
// passes a barrier, and every load of a handed-off node is an `sc1` load.  Counters are zeroed by a memset node in front of every launch.
// Spins are bounded: a poll that gives up sets the launch's error word, every later poll sees it and gives up at once, the grid drains,
// and the host reports P2MT_EHIP when it next reads the tree.
#include "tree_common.hip.h"

#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

using namespace p2mt_dev;

namespace {

using p2mt::TreeLayout;

enum : uint32_t { KS = 0, KU = 1, KQ = 2, KW = 3 };

struct PlanItem {
  uint32_t w;   // kind | h << 4 | n << 12 : kind of work, level of the nodes it produces, how many (<= 256)
  uint32_t j0;  // first node it produces, as an index into its level of THIS subtree (S: first subtree root)
};

constexpr unsigned kMaxPlanLevels = 34;
struct PlanArgs {
  const PlanItem* items;
  uint32_t n_items, H, spin_limit, prof;
  uint32_t cnt_off[kMaxPlanLevels];  // first counter of level h
  uint8_t csh[kMaxPlanLevels + 6];   // log2(nodes per counter) of level h (6 for S / U / Q levels, 2 for W levels)
};

typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) u32 gu32;
GL_DEV u64 ld_sc1(const u64* p) { return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
GL_DEV void st_sc1(u64* p, u64 v) { __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
GL_DEV u32 ld_sc1(const u32* p) { return __hip_atomic_load((const gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
GL_DEV void add_agent(u32* p, u32 v) { (void)__hip_atomic_fetch_add((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
GL_DEV void load_hash_sc1(const u64* p, u64 (&h)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) h[k] = ld_sc1(p + k);
}
GL_DEV void store_hash_sc1(u64* p, const u64 (&h)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) st_sc1(p + k, h[k]);
}
// every storing wave, after its last handed-off store and before it signals
GL_DEV void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// slot of node j (index into level h of the whole structure)
GL_DEV u64* node_ptr(const TreeLayout& lay, unsigned h, size_t j) {
  if (lay.kind == 0) return lay.base + 4 * node_pos(((j + 1) << h) - 1, h);  // MMR.elements, post-order
  // MerkleTree.tree, level-major: level h of an n-leaf tree starts 2n - (2n >> h) digests in; the root has a home of its own
  if (((size_t)1 << h) == lay.n) return lay.root;
  return lay.base + 4 * (2 * lay.n - ((2 * lay.n) >> h) + j);
}

// state words: [0] ticket, [1] error (0 = none, else 1 + the ticket whose poll gave up), [16 ..) chunk counters
constexpr unsigned kStateHdr = 16;

// Wait until nodes [jc0, jc0 + nc) of level hc (indices into the level of this subtree) are published.  All 256 threads call it.
GL_DEV void wait_children(const PlanArgs& pa, u32* __restrict__ state, unsigned hc, u32 jc0, u32 nc, u32 ticket) {
  if (threadIdx.x < 64) {
    const unsigned lane = threadIdx.x;
    const unsigned sh = pa.csh[hc];
    const u32 n_level = 1u << (pa.H - hc);
    const u32 c0 = jc0 >> sh, c1 = (jc0 + nc - 1) >> sh;  // at most 8 counters
    const u32 c = c0 + lane;
    const bool mine = c <= c1 && lane < 62;
    const u32 per = 1u << sh;
    u32 want = 0;
    const u32* p = state + 1;  // lane 63 watches the error word (want 0)
    if (mine) {
      want = (n_level - (c << sh)) < per ? n_level - (c << sh) : per;
      p = state + kStateHdr + pa.cnt_off[hc] + c;
    }
    const bool polls = mine || lane == 63;
    unsigned spins = 0;
    for (;;) {
      const u32 v = polls ? ld_sc1(p) : want;
      const unsigned long long ne = __ballot(v != want);
      if (ne == 0) break;
      if ((ne >> 63) != 0) break;                 // another workgroup gave up: give up at once, the grid drains
      if (++spins > pa.spin_limit) {              // (wave-uniform)
        if (lane == 0) __hip_atomic_store((gu32*)(state + 1), 1u + ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();  // between the poll and EVERY load of the published nodes
}

// The two latency layouts as functions of their own: inlined into the kernel they shared its register allocation and spilled (39
// scratch accesses on the chain of a quad permutation -- every reload an L2 round trip that a latency-bound wave cannot hide).
GL_DEV u64 quad_item(const TreeLayout& lay, size_t first_leaf, const PlanArgs& pa, u32* __restrict__ state,
                                                   const PermCtx& ctx, u32 h, u32 n, u32 j0, u32 t) {
  u32* const cnt = state + kStateHdr;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wait_children(pa, state, h - 1, 2 * j0, 2 * n, t);
  const u64 t_ready = pa.prof ? wall_clock64() : 0;
  const u32 i = threadIdx.x >> 2;
  if (i < n) {  // quad-uniform
    poseidon_quad::Lane ln;
    poseidon_quad::lane_init(ln, ctx.rc);
    const size_t j = (first_leaf >> h) + j0 + i;
    const u64* lp = node_ptr(lay, h - 1, 2 * j);
    const u64* rp = node_ptr(lay, h - 1, 2 * j + 1);
    u64 x[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const u32 w = 3 * ln.q + k;
      x[k] = w < 4 ? ld_sc1(lp + w) : (w < 8 ? ld_sc1(rp + (w - 4)) : 0);
    }
    poseidon_quad::permute(x, ln);
    u64* out = node_ptr(lay, h, j);
    if (ln.q == 0) {
      st_sc1(out + 0, gl::canon(x[0]));
      st_sc1(out + 1, gl::canon(x[1]));
      st_sc1(out + 2, gl::canon(x[2]));
    } else if (ln.q == 1) {
      st_sc1(out + 3, gl::canon(x[0]));
    }
  }
  drain_stores();
  if (lane == 0 && 16 * wave < n) add_agent(cnt + pa.cnt_off[h] + ((j0 + 16 * wave) >> 6), n - 16 * wave < 16 ? n - 16 * wave : 16);
  return t_ready;
}

GL_DEV u64 wave_item(const TreeLayout& lay, size_t first_leaf, const PlanArgs& pa, u32* __restrict__ state,
                                                   const PermCtx& ctx, u32 h, u32 n, u32 j0, u32 t, u64* smem) {
  u32* const cnt = state + kStateHdr;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const PermCtx c2 = stage_round_constants(smem, ctx);  // (before the wait: it overlaps the producers)
  wait_children(pa, state, h - 1, 2 * j0, 2 * n, t);
  const u64 t_ready = pa.prof ? wall_clock64() : 0;
  if (wave < n) {  // wave-uniform
    const size_t j = (first_leaf >> h) + j0 + wave;
    const u64* lp = node_ptr(lay, h - 1, 2 * j);
    const u64* rp = node_ptr(lay, h - 1, 2 * j + 1);
    u64 x = lane < 4 ? ld_sc1(lp + lane) : (lane < 8 ? ld_sc1(rp + (lane - 4)) : 0);
    x = permute_wave(x, c2);
    if (lane < 4) st_sc1(node_ptr(lay, h, j) + lane, gl::canon(x));
    drain_stores();
    if (lane == 0) add_agent(cnt + pa.cnt_off[h] + ((j0 + wave) >> 2), 1);
  }
  return t_ready;
}

// One launch = one plan.  grid = n_items workgroups of 256 lanes; four waves per SIMD like k_mmr_subtree.
__global__ __launch_bounds__(256, 4) void k_tree_plan(const u64* __restrict__ leaves, TreeLayout lay, size_t first_leaf, PlanArgs pa,
                                                      u32* __restrict__ state, u64* __restrict__ prof, PermCtx ctx) {
  __shared__ __attribute__((aligned(16))) u64 smem[3 * 256 * 4];  // S: per-lane stack of pending left siblings; W: round constants
  __shared__ u32 s_ticket;
  poseidon_fast::MfmaCtx mc;  // made while every lane is active (an MFMA ignores EXEC)
  poseidon_fast::mfma32_ctx_init(mc);
  if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add((gu32*)state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const u32 t = (u32)__builtin_amdgcn_readfirstlane((int)s_ticket);
  if (t >= pa.n_items) return;  // (never: the grid has exactly n_items workgroups)
  const u64 t_start = pa.prof ? wall_clock64() : 0;
  const PlanItem it = pa.items[t];
  const u32 kind = it.w & 15u, h = (it.w >> 4) & 63u, n = it.w >> 12, j0 = it.j0;
  u32* const cnt = state + kStateHdr;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u64 t_ready = 0;

  {  // (kinds S and U only: the Q / W items of a plan run in k_tree_top)
    const bool is_s = kind == KS;
    if (!is_s) wait_children(pa, state, h - 1, 2 * j0, 2 * n, t);
    if (pa.prof) t_ready = wall_clock64();
    // no lane leaves before the loop (matrix-pipe MDS): lanes past the end redo the last one and store nothing
    const bool live = threadIdx.x < n;
    const u32 i = live ? threadIdx.x : n - 1;
    u64 (*stack)[256 * 4] = reinterpret_cast<u64 (*)[256 * 4]>(smem);
    u64 cur[4] = {0, 0, 0, 0};
    unsigned pairs_done = 0, hh = 0, merges = 0, n_steps;
    size_t first_leaf_l = 0;
    const u64* lp = leaves;
    u64* u_dst = nullptr;
    if (is_s) {
      const size_t sub = (size_t)j0 + i;  // this lane's subtree
      first_leaf_l = first_leaf + (sub << h);
      lp = leaves + (sub << h);
      n_steps = (1u << h) - 1;
    } else {
      const size_t j = (first_leaf >> h) + j0 + i;
      u64 l[4];
      load_hash_sc1(node_ptr(lay, h - 1, 2 * j), l);
      load_hash_sc1(node_ptr(lay, h - 1, 2 * j + 1), cur);
      store_hash(&stack[0][threadIdx.x * 4], l);
      u_dst = node_ptr(lay, h, j);
      merges = 1;
      hh = h - 1;
      n_steps = 1;
    }
#pragma unroll 1
    for (unsigned step = 0; step < n_steps; ++step) {
      u64 o[4];
      u64* dst;
      if (merges == 0) {  // hash the next leaf pair (S only)
        const u64 a = gl::canon(lp[2 * pairs_done]), b = gl::canon(lp[2 * pairs_done + 1]);
        const size_t leaf = first_leaf_l + 2 * pairs_done;
        if (live) {
          const u64 la[4] = {a, 0, 0, 0}, lb[4] = {b, 0, 0, 0};
          store_hash(node_ptr(lay, 0, leaf), la);  // hash_or_noop([leaf]) = [leaf, 0, 0, 0]
          store_hash(node_ptr(lay, 0, leaf + 1), lb);
        }
        two_to_one_r<IMPL_FAST, 5, true>(ctx, o, [&](u64 (&ll)[4], u64 (&rr)[4]) {
          ll[0] = gl::canon(lp[2 * pairs_done]); ll[1] = ll[2] = ll[3] = 0;
          rr[0] = gl::canon(lp[2 * pairs_done + 1]); rr[1] = rr[2] = rr[3] = 0;
        }, &mc);
        merges = (unsigned)__builtin_ctz(~pairs_done);
        pairs_done += 1;
        hh = 1;
        dst = node_ptr(lay, 1, leaf >> 1);
      } else {  // merge the pending left sibling of height hh with cur
        const u64* sp = &stack[is_s ? hh - 1 : 0][threadIdx.x * 4];
        two_to_one_r<IMPL_FAST, 5>(ctx, o, [&](u64 (&ll)[4], u64 (&rr)[4]) {
          load_hash(sp, ll);
#pragma unroll
          for (int k = 0; k < 4; ++k) rr[k] = cur[k];
        }, &mc);
        merges -= 1;
        hh += 1;
        dst = is_s ? node_ptr(lay, hh, ((first_leaf_l + 2 * pairs_done) >> hh) - 1) : u_dst;
      }
      if (live) {
        if (hh == h) store_hash_sc1(dst, o);  // the nodes another workgroup will read: write-through
        else store_hash(dst, o);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) cur[k] = o[k];
      if (is_s && merges == 0 && hh < h) store_hash(&stack[hh - 1][threadIdx.x * 4], cur);  // becomes a pending left sibling
    }
    drain_stores();
    if (lane == 0 && 64 * wave < n) add_agent(cnt + pa.cnt_off[h] + ((j0 + 64 * wave) >> 6), n - 64 * wave < 64 ? n - 64 * wave : 64);
  }
  if (pa.prof && threadIdx.x == 0) {
    const u32 xcc = (u32)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);   // HW_REG_XCC_ID[3:0]
    const u32 hwid = (u32)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);  // HW_REG_HW_ID
    prof[4 * (size_t)t + 0] = t_start;
    prof[4 * (size_t)t + 1] = t_ready;
    prof[4 * (size_t)t + 2] = wall_clock64();
    prof[4 * (size_t)t + 3] = ((u64)xcc << 32) | hwid;
  }
}

// The latency layouts (Q: four lanes per node, W: one wavefront per node) run in a launch of their own behind k_tree_plan: inlined
// into it they shared its register budget (128 VGPRs for four waves per SIMD) and spilled -- 39 scratch accesses on the chain of a quad
// permutation, every reload an L2 round trip that a latency-bound wave cannot hide (Q items took 55-100 us instead of 13-26).  Same
// plan, same counters: the items of this launch are the plan's tail [first, n_items), tickets come from state word 2, and the counters
// the first launch left behind satisfy the waits on its nodes at once.
__global__ __launch_bounds__(256) void k_tree_top(TreeLayout lay, size_t first_leaf, PlanArgs pa, uint32_t first, u32* __restrict__ state,
                                                  u64* __restrict__ prof, PermCtx ctx) {
  __shared__ u64 rc_lds[kWaveRcWords];
  __shared__ u32 s_ticket;
  if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add((gu32*)(state + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const u32 t = first + (u32)__builtin_amdgcn_readfirstlane((int)s_ticket);
  if (t >= pa.n_items) return;  // (never)
  const u64 t_start = pa.prof ? wall_clock64() : 0;
  const PlanItem it = pa.items[t];
  const u32 kind = it.w & 15u, h = (it.w >> 4) & 63u, n = it.w >> 12, j0 = it.j0;
  const u64 t_ready = kind == KQ ? quad_item(lay, first_leaf, pa, state, ctx, h, n, j0, t)
                                 : wave_item(lay, first_leaf, pa, state, ctx, h, n, j0, t, rc_lds);
  if (pa.prof && threadIdx.x == 0) {
    const u32 xcc = (u32)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);   // HW_REG_XCC_ID[3:0]
    const u32 hwid = (u32)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);  // HW_REG_HW_ID
    prof[4 * (size_t)t + 0] = t_start;
    prof[4 * (size_t)t + 1] = t_ready;
    prof[4 * (size_t)t + 2] = wall_clock64();
    prof[4 * (size_t)t + 3] = ((u64)xcc << 32) | hwid;
  }
}

// ------------------------------------------------------------------------------------------------ host: plans
struct PlanKey {
  unsigned H, lv, tq, tw, order;
  bool operator==(const PlanKey& o) const { return H == o.H && lv == o.lv && tq == o.tq && tw == o.tw && order == o.order; }
};
struct Plan {
  PlanKey key;
  PlanItem* d_items = nullptr;
  std::vector<PlanItem> items;
  uint32_t cnt_off[kMaxPlanLevels] = {};
  uint8_t csh[kMaxPlanLevels + 6] = {};
  uint32_t n_counters = 0;
  uint32_t n_bulk = 0;  // items [0, n_bulk) are of kinds S / U (k_tree_plan), the rest Q / W (k_tree_top)
};
std::mutex g_plan_mutex;
std::vector<Plan*> g_plans;  // a handful per process (one per subtree size in use); never freed before exit

struct Knobs {
  int enabled, min_log, order, tq, tw, spin_limit;
};
Knobs& knobs() {
  static Knobs k = [] {
    auto geti = [](const char* name, int dflt) {
      const char* e = getenv(name);
      return e ? atoi(e) : dflt;
    };
    Knobs v;
    v.enabled = geti("P2MT_PLAN", 0);  // measured: no faster than the separate launches (profiles/r05_one_launch_build.txt)
    v.min_log = geti("P2MT_PLAN_MIN_LOG", 18);
    v.order = geti("P2MT_PLAN_ORDER", 0);
    v.tq = geti("P2MT_PLAN_TQ", 16);
    v.tw = geti("P2MT_PLAN_TW", 12);
    v.spin_limit = geti("P2MT_PLAN_SPIN", 1 << 18);
    return v;
  }();
  return k;
}

inline PlanItem mk_item(uint32_t kind, unsigned h, uint32_t n, uint32_t j0) { return PlanItem{kind | (h << 4) | (n << 12), j0}; }

// Items of a 2^H-leaf subtree whose stage 1 builds 2^lv leaves per lane, in ticket order.
//   order 0: stage 1 first, then level by level (what the separate launches did, with per-chunk hand-offs instead of kernel boundaries)
//   order 1: stage 1 in groups of `grp` items, each followed by the upper items whose inputs ended `lag` tickets earlier
int build_plan(const PlanKey& key, Plan* p) {
  const unsigned H = key.H, lv = key.lv;
  if (H < lv + 8 || H > 31) return p2mt::fail(P2MT_EINVAL, "tree plan: subtree size out of range");
  std::vector<PlanItem> s_items, upper;
  std::vector<std::vector<PlanItem>> by_level(H + 1);
  const uint32_t n_s = 1u << (H - lv - 8);
  for (uint32_t b = 0; b < n_s; ++b) s_items.push_back(mk_item(KS, lv, 256, 256 * b));
  for (unsigned h = 0; h <= H; ++h) p->csh[h] = 6;
  for (unsigned h = lv + 1; h <= H; ++h) {
    const uint32_t n_level = 1u << (H - h);
    const uint32_t kind = n_level > (1u << key.tq) ? KU : (n_level > (1u << key.tw) ? KQ : KW);
    const uint32_t per = kind == KU ? 256 : (kind == KQ ? 64 : 4);
    if (kind == KW) p->csh[h] = 2;
    for (uint32_t j = 0; j < n_level; j += per) by_level[h].push_back(mk_item(kind, h, std::min(per, n_level - j), j));
  }
  uint32_t off = 0;
  for (unsigned h = 0; h <= H; ++h) {
    p->cnt_off[h] = off;
    if (h >= lv) off += std::max<uint32_t>(1u, (1u << (H - h)) >> p->csh[h]);
  }
  p->n_counters = off;
  std::vector<PlanItem>& out = p->items;
  out.clear();
  unsigned h_top = lv + 1;  // first level of the latency layouts (Q / W): they always come last, level by level (k_tree_top)
  while (h_top <= H && !by_level[h_top].empty() && (by_level[h_top][0].w & 15u) == KU) ++h_top;
  if (key.order == 0) {
    out = s_items;
    for (unsigned h = lv + 1; h < h_top; ++h) out.insert(out.end(), by_level[h].begin(), by_level[h].end());
  } else {
    // order 1: a U item becomes eligible once the last stage-1 item under it is `lag` stage-1 tickets old (one generation of resident
    // workgroups plus a margin: its producers have finished, the consumer does not hold a slot polling), one more `lag_u` per level
    // above (the time a U item takes, in stage-1 tickets).  What is left when stage 1 runs out follows level by level.
    const uint32_t lag = 1024 + 64, lag_u = 96;
    std::vector<size_t> next(H + 1, 0);
    for (uint32_t b = 0; b < n_s; ++b) {
      out.push_back(s_items[b]);
      for (unsigned h = lv + 1; h < h_top; ++h) {
        const uint64_t need = (uint64_t)lag + (uint64_t)lag_u * (h - lv - 1);
        while (next[h] < by_level[h].size()) {
          const PlanItem& it = by_level[h][next[h]];
          // a level-h item producing nodes [j, j + n) covers stage-1 items [.., ceil(((j + n) << (h - lv)) / 256))
          const uint64_t last_s = ((((uint64_t)it.j0 + (it.w >> 12)) << (h - lv)) + 255) >> 8;
          if (last_s + need > (uint64_t)b + 1) break;
          out.push_back(it);
          ++next[h];
        }
      }
    }
    for (unsigned h = lv + 1; h < h_top; ++h)
      for (; next[h] < by_level[h].size(); ++next[h]) out.push_back(by_level[h][next[h]]);
  }
  p->n_bulk = (uint32_t)out.size();
  for (unsigned h = h_top; h <= H; ++h) out.insert(out.end(), by_level[h].begin(), by_level[h].end());
  // every item's inputs must come from smaller tickets (the no-deadlock argument): check it
  {
    std::vector<std::vector<uint32_t>> owner(H + 1);  // owner[h][counter] = largest ticket that adds to it
    for (unsigned h = lv; h <= H; ++h) owner[h].assign(std::max<uint32_t>(1u, (1u << (H - h)) >> p->csh[h]), 0);
    std::vector<std::vector<uint8_t>> seen(H + 1);
    for (unsigned h = lv; h <= H; ++h) seen[h].assign(owner[h].size(), 0);
    for (uint32_t t = 0; t < out.size(); ++t) {
      const uint32_t kind = out[t].w & 15u, h = (out[t].w >> 4) & 63u, n = out[t].w >> 12, j0 = out[t].j0;
      if (kind != KS) {
        const unsigned sh = p->csh[h - 1];
        for (uint32_t c = (2 * j0) >> sh; c <= (2 * j0 + 2 * n - 1) >> sh; ++c)
          if (!seen[h - 1][c] || owner[h - 1][c] >= t) return p2mt::fail(P2MT_EINVAL, "tree plan: an item precedes its inputs (plan builder bug)");
        if ((((2 * j0 + 2 * n - 1) >> sh) - ((2 * j0) >> sh)) >= 8) return p2mt::fail(P2MT_EINVAL, "tree plan: more than 8 input counters");
      }
      const unsigned sh = p->csh[h];
      for (uint32_t c = j0 >> sh; c <= (j0 + n - 1) >> sh; ++c) {
        seen[h][c] = 1;
        owner[h][c] = std::max(owner[h][c], t);
      }
    }
  }
  return P2MT_OK;
}

int get_plan(const PlanKey& key, Plan** out) {
  std::lock_guard<std::mutex> lock(g_plan_mutex);
  for (Plan* p : g_plans)
    if (p->key == key) {
      *out = p;
      return P2MT_OK;
    }
  Plan* p = new Plan();
  p->key = key;
  int rc = build_plan(key, p);
  if (rc == P2MT_OK && hipMalloc((void**)&p->d_items, p->items.size() * sizeof(PlanItem)) != hipSuccess)
    rc = p2mt::fail(P2MT_ENOMEM, "tree plan: hipMalloc failed");
  if (rc == P2MT_OK &&
      hipMemcpy(p->d_items, p->items.data(), p->items.size() * sizeof(PlanItem), hipMemcpyHostToDevice) != hipSuccess)
    rc = p2mt::fail(P2MT_EHIP, "tree plan: upload failed");
  if (rc != P2MT_OK) {
    if (p->d_items) (void)hipFree(p->d_items);
    delete p;
    return rc;
  }
  g_plans.push_back(p);
  *out = p;
  return P2MT_OK;
}

// per-item device timestamps of the last profiled launch (p2mt_debug_plan_profile)
thread_local u64* tl_prof = nullptr;
thread_local size_t tl_prof_items = 0, tl_prof_cap = 0;
thread_local int tl_prof_on = 0;
thread_local std::vector<PlanItem>* tl_prof_plan = nullptr;

}  // namespace

namespace p2mt {

bool tree_plan_wanted(unsigned H) {
  const Knobs& k = knobs();
  return k.enabled && rt().mds == 2 && rt().partial == 0 && rt().subtree_block == 256 && H >= (unsigned)k.min_log && H <= 31;
}

static int plan_for(unsigned H, Plan** p) {
  const Knobs& k = knobs();
  // the stage-1 subtree size follows the rule of the separate launches
  unsigned lv = subtree_levels_for((size_t)1 << H);
  if (lv < 2 || lv > 4) lv = 4;
  PlanKey key{H, lv, (unsigned)k.tq, (unsigned)k.tw, (unsigned)k.order};
  return get_plan(key, p);
}

int tree_plan_state_bytes(unsigned H, size_t* bytes_out) {
  Plan* p = nullptr;
  P2MT_TRY(plan_for(H, &p));
  *bytes_out = ((size_t)(kStateHdr + p->n_counters) * 4 + 15) & ~(size_t)15;
  return P2MT_OK;
}

int tree_plan_launch(const TreeLayout& lay, const uint64_t* d_leaves, size_t first_leaf, unsigned H, uint32_t* d_state) {
  const Knobs& k = knobs();
  Plan* p = nullptr;
  P2MT_TRY(plan_for(H, &p));
  const size_t state_bytes = ((size_t)(kStateHdr + p->n_counters) * 4 + 15) & ~(size_t)15;
  if (!d_state) P2MT_TRY(scratch_get_shared(kScratchPlan, state_bytes, (void**)&d_state));
  hipStream_t st = rt().stream;
  P2MT_HIP(hipMemsetAsync(d_state, 0, state_bytes, st));
  PlanArgs pa;
  pa.items = p->d_items;
  pa.n_items = (uint32_t)p->items.size();
  pa.H = H;
  pa.spin_limit = (uint32_t)k.spin_limit;
  pa.prof = 0;
  memcpy(pa.cnt_off, p->cnt_off, sizeof pa.cnt_off);
  memcpy(pa.csh, p->csh, sizeof pa.csh);
  u64* d_prof = nullptr;
  if (tl_prof_on) {
    const size_t need = 4 * p->items.size() * 8;
    if (need > tl_prof_cap) {
      if (tl_prof) (void)hipFree(tl_prof);
      tl_prof = nullptr;
      tl_prof_cap = 0;
      P2MT_HIP(hipMalloc((void**)&tl_prof, need));
      tl_prof_cap = need;
    }
    P2MT_HIP(hipMemsetAsync(tl_prof, 0, need, st));
    d_prof = tl_prof;
    tl_prof_items = p->items.size();
    tl_prof_plan = &p->items;
    pa.prof = 1;
  }
  const int prof_slot = prof_begin();  // the dominant launch
  if (p->n_bulk)
    hipLaunchKernelGGL(k_tree_plan, dim3(p->n_bulk), dim3(256), 0, st, (const u64*)d_leaves, lay, first_leaf, pa, d_state, d_prof,
                       perm_ctx());
  P2MT_LAUNCH_CHECK();
  prof_end(prof_slot);
  if (pa.n_items > p->n_bulk)
    hipLaunchKernelGGL(k_tree_top, dim3(pa.n_items - p->n_bulk), dim3(256), 0, st, lay, first_leaf, pa, p->n_bulk, d_state, d_prof,
                       perm_ctx());
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}

}  // namespace p2mt

// ---------------------------------------------------------------- debug / measurement hooks (not part of the reference's surface)
extern "C" int p2mt_debug_plan_knobs(int enabled, int min_log, int order, int tq, int tw) {
  return p2mt::abi_guard([&]() -> int {
  Knobs& k = knobs();
  if (enabled >= 0) k.enabled = enabled;
  if (min_log >= 0) k.min_log = min_log;
  if (order >= 0) k.order = order;
  if (tq >= 0) k.tq = tq;
  if (tw >= 0) k.tw = tw;
  return P2MT_OK;
  });
}

extern "C" int p2mt_debug_plan_profile(int on) {
  return p2mt::abi_guard([&]() -> int {
  tl_prof_on = on ? 1 : 0;
  return P2MT_OK;
  });
}

// rows of the last profiled launch of this thread: [ticket][6] = kind, level, first node, start, ready, end (device clock ticks of 10 ns,
// relative to the first start) and [ticket][6..8) = XCC id, HW_ID; returns the number of items (<= max_items rows are written)
extern "C" int64_t p2mt_debug_plan_profile_read(uint64_t* rows_out, size_t max_items) {
  int64_t n_out = 0;
  int rc = p2mt::abi_guard([&]() -> int {
  if (!tl_prof || !tl_prof_plan) return p2mt::fail(P2MT_EINVAL, "no profiled plan launch on this thread");
  P2MT_HIP(hipStreamSynchronize(p2mt::rt().stream));
  std::vector<uint64_t> raw(4 * tl_prof_items);
  P2MT_HIP(hipMemcpy(raw.data(), tl_prof, raw.size() * 8, hipMemcpyDeviceToHost));
  uint64_t t0 = ~0ull;
  for (size_t i = 0; i < tl_prof_items; ++i) t0 = std::min(t0, raw[4 * i]);
  const size_t n = std::min(tl_prof_items, max_items);
  for (size_t i = 0; i < n && rows_out; ++i) {
    const PlanItem& it = (*tl_prof_plan)[i];
    uint64_t* r = rows_out + 8 * i;
    r[0] = it.w & 15u;
    r[1] = (it.w >> 4) & 63u;
    r[2] = it.j0;
    r[3] = raw[4 * i] - t0;
    r[4] = raw[4 * i + 1] ? raw[4 * i + 1] - t0 : 0;
    r[5] = raw[4 * i + 2] - t0;
    r[6] = raw[4 * i + 3] >> 32;
    r[7] = raw[4 * i + 3] & 0xFFFFFFFFull;
  }
  n_out = (int64_t)tl_prof_items;
  return P2MT_OK;
  });
  return rc == P2MT_OK ? n_out : rc;
}
