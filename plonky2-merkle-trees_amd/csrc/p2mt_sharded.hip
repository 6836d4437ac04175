// p2mt_sharded.hip -- leaf-range sharding of one 2^k-leaf MMR over the GPUs of a node, behind the C ABI (north_star:
// "tree build partitions across the 8 GPUs of one node with a single RCCL all-gather over xGMI of the per-shard subtree roots before
// the final peaks ... through a thin extern "C" FFI").
//
// The reference's build is the serial loop /root/reference/src/mmr/merkle_mountain_ranges.rs:89-120; here one process per GPU owns
// leaves [rank * n_local, (rank + 1) * n_local) (n_local and world powers of two), builds that perfect subtree with no data-path
// collective, and the only exchange of the path -- world x 32 bytes -- happens on the library stream between two launches of this
// library:  p2mt_mmr_root_dev -> ncclAllGather (RCCL over xGMI) -> k_combine_roots (the log2(world) top levels, redundantly on every
// rank).  Nothing visits the host until a caller asks for the root.
//
// RCCL is reached through dlopen("librccl.so.1") and six symbols: a process that already holds an RCCL (a PyTorch process does) is
// served by that copy, a process without one loads /opt/rocm's, and a machine without RCCL gets P2MT_EHIP from the entry points that
// need it -- single-GPU users of the library never load it.  Who owns the communicator is said per constructor below.
// For transports that are not RCCL (the gloo CPU tests, an MPI caller) the exchange can be a host callback instead.
#include "runtime.h"

#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

using p2mt::rt;

namespace {

// the slice of rccl.h this file needs (ABI-stable across NCCL 2.x / RCCL)
typedef void* nccl_comm_t;
struct nccl_unique_id {
  char internal[128];
};
enum { kNcclSuccess = 0, kNcclUint64 = 5 };
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(nccl_unique_id*) = nullptr;
  int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*CommCount)(nccl_comm_t, int*) = nullptr;
  int (*CommUserRank)(nccl_comm_t, int*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
std::mutex g_rccl_mutex;
Rccl g_rccl;
int rccl_load(Rccl** out) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!g_rccl.h) {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the copy the process already holds, if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return p2mt::fail(P2MT_EHIP, "RCCL (librccl.so.1) is not available on this machine: multi-GPU entry points need it");
    Rccl r;
    r.h = h;
    *(void**)&r.GetUniqueId = dlsym(h, "ncclGetUniqueId");
    *(void**)&r.CommInitRank = dlsym(h, "ncclCommInitRank");
    *(void**)&r.CommDestroy = dlsym(h, "ncclCommDestroy");
    *(void**)&r.CommCount = dlsym(h, "ncclCommCount");
    *(void**)&r.CommUserRank = dlsym(h, "ncclCommUserRank");
    *(void**)&r.AllGather = dlsym(h, "ncclAllGather");
    *(void**)&r.GetErrorString = dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommCount || !r.CommUserRank || !r.AllGather)
      return p2mt::fail(P2MT_EHIP, "librccl.so.1 lacks a symbol this library needs");
    g_rccl = r;
  }
  *out = &g_rccl;
  return P2MT_OK;
}
int rccl_fail(const Rccl* r, int code, const char* what) {
  snprintf(p2mt::err_buf(), p2mt::kErrLen, "%s failed: %s", what, r->GetErrorString ? r->GetErrorString(code) : "RCCL error");
  return P2MT_EHIP;
}

bool pow2(size_t x) { return x && !(x & (x - 1)); }

}  // namespace

struct p2mt_sharded_mmr {
  p2mt_mmr* local = nullptr;
  size_t n_local = 0;
  int rank = 0, world = 1;
  nccl_comm_t comm = nullptr;
  bool owns_comm = false;
  p2mt_allgather32_fn exchange = nullptr;  // host transport instead of RCCL (set_exchange)
  void* exchange_user = nullptr;
  uint64_t* d_all = nullptr;  // device: [world roots | world - 1 top nodes | root] x 4 words, then this rank's root (4 words)
  std::vector<uint64_t> h_all;  // host copy after p2mt_sharded_mmr_root
  bool built = false, fetched = false;
};

static int sharded_alloc(p2mt_sharded_mmr** out, size_t n_local, int rank, int world) {
  P2MT_TRY(p2mt::ensure_init());
  if (!out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  *out = nullptr;
  if (!pow2(n_local) || world < 1 || world > 1024 || !pow2((size_t)world) || rank < 0 || rank >= world)
    return p2mt::fail(P2MT_EINVAL, "sharded MMR: n_local and world must be powers of two (world <= 1024), 0 <= rank < world");
  if (((n_local * (size_t)world) >> 40) != 0) return p2mt::fail(P2MT_ERANGE, "sharded MMR too large");
  p2mt_sharded_mmr* s = new (std::nothrow) p2mt_sharded_mmr();
  if (!s) return p2mt::fail(P2MT_ENOMEM, "out of host memory");
  s->n_local = n_local, s->rank = rank, s->world = world;
  int rc = p2mt_mmr_create(&s->local);
  if (rc == P2MT_OK) rc = p2mt_mmr_reserve(s->local, n_local);
  if (rc == P2MT_OK && hipMalloc((void**)&s->d_all, (size_t)(2 * world + 1) * 32) != hipSuccess) {
    (void)hipGetLastError();
    rc = p2mt::fail(P2MT_ENOMEM, "hipMalloc(shard roots) failed");
  }
  if (rc != P2MT_OK) {
    if (s->local) (void)p2mt_mmr_destroy(s->local);
    delete s;
    return rc;
  }
  s->h_all.assign((size_t)2 * world * 4, 0);
  *out = s;
  return P2MT_OK;
}

// The caller's communicator: rank and size are checked against the arguments; the library enqueues ONE ncclAllGather per build on its
// own stream with it and never destroys it.  nccl_comm may be NULL when world == 1 (nothing is exchanged then).
extern "C" int p2mt_sharded_mmr_create(p2mt_sharded_mmr** out, size_t n_local, int rank, int world, void* nccl_comm) {
  return p2mt::abi_guard([&]() -> int {
  if (world > 1 && !nccl_comm) return p2mt::fail(P2MT_EINVAL, "sharded MMR: world > 1 needs a communicator (or p2mt_sharded_mmr_create_with_id / _set_exchange)");
  if (nccl_comm) {
    Rccl* r;
    P2MT_TRY(rccl_load(&r));
    int cnt = -1, ur = -1;
    int e = r->CommCount(nccl_comm, &cnt);
    if (e != kNcclSuccess) return rccl_fail(r, e, "ncclCommCount");
    e = r->CommUserRank(nccl_comm, &ur);
    if (e != kNcclSuccess) return rccl_fail(r, e, "ncclCommUserRank");
    if (cnt != world || ur != rank) return p2mt::fail(P2MT_EINVAL, "sharded MMR: the communicator's size / rank differ from the arguments");
  }
  P2MT_TRY(sharded_alloc(out, n_local, rank, world));
  (*out)->comm = nccl_comm;
  return P2MT_OK;
  });
}

// ncclGetUniqueId for callers that have no RCCL binding of their own: rank 0 calls this and sends the 128 bytes to every rank over
// whatever channel it has; then every rank calls p2mt_sharded_mmr_create_with_id.
extern "C" int p2mt_nccl_unique_id(void* id_out) {
  return p2mt::abi_guard([&]() -> int {
  if (!id_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  P2MT_TRY(p2mt::ensure_init());
  Rccl* r;
  P2MT_TRY(rccl_load(&r));
  nccl_unique_id id;
  const int e = r->GetUniqueId(&id);
  if (e != kNcclSuccess) return rccl_fail(r, e, "ncclGetUniqueId");
  memcpy(id_out, &id, sizeof id);
  return P2MT_OK;
  });
}

// A communicator of the library's own (ncclCommInitRank: collective, every rank of `world` must make this call with the same id);
// destroyed with the handle.
extern "C" int p2mt_sharded_mmr_create_with_id(p2mt_sharded_mmr** out, size_t n_local, int rank, int world, const void* nccl_unique_id_128) {
  return p2mt::abi_guard([&]() -> int {
  if (!nccl_unique_id_128) return p2mt::fail(P2MT_EINVAL, "null pointer");
  P2MT_TRY(sharded_alloc(out, n_local, rank, world));
  Rccl* r;
  int rc = rccl_load(&r);
  if (rc == P2MT_OK) {
    nccl_unique_id id;
    memcpy(&id, nccl_unique_id_128, sizeof id);
    nccl_comm_t comm = nullptr;
    const int e = r->CommInitRank(&comm, world, id, rank);
    if (e != kNcclSuccess) rc = rccl_fail(r, e, "ncclCommInitRank");
    else (*out)->comm = comm, (*out)->owns_comm = true;
  }
  if (rc != P2MT_OK) {
    (void)p2mt_sharded_mmr_destroy(*out);
    *out = nullptr;
  }
  return rc;
  });
}

// the same handle with a host transport from the start (no RCCL involved at all): see p2mt_sharded_mmr_set_exchange
extern "C" int p2mt_sharded_mmr_create_exchange(p2mt_sharded_mmr** out, size_t n_local, int rank, int world, p2mt_allgather32_fn fn,
                                                void* user) {
  return p2mt::abi_guard([&]() -> int {
  if (!fn) return p2mt::fail(P2MT_EINVAL, "null callback");
  P2MT_TRY(sharded_alloc(out, n_local, rank, world));
  (*out)->exchange = fn;
  (*out)->exchange_user = user;
  return P2MT_OK;
  });
}

// A host transport in place of RCCL: `fn(user, mine, all)` must fill all[world][4] with every rank's 4 words (an all-gather of 32
// bytes over whatever the caller has: MPI, gloo, a socket).  The roots then travel through the host: one read-back before the call,
// one upload after it.
extern "C" int p2mt_sharded_mmr_set_exchange(p2mt_sharded_mmr* s, p2mt_allgather32_fn fn, void* user) {
  return p2mt::abi_guard([&]() -> int {
  if (!s) return p2mt::fail(P2MT_EINVAL, "null handle");
  s->exchange = fn;
  s->exchange_user = user;
  return P2MT_OK;
  });
}

extern "C" int p2mt_sharded_mmr_destroy(p2mt_sharded_mmr* s) {
  return p2mt::abi_guard([&]() -> int {
  if (!s) return P2MT_OK;
  (void)hipStreamSynchronize(rt().stream);
  if (s->comm && s->owns_comm) {
    Rccl* r;
    if (rccl_load(&r) == P2MT_OK) (void)r->CommDestroy(s->comm);
  }
  if (s->d_all) (void)hipFree(s->d_all);
  if (s->local) (void)p2mt_mmr_destroy(s->local);
  delete s;
  return P2MT_OK;
  });
}

extern "C" p2mt_mmr* p2mt_sharded_mmr_local(p2mt_sharded_mmr* s) { return s ? s->local : nullptr; }

// the exchange and the top levels, behind a finished local build; on the library stream, nothing synchronised (RCCL transport)
static int sharded_finish(p2mt_sharded_mmr* s) {
  const size_t w = (size_t)s->world;
  uint64_t* d_mine = s->d_all + 4 * (2 * w);
  uint64_t* d_roots = s->d_all;
  uint64_t* d_top = s->d_all + 4 * w;
  uint64_t* d_root = s->d_all + 4 * (2 * w - 1);
  P2MT_TRY(p2mt_mmr_root_dev(s->local, d_mine));  // a perfect subtree: its one peak is its root
  hipStream_t st = rt().stream;
  if (s->world == 1) {
    P2MT_HIP(hipMemcpyAsync(d_roots, d_mine, 32, hipMemcpyDeviceToDevice, st));
  } else if (s->exchange) {
    uint64_t mine[4];
    P2MT_HIP(hipMemcpyAsync(mine, d_mine, 32, hipMemcpyDeviceToHost, st));
    P2MT_HIP(hipStreamSynchronize(st));
    std::vector<uint64_t> all(4 * w, 0);
    if (s->exchange(s->exchange_user, mine, all.data()) != 0) return p2mt::fail(P2MT_EHIP, "sharded MMR: the caller's exchange failed");
    P2MT_HIP(hipMemcpyAsync(d_roots, all.data(), 32 * w, hipMemcpyHostToDevice, st));
    P2MT_HIP(hipStreamSynchronize(st));  // `all` dies with this scope
  } else {
    if (!s->comm) return p2mt::fail(P2MT_EINVAL, "sharded MMR: no communicator and no exchange callback");
    Rccl* r;
    P2MT_TRY(rccl_load(&r));
    const int e = r->AllGather(d_mine, d_roots, 4, kNcclUint64, s->comm, st);  // THE collective of the path: world x 32 bytes
    if (e != kNcclSuccess) return rccl_fail(r, e, "ncclAllGather");
  }
  P2MT_TRY(p2mt_mmr_combine_shard_roots_dev(d_roots, w, w > 1 ? d_top : nullptr, d_root));
  s->built = true;
  s->fetched = false;
  return P2MT_OK;
}

// reset + extend of this rank's n_local leaves (device-resident) + the exchange + the top levels; enqueue only with RCCL
extern "C" int p2mt_sharded_mmr_build_dev(p2mt_sharded_mmr* s, const uint64_t* d_local_leaves) {
  return p2mt::abi_guard([&]() -> int {
  if (!s || !d_local_leaves) return p2mt::fail(P2MT_EINVAL, "null pointer");
  s->built = false;
  P2MT_TRY(p2mt_mmr_reset(s->local));
  P2MT_TRY(p2mt_mmr_extend_dev(s->local, d_local_leaves, s->n_local));
  return sharded_finish(s);
  });
}
extern "C" int p2mt_sharded_mmr_build(p2mt_sharded_mmr* s, const uint64_t* local_leaves) {
  return p2mt::abi_guard([&]() -> int {
  if (!s || !local_leaves) return p2mt::fail(P2MT_EINVAL, "null pointer");
  s->built = false;
  P2MT_TRY(p2mt_mmr_reset(s->local));
  P2MT_TRY(p2mt_mmr_extend(s->local, local_leaves, s->n_local));
  return sharded_finish(s);
  });
}
// the exchange alone, for a caller that extended the local shard itself (p2mt_sharded_mmr_local)
extern "C" int p2mt_sharded_mmr_finish(p2mt_sharded_mmr* s) {
  return p2mt::abi_guard([&]() -> int {
  if (!s) return p2mt::fail(P2MT_EINVAL, "null handle");
  if (p2mt_mmr_num_leaves(s->local) != s->n_local) return p2mt::fail(P2MT_EINVAL, "sharded MMR: the local shard does not hold n_local leaves");
  return sharded_finish(s);
  });
}

static int sharded_fetch(p2mt_sharded_mmr* s) {
  if (!s->built) return p2mt::fail(P2MT_EINVAL, "sharded MMR: no finished build");
  if (s->fetched) return P2MT_OK;
  P2MT_HIP(hipMemcpyAsync(s->h_all.data(), s->d_all, (size_t)2 * s->world * 32, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  s->fetched = true;
  return P2MT_OK;
}

// one read-back: the root of the whole MMR, optionally every shard's root and the world - 1 top nodes (level-major, bottom-up)
extern "C" int p2mt_sharded_mmr_root(p2mt_sharded_mmr* s, uint64_t* root_out, uint64_t* shard_roots_out, uint64_t* top_nodes_out) {
  return p2mt::abi_guard([&]() -> int {
  if (!s || !root_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  P2MT_TRY(sharded_fetch(s));
  const size_t w = (size_t)s->world;
  memcpy(root_out, s->h_all.data() + 4 * (2 * w - 1), 32);
  // (an all-ones word is not a field element: a shard whose one-launch build gave up hands that in as its root, p2mt_mmr.hip)
  for (size_t k = 0; k < 4 * w; ++k)
    if (s->h_all[k] == ~0ull) return p2mt::fail(P2MT_EHIP, "sharded MMR: a shard's build failed (its root is the failure mark)");
  if (shard_roots_out) memcpy(shard_roots_out, s->h_all.data(), 32 * w);
  if (top_nodes_out && w > 1) memcpy(top_nodes_out, s->h_all.data() + 4 * w, 32 * (w - 1));
  return P2MT_OK;
  });
}

// MMR::get_proof for a leaf THIS rank owns (global index): the log2(n_local) siblings inside the shard come from the local MMR, the
// log2(world) above it from the gathered roots; the one peak is the root.  (A proof is < 2 KB: the owner makes it and sends it.)
extern "C" int p2mt_sharded_mmr_proof(p2mt_sharded_mmr* s, size_t global_leaf, uint64_t* siblings_out, uint8_t* lefts_out, int* n_siblings,
                                      uint64_t* root_out) {
  return p2mt::abi_guard([&]() -> int {
  if (!s || !siblings_out || !lefts_out || !n_siblings || !root_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  const size_t owner = global_leaf / s->n_local, local_idx = global_leaf % s->n_local;
  if (owner >= (size_t)s->world) return p2mt::fail(P2MT_EINVAL, "sharded MMR: leaf index out of range");
  if (owner != (size_t)s->rank) return p2mt::fail(P2MT_EINVAL, "sharded MMR: the leaf belongs to another rank's shard (its owner makes the proof)");
  P2MT_TRY(sharded_fetch(s));
  const int64_t mmr_index = p2mt_get_mmr_index(local_idx);
  if (mmr_index < 0) return p2mt::fail(P2MT_ERANGE, "get_mmr_index overflow");
  uint64_t peaks[4 * P2MT_MAX_PROOF_LEN];
  int ns = 0, npk = 0;
  P2MT_TRY(p2mt_mmr_proof(s->local, (size_t)mmr_index, siblings_out, lefts_out, &ns, peaks, &npk, nullptr));
  const size_t w = (size_t)s->world;
  const uint64_t* level = s->h_all.data();  // shard roots, then the top nodes level by level
  size_t idx = owner, cnt = w, off = 0;
  while (cnt > 1) {
    if (ns >= P2MT_MAX_PROOF_LEN) return p2mt::fail(P2MT_ERANGE, "proof longer than P2MT_MAX_PROOF_LEN");
    memcpy(siblings_out + 4 * ns, level + 4 * (idx ^ 1), 32);
    lefts_out[ns] = (uint8_t)(idx & 1);
    ++ns;
    level = s->h_all.data() + 4 * (w + off);
    off += cnt / 2;
    idx >>= 1;
    cnt /= 2;
  }
  *n_siblings = ns;
  memcpy(root_out, s->h_all.data() + 4 * (2 * w - 1), 32);
  return P2MT_OK;
  });
}
