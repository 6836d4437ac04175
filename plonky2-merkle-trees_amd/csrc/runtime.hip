// runtime.hip -- process-global runtime of libp2mt_hip.so: device selection, stream, error text, HIP-event timer.
#include "runtime.h"
#include "poseidon_constants.h"  // host copy of the tables (P2MT_QUAL defaults to static const)

#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <mutex>
#include <thread>

namespace p2mt {

Runtime& rt() {
  static Runtime r;
  return r;
}

namespace {
thread_local hipStream_t tl_stream = nullptr;
thread_local char tl_err[kErrLen] = {0};
}  // namespace
StreamRef::operator hipStream_t() const { return tl_stream; }
StreamRef& StreamRef::operator=(hipStream_t s) {
  tl_stream = s;
  return *this;
}
char* err_buf() { return tl_err; }

int fail_hip(hipError_t e, const char* what, const char* file, int line) {
  snprintf(err_buf(), kErrLen, "HIP error '%s' in %s (%s:%d)", hipGetErrorString(e), what, file, line);
  (void)hipGetLastError();  // clear the sticky error
  return P2MT_EHIP;
}

int fail(int code, const char* msg) {
  snprintf(err_buf(), kErrLen, "%s", msg);
  return code;
}

namespace {
thread_local void* tl_scratch_ptr[kScratchCount] = {};  // per thread: concurrent provers must not share scratch
thread_local size_t tl_scratch_cap[kScratchCount] = {};
thread_local uint64_t tl_scratch_epoch = 0;  // bumped whenever a scratch buffer of this thread is (re)allocated or freed
}  // namespace

// A worker thread that ends without p2mt_thread_stream_destroy() still gives its scratch back: the holder's destructor
// runs at thread exit.  (Not for the thread that loaded the library: at process exit the HIP runtime may already be gone.)
namespace {
const std::thread::id g_loader_thread = std::this_thread::get_id();
struct ScratchHolder {
  bool armed = false;
  ~ScratchHolder() {
    if (!armed || std::this_thread::get_id() == g_loader_thread) return;
    for (int k = 0; k < kScratchCount; ++k)
      if (tl_scratch_ptr[k]) {
        (void)hipFree(tl_scratch_ptr[k]);  // hipFree synchronises the device: nothing still reads the buffer
        tl_scratch_ptr[k] = nullptr;
      }
  }
};
thread_local ScratchHolder tl_scratch_holder;
}  // namespace

uint64_t scratch_epoch() { return tl_scratch_epoch; }

namespace {
thread_local size_t tl_scratch_req[kScratchCount] = {};
thread_local BatchCtx tl_batch;
__global__ __launch_bounds__(256) void k_batch_broadcast(uint64_t* p, size_t words, size_t stride_words) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < words) p[(size_t)(blockIdx.z + 1) * stride_words + i] = p[i];
}
__device__ __forceinline__ uint64_t* rebase(uint64_t* p, const BatchArg& ba) {
  return ((uint64_t)p - ba.base) < ba.span ? reinterpret_cast<uint64_t*>((uint64_t)p + (uint64_t)blockIdx.z * ba.stride) : p;
}
__global__ __launch_bounds__(256) void k_batch_copy(uint64_t* dst, const uint64_t* src, size_t words, BatchArg ba) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < words) rebase(dst, ba)[i] = rebase(const_cast<uint64_t*>(src), ba)[i];
}
__global__ __launch_bounds__(256) void k_batch_fill(uint64_t* dst, uint64_t v, size_t words, BatchArg ba) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < words) rebase(dst, ba)[i] = v;
}
}  // namespace
int batch_copy(void* dst, const void* src, size_t bytes) {
  const BatchCtx& b = tl_batch;
  if (b.B <= 1) {
    P2MT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, rt().stream));
    return P2MT_OK;
  }
  if (bytes & 7) return fail(P2MT_EINVAL, "batch_copy: size not a multiple of 8");
  hipLaunchKernelGGL(k_batch_copy, dim3((unsigned)((bytes / 8 + 255) / 256), 1, b.B), dim3(256), 0, rt().stream, (uint64_t*)dst,
                     (const uint64_t*)src, bytes / 8, b.arg);
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}
int batch_fill(void* dst, int byte_value, size_t bytes) {
  const BatchCtx& b = tl_batch;
  if (b.B <= 1) {
    P2MT_HIP(hipMemsetAsync(dst, byte_value, bytes, rt().stream));
    return P2MT_OK;
  }
  if (bytes & 7) return fail(P2MT_EINVAL, "batch_fill: size not a multiple of 8");
  hipLaunchKernelGGL(k_batch_fill, dim3((unsigned)((bytes / 8 + 255) / 256), 1, b.B), dim3(256), 0, rt().stream, (uint64_t*)dst,
                     0x0101010101010101ull * (uint64_t)(byte_value & 0xFF), bytes / 8, b.arg);
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}
void scratch_track_reset() {
  for (int k = 0; k < kScratchCount; ++k) tl_scratch_req[k] = 0;
}
size_t scratch_track_max(int slot) { return tl_scratch_req[slot]; }
BatchCtx& batch() { return tl_batch; }

int batch_broadcast(void* p, size_t bytes) {
  const BatchCtx& b = tl_batch;
  if (b.B <= 1 || bytes == 0) return P2MT_OK;
  if ((uint64_t)p - b.arg.base >= b.arg.span || (bytes & 7) || ((uint64_t)p & 7)) return fail(P2MT_EINVAL, "batch_broadcast: not a per-proof buffer");
  const size_t words = bytes / 8;
  hipLaunchKernelGGL(k_batch_broadcast, dim3((unsigned)((words + 255) / 256), 1, b.B - 1), dim3(256), 0, rt().stream, (uint64_t*)p,
                     words, (size_t)(b.arg.stride / 8));
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}

static int scratch_get_impl(int slot, size_t bytes, void** out, bool shared);
int scratch_get(int slot, size_t bytes, void** out) { return scratch_get_impl(slot, bytes, out, false); }
int scratch_get_shared(int slot, size_t bytes, void** out) { return scratch_get_impl(slot, bytes, out, true); }

static int scratch_get_impl(int slot, size_t bytes, void** out, bool shared) {
  void** ptr = tl_scratch_ptr;
  size_t* cap = tl_scratch_cap;
  if (bytes == 0) bytes = 8;
  if (!shared && bytes > tl_scratch_req[slot]) tl_scratch_req[slot] = bytes;
  if (tl_batch.arena && !shared) {  // inside a batch: block 0's arena, sized from a tracked single-proof run
    if (bytes > tl_batch.slot_cap[slot]) return fail(P2MT_EINVAL, "batch: scratch request exceeds the per-proof arena");
    *out = tl_batch.arena + tl_batch.slot_off[slot];
    return P2MT_OK;
  }
  if (bytes > cap[slot]) {
    if (ptr[slot]) {
      (void)hipStreamSynchronize(rt().stream);  // earlier kernels may still read the old buffer
      (void)hipFree(ptr[slot]);
      ptr[slot] = nullptr;
      cap[slot] = 0;
    }
    const size_t want = bytes + bytes / 4;
    ++tl_scratch_epoch;
    tl_scratch_holder.armed = true;
    if (hipMalloc(&ptr[slot], want) != hipSuccess) {
      (void)hipGetLastError();
      return fail(P2MT_ENOMEM, "hipMalloc(scratch) failed");
    }
    cap[slot] = want;
  }
  *out = ptr[slot];
  return P2MT_OK;
}

// free the calling thread's scratch buffers (worker threads call it before they exit)
void scratch_release_thread() {
  (void)hipStreamSynchronize(rt().stream);
  ++tl_scratch_epoch;
  for (int k = 0; k < kScratchCount; ++k) {
    if (tl_scratch_ptr[k]) (void)hipFree(tl_scratch_ptr[k]);
    tl_scratch_ptr[k] = nullptr;
    tl_scratch_cap[k] = 0;
  }
}

int prof_begin() {
  Runtime& r = rt();
  if (!r.profile || r.prof_n >= Runtime::kMaxProf) return -1;
  const int i = r.prof_n++;
  if (!r.prof_ev[2 * i]) {
    if (hipEventCreate(&r.prof_ev[2 * i]) != hipSuccess || hipEventCreate(&r.prof_ev[2 * i + 1]) != hipSuccess) return -1;
  }
  (void)hipEventRecord(r.prof_ev[2 * i], r.stream);
  return i;
}

void prof_end(int slot) {
  if (slot >= 0) (void)hipEventRecord(rt().prof_ev[2 * slot + 1], rt().stream);
}

unsigned subtree_levels_for(size_t n_leaves) {
  const Runtime& r = rt();
  if (!r.subtree_auto || r.subtree_levels == 0) return r.subtree_levels;
  // the A/B instantiations (sparse partial rounds, matrix-pipe MDS, 64- / 128-lane workgroups) exist for 2^4-leaf subtrees only
  if (r.partial != 0 || r.subtree_block != 256) return 4;
  unsigned lv = 4;
  // 2^4 leaves per lane from 2^24 leaves up, 2^3 from 2^23, 2^2 below: the smallest subtree that still leaves 2^20 lanes (four rounds of
  // four wavefronts per SIMD) wins at every size since round 4's faster level kernels (2^21 / 2^22 / 2^23 leaves: 0.88 / 1.44 / 2.56 ms
  // against 0.91 / 1.56 / 2.68 with the round-3 rule of 2^18 lanes; profiles/r04_subtree_size_sweep.txt)
  while (lv > 2 && (n_leaves >> lv) < ((size_t)1 << 20)) --lv;
  return lv;
}

PermCtx perm_ctx() {
  const Runtime& r = rt();
  return PermCtx{r.d_rc, r.force_fallback ? ~0ull : 0ull, r.d_rc + (646 + 22 * 11 * 2 + 121 * 2 + 84 * 7)};  // kP3K
}

int ensure_init() {
  if (rt().initialised) return P2MT_OK;  // (set last, under the lock, by p2mt_init)
  return p2mt_init(rt().device);
}

}  // namespace p2mt

using p2mt::rt;

extern "C" int p2mt_device_count(void) {
  return p2mt::abi_guard([&]() -> int {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
  });
}

// The process-global state is written under one lock; a process drives ONE device (one process per GPU): the twiddle / coset
// tables cached by the commit step live on the device of the first init, so a later init on another device is refused instead
// of leaving them pointing at the old one.
static std::mutex g_init_mutex;
extern "C" int p2mt_init(int device) {
  return p2mt::abi_guard([&]() -> int {
  std::lock_guard<std::mutex> lock(g_init_mutex);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    (void)hipGetLastError();
    return p2mt::fail(P2MT_EHIP, "no HIP device available: this library has no CPU fallback");
  }
  if (device < 0 || device >= n) return p2mt::fail(P2MT_EINVAL, "device index out of range");
  if (rt().initialised && rt().device != device)
    return p2mt::fail(P2MT_EINVAL, "p2mt_init: already initialised on another device (one process drives one GPU)");
  P2MT_HIP(hipSetDevice(device));
  if (rt().initialised) return P2MT_OK;
  rt().device = device;
  if (rt().ev_start) (void)hipEventDestroy(rt().ev_start);
  if (rt().ev_stop) (void)hipEventDestroy(rt().ev_stop);
  P2MT_HIP(hipEventCreate(&rt().ev_start));
  P2MT_HIP(hipEventCreate(&rt().ev_stop));
  if (rt().d_rc) (void)hipFree(rt().d_rc);
  {  // 360 round constants + first-round S-box outputs of words that are zero on entry, (rc[i])^7: [360..364) words 8..11 (the
     // capacity of two_to_one), [364..367) words 1..3 and [367..370) words 5..7 (two_to_one of two leaf digests [leaf, 0, 0, 0])
    // behind them the tables of the sparse partial rounds (poseidon_fast.hip.h kSp*: FIRST, K, V, then W_HAT and INIT as limbs of
    // 22 + 22 + 20 bits in four u32 per constant)
    // and the tables of the batched partial rounds (poseidon_fast.hip.h kP3Tab / kP3K): M^3, row 0 of M^2, M m0 as u32; per
    // group of three rounds the constants c1[0], (M c1 + c2)[0] and M^2 c1 + M c2 + c3 (mod p)
    // and, last, for two_to_one of two leaf digests the share of the ten constant first-round S-box outputs in round 0's MDS layer,
    // with round 1's constants in (poseidon_fast.hip.h kLeafPairK0): 12 words
    // and the tables of the four-round groups (poseidon_fast.hip.h kPGTab): per group of G rounds G + 11 rows of 16 u32 and 16 u64 addends
    constexpr int kPGTab = 1372 + 84 * 7 + 14 * 7 + 98 + 12, kPG4 = 8 * (4 + 11) + 16, kPG3 = 8 * (3 + 11) + 16;
    static_assert(kPGTab % 8 == 0, "64-byte rows");
    static uint64_t table[kPGTab + 5 * kPG4 + 1 * kPG3];
    static_assert(kPGTab == 646 + 22 * 11 * 2 + 121 * 2 + 84 * 7 + 98 + 98 + 12, "layout of poseidon_fast.hip.h");
    memcpy(table, POSEIDON_RC, sizeof(POSEIDON_RC));
    memcpy(table + 370, POSEIDON_FAST_FIRST, sizeof(POSEIDON_FAST_FIRST));
    memcpy(table + 382, POSEIDON_FAST_K, sizeof(POSEIDON_FAST_K));
    memcpy(table + 404, POSEIDON_FAST_V, sizeof(POSEIDON_FAST_V));
    {
      uint32_t* w = reinterpret_cast<uint32_t*>(table + 646);
      auto limbs = [](uint64_t c, uint32_t* out) {
        out[0] = (uint32_t)(c & 0x3FFFFF), out[1] = (uint32_t)((c >> 22) & 0x3FFFFF), out[2] = (uint32_t)(c >> 44), out[3] = 0;
      };
      for (int i = 0; i < 242; ++i) limbs(POSEIDON_FAST_W_HAT[i], w + 4 * i);
      for (int i = 0; i < 121; ++i) limbs(POSEIDON_FAST_INIT[i], w + 4 * (242 + i));
    }
    {
      typedef unsigned __int128 u128;
      const u128 p = 0xFFFFFFFF00000001ULL;
      const uint64_t circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
      uint64_t M[12][12], M2[12][12], M3[12][12];
      for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) M[r][c] = circ[((c - r) % 12 + 12) % 12] + ((r == 0 && c == 0) ? 8 : 0);
      for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) {
          uint64_t a = 0;
          for (int k = 0; k < 12; ++k) a += M[r][k] * M[k][c];
          M2[r][c] = a;
        }
      for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) {
          uint64_t a = 0;
          for (int k = 0; k < 12; ++k) a += M2[r][k] * M[k][c];
          M3[r][c] = a;  // < 2^25: every term of a row is one 32 x 32 multiply
        }
      for (int g = 0; g < 7; ++g) {  // one copy per group (partial_rounds3 says why)
        uint32_t* t = reinterpret_cast<uint32_t*>(table + 1372 + 84 * g);
        for (int r = 0; r < 12; ++r)
          for (int c = 0; c < 12; ++c) t[12 * r + c] = (uint32_t)M3[r][c];
        for (int c = 0; c < 12; ++c) t[144 + c] = (uint32_t)M2[0][c];
        for (int r = 0; r < 12; ++r) t[156 + r] = (uint32_t)M2[r][0];  // (M m0)[r] = sum_k M[r][k] M[k][0] = M^2[r][0]
      }
      auto matvec = [&](const uint64_t (&A)[12][12], const uint64_t* v, uint64_t* out) {
        for (int r = 0; r < 12; ++r) {
          u128 a = 0;
          for (int c = 0; c < 12; ++c) a = (a + (u128)A[r][c] * v[c]) % p;
          out[r] = (uint64_t)a;
        }
      };
      {  // per-lane rows of the 12-lane layout (kP3W): 12 matrix entries, then the coefficients of d1 and d2
        uint32_t* w = reinterpret_cast<uint32_t*>(table + 1372 + 84 * 7 + 14 * 7);
        for (int L = 0; L < 14; ++L)
          for (int c = 0; c < 14; ++c) {
            uint64_t v;
            if (c < 12) v = L == 0 ? M[0][c] : (L == 1 ? M2[0][c] : M3[L - 2][c]);
            else if (c == 12) v = L == 0 ? 0 : (L == 1 ? M[0][0] : M2[L - 2][0]);  // d1: m0[0] for v2[0], (M m0)[r] for v3[r]
            else v = L < 2 ? 0 : M[L - 2][0];                                       // d2: m0[r] for v3[r]
            w[14 * L + c] = (uint32_t)v;
          }
      }
      for (int g = 0; g < 7; ++g) {
        const int r0 = 4 + 3 * g;  // the group's first round; c1, c2, c3 = the constants of rounds r0+1, r0+2, r0+3
        const uint64_t *c1 = POSEIDON_RC + 12 * (r0 + 1), *c2 = POSEIDON_RC + 12 * (r0 + 2), *c3 = POSEIDON_RC + 12 * (r0 + 3);
        uint64_t mc1[12], m2c1[12], mc2[12];
        matvec(M, c1, mc1);
        matvec(M2, c1, m2c1);
        matvec(M, c2, mc2);
        uint64_t* k = table + 1372 + 84 * 7 + 14 * g;
        k[0] = c1[0];
        k[1] = (uint64_t)(((u128)mc1[0] + c2[0]) % p);
        for (int r = 0; r < 12; ++r) k[2 + r] = (uint64_t)(((u128)m2c1[r] + mc2[r] + c3[r]) % p);
      }
    }
    {  // partial_rounds_g: groups of G rounds starting at round r0 (the rounds are numbered 0..29; 4..25 are the partial ones)
      typedef unsigned __int128 u128;
      const u128 p = 0xFFFFFFFF00000001ULL;
      const uint64_t circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
      uint64_t Mp[5][12][12];  // Mp[k] = M^k (integers: M^4 has 29-bit entries)
      for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) {
          Mp[0][r][c] = r == c;
          Mp[1][r][c] = circ[((c - r) % 12 + 12) % 12] + ((r == 0 && c == 0) ? 8 : 0);
        }
      for (int k = 2; k <= 4; ++k)
        for (int r = 0; r < 12; ++r)
          for (int c = 0; c < 12; ++c) {
            uint64_t a = 0;
            for (int j = 0; j < 12; ++j) a += Mp[k - 1][r][j] * Mp[1][j][c];
            Mp[k][r][c] = a;
          }
      {  // what the overflow handling of the groups of four rests on: entries below 2^32, and any eleven of a row's twelve terms plus
         // the d terms and the addend stay below 2^64 -- only the last link of a chain can carry out
        const u128 X = 0xFFFFFFFFu;
        for (int r = 0; r < 12; ++r) {
          u128 tot = X;  // the addend's half
          uint64_t mn = ~0ull;
          for (int c = 0; c < 12; ++c) {
            if (Mp[4][r][c] >> 32) return p2mt::fail(P2MT_EINVAL, "p2mt_init: M^4 entry does not fit 32 bits");
            tot += (u128)Mp[4][r][c] * X;
            mn = Mp[4][r][c] < mn ? Mp[4][r][c] : mn;
          }
          tot += (u128)(Mp[3][r][0] + Mp[2][r][0] + Mp[1][r][0]) * X;
          if ((tot - (u128)mn * X) >> 64) return p2mt::fail(P2MT_EINVAL, "p2mt_init: a chain of the four-round groups can overflow early");
        }
      }
      auto build_group = [&](uint64_t* g, int G, int r0) {
        uint32_t* T = reinterpret_cast<uint32_t*>(g);
        memset(g, 0, 8 * (size_t)(8 * (G + 11) + 16));
        for (int i = 1; i < G; ++i) {  // row i - 1: S-box input of the group's round i = row 0 of M^i y + sum_k d_k M^(i-k)[0][0] + ...
          for (int c = 0; c < 12; ++c) T[16 * (i - 1) + c] = (uint32_t)Mp[i][0][c];
          for (int k = 1; k <= i - 2; ++k) T[16 * (i - 1) + 12 + (k - 1)] = (uint32_t)Mp[i - k][0][0];  // (d_(i-1) m0[0] is an immediate)
        }
        for (int r = 0; r < 12; ++r) {
          for (int c = 0; c < 12; ++c) T[16 * (G - 1 + r) + c] = (uint32_t)Mp[G][r][c];
          for (int k = 1; k <= G - 2; ++k) T[16 * (G - 1 + r) + 12 + (k - 1)] = (uint32_t)Mp[G - k][r][0];
        }
        uint64_t* K = g + 8 * (G + 11);
        auto addend = [&](int i, int r) -> uint64_t {  // (sum_{t=1..i} M^(i-t) c_t)[r] mod p, c_t = the constants of round r0 + t
          u128 a = 0;
          for (int t = 1; t <= i; ++t)
            for (int c = 0; c < 12; ++c) a = (a + (u128)Mp[i - t][r][c] * POSEIDON_RC[12 * (r0 + t) + c]) % p;
          return (uint64_t)a;
        };
        for (int i = 1; i < G; ++i) K[i - 1] = addend(i, 0);
        for (int r = 0; r < 12; ++r) K[G - 1 + r] = addend(G, r);
      };
      for (int g = 0; g < 5; ++g) build_group(table + kPGTab + kPG4 * g, 4, 3 + 4 * g);  // MDS layers of rounds 3-6, 7-10, .., 19-22
      build_group(table + kPGTab + 5 * kPG4, 3, 23);                                      // 23-25
    }
    const int word_of[10] = {8, 9, 10, 11, 1, 2, 3, 5, 6, 7};
    for (int i = 0; i < 10; ++i) {
      const unsigned __int128 p = 0xFFFFFFFF00000001ULL;
      unsigned __int128 x = POSEIDON_RC[word_of[i]], x2 = x * x % p, x4 = x2 * x2 % p, x3 = x2 * x % p;
      table[360 + i] = (uint64_t)(x4 * x3 % p);
    }
    {  // kLeafPairK0[r] = sum_{k not in {0, 4}} MDS[r][k] (rc[k])^7 + rc[12 + r]  (mod p)
      typedef unsigned __int128 u128;
      const u128 p = 0xFFFFFFFF00000001ULL;
      const uint64_t circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
      uint64_t y[12] = {0};
      for (int i = 0; i < 10; ++i) y[word_of[i]] = table[360 + i];
      for (int r = 0; r < 12; ++r) {
        u128 a = POSEIDON_RC[12 + r];
        for (int k = 0; k < 12; ++k) {
          if (k == 0 || k == 4) continue;
          const uint64_t m = circ[((k - r) % 12 + 12) % 12] + ((r == 0 && k == 0) ? 8 : 0);
          a = (a + (u128)m * y[k]) % p;
        }
        table[1372 + 84 * 7 + 14 * 7 + 98 + r] = (uint64_t)a;
      }
    }
    P2MT_HIP(hipMalloc((void**)&rt().d_rc, sizeof(table)));
    P2MT_HIP(hipMemcpy(rt().d_rc, table, sizeof(table), hipMemcpyHostToDevice));
  }
  if (const char* e = getenv("P2MT_TILE_LOG")) {
    const int v = atoi(e);
    if (v >= 9 && v <= 11) rt().tile_log = (unsigned)v;
  }
  if (const char* e = getenv("P2MT_QUAD")) rt().use_quad = atoi(e) != 0;
  if (const char* e = getenv("P2MT_THROUGHPUT")) rt().throughput = atoi(e) != 0;
  if (const char* e = getenv("P2MT_LDE12")) rt().use_lde12 = atoi(e);
  if (const char* e = getenv("P2MT_SUBTREE_BLOCK")) {
    const int v = atoi(e);
    if (v == 64 || v == 128 || v == 256) rt().subtree_block = (unsigned)v;
  }
  if (const char* e = getenv("P2MT_SUBTREE")) {
    const int v = atoi(e);
    rt().subtree_levels = (v >= 2 && v <= 5) ? (unsigned)v : 0;
    rt().subtree_auto = false;
  }
  if ((rt().subtree_levels == 2 || rt().subtree_levels == 3) && rt().subtree_block != 256) {
    rt().subtree_block = 256;  // the small subtrees are instantiated for 256-lane workgroups only
  }
  rt().initialised = true;
  return P2MT_OK;
  });
}

extern "C" int p2mt_thread_stream_create(void) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  P2MT_HIP(hipSetDevice(rt().device));  // the current device is per thread too
  hipStream_t s = nullptr;
  P2MT_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  rt().stream = s;
  return P2MT_OK;
  });
}

extern "C" int p2mt_thread_stream_destroy(void) {
  return p2mt::abi_guard([&]() -> int {
  hipStream_t s = rt().stream;
  p2mt::scratch_release_thread();
  if (s) {
    P2MT_HIP(hipStreamSynchronize(s));
    P2MT_HIP(hipStreamDestroy(s));
  }
  rt().stream = nullptr;
  return P2MT_OK;
  });
}

extern "C" int p2mt_set_throughput_mode(int on) {
  return p2mt::abi_guard([&]() -> int {
  rt().throughput = on != 0;
  return P2MT_OK;
  });
}

extern "C" int p2mt_set_stream(void* hip_stream) {
  return p2mt::abi_guard([&]() -> int {
  rt().stream = static_cast<hipStream_t>(hip_stream);
  return P2MT_OK;
  });
}

extern "C" int p2mt_get_stream(void** hip_stream_out) {
  return p2mt::abi_guard([&]() -> int {
  if (!hip_stream_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  *hip_stream_out = static_cast<void*>(rt().stream);
  return P2MT_OK;
  });
}

extern "C" int p2mt_sync(void) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

extern "C" const char* p2mt_last_error(void) { return p2mt::err_buf(); }

extern "C" int p2mt_set_variant(int mds, int partial) {
  return p2mt::abi_guard([&]() -> int {
  // partial: 0 dense, 1 sparse partial rounds; 2 / 3 (with mds == 2 only) = dense with all / only the partial rounds' MDS layers on the matrix pipe, 4 = one MDS
  // layer per partial round (round 2's form; the default batches four partial rounds per MDS application, 7 = three) -- stage-1 A/B
  if (mds < 0 || mds > 2 || partial < 0 || partial > (mds == 2 ? 8 : 1)) return p2mt::fail(P2MT_EINVAL, "variant out of range");
  rt().mds = mds;
  rt().partial = partial;
  return P2MT_OK;
  });
}

extern "C" int p2mt_debug_force_fallback(int on) {
  return p2mt::abi_guard([&]() -> int {
  rt().force_fallback = on ? 1 : 0;
  return P2MT_OK;
  });
}

extern "C" int p2mt_get_variant(int* mds, int* partial) {
  return p2mt::abi_guard([&]() -> int {
  if (mds) *mds = rt().mds;
  if (partial) *partial = rt().partial;
  return P2MT_OK;
  });
}

extern "C" int p2mt_mmr_stage1_levels(size_t n_leaves) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());  // a negative status, never a level count, when the library cannot initialise
  if (rt().mds != 2) return (int)rt().tile_log - 6;
  const unsigned lv = p2mt::subtree_levels_for(n_leaves);
  return lv ? (int)lv : (int)rt().tile_log - 6;
  });
}

extern "C" int p2mt_get_build_config(int* subtree_levels, int* tile_log, int* subtree_block) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());  // the knobs are read from the environment at init
  if (subtree_levels) *subtree_levels = rt().mds == 2 ? (int)rt().subtree_levels : 0;
  if (tile_log) *tile_log = rt().mds == 2 ? (int)rt().tile_log : 11;
  if (subtree_block) *subtree_block = (int)rt().subtree_block;
  return P2MT_OK;
  });
}

extern "C" int p2mt_profile_enable(int on) {
  return p2mt::abi_guard([&]() -> int {
  rt().profile = on != 0;
  rt().prof_n = 0;
  return P2MT_OK;
  });
}

// Sum and count of the HIP-event durations recorded around the dominant kernel launches since the last
// p2mt_profile_enable(1); synchronises the stream.
extern "C" int p2mt_profile_read(float* total_ms, int* launches) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  float sum = 0;
  for (int i = 0; i < rt().prof_n; ++i) {
    float ms = 0;
    P2MT_HIP(hipEventElapsedTime(&ms, rt().prof_ev[2 * i], rt().prof_ev[2 * i + 1]));
    sum += ms;
  }
  if (total_ms) *total_ms = sum;
  if (launches) *launches = rt().prof_n;
  rt().prof_n = 0;
  return P2MT_OK;
  });
}

extern "C" int p2mt_timer_start(void) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  P2MT_HIP(hipEventRecord(rt().ev_start, rt().stream));
  return P2MT_OK;
  });
}

extern "C" int p2mt_timer_stop(float* elapsed_ms) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  P2MT_HIP(hipEventRecord(rt().ev_stop, rt().stream));
  P2MT_HIP(hipEventSynchronize(rt().ev_stop));
  float ms = 0;
  P2MT_HIP(hipEventElapsedTime(&ms, rt().ev_start, rt().ev_stop));
  if (elapsed_ms) *elapsed_ms = ms;
  return P2MT_OK;
  });
}

// ---------------------------------------------------------------------------------------------------- host link (runtime.h)
namespace p2mt {
struct HostLink {
  uint64_t* h_buf = nullptr;  // mapped pinned, coherent: [max_words data | sequence word]
  size_t max_words = 0;
  uint32_t seq = 0;
};
namespace {
typedef __attribute__((address_space(1))) uint32_t hl_gu32;
__global__ __launch_bounds__(256) void k_hostlink_publish(const uint64_t* __restrict__ src0, uint32_t n0, const uint64_t* __restrict__ src1,
                                                          uint32_t n1, uint64_t* __restrict__ h_dst, uint32_t* __restrict__ h_seq, uint32_t seq) {
  for (uint32_t k = threadIdx.x; k < n0 + n1; k += 256) h_dst[k] = k < n0 ? src0[k] : src1[k - n0];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct PutVals {
  uint64_t v[32];
};
__global__ __launch_bounds__(64) void k_hostlink_put(PutVals pv, uint32_t n, uint64_t* __restrict__ dst) {
  if (threadIdx.x < n) dst[threadIdx.x] = pv.v[threadIdx.x];
}
}  // namespace

int hostlink_create(HostLink** out, size_t max_words) {
  HostLink* l = new (std::nothrow) HostLink();
  if (!l) return fail(P2MT_ENOMEM, "host link: out of host memory");
  if (hipHostMalloc((void**)&l->h_buf, (max_words + 2) * 8, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    (void)hipGetLastError();
    delete l;
    return fail(P2MT_ENOMEM, "host link: hipHostMalloc failed");
  }
  memset(l->h_buf, 0, (max_words + 2) * 8);
  l->max_words = max_words;
  *out = l;
  return P2MT_OK;
}
void hostlink_destroy(HostLink* l) {
  if (!l) return;
  if (l->h_buf) (void)hipHostFree(l->h_buf);
  delete l;
}
int hostlink_fetch2(HostLink* l, const uint64_t* d_src0, size_t n0, const uint64_t* d_src1, size_t n1, const uint64_t** h_out) {
  if (!l || n0 + n1 > l->max_words) return fail(P2MT_EINVAL, "host link: fetch larger than the link's buffer");
  const uint32_t seq = ++l->seq;
  uint32_t* h_seq = reinterpret_cast<uint32_t*>(l->h_buf + l->max_words);
  hipLaunchKernelGGL(k_hostlink_publish, dim3(1), dim3(256), 0, rt().stream, d_src0, (uint32_t)n0, d_src1, (uint32_t)n1, l->h_buf, h_seq, seq);
  P2MT_LAUNCH_CHECK();
  // spin on the sequence word; every ~50 us ask the runtime whether the stream died (a failed launch would never raise it)
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spins = 0;; ++spins) {
    if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) == seq) break;
    __builtin_ia32_pause();
    if ((spins & 0xFFFF) == 0xFFFF) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((t1.tv_sec - t0.tv_sec) > 20) return fail(P2MT_EHIP, "host link: the device never delivered (20 s)");
      const hipError_t e = hipStreamQuery(rt().stream);
      if (e != hipSuccess && e != hipErrorNotReady) return fail_hip(e, "hipStreamQuery(host link)", __FILE__, __LINE__);
      if (e == hipSuccess && __atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != seq)
        return fail(P2MT_EHIP, "host link: the stream drained without delivering");
    }
  }
  *h_out = l->h_buf;
  return P2MT_OK;
}
int hostlink_fetch(HostLink* l, const uint64_t* d_src, size_t n, const uint64_t** h_out) {
  return hostlink_fetch2(l, d_src, n, nullptr, 0, h_out);
}
int hostlink_put(const uint64_t* vals, size_t n, uint64_t* d_dst) {
  if (n > 32 || !d_dst) return fail(P2MT_EINVAL, "host link: put of more than 32 words");
  if (n == 0) return P2MT_OK;
  PutVals pv;
  memcpy(pv.v, vals, n * 8);
  hipLaunchKernelGGL(k_hostlink_put, dim3(1), dim3(64), 0, rt().stream, pv, (uint32_t)n, d_dst);
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}
}  // namespace p2mt
