// p2mt_commit.hip -- placeholder until the NTT / LDE / Merkle-cap kernels land (next commit).
#include "runtime.h"
#define NOT_YET return p2mt::fail(P2MT_EINVAL, "commit kernels not built yet")
extern "C" int p2mt_ntt_batch(uint64_t*, unsigned, size_t, int) { NOT_YET; }
extern "C" int p2mt_ntt_batch_dev(uint64_t*, unsigned, size_t, int) { NOT_YET; }
extern "C" int p2mt_coset_lde_batch(const uint64_t*, unsigned, unsigned, uint64_t, size_t, uint64_t*) { NOT_YET; }
extern "C" int p2mt_coset_lde_batch_dev(const uint64_t*, unsigned, unsigned, uint64_t, size_t, uint64_t*) { NOT_YET; }
extern "C" int p2mt_merkle_cap_commit(const uint64_t*, size_t, size_t, unsigned, uint64_t*, uint64_t*) { NOT_YET; }
extern "C" int p2mt_merkle_cap_commit_dev(const uint64_t*, size_t, size_t, unsigned, uint64_t*, uint64_t*) { NOT_YET; }
extern "C" int p2mt_polynomial_batch_commit(const uint64_t*, int, size_t, unsigned, unsigned, unsigned, uint64_t*, uint64_t*, uint64_t*) { NOT_YET; }
extern "C" int p2mt_polynomial_batch_commit_dev(const uint64_t*, int, size_t, unsigned, unsigned, unsigned, uint64_t*, uint64_t*, uint64_t*) { NOT_YET; }
