// host_poseidon.h -- the Poseidon permutation and plonky2's Challenger on a HOST core (product code, not the oracle).
//
// Why the product has a CPU permutation at all: a transcript is a chain of dependent permutations.  One wavefront of the GPU does one
// in 6.9 us (tree_common.hip.h permute_wave: the 12-lane layout at the dependent-issue rate of a lone wave), a host core in
// 0.9-1.4 us -- and plonky2 itself keeps the Challenger on the host (plonky2 @3b21b87 iop/challenger.rs, reached from
// /root/reference/src/mmr/mmr_plonky2_verifier.rs:148-150 and mmr_plonky2_verifier_1_recursion.rs:217-220).  For a single
// verification every challenge is a function of proof words the host already holds; for a single prove the host needs one 512-byte
// cap per phase.  The batched prover keeps the device transcript (lane-parallel there).  Self-contained: the tables of poseidon_constants.h only.
//
// Same function as poseidon.hip.h (width 12, x^7, 4 + 22 + 4 rounds, circulant MDS + diagonal, constants of poseidon_constants.h),
// bit-identical: tests/test_host_transcript.py (CPU, against the oracle) and tests/test_circuit_gpu.py (against the device).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace host_poseidon {

typedef uint64_t u64;

// the permutation, in place; input any u64 words, output canonical.  (host_poseidon.hip: scalar code in two spellings of its
// corrections, and -- where the CPU has AVX-512 -- the eight full rounds with the state across the lanes of two zmm registers; the
// candidates are timed once, on a dependent chain, on the machine that runs them: 1.35 us scalar / 0.86 us with AVX-512 on an EPYC 9575F)
void permute(u64 (&s)[12]);
// hash_n_to_hash_no_pad (overwrite-mode sponge, rate 8): out = state[0..4) after absorbing `n` elements
void hash_no_pad(const u64* in, size_t n, u64 (&out)[4]);

// plonky2's Challenger (iop/challenger.rs): duplex sponge with an input buffer (absorbed 8 at a time, overwriting) and an output
// buffer popped from the back.  Mirrors k_challenger (p2mt_fri.hip) word for word.
struct Challenger {
  u64 state[12] = {0};
  u64 in[8] = {0}, out[8] = {0};
  unsigned n_in = 0, n_out = 0;
  void duplex();
  void observe(u64 x);
  void observe(const u64* x, size_t n) {
    for (size_t i = 0; i < n; ++i) observe(x[i]);
  }
  u64 squeeze();
  void squeeze(u64* dst, size_t n) {
    for (size_t i = 0; i < n; ++i) dst[i] = squeeze();
  }
};

}  // namespace host_poseidon
