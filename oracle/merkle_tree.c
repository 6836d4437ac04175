/* oracle/merkle_tree.c -- TEST INFRASTRUCTURE.
 * Line-by-line restatement of /root/reference/src/simple_merkle_tree/simple_merkle_tree.rs:18-109. */
#include <string.h>
#include "oracle.h"

static int log2_strict(size_t n) { /* plonky2_util::log2_strict: panics unless n is a power of two */
  if (n == 0 || (n & (n - 1))) return -1;
  int k = 0;
  while ((((size_t)1) << k) < n) ++k;
  return k;
}

/* MerkleTree::build (simple_merkle_tree.rs:28-51). tree = levels 0..k-1, root separate. */
int oracle_merkle_build(const uint64_t *leaves, size_t n, uint64_t *levels_out, uint64_t root[4]) {
  int count_levels = log2_strict(n); /* :30 panics if not a power of 2 */
  if (count_levels < 1) return -1;   /* n == 1: `0..(0-1)` usize underflow panic at :38 (Quirk Q6) */
  /* :33 level0 = hash_or_noop(&[leaf]) */
  for (size_t i = 0; i < n; ++i) oracle_hash_or_noop(&leaves[i], 1, &levels_out[4 * i]);
  /* :38-41 next_level_hashes (:21-25): chunks(2) -> two_to_one */
  uint64_t *cur = levels_out;
  size_t cur_n = n;
  for (int i = 0; i < count_levels - 1; ++i) {
    uint64_t *next = cur + 4 * cur_n;
    for (size_t j = 0; j < cur_n / 2; ++j) oracle_two_to_one(&cur[8 * j], &cur[8 * j + 4], &next[4 * j]);
    cur = next;
    cur_n /= 2;
  }
  /* :44-45 root = two_to_one(last[0], last[1]) */
  oracle_two_to_one(&cur[0], &cur[4], root);
  return count_levels;
}

static const uint64_t *level_ptr(const uint64_t *levels, size_t n, int level) {
  const uint64_t *p = levels;
  for (int i = 0; i < level; ++i) p += 4 * (n >> i);
  return p;
}

/* get_merkle_proof (:55-74): sibling per level, bottom-up; count_levels hashes */
int oracle_merkle_get_proof(const uint64_t *levels, size_t n, size_t leaf_index, uint64_t *proof_out) {
  int count_levels = log2_strict(n);
  if (count_levels < 1 || leaf_index >= n) return -1; /* assert! :56 */
  size_t updated_index = leaf_index;
  for (int i = 0; i < count_levels; ++i) {
    const uint64_t *level_i = level_ptr(levels, n, i);
    size_t sel = (updated_index & 1) ? updated_index - 1 : updated_index + 1; /* :64-68 */
    memcpy(&proof_out[4 * i], &level_i[4 * sel], 32);
    updated_index /= 2;
  }
  return count_levels;
}

/* get_in_between_hashes (:76-86): ancestors of the leaf at levels 1..k-1, then the root */
int oracle_merkle_get_in_between_hashes(const uint64_t *levels, const uint64_t root[4], size_t n,
                                        size_t leaf_index, uint64_t *out) {
  int count_levels = log2_strict(n);
  if (count_levels < 1 || leaf_index >= n) return -1; /* assert! :77 */
  size_t index = leaf_index / 2;
  int k = 0;
  for (int i = 1; i < count_levels; ++i) {
    memcpy(&out[4 * k++], &level_ptr(levels, n, i)[4 * index], 32);
    index /= 2;
  }
  memcpy(&out[4 * k++], root, 32);
  return k;
}

/* verify_merkle_proof (:91-109) */
int oracle_verify_merkle_proof(uint64_t leaf, size_t leaf_index, const uint64_t root[4],
                               const uint64_t *hashes, size_t n_hashes) {
  uint64_t next_hash[4], tmp[4];
  oracle_hash_or_noop(&leaf, 1, next_hash); /* :93 */
  size_t updated_index = leaf_index;
  for (size_t i = 0; i < n_hashes; ++i) {
    if ((updated_index & 1) == 0) oracle_two_to_one(next_hash, &hashes[4 * i], tmp); /* :100 */
    else oracle_two_to_one(&hashes[4 * i], next_hash, tmp);                           /* :102 */
    memcpy(next_hash, tmp, 32);
    updated_index /= 2;
  }
  return memcmp(next_hash, root, 32) == 0; /* :108 */
}
