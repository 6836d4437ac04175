/* oracle/mmr.c -- TEST INFRASTRUCTURE.
 * Line-by-line restatement of /root/reference/src/mmr/merkle_mountain_ranges.rs:39-270
 * (array-only MMR in post-order), quirks included (Q1, Q2, Q5, Q6). */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

struct oracle_mmr { /* struct MMR { elements: Vec<HashOut> }  (:8-12) */
  uint64_t *elements;
  size_t len, cap;
};

static int leading_zeros_usize(size_t x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }

/* get_heights_bitmap_for_mmr_size (:39-81) */
uint64_t oracle_get_heights_bitmap_for_mmr_size(size_t mmr_size, size_t *remainder) {
  if (mmr_size == 0) { /* :40-42 */
    if (remainder) *remainder = 0;
    return 0;
  }
  size_t all_peaks_set = (~(size_t)0) >> leading_zeros_usize(mmr_size); /* :44 */
  size_t subtree_size = all_peaks_set;                                  /* :63 */
  size_t updated_mmr_size = mmr_size;
  uint64_t peaks = 0;
  while (subtree_size > 0) { /* :69-79 */
    peaks <<= 1;
    if (updated_mmr_size >= subtree_size) {
      peaks |= 1;
      updated_mmr_size -= subtree_size;
    }
    subtree_size >>= 1;
  }
  if (remainder) *remainder = updated_mmr_size;
  return peaks;
}

/* get_mmr_index (:257-270): sum over set bits i of (2^(i+1) - 1); i32 arithmetic => n < 2^30 */
size_t oracle_get_mmr_index(size_t leaf_normal_index) {
  size_t index = leaf_normal_index;
  unsigned height = 1;
  int64_t res = 0;
  while (index > 0) {
    if (index & 1) {
      if (height >= 31) return (size_t)-1; /* 2i32.pow(31) overflows: the reference panics */
      res += ((int64_t)1 << height) - 1;
      if (res > INT32_MAX) return (size_t)-1;
    }
    height += 1;
    index >>= 1;
  }
  return (size_t)res;
}

oracle_mmr *oracle_mmr_new(void) { return (oracle_mmr *)calloc(1, sizeof(oracle_mmr)); } /* :84-86 */

void oracle_mmr_free(oracle_mmr *m) {
  if (!m) return;
  free(m->elements);
  free(m);
}

static void push(oracle_mmr *m, const uint64_t h[4]) {
  if (m->len == m->cap) {
    m->cap = m->cap ? m->cap * 2 : 64;
    m->elements = (uint64_t *)realloc(m->elements, m->cap * 32);
  }
  memcpy(&m->elements[4 * m->len], h, 32);
  m->len += 1;
}

/* MMR::add_leaf (:89-120) */
void oracle_mmr_add_leaf(oracle_mmr *m, uint64_t leaf) {
  uint64_t next_hash[4], tmp[4];
  if (m->len == 0) { /* :90-93 */
    oracle_hash_or_noop(&leaf, 1, next_hash);
    push(m, next_hash);
    return;
  }
  oracle_hash_or_noop(&leaf, 1, next_hash); /* :96 */
  size_t pos;
  uint64_t peaks = oracle_get_heights_bitmap_for_mmr_size(m->len, &pos); /* :102 */
  size_t current_pos = m->len;
  push(m, next_hash); /* :104 */
  unsigned height = 1;
  while (peaks > 0) { /* :106-119 */
    if ((peaks & 1) == 1) {
      size_t prev_peak_index = current_pos - ((((size_t)1) << height) - 1); /* :109 */
      oracle_two_to_one(&m->elements[4 * prev_peak_index], next_hash, tmp);    /* :111 */
      memcpy(next_hash, tmp, 32);
      push(m, next_hash);
    } else {
      break;
    }
    peaks >>= 1;
    height += 1;
    current_pos += 1;
  }
}

void oracle_mmr_add_leaves(oracle_mmr *m, const uint64_t *leaves, size_t n) {
  for (size_t i = 0; i < n; ++i) oracle_mmr_add_leaf(m, leaves[i]);
}

/* Level-parallel build of the 2^k-leaf MMR (one perfect tree) into its post-order array: node of height h whose
 * last leaf is L sits at 2L - popcount(L) + h.  Same values as the add_leaf loop (tested). */
#ifdef _OPENMP
#include <omp.h>
#endif
int oracle_mmr_build_pow2_parallel(const uint64_t *leaves, size_t n, uint64_t *el, int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
  threads = omp_get_max_threads();
#else
  threads = 1;
#endif
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)n; ++i) {
    size_t pos = 2 * (size_t)i - (size_t)__builtin_popcountll((unsigned long long)i);
    oracle_hash_or_noop(&leaves[i], 1, &el[4 * pos]);
  }
  for (unsigned h = 1; ((size_t)1 << h) <= n; ++h) {
    const long long cnt = (long long)(n >> h);
#pragma omp parallel for schedule(static)
    for (long long j = 0; j < cnt; ++j) {
      size_t last = (((size_t)j + 1) << h) - 1;
      size_t pos = 2 * last - (size_t)__builtin_popcountll((unsigned long long)last) + h;
      oracle_two_to_one(&el[4 * (pos - ((size_t)1 << h))], &el[4 * (pos - 1)], &el[4 * pos]);
    }
  }
  return threads;
}

size_t oracle_mmr_len(const oracle_mmr *m) { return m->len; }
const uint64_t *oracle_mmr_elements(const oracle_mmr *m) { return m->elements; }

/* MMR::get_peaks (:179-200) */
int oracle_mmr_get_peaks(const oracle_mmr *m, uint64_t *peaks_out) {
  size_t mmr_len = m->len;
  if (mmr_len == 0 || mmr_len > 0xFFFFFFFFULL) return -1; /* to_u32().unwrap() / shift overflow :184 (Q6) */
  size_t max_tree_size = (size_t)(0xFFFFFFFFu >> __builtin_clz((uint32_t)mmr_len));
  size_t current_index = mmr_len;
  size_t peak_pos = 0;
  int n = 0;
  while (max_tree_size > 0) { /* :188-198 */
    if (current_index >= max_tree_size) {
      peak_pos += max_tree_size;
      memcpy(&peaks_out[4 * n++], &m->elements[4 * (peak_pos - 1)], 32);
      current_index -= max_tree_size;
    }
    max_tree_size >>= 1;
  }
  return n;
}

/* MMR::bagging_the_peaks (:122-127) */
int oracle_mmr_bagging_the_peaks(const oracle_mmr *m, uint64_t root[4]) {
  uint64_t peaks[4 * 64];
  int n = oracle_mmr_get_peaks(m, peaks);
  if (n < 0) return -1;
  oracle_hash_or_noop(peaks, (size_t)n * 4, root); /* :125: one peak => no-op copy (Q2) */
  return 0;
}

/* add_right_elm (:129-144) */
static void add_right_elm(size_t curr_index, unsigned height, const oracle_mmr *m, uint64_t *sib,
                          uint8_t *lefts, int *n, size_t *curr_index_mut, int *intree_mut) {
  size_t next_elm_index = curr_index + ((((size_t)1) << (height + 1)) - 1);
  if (next_elm_index < m->len - 1) {
    memcpy(&sib[4 * *n], &m->elements[4 * next_elm_index], 32);
    lefts[*n] = 0;
    *n += 1;
    *curr_index_mut = next_elm_index + 1;
  } else {
    *intree_mut = 0;
  }
}

/* MMR::get_subtree_proof_elm (:147-176) */
int oracle_mmr_get_subtree_proof_elm(const oracle_mmr *m, size_t mmr_index, uint64_t *sib, uint8_t *lefts) {
  if (m->len == 0 || mmr_index >= m->len) return -1;
  int n = 0;
  size_t curr_index = mmr_index;
  int intree = 1;
  unsigned height = 0;
  while (intree) {
    size_t span = (((size_t)1) << (height + 1)) - 1;
    if (curr_index >= span) { /* :157 */
      size_t prev_elm_index = curr_index - span;
      size_t rem;
      oracle_get_heights_bitmap_for_mmr_size(prev_elm_index, &rem);
      if (rem == height) { /* :161 previous element is at the same height => left sibling */
        memcpy(&sib[4 * n], &m->elements[4 * prev_elm_index], 32);
        lefts[n] = 1;
        n += 1;
        curr_index += 1;
      } else {
        add_right_elm(curr_index, height, m, sib, lefts, &n, &curr_index, &intree);
      }
    } else {
      add_right_elm(curr_index, height, m, sib, lefts, &n, &curr_index, &intree);
    }
    height += 1;
  }
  return n;
}

/* MMR::get_proof (:209-223) */
int oracle_mmr_get_proof(const oracle_mmr *m, size_t mmr_index, uint64_t *sib, uint8_t *lefts, int *n_sib,
                         uint64_t *peaks_out, int *n_peaks, size_t *mmr_size) {
  int ns = oracle_mmr_get_subtree_proof_elm(m, mmr_index, sib, lefts);
  if (ns < 0) return -1;
  int np = oracle_mmr_get_peaks(m, peaks_out);
  if (np < 0) return -1;
  *n_sib = ns;
  *n_peaks = np;
  if (mmr_size) *mmr_size = m->len;
  return 0;
}

/* MMR_proof::verify (:232-252) */
int oracle_mmr_proof_verify(const uint64_t *sib, const uint8_t *lefts, int n_sib, const uint64_t *peaks,
                            int n_peaks, uint64_t leaf, const uint64_t root[4]) {
  uint64_t next_hash[4], tmp[4];
  oracle_hash_or_noop(&leaf, 1, next_hash); /* :233 */
  for (int i = 0; i < n_sib; ++i) {         /* :236-242 */
    if (lefts[i]) oracle_two_to_one(&sib[4 * i], next_hash, tmp);
    else oracle_two_to_one(next_hash, &sib[4 * i], tmp);
    memcpy(next_hash, tmp, 32);
  }
  int found = 0; /* :245 assert!(self.peaks.contains(&next_hash)) -- PANICS, does not return false (Q5) */
  for (int i = 0; i < n_peaks; ++i) found |= memcmp(&peaks[4 * i], next_hash, 32) == 0;
  if (!found) return -1;
  uint64_t calc_root[4];
  oracle_hash_or_noop(peaks, (size_t)n_peaks * 4, calc_root); /* :248-249 */
  return memcmp(calc_root, root, 32) == 0;                    /* :251 */
}
