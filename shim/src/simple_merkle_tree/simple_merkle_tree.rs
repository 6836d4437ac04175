//! `MerkleTree` with the reference's fields and signatures (/root/reference/src/simple_merkle_tree/simple_merkle_tree.rs:11-109);
//! `build` is ONE call into the library (p2mt_merkle_build_pow2: all levels on the GPU), the getters are the reference's own
//! index arithmetic on `self.tree`, `verify_merkle_proof` folds on the GPU (p2mt_verify_merkle_proof_batch with m = 1).
use crate::{canonical, ffi, hash_from, hash_words, ok};
use plonky2::field::goldilocks_field::GoldilocksField;
use plonky2::field::types::PrimeField64;
use plonky2::hash::hash_types::HashOut;

#[derive(Debug, Clone)]
pub struct MerkleTree {
    pub count_levels: usize,
    pub tree: Vec<Vec<HashOut<GoldilocksField>>>, // levels 0 .. count_levels-1, as in the reference (:13)
    pub root: HashOut<GoldilocksField>,
}

impl MerkleTree {
    /// :28-51.  Panics (status -1) unless `leaves.len()` is a power of two >= 2, as `log2_strict` (:30) / the underflow at :38 do.
    pub fn build(leaves: Vec<GoldilocksField>) -> Self {
        let n = leaves.len();
        let words = canonical(&leaves);
        let mut levels = vec![0u64; 4 * (2 * n).saturating_sub(2).max(1)];
        let mut root = [0u64; 4];
        ok(unsafe { ffi::p2mt_merkle_build_pow2(words.as_ptr(), n, levels.as_mut_ptr(), root.as_mut_ptr()) });
        let count_levels = n.trailing_zeros() as usize;
        let mut tree = Vec::with_capacity(count_levels);
        let mut off = 0usize;
        for i in 0..count_levels {
            let len = n >> i;
            tree.push((0..len).map(|j| {
                let w = &levels[4 * (off + j)..4 * (off + j) + 4];
                hash_from([w[0], w[1], w[2], w[3]])
            }).collect());
            off += len;
        }
        MerkleTree { count_levels, tree, root: hash_from(root) }
    }

    /// :55-74 (consumes `self`, like the reference).
    pub fn get_merkle_proof(self, leaf_index: usize) -> Vec<HashOut<GoldilocksField>> {
        assert!(leaf_index < self.tree[0].len());
        let mut idx = leaf_index;
        (0..self.count_levels).map(|i| {
            let h = self.tree[i][idx ^ 1];
            idx /= 2;
            h
        }).collect()
    }

    /// :76-86
    pub fn get_in_between_hashes(self, leaf_index: usize) -> Vec<HashOut<GoldilocksField>> {
        assert!(leaf_index < self.tree[0].len());
        let mut index = leaf_index / 2;
        let mut hashes = Vec::new();
        for i in 1..self.count_levels {
            hashes.push(self.tree[i][index]);
            index /= 2;
        }
        hashes.push(self.root);
        hashes
    }
}

/// :91-109
pub fn verify_merkle_proof(leaf: GoldilocksField, leaf_index: usize, root: HashOut<GoldilocksField>,
                           hashes: Vec<HashOut<GoldilocksField>>) -> bool {
    let leaf_w = [leaf.to_canonical_u64()];
    let idx = [leaf_index as u64];
    let root_w = hash_words(&root);
    let path: Vec<u64> = hashes.iter().flat_map(hash_words).collect();
    let mut result = [0u8; 1];
    ok(unsafe {
        ffi::p2mt_verify_merkle_proof_batch(leaf_w.as_ptr(), idx.as_ptr(), root_w.as_ptr(), path.as_ptr(), hashes.len(), 1,
                                            result.as_mut_ptr())
    });
    result[0] != 0
}
