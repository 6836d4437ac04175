//! The reference crate's module layout (/root/reference/src/lib.rs:1-2) over the MI355X library:
//! `simple_merkle_tree::simple_merkle_tree::{MerkleTree, verify_merkle_proof}` and
//! `mmr::merkle_mountain_ranges::{MMR, MMR_proof, get_mmr_index, get_heights_bitmap_for_mmr_size}` keep their names and
//! signatures; every hash runs in libp2mt_hip.so.  A caller switches by changing the crate name in its `use` lines; the two
//! call-site edits that remain are listed in INTEGRATION.md (the `elements` field, by-reference getters).
//! The circuit modules of the reference (`mmr::{common, mmr_plonky2_verifier, mmr_plonky2_verifier_1_recursion}`) are NOT re-hosted
//! here: they compile unchanged against `shim/plonky2`.
pub use p2mt_sys as ffi;

pub mod simple_merkle_tree {
    pub mod simple_merkle_tree;
}
pub mod mmr {
    pub mod merkle_mountain_ranges;
}

use plonky2::field::goldilocks_field::GoldilocksField;
use plonky2::field::types::{Field, PrimeField64};
use plonky2::hash::hash_types::HashOut;

pub(crate) use p2mt_sys::ok;

/// One-time device selection (p2mt_init(0) happens implicitly on first use; call this to pick another GPU).
pub fn init(device: i32) {
    ok(unsafe { ffi::p2mt_init(device) })
}

pub(crate) fn hash_from(words: [u64; 4]) -> HashOut<GoldilocksField> {
    HashOut { elements: words.map(GoldilocksField::from_canonical_u64) }
}

pub(crate) fn hash_words(h: &HashOut<GoldilocksField>) -> [u64; 4] {
    h.elements.map(|e| e.to_canonical_u64())
}

/// Vec<GoldilocksField> -> canonical words (GoldilocksField may hold non-canonical values internally).
pub(crate) fn canonical(leaves: &[GoldilocksField]) -> Vec<u64> {
    leaves.iter().map(|l| l.to_canonical_u64()).collect()
}
