//! `MMR` / `MMR_proof` with the reference's method names and signatures
//! (/root/reference/src/mmr/merkle_mountain_ranges.rs:8-270).  `elements` lives in HBM behind a handle; `add_leaf` is
//! write-combined by the library (queued, flushed as one bulk extend before the MMR is next observed), `extend` /
//! `from_leaves` are the bulk constructors the reference lacks.
#![allow(non_camel_case_types)]
use crate::{canonical, ffi, hash_from, hash_words, ok};
use plonky2::field::goldilocks_field::GoldilocksField;
use plonky2::field::types::PrimeField64;
use plonky2::hash::hash_types::HashOut;

pub struct MMR {
    h: *mut ffi::p2mt_mmr,
}

#[derive(Debug, Clone)]
pub struct MMR_proof {
    pub mmr_size: usize,
    pub merkle_proof: Vec<(HashOut<GoldilocksField>, bool)>,
    pub peaks: Vec<HashOut<GoldilocksField>>,
}

/// :25-28 (declared by the reference, private fields, constructed nowhere: kept so that the type name resolves)
pub struct MMR_extended_proof {
    #[allow(dead_code)]
    mmr_proof: MMR_proof,
    #[allow(dead_code)]
    root_subtree: HashOut<GoldilocksField>,
}

/// :39-81
pub fn get_heights_bitmap_for_mmr_size(mmr_size: usize) -> (u64, usize) {
    let mut rem = 0usize;
    let bitmap = unsafe { ffi::p2mt_get_heights_bitmap_for_mmr_size(mmr_size, &mut rem) };
    (bitmap, rem)
}

/// :257-270 (panics where the reference's i32 arithmetic overflows)
pub fn get_mmr_index(leaf_normal_index: usize) -> usize {
    let r = unsafe { ffi::p2mt_get_mmr_index(leaf_normal_index) };
    assert!(r >= 0, "get_mmr_index: i32 overflow (n >= 2^30)");
    r as usize
}

impl MMR {
    /// :85-87
    pub fn new() -> Self {
        let mut h = std::ptr::null_mut();
        ok(unsafe { ffi::p2mt_mmr_create(&mut h) });
        MMR { h }
    }

    /// bulk constructor (not in the reference): identical `elements` to `for l in leaves { add_leaf(l) }`
    pub fn from_leaves(leaves: &[GoldilocksField]) -> Self {
        let mut m = MMR::new();
        m.extend(leaves);
        m
    }

    /// :89-120
    pub fn add_leaf(&mut self, leaf: GoldilocksField) {
        ok(unsafe { ffi::p2mt_mmr_add_leaf(self.h, leaf.to_canonical_u64()) })
    }

    pub fn extend(&mut self, leaves: &[GoldilocksField]) {
        let words = canonical(leaves);
        ok(unsafe { ffi::p2mt_mmr_extend(self.h, words.as_ptr(), words.len()) })
    }

    /// The reference's PUBLIC FIELD `elements` (:11) is a method here: the array lives in HBM (1 GB at 2^24 leaves) and a field
    /// would need a host mirror kept coherent on every add_leaf.  This is the one source edit a caller makes:
    /// `mmr.elements.len()` -> `mmr.elements().len()` (the reference's own test does this at :339).
    pub fn elements(&self) -> Vec<HashOut<GoldilocksField>> {
        let len = unsafe { ffi::p2mt_mmr_len(self.h) };
        let mut words = vec![0u64; 4 * len];
        ok(unsafe { ffi::p2mt_mmr_copy_elements(self.h, 0, len, words.as_mut_ptr()) });
        words.chunks_exact(4).map(|w| hash_from([w[0], w[1], w[2], w[3]])).collect()
    }

    /// :122-127.  (`&self`: the reference's by-value getters force whole-array clones, quirk Q7; `self.clone().x()` call
    /// sites compile unchanged through `Clone` below.)
    pub fn bagging_the_peaks(&self) -> HashOut<GoldilocksField> {
        let mut r = [0u64; 4];
        ok(unsafe { ffi::p2mt_mmr_root(self.h, r.as_mut_ptr()) });
        hash_from(r)
    }

    /// :179-200
    pub fn get_peaks(&self) -> Vec<HashOut<GoldilocksField>> {
        let mut peaks = [0u64; 4 * ffi::P2MT_MAX_PROOF_LEN];
        let mut n = 0i32;
        ok(unsafe { ffi::p2mt_mmr_peaks(self.h, peaks.as_mut_ptr(), &mut n) });
        peaks[..4 * n as usize].chunks_exact(4).map(|w| hash_from([w[0], w[1], w[2], w[3]])).collect()
    }

    /// :203-205
    pub fn get_proof_normal_index(&self, normal_index: usize) -> MMR_proof {
        self.get_proof(get_mmr_index(normal_index))
    }

    /// :209-223
    pub fn get_proof(&self, mmr_index: usize) -> MMR_proof {
        let mut sib = [0u64; 4 * ffi::P2MT_MAX_PROOF_LEN];
        let mut lefts = [0u8; ffi::P2MT_MAX_PROOF_LEN];
        let mut peaks = [0u64; 4 * ffi::P2MT_MAX_PROOF_LEN];
        let (mut ns, mut np, mut size) = (0i32, 0i32, 0usize);
        ok(unsafe {
            ffi::p2mt_mmr_proof(self.h, mmr_index, sib.as_mut_ptr(), lefts.as_mut_ptr(), &mut ns, peaks.as_mut_ptr(), &mut np,
                                &mut size)
        });
        MMR_proof {
            mmr_size: size,
            merkle_proof: (0..ns as usize).map(|i| {
                let w = &sib[4 * i..4 * i + 4];
                (hash_from([w[0], w[1], w[2], w[3]]), lefts[i] != 0)
            }).collect(),
            peaks: peaks[..4 * np as usize].chunks_exact(4).map(|w| hash_from([w[0], w[1], w[2], w[3]])).collect(),
        }
    }

    /// :147-176 (associated function taking the MMR by value, as in the reference)
    pub fn get_subtree_proof_elm(mmr: MMR, mmr_index: usize) -> Vec<(HashOut<GoldilocksField>, bool)> {
        mmr.get_proof(mmr_index).merkle_proof
    }
}

/// One shard of an MMR whose 2^k leaves are split over the GPUs of a node: `rank` of `world` owns leaves
/// [rank * n_local, (rank + 1) * n_local).  One process per GPU; the ONLY exchange of a build -- world x 32 bytes -- is one
/// ncclAllGather on the library's stream (p2mt_sharded_mmr_*, csrc/p2mt_sharded.hip).
pub struct ShardedMMR {
    h: *mut ffi::p2mt_sharded_mmr,
    n_local: usize,
}
impl ShardedMMR {
    /// `nccl_unique_id`: the 128 bytes rank 0 got from `ShardedMMR::unique_id()` and sent to every rank (MPI_Bcast, a file, a
    /// socket); collective -- every rank calls this.  The communicator belongs to the handle.
    pub fn new(n_local: usize, rank: i32, world: i32, nccl_unique_id: &[u8; 128]) -> Self {
        let mut h = std::ptr::null_mut();
        ok(unsafe { ffi::p2mt_sharded_mmr_create_with_id(&mut h, n_local, rank, world, nccl_unique_id.as_ptr() as *const _) });
        ShardedMMR { h, n_local }
    }
    /// the same over a communicator the caller made with its own RCCL binding (never destroyed by the library)
    pub unsafe fn with_comm(n_local: usize, rank: i32, world: i32, nccl_comm: *mut std::os::raw::c_void) -> Self {
        let mut h = std::ptr::null_mut();
        ok(ffi::p2mt_sharded_mmr_create(&mut h, n_local, rank, world, nccl_comm));
        ShardedMMR { h, n_local }
    }
    pub fn unique_id() -> [u8; 128] {
        let mut id = [0u8; 128];
        ok(unsafe { ffi::p2mt_nccl_unique_id(id.as_mut_ptr() as *mut _) });
        id
    }
    /// `for l in my_leaves { add_leaf(l) }` on this rank's shard + the exchange + the top levels; returns the root of the WHOLE MMR
    pub fn build(&mut self, my_leaves: &[GoldilocksField]) -> HashOut<GoldilocksField> {
        assert!(my_leaves.len() == self.n_local);
        let words = canonical(my_leaves);
        ok(unsafe { ffi::p2mt_sharded_mmr_build(self.h, words.as_ptr()) });
        let mut r = [0u64; 4];
        ok(unsafe { ffi::p2mt_sharded_mmr_root(self.h, r.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()) });
        hash_from(r)
    }
    /// MMR::get_proof_normal_index for a leaf this rank owns (global index): siblings inside the shard + above it, the root as the peak
    pub fn get_proof_normal_index(&self, global_leaf: usize) -> MMR_proof {
        let mut sib = [0u64; 4 * ffi::P2MT_MAX_PROOF_LEN];
        let mut lefts = [0u8; ffi::P2MT_MAX_PROOF_LEN];
        let (mut ns, mut root) = (0i32, [0u64; 4]);
        ok(unsafe { ffi::p2mt_sharded_mmr_proof(self.h, global_leaf, sib.as_mut_ptr(), lefts.as_mut_ptr(), &mut ns, root.as_mut_ptr()) });
        MMR_proof {
            mmr_size: 0,  // (filled by callers that need it: 2 * n_local * world - 1)
            merkle_proof: (0..ns as usize).map(|i| {
                let w = &sib[4 * i..4 * i + 4];
                (hash_from([w[0], w[1], w[2], w[3]]), lefts[i] != 0)
            }).collect(),
            peaks: vec![hash_from(root)],
        }
    }
}
impl Drop for ShardedMMR {
    fn drop(&mut self) {
        unsafe { ffi::p2mt_sharded_mmr_destroy(self.h) };
    }
}

impl MMR {
    /// `MMR::from_leaves` for ONE rank of a sharded build: see `ShardedMMR` (the shard itself stays behind the returned handle;
    /// `.1` is the root of the whole MMR)
    pub fn from_leaves_sharded(my_leaves: &[GoldilocksField], rank: i32, world: i32, nccl_unique_id: &[u8; 128]) -> (ShardedMMR, HashOut<GoldilocksField>) {
        let mut s = ShardedMMR::new(my_leaves.len(), rank, world, nccl_unique_id);
        let root = s.build(my_leaves);
        (s, root)
    }

    /// `elements()` into page-locked memory of the library (the link's rate instead of a pageable copy's: 19 ms instead of 64 for the
    /// 1.07 GB of a 2^24-leaf MMR); the returned buffer frees itself
    pub fn elements_pinned(&self) -> PinnedWords {
        let len = unsafe { ffi::p2mt_mmr_len(self.h) };
        let mut p = std::ptr::null_mut();
        ok(unsafe { ffi::p2mt_host_alloc_pinned(32 * len, &mut p) });
        ok(unsafe { ffi::p2mt_mmr_copy_elements_async(self.h, 0, len, p as *mut u64) });
        ok(unsafe { ffi::p2mt_sync() });
        PinnedWords { p: p as *mut u64, n: 4 * len }
    }
}

/// page-locked host memory of the library holding HashOut records as words ([u64; 4] each)
pub struct PinnedWords {
    p: *mut u64,
    n: usize,
}
impl PinnedWords {
    pub fn as_slice(&self) -> &[u64] {
        unsafe { std::slice::from_raw_parts(self.p, self.n) }
    }
}
impl Drop for PinnedWords {
    fn drop(&mut self) {
        unsafe { ffi::p2mt_host_free_pinned(self.p as *mut _) };
    }
}

impl Clone for MMR {
    /// `#[derive(Clone)]` of the reference (:7): a second device-resident copy of the array
    fn clone(&self) -> Self {
        let len = unsafe { ffi::p2mt_mmr_len(self.h) };
        let mut words = vec![0u64; 4 * len];
        ok(unsafe { ffi::p2mt_mmr_copy_elements(self.h, 0, len, words.as_mut_ptr()) });
        // the leaves are the elements of height 0: leaf i sits at 2i - popcount(i)
        let n = unsafe { ffi::p2mt_mmr_num_leaves(self.h) };
        let leaves: Vec<u64> = (0..n).map(|i| words[4 * (2 * i - (i.count_ones() as usize))]).collect();
        let m = MMR::new();
        ok(unsafe { ffi::p2mt_mmr_extend(m.h, leaves.as_ptr(), leaves.len()) });
        m
    }
}

impl Drop for MMR {
    fn drop(&mut self) {
        unsafe { ffi::p2mt_mmr_destroy(self.h) };
    }
}

impl MMR_proof {
    /// :232-252.  Panics (status -5) where the reference's `assert!(self.peaks.contains(&next_hash))` fires (:245, quirk Q5).
    pub fn verify(self, leaf: GoldilocksField, root: HashOut<GoldilocksField>) -> bool {
        let sib: Vec<u64> = self.merkle_proof.iter().flat_map(|(h, _)| hash_words(h)).collect();
        let lefts: Vec<u8> = self.merkle_proof.iter().map(|(_, l)| *l as u8).collect();
        let peaks: Vec<u64> = self.peaks.iter().flat_map(hash_words).collect();
        let root_w = hash_words(&root);
        let mut result = 0i32;
        ok(unsafe {
            ffi::p2mt_mmr_proof_verify(sib.as_ptr(), lefts.as_ptr(), lefts.len() as i32, peaks.as_ptr(), self.peaks.len() as i32,
                                       leaf.to_canonical_u64(), root_w.as_ptr(), &mut result)
        });
        result != 0
    }
}

#[cfg(test)]
mod tests {
    use super::*;
    use plonky2::field::types::Field;

    // the reference's own tables (merkle_mountain_ranges.rs:280-301, :307-327)
    #[test]
    fn index_tables() {
        for (size, bitmap) in [(1usize, 1u64), (3, 2), (4, 3), (7, 4), (10, 6), (15, 8), (22, 12), (25, 14), (26, 15), (31, 16), (32, 17),
                               (34, 18), (35, 19), (38, 20), (41, 22), (42, 23)] {
            assert_eq!(get_heights_bitmap_for_mmr_size(size), (bitmap, 0));
        }
        for (n, idx) in [(0usize, 0usize), (1, 1), (2, 3), (3, 4), (4, 7), (5, 8), (6, 10), (7, 11), (8, 15), (9, 16), (10, 18), (11, 19),
                         (12, 22), (13, 23), (14, 25), (15, 26)] {
            assert_eq!(get_mmr_index(n), idx);
        }
    }

    #[test]
    fn proofs_of_every_leaf_verify() {
        let leaves: Vec<GoldilocksField> = (0..70u64).map(GoldilocksField::from_canonical_u64).collect();
        let mut mmr = MMR::new();
        for l in &leaves {
            mmr.add_leaf(*l);
        }
        assert_eq!(mmr.elements(), MMR::from_leaves(&leaves).elements());
        let root = mmr.bagging_the_peaks();
        for (i, l) in leaves.iter().enumerate() {
            assert!(mmr.get_proof_normal_index(i).verify(*l, root));
        }
    }
}
