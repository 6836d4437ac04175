//! /root/reference/src/mmr/common.rs:5-58 over the library's circuit builder: same names, argument order and -- this is what fixes
//! the circuit -- the same builder calls in the same order.
use crate::plonk::{BoolTarget, CircuitBuilder, HashOutTarget};

pub const GOLDILOCKS_FIELD_ORDER: u64 = 18446744069414584321;

/// common.rs:5-17
pub fn equal(builder: &mut CircuitBuilder, first: HashOutTarget, second: HashOutTarget) -> BoolTarget {
    let elm0 = builder.is_equal(first.elements[0], second.elements[0]);
    let elm1 = builder.is_equal(first.elements[1], second.elements[1]);
    let elm2 = builder.is_equal(first.elements[2], second.elements[2]);
    let elm3 = builder.is_equal(first.elements[3], second.elements[3]);
    let elm0_or_elm1 = builder.or(elm0, elm1);
    let elm2_or_elm3 = builder.or(elm2, elm3);
    builder.or(elm0_or_elm1, elm2_or_elm3)
}

/// common.rs:19-39
pub fn or_list(builder: &mut CircuitBuilder, ins: Vec<BoolTarget>) -> BoolTarget {
    assert!(ins.len() > 0);
    if ins.len() == 1 {
        ins[0]
    } else if ins.len() == 2 {
        builder.or(ins[0], ins[1])
    } else {
        let mut pairs: Vec<BoolTarget> = Vec::new();
        for pair in ins.chunks(2) {
            if pair.len() > 1 {
                pairs.push(builder.or(pair[0], pair[1]));
            } else {
                pairs.push(pair[0]);
            }
        }
        or_list(builder, pairs)
    }
}

/// common.rs:43-58: option1 if pick_left else option2
pub fn pick_hash(builder: &mut CircuitBuilder, option1: HashOutTarget, option2: HashOutTarget, pick_left: BoolTarget) -> HashOutTarget {
    let opposite = builder.not(pick_left);
    let t0 = builder.mul(option2.elements[0], opposite.target);
    let t1 = builder.mul(option2.elements[1], opposite.target);
    let t2 = builder.mul(option2.elements[2], opposite.target);
    let t3 = builder.mul(option2.elements[3], opposite.target);
    let hash_elm0 = builder.mul_add(option1.elements[0], pick_left.target, t0);
    let hash_elm1 = builder.mul_add(option1.elements[1], pick_left.target, t1);
    let hash_elm2 = builder.mul_add(option1.elements[2], pick_left.target, t2);
    let hash_elm3 = builder.mul_add(option1.elements[3], pick_left.target, t3);
    HashOutTarget { elements: [hash_elm0, hash_elm1, hash_elm2, hash_elm3] }
}
