// gates_recursion.hip.h -- constraints of the gate types plonky2's in-circuit verifier instantiates (builder.verify_proof,
// /root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:101-104): BaseSumGate<2>{63}, ArithmeticExtensionGate{10},
// MulExtensionGate{13}, ReducingGate{43}, ReducingExtensionGate{32}, RandomAccessGate{bits 4, 4 copies, 2 extra constants},
// CosetInterpolationGate{subgroup_bits 4, degree 6}, PoseidonMdsGate -- under standard_recursion_config, D = 2.
//
// Restates plonky2 (git rev 3b21b87d, NOT in /root/reference; parity unpinned) gates/{base_sum, arithmetic_extension,
// multiplication_extension, reducing, reducing_extension, random_access, coset_interpolation, poseidon_mds}.rs eval_unfiltered.
// One source for both evaluation fields, as in plonky2: the prover's quotient kernel instantiates it over the base field on the
// LDE coset (device, one lane per point), the verifier over the quadratic extension at zeta (host).  A gate's wires hold an
// extension element as two consecutive wires; over a field FE the pair is an element of the algebra FE[X]/(X^2 - 7).
//
// F supplies: typedef T; add, sub, mul (T x T); mulc / addc / subc (T x canonical u64); fromc (u64 -> T).
// W: wire accessor (column -> T).  Emit: (constraint index, value).
#pragma once
#include <stdint.h>

#define GR_HD __host__ __device__ __forceinline__

namespace gates_rec {

// two_adic_subgroup(4) and its barycentric weights 1 / prod_{j != i} (x_i - x_j)  (= x_i / 16)
__device__ __constant__ const uint64_t kCosetDomainDev[16] = {
    0x0000000000000001ull, 0xefffffff00000001ull, 0xfffffffeff000001ull, 0x000ffffffff00000ull, 0x0001000000000000ull, 0x0000000000001000ull,
    0xfffffeff00000101ull, 0xffffffef00000001ull, 0xffffffff00000000ull, 0x1000000000000000ull, 0x0000000001000000ull, 0xffefffff00100001ull,
    0xfffeffff00000001ull, 0xfffffffefffff001ull, 0x000000ffffffff00ull, 0x0000001000000000ull};
__device__ __constant__ const uint64_t kCosetWeightsDev[16] = {
    0xefffffff10000001ull, 0xfeffffff00000001ull, 0xfffffffefff00001ull, 0x0000ffffffff0000ull, 0x0000100000000000ull, 0x0000000000000100ull,
    0xffffffef00000011ull, 0xfffffffe00000001ull, 0x0ffffffff0000000ull, 0x0100000000000000ull, 0x0000000000100000ull, 0xfffeffff00010001ull,
    0xffffefff00000001ull, 0xfffffffeffffff01ull, 0x0000000ffffffff0ull, 0x0000000100000000ull};
static const uint64_t kCosetDomainHost[16] = {
    0x0000000000000001ull, 0xefffffff00000001ull, 0xfffffffeff000001ull, 0x000ffffffff00000ull, 0x0001000000000000ull, 0x0000000000001000ull,
    0xfffffeff00000101ull, 0xffffffef00000001ull, 0xffffffff00000000ull, 0x1000000000000000ull, 0x0000000001000000ull, 0xffefffff00100001ull,
    0xfffeffff00000001ull, 0xfffffffefffff001ull, 0x000000ffffffff00ull, 0x0000001000000000ull};
static const uint64_t kCosetWeightsHost[16] = {
    0xefffffff10000001ull, 0xfeffffff00000001ull, 0xfffffffefff00001ull, 0x0000ffffffff0000ull, 0x0000100000000000ull, 0x0000000000000100ull,
    0xffffffef00000011ull, 0xfffffffe00000001ull, 0x0ffffffff0000000ull, 0x0100000000000000ull, 0x0000000000100000ull, 0xfffeffff00010001ull,
    0xffffefff00000001ull, 0xfffffffeffffff01ull, 0x0000000ffffffff0ull, 0x0000000100000000ull};
GR_HD uint64_t coset_domain(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return kCosetDomainDev[i];
#else
  return kCosetDomainHost[i];
#endif
}
GR_HD uint64_t coset_weight(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return kCosetWeightsDev[i];
#else
  return kCosetWeightsHost[i];
#endif
}
// MDS_MATRIX_CIRC / MDS_MATRIX_DIAG (hash/poseidon_goldilocks.rs)
GR_HD uint64_t mds_circ(int i) {
  constexpr uint64_t c[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  return c[i];
}

template <class F>
struct Alg {
  typename F::T a, b;
};
template <class F, class W>
GR_HD Alg<F> alg_w(const W& w, int at) {
  return Alg<F>{w(at), w(at + 1)};
}
template <class F>
GR_HD Alg<F> alg_add(Alg<F> x, Alg<F> y) {
  return Alg<F>{F::add(x.a, y.a), F::add(x.b, y.b)};
}
template <class F>
GR_HD Alg<F> alg_sub(Alg<F> x, Alg<F> y) {
  return Alg<F>{F::sub(x.a, y.a), F::sub(x.b, y.b)};
}
template <class F>
GR_HD Alg<F> alg_mul(Alg<F> x, Alg<F> y) {
  return Alg<F>{F::add(F::mul(x.a, y.a), F::mulc(F::mul(x.b, y.b), 7)), F::add(F::mul(x.a, y.b), F::mul(x.b, y.a))};
}
template <class F>
GR_HD Alg<F> alg_scale(Alg<F> x, typename F::T s) {
  return Alg<F>{F::mul(x.a, s), F::mul(x.b, s)};
}
template <class F>
GR_HD Alg<F> alg_scalec(Alg<F> x, uint64_t c) {
  return Alg<F>{F::mulc(x.a, c), F::mulc(x.b, c)};
}

// ArithmeticExtensionGate { num_ops: 10 }: per op, wires m0 | m1 | addend | output; output - (c0 m0 m1 + c1 addend)
template <class F, class W, class Emit>
GR_HD void arithmetic_ext_gate(const W& w, typename F::T c0, typename F::T c1, const Emit& emit) {
#pragma unroll 1
  for (int i = 0; i < 10; ++i) {
    const Alg<F> m0 = alg_w<F>(w, 8 * i), m1 = alg_w<F>(w, 8 * i + 2), ad = alg_w<F>(w, 8 * i + 4), o = alg_w<F>(w, 8 * i + 6);
    const Alg<F> c = alg_sub<F>(o, alg_add<F>(alg_scale<F>(alg_mul<F>(m0, m1), c0), alg_scale<F>(ad, c1)));
    emit(2 * i, c.a);
    emit(2 * i + 1, c.b);
  }
}

// MulExtensionGate { num_ops: 13 }: per op, wires m0 | m1 | output; output - c0 m0 m1
template <class F, class W, class Emit>
GR_HD void mul_ext_gate(const W& w, typename F::T c0, const Emit& emit) {
#pragma unroll 1
  for (int i = 0; i < 13; ++i) {
    const Alg<F> m0 = alg_w<F>(w, 6 * i), m1 = alg_w<F>(w, 6 * i + 2), o = alg_w<F>(w, 6 * i + 4);
    const Alg<F> c = alg_sub<F>(o, alg_scale<F>(alg_mul<F>(m0, m1), c0));
    emit(2 * i, c.a);
    emit(2 * i + 1, c.b);
  }
}

// BaseSumGate<2> { num_limbs: 63 }: wire 0 = sum, wires 1..63 = limbs; reduce_with_powers(limbs, 2) - sum, then limb (limb - 1)
template <class F, class W, class Emit>
GR_HD void base_sum_gate(const W& w, const Emit& emit) {
  typename F::T acc = F::fromc(0);
#pragma unroll 1
  for (int i = 63; i-- > 0;) acc = F::add(F::mulc(acc, 2), w(1 + i));
  emit(0, F::sub(acc, w(0)));
#pragma unroll 1
  for (int i = 0; i < 63; ++i) {
    const typename F::T l = w(1 + i);
    emit(1 + i, F::mul(l, F::subc(l, 1)));
  }
}

// ReducingGate { num_coeffs: 43 }: output 0-1, alpha 2-3, old_acc 4-5, coeffs 6..48 (base field), accs 49.. (the last acc is the
// output); acc_i = acc_{i-1} alpha + coeff_i
template <class F, class W, class Emit>
GR_HD void reducing_gate(const W& w, const Emit& emit) {
  const Alg<F> alpha = alg_w<F>(w, 2);
  Alg<F> acc = alg_w<F>(w, 4);
#pragma unroll 1
  for (int i = 0; i < 43; ++i) {
    const Alg<F> next = i == 42 ? alg_w<F>(w, 0) : alg_w<F>(w, 49 + 2 * i);
    Alg<F> c = alg_mul<F>(acc, alpha);
    c.a = F::add(c.a, w(6 + i));
    c = alg_sub<F>(next, c);
    emit(2 * i, c.a);
    emit(2 * i + 1, c.b);
    acc = next;
  }
}

// ReducingExtensionGate { num_coeffs: 32 }: output 0-1, alpha 2-3, old_acc 4-5, coeffs 6..69 (2 each), accs 70..
template <class F, class W, class Emit>
GR_HD void reducing_ext_gate(const W& w, const Emit& emit) {
  const Alg<F> alpha = alg_w<F>(w, 2);
  Alg<F> acc = alg_w<F>(w, 4);
#pragma unroll 1
  for (int i = 0; i < 32; ++i) {
    const Alg<F> next = i == 31 ? alg_w<F>(w, 0) : alg_w<F>(w, 70 + 2 * i);
    const Alg<F> c = alg_sub<F>(next, alg_add<F>(alg_mul<F>(acc, alpha), alg_w<F>(w, 6 + 2 * i)));
    emit(2 * i, c.a);
    emit(2 * i + 1, c.b);
    acc = next;
  }
}

// RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2 }: copy c: access_index 18c, claimed 18c+1, list 18c+2..18c+17;
// extra constants on wires 72, 73; bits of copy c on wires 74+4c..  Per copy: 4 x bit (bit - 1), index, claimed element.
template <class F, class W, class Emit>
GR_HD void random_access_gate(const W& w, typename F::T gc0, typename F::T gc1, const Emit& emit) {
  typedef typename F::T T;
#pragma unroll 1
  for (int c = 0; c < 4; ++c) {
    T bits[4];
    for (int i = 0; i < 4; ++i) {
      bits[i] = w(74 + 4 * c + i);
      emit(6 * c + i, F::mul(bits[i], F::subc(bits[i], 1)));
    }
    T idx = F::fromc(0);
    for (int i = 4; i-- > 0;) idx = F::add(F::add(idx, idx), bits[i]);
    emit(6 * c + 4, F::sub(idx, w(18 * c)));
    // fold the list pairwise on bit 0, then 1, ...: written depth-first so that 5 values are live instead of 16
    T lvl3[2];
    for (int h = 0; h < 2; ++h) {
      T lvl2[2];
      for (int q = 0; q < 2; ++q) {
        T lvl1[2];
        for (int p = 0; p < 2; ++p) {
          const int at = 18 * c + 2 + 8 * h + 4 * q + 2 * p;
          const T x = w(at), y = w(at + 1);
          lvl1[p] = F::add(x, F::mul(bits[0], F::sub(y, x)));
        }
        lvl2[q] = F::add(lvl1[0], F::mul(bits[1], F::sub(lvl1[1], lvl1[0])));
      }
      lvl3[h] = F::add(lvl2[0], F::mul(bits[2], F::sub(lvl2[1], lvl2[0])));
    }
    const T sel = F::add(lvl3[0], F::mul(bits[3], F::sub(lvl3[1], lvl3[0])));
    emit(6 * c + 5, F::sub(sel, w(18 * c + 1)));
  }
  emit(24, F::sub(gc0, w(72)));
  emit(25, F::sub(gc1, w(73)));
}

// CosetInterpolationGate { subgroup_bits: 4, degree: 6 } (2 intermediates): shift 0, values 1..32, evaluation point 33-34,
// evaluation value 35-36, intermediate evals 37..40, intermediate products 41..44, shifted evaluation point 45-46.  Barycentric
// interpolation on the subgroup <g_16> at x = point / shift, chunked (6, 5, 5 points) so that no constraint exceeds degree 6.
template <class F, class W>
GR_HD void partial_interpolate(const W& w, int from, int to, Alg<F> x, Alg<F>& eval, Alg<F>& prod) {
#pragma unroll 1
  for (int i = from; i < to; ++i) {
    Alg<F> term = x;
    term.a = F::subc(term.a, coset_domain(i));
    const Alg<F> weighted = alg_scalec<F>(alg_w<F>(w, 1 + 2 * i), coset_weight(i));
    eval = alg_add<F>(alg_mul<F>(eval, term), alg_mul<F>(weighted, prod));
    prod = alg_mul<F>(prod, term);
  }
}
template <class F, class W, class Emit>
GR_HD void coset_interpolation_gate(const W& w, const Emit& emit) {
  const Alg<F> point = alg_w<F>(w, 33), shifted = alg_w<F>(w, 45);
  Alg<F> c = alg_sub<F>(point, alg_scale<F>(shifted, w(0)));
  emit(0, c.a);
  emit(1, c.b);
  Alg<F> eval{F::fromc(0), F::fromc(0)}, prod{F::fromc(1), F::fromc(0)};
  partial_interpolate<F>(w, 0, 6, shifted, eval, prod);
#pragma unroll 1
  for (int i = 0; i < 2; ++i) {
    const Alg<F> ie = alg_w<F>(w, 37 + 2 * i), ip = alg_w<F>(w, 41 + 2 * i);
    c = alg_sub<F>(ie, eval);
    emit(2 + 4 * i, c.a);
    emit(3 + 4 * i, c.b);
    c = alg_sub<F>(ip, prod);
    emit(4 + 4 * i, c.a);
    emit(5 + 4 * i, c.b);
    eval = ie;
    prod = ip;
    const int start = 1 + 5 * (i + 1), end = start + 5 < 16 ? start + 5 : 16;
    partial_interpolate<F>(w, start, end, shifted, eval, prod);
  }
  c = alg_sub<F>(alg_w<F>(w, 35), eval);
  emit(10, c.a);
  emit(11, c.b);
}

// PoseidonMdsGate: inputs 0..23 (2 each), outputs 24..47; output = MDS * input over the algebra
template <class F, class W, class Emit>
GR_HD void poseidon_mds_gate(const W& w, const Emit& emit) {
#pragma unroll 1
  for (int r = 0; r < 12; ++r) {
    Alg<F> acc = r == 0 ? alg_scalec<F>(alg_w<F>(w, 0), 8) : Alg<F>{F::fromc(0), F::fromc(0)};
#pragma unroll 1
    for (int i = 0; i < 12; ++i) {
      const int src = i + r >= 12 ? i + r - 12 : i + r;
      acc = alg_add<F>(acc, alg_scalec<F>(alg_w<F>(w, 2 * src), mds_circ(i)));
    }
    const Alg<F> c = alg_sub<F>(acc, alg_w<F>(w, 24 + 2 * r));
    emit(2 * r, c.a);
    emit(2 * r + 1, c.b);
  }
}

}  // namespace gates_rec
