/* oracle/plonk_eval.inc.h -- TEST INFRASTRUCTURE.  [parity unpinned]
 *
 * Gate constraints and the vanishing polynomial of the circuits the reference builds with plonky2's CircuitBuilder
 * (/root/reference/src/mmr/mmr_plonky2_verifier.rs:13-91, common.rs:5-58), generic over the evaluation field: the file is
 * included twice by plonk.c, once over the base field (the prover's quotient on the LDE coset) and once over the
 * quadratic extension (the verifier's check at zeta).  Restates plonky2 (git rev 3b21b87d, NOT in /root/reference)
 * gates/{poseidon,arithmetic_base,constant,public_input,noop}.rs eval_unfiltered, gates/gate.rs eval_filtered,
 * gates/selectors.rs, plonk/vanishing_poly.rs eval_vanishing_poly(_base_batch) and plonk_common.rs from their published
 * algorithm.
 *
 * The including file defines: FE (element type), FN(name) (suffixing), FE_ADD/FE_SUB/FE_MUL (FE x FE), FE_MULC (FE x
 * canonical u64), FE_ADDC / FE_SUBC (FE +- canonical u64), FE_FROMC (u64 -> FE).
 */

/* x^7 */
static inline FE FN(sbox7)(FE x) {
  FE x2 = FE_MUL(x, x), x4 = FE_MUL(x2, x2), x3 = FE_MUL(x2, x);
  return FE_MUL(x4, x3);
}

static void FN(mds_layer)(FE s[12]) {
  FE out[12];
  for (int r = 0; r < 12; ++r) {
    FE acc = FE_MULC(s[r], POSEIDON_MDS_DIAG[r]);
    for (int i = 0; i < 12; ++i) acc = FE_ADD(acc, FE_MULC(s[(i + r) % 12], POSEIDON_MDS_CIRC[i]));
    out[r] = acc;
  }
  memcpy(s, out, sizeof out);
}

/* PoseidonGate::eval_unfiltered: 123 constraints.  Wires: 0-11 in, 12-23 out, 24 swap, 25-28 delta, 29-64 full-round
 * S-box inputs of rounds 1-3, 65-86 partial-round S-box inputs, 87-134 S-box inputs of the last four full rounds.
 * plonky2 evaluates the partial rounds in its sparse ("fast") form; that form is a linear re-association of the spec form
 * used here, so both give the same constraint polynomials. */
static void FN(poseidon_gate_eval)(const FE *w, FE *out) {
  int k = 0;
  const FE swap = w[24];
  out[k++] = FE_MUL(swap, FE_SUBC(swap, 1));
  for (int i = 0; i < 4; ++i) out[k++] = FE_SUB(FE_MUL(swap, FE_SUB(w[i + 4], w[i])), w[25 + i]);
  FE s[12];
  for (int i = 0; i < 4; ++i) {
    s[i] = FE_ADD(w[i], w[25 + i]);
    s[i + 4] = FE_SUB(w[i + 4], w[25 + i]);
  }
  for (int i = 8; i < 12; ++i) s[i] = w[i];
  for (int r = 0; r < POSEIDON_ROUNDS; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = FE_ADDC(s[i], POSEIDON_RC[12 * r + i]);
    if (r < 4) {
      if (r >= 1)
        for (int i = 0; i < 12; ++i) {
          const FE in = w[29 + 12 * (r - 1) + i];
          out[k++] = FE_SUB(s[i], in);
          s[i] = in;
        }
      for (int i = 0; i < 12; ++i) s[i] = FN(sbox7)(s[i]);
    } else if (r < 26) {
      const FE in = w[65 + (r - 4)];
      out[k++] = FE_SUB(s[0], in);
      s[0] = FN(sbox7)(in);
    } else {
      for (int i = 0; i < 12; ++i) {
        const FE in = w[87 + 12 * (r - 26) + i];
        out[k++] = FE_SUB(s[i], in);
        s[i] = in;
      }
      for (int i = 0; i < 12; ++i) s[i] = FN(sbox7)(s[i]);
    }
    FN(mds_layer)(s);
  }
  for (int i = 0; i < 12; ++i) out[k++] = FE_SUB(s[i], w[12 + i]);
}

/* ---- the extension algebra FE[X]/(X^2 - 7): what plonky2 calls ExtensionAlgebra when FE is itself the extension field (the
 * verifier at zeta) and what is simply the quadratic extension when FE is the base field (the prover on the LDE coset).  The
 * recursion gates' wires hold extension elements as D = 2 consecutive wires. */
typedef struct { FE a, b; } FN(alg);
static inline FN(alg) FN(alg_w)(const FE *w, int at) { return (FN(alg)){w[at], w[at + 1]}; }
static inline FN(alg) FN(alg_add)(FN(alg) x, FN(alg) y) { return (FN(alg)){FE_ADD(x.a, y.a), FE_ADD(x.b, y.b)}; }
static inline FN(alg) FN(alg_sub)(FN(alg) x, FN(alg) y) { return (FN(alg)){FE_SUB(x.a, y.a), FE_SUB(x.b, y.b)}; }
static inline FN(alg) FN(alg_mul)(FN(alg) x, FN(alg) y) {
  return (FN(alg)){FE_ADD(FE_MUL(x.a, y.a), FE_MULC(FE_MUL(x.b, y.b), 7)), FE_ADD(FE_MUL(x.a, y.b), FE_MUL(x.b, y.a))};
}
static inline FN(alg) FN(alg_scale)(FN(alg) x, FE s) { return (FN(alg)){FE_MUL(x.a, s), FE_MUL(x.b, s)}; }
static inline FN(alg) FN(alg_scalec)(FN(alg) x, uint64_t c) { return (FN(alg)){FE_MULC(x.a, c), FE_MULC(x.b, c)}; }
static inline FN(alg) FN(alg_subc)(FN(alg) x, uint64_t c) { return (FN(alg)){FE_SUBC(x.a, c), x.b}; }

/* ArithmeticExtensionGate { num_ops: 10 }: per op, wires m0 | m1 | addend | output (2 each); output - (c0 m0 m1 + c1 addend) */
static int FN(arithmetic_ext_gate_eval)(const FE *w, const FE *gc, FE *out) {
  int k = 0;
  for (int i = 0; i < 10; ++i) {
    const FN(alg) m0 = FN(alg_w)(w, 8 * i), m1 = FN(alg_w)(w, 8 * i + 2), ad = FN(alg_w)(w, 8 * i + 4), o = FN(alg_w)(w, 8 * i + 6);
    const FN(alg) c = FN(alg_sub)(o, FN(alg_add)(FN(alg_scale)(FN(alg_mul)(m0, m1), gc[0]), FN(alg_scale)(ad, gc[1])));
    out[k++] = c.a;
    out[k++] = c.b;
  }
  return k;
}

/* MulExtensionGate { num_ops: 13 }: per op, wires m0 | m1 | output; output - c0 m0 m1 */
static int FN(mul_ext_gate_eval)(const FE *w, const FE *gc, FE *out) {
  int k = 0;
  for (int i = 0; i < 13; ++i) {
    const FN(alg) m0 = FN(alg_w)(w, 6 * i), m1 = FN(alg_w)(w, 6 * i + 2), o = FN(alg_w)(w, 6 * i + 4);
    const FN(alg) c = FN(alg_sub)(o, FN(alg_scale)(FN(alg_mul)(m0, m1), gc[0]));
    out[k++] = c.a;
    out[k++] = c.b;
  }
  return k;
}

/* BaseSumGate<2> { num_limbs: 63 }: wire 0 = sum, wires 1..63 = limbs; sum constraint, then limb (limb - 1) per limb */
static int FN(base_sum_gate_eval)(const FE *w, FE *out) {
  int k = 0;
  FE acc = FE_FROMC(0);
  for (int i = 63; i-- > 0;) acc = FE_ADD(FE_MULC(acc, 2), w[1 + i]); /* reduce_with_powers(limbs, 2) */
  out[k++] = FE_SUB(acc, w[0]);
  for (int i = 0; i < 63; ++i) out[k++] = FE_MUL(w[1 + i], FE_SUBC(w[1 + i], 1));
  return k;
}

/* ReducingGate { num_coeffs: 43 }: output 0-1, alpha 2-3, old_acc 4-5, coeffs 6..48 (base field), accs 49.. (the last acc is
 * the output); acc_i = acc_{i-1} alpha + coeff_i */
static int FN(reducing_gate_eval)(const FE *w, FE *out) {
  int k = 0;
  const FN(alg) alpha = FN(alg_w)(w, 2);
  FN(alg) acc = FN(alg_w)(w, 4);
  for (int i = 0; i < 43; ++i) {
    const FN(alg) next = i == 42 ? FN(alg_w)(w, 0) : FN(alg_w)(w, 49 + 2 * i);
    FN(alg) c = FN(alg_mul)(acc, alpha);
    c.a = FE_ADD(c.a, w[6 + i]);
    c = FN(alg_sub)(next, c);
    out[k++] = c.a;
    out[k++] = c.b;
    acc = next;
  }
  return k;
}

/* ReducingExtensionGate { num_coeffs: 32 }: output 0-1, alpha 2-3, old_acc 4-5, coeffs 6..69 (2 each), accs 70.. */
static int FN(reducing_ext_gate_eval)(const FE *w, FE *out) {
  int k = 0;
  const FN(alg) alpha = FN(alg_w)(w, 2);
  FN(alg) acc = FN(alg_w)(w, 4);
  for (int i = 0; i < 32; ++i) {
    const FN(alg) next = i == 31 ? FN(alg_w)(w, 0) : FN(alg_w)(w, 70 + 2 * i);
    const FN(alg) c = FN(alg_sub)(next, FN(alg_add)(FN(alg_mul)(acc, alpha), FN(alg_w)(w, 6 + 2 * i)));
    out[k++] = c.a;
    out[k++] = c.b;
    acc = next;
  }
  return k;
}

/* RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2 }: copy c: access_index 18c, claimed 18c+1, list 18c+2..18c+17;
 * extra constants on wires 72, 73; bits of copy c on wires 74+4c.. */
static int FN(random_access_gate_eval)(const FE *w, const FE *gc, FE *out) {
  int k = 0;
  for (int c = 0; c < 4; ++c) {
    const FE *bits = w + 74 + 4 * c;
    for (int i = 0; i < 4; ++i) out[k++] = FE_MUL(bits[i], FE_SUBC(bits[i], 1));
    FE idx = FE_FROMC(0);
    for (int i = 4; i-- > 0;) idx = FE_ADD(FE_ADD(idx, idx), bits[i]);
    out[k++] = FE_SUB(idx, w[18 * c]);
    FE list[16];
    for (int i = 0; i < 16; ++i) list[i] = w[18 * c + 2 + i];
    for (int b = 0, len = 16; b < 4; ++b, len >>= 1)
      for (int j = 0; j < len / 2; ++j) list[j] = FE_ADD(list[2 * j], FE_MUL(bits[b], FE_SUB(list[2 * j + 1], list[2 * j])));
    out[k++] = FE_SUB(list[0], w[18 * c + 1]);
  }
  for (int i = 0; i < 2; ++i) out[k++] = FE_SUB(gc[i], w[72 + i]);
  return k;
}

/* CosetInterpolationGate { subgroup_bits: 4, degree: 6 } (2 intermediates): shift 0, values 1..32, evaluation point 33-34,
 * evaluation value 35-36, intermediate evals 37..40, intermediate products 41..44, shifted evaluation point 45-46.
 * Barycentric interpolation on the subgroup <g_16>, chunked so that no constraint exceeds degree 6. */
#ifndef ORACLE_COSET_INTERP_TABLES
#define ORACLE_COSET_INTERP_TABLES
static uint64_t coset_interp_domain[16], coset_interp_weights[16];
static void coset_interp_tables(void) {
  if (coset_interp_domain[0]) return;
  uint64_t d[16];
  const uint64_t g = gl_primitive_root_of_unity(4);
  d[0] = 1;
  for (int i = 1; i < 16; ++i) d[i] = gl_mul(d[i - 1], g);
  for (int i = 0; i < 16; ++i) { /* w_i = 1 / prod_{j != i} (x_i - x_j) */
    uint64_t p = 1;
    for (int j = 0; j < 16; ++j)
      if (j != i) p = gl_mul(p, gl_sub(d[i], d[j]));
    coset_interp_weights[i] = gl_inv(p);
  }
  for (int i = 15; i >= 0; --i) coset_interp_domain[i] = d[i]; /* [0] last: it is the "initialised" marker */
}
#endif
static void FN(partial_interpolate)(const FE *w, int from, int to, FN(alg) x, FN(alg) *eval, FN(alg) *prod) {
  for (int i = from; i < to; ++i) {
    const FN(alg) term = FN(alg_subc)(x, coset_interp_domain[i]);
    const FN(alg) weighted = FN(alg_scalec)(FN(alg_w)(w, 1 + 2 * i), coset_interp_weights[i]);
    *eval = FN(alg_add)(FN(alg_mul)(*eval, term), FN(alg_mul)(weighted, *prod));
    *prod = FN(alg_mul)(*prod, term);
  }
}
static int FN(coset_interpolation_gate_eval)(const FE *w, FE *out) {
  coset_interp_tables();
  int k = 0;
  const FN(alg) point = FN(alg_w)(w, 33), shifted = FN(alg_w)(w, 45);
  FN(alg) c = FN(alg_sub)(point, FN(alg_scale)(shifted, w[0]));
  out[k++] = c.a;
  out[k++] = c.b;
  FN(alg) eval = {FE_FROMC(0), FE_FROMC(0)}, prod = {FE_FROMC(1), FE_FROMC(0)};
  FN(partial_interpolate)(w, 0, 6, shifted, &eval, &prod);
  for (int i = 0; i < 2; ++i) {
    const FN(alg) ie = FN(alg_w)(w, 37 + 2 * i), ip = FN(alg_w)(w, 41 + 2 * i);
    c = FN(alg_sub)(ie, eval);
    out[k++] = c.a;
    out[k++] = c.b;
    c = FN(alg_sub)(ip, prod);
    out[k++] = c.a;
    out[k++] = c.b;
    eval = ie;
    prod = ip;
    const int start = 1 + 5 * (i + 1), end = start + 5 < 16 ? start + 5 : 16;
    FN(partial_interpolate)(w, start, end, shifted, &eval, &prod);
  }
  c = FN(alg_sub)(FN(alg_w)(w, 35), eval);
  out[k++] = c.a;
  out[k++] = c.b;
  return k;
}

/* PoseidonMdsGate: inputs 0..23 (2 each), outputs 24..47; output = MDS * input over the algebra */
static int FN(poseidon_mds_gate_eval)(const FE *w, FE *out) {
  int k = 0;
  for (int r = 0; r < 12; ++r) {
    FN(alg) acc = FN(alg_scalec)(FN(alg_w)(w, 2 * r), POSEIDON_MDS_DIAG[r]);
    for (int i = 0; i < 12; ++i) acc = FN(alg_add)(acc, FN(alg_scalec)(FN(alg_w)(w, 2 * ((i + r) % 12)), POSEIDON_MDS_CIRC[i]));
    const FN(alg) c = FN(alg_sub)(acc, FN(alg_w)(w, 24 + 2 * r));
    out[k++] = c.a;
    out[k++] = c.b;
  }
  return k;
}

/* eval_unfiltered of one gate type: gc = the gate's own constants (after the selectors).  Returns the constraint count. */
static int FN(gate_eval_unfiltered)(unsigned kind, unsigned num_constants, unsigned num_routed, const FE *gc, const FE *w,
                                    const uint64_t pi_hash[4], FE *c) {
  int nc = 0;
  switch (kind) {
    case ORACLE_GATE_NOOP: break;
    case ORACLE_GATE_CONSTANT: /* local_constants[i] - wires[i] */
      for (unsigned i = 0; i < num_constants; ++i) c[nc++] = FE_SUB(gc[i], w[i]);
      break;
    case ORACLE_GATE_PUBLIC_INPUT: /* wires[i] - public_inputs_hash[i] */
      for (int i = 0; i < 4; ++i) c[nc++] = FE_SUBC(w[i], pi_hash[i]);
      break;
    case ORACLE_GATE_ARITHMETIC: /* output - (m0 m1 c0 + addend c1), num_routed / 4 operations per row */
      for (unsigned i = 0; i < num_routed / 4; ++i) {
        const FE prod = FE_MUL(FE_MUL(w[4 * i], w[4 * i + 1]), gc[0]);
        c[nc++] = FE_SUB(w[4 * i + 3], FE_ADD(prod, FE_MUL(w[4 * i + 2], gc[1])));
      }
      break;
    case ORACLE_GATE_POSEIDON:
      FN(poseidon_gate_eval)(w, c);
      nc = 123;
      break;
    case ORACLE_GATE_BASE_SUM: nc = FN(base_sum_gate_eval)(w, c); break;
    case ORACLE_GATE_ARITHMETIC_EXT: nc = FN(arithmetic_ext_gate_eval)(w, gc, c); break;
    case ORACLE_GATE_MUL_EXT: nc = FN(mul_ext_gate_eval)(w, gc, c); break;
    case ORACLE_GATE_REDUCING: nc = FN(reducing_gate_eval)(w, c); break;
    case ORACLE_GATE_REDUCING_EXT: nc = FN(reducing_ext_gate_eval)(w, c); break;
    case ORACLE_GATE_RANDOM_ACCESS: nc = FN(random_access_gate_eval)(w, gc, c); break;
    case ORACLE_GATE_COSET_INTERPOLATION: nc = FN(coset_interpolation_gate_eval)(w, c); break;
    case ORACLE_GATE_POSEIDON_MDS: nc = FN(poseidon_mds_gate_eval)(w, c); break;
  }
  return nc;
}

/* compute_filter: prod_{i in group, i != row} (i - s) * (UNUSED_SELECTOR - s if there are several groups) */
static FE FN(gate_filter)(const oracle_plonk_desc *d, unsigned g, FE s) {
  FE f = FE_FROMC(1);
  for (unsigned i = d->group_start[g]; i < d->group_end[g]; ++i)
    if (i != g) f = FE_MUL(f, FE_SUB(FE_FROMC(i), s));
  if (d->num_selectors > 1) f = FE_MUL(f, FE_SUB(FE_FROMC(0xFFFFFFFFULL), s));
  return f;
}

/* evaluate_gate_constraints: terms[j] = sum over gate types of filter * constraint_j.  consts = selectors then the
 * gates' own constants. */
static void FN(gate_constraints)(const oracle_plonk_desc *d, const FE *consts, const FE *w, const uint64_t pi_hash[4],
                                 FE *terms /* [ORACLE_PLONK_NUM_GATE_CONSTRAINTS] */) {
  const FE *gc = consts + d->num_selectors;
  for (int j = 0; j < ORACLE_PLONK_NUM_GATE_CONSTRAINTS; ++j) terms[j] = FE_FROMC(0);
  for (unsigned g = 0; g < d->num_gates; ++g) {
    FE c[ORACLE_PLONK_NUM_GATE_CONSTRAINTS];
    const int nc = FN(gate_eval_unfiltered)(d->gate_kind[g], d->num_constants, d->num_routed, gc, w, pi_hash, c);
    const FE f = FN(gate_filter)(d, g, consts[d->gate_selector[g]]);
    for (int j = 0; j < nc; ++j) terms[j] = FE_ADD(terms[j], FE_MUL(f, c[j]));
  }
}

/* eval_vanishing_poly at one point x: out[c] = sum_k terms[k] alpha_c^k over
 * terms = [L_0(x) (Z_c(x) - 1)]_c ++ [partial-product checks]_c ++ gate constraints.
 * l0_x = L_0(x) is supplied by the caller (it needs a division). */
static void FN(eval_vanishing)(const oracle_plonk_desc *d, const uint64_t *k_is, FE x, FE l0_x, const FE *consts, const FE *sigmas,
                               const FE *w, const FE *zs, const FE *next_zs, const FE *pps, const uint64_t pi_hash[4],
                               const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas, FE *out) {
  const unsigned nch = d->num_challenges, chunk = d->quotient_degree_factor;
  const unsigned num_chunks = (d->num_routed + chunk - 1) / chunk, num_prods = num_chunks - 1;
  const unsigned n_terms = nch + nch * num_chunks + ORACLE_PLONK_NUM_GATE_CONSTRAINTS;
  FE terms[16 + 16 * 64 + ORACLE_PLONK_NUM_GATE_CONSTRAINTS];
  unsigned k = 0;
  for (unsigned c = 0; c < nch; ++c) terms[k++] = FE_MUL(l0_x, FE_SUBC(zs[c], 1));
  for (unsigned c = 0; c < nch; ++c) {
    const FE bx = FE_MULC(x, betas[c]);
    for (unsigned q = 0; q < num_chunks; ++q) {
      FE num = FE_FROMC(1), den = FE_FROMC(1);
      for (unsigned j = q * chunk; j < d->num_routed && j < (q + 1) * chunk; ++j) {
        const FE wg = FE_ADDC(w[j], gammas[c]);
        num = FE_MUL(num, FE_ADD(wg, FE_MULC(bx, k_is[j])));
        den = FE_MUL(den, FE_ADD(wg, FE_MULC(sigmas[j], betas[c])));
      }
      const FE prev = q == 0 ? zs[c] : pps[c * num_prods + q - 1];
      const FE next = q == num_prods ? next_zs[c] : pps[c * num_prods + q];
      terms[k++] = FE_SUB(FE_MUL(prev, num), FE_MUL(next, den));
    }
  }
  FN(gate_constraints)(d, consts, w, pi_hash, terms + k);
  for (unsigned c = 0; c < nch; ++c) { /* reduce_with_powers: Horner from the last term */
    FE acc = FE_FROMC(0);
    for (unsigned t = n_terms; t-- > 0;) acc = FE_ADD(FE_MULC(acc, alphas[c]), terms[t]);
    out[c] = acc;
  }
}
