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
    int nc = 0;
    switch (d->gate_kind[g]) {
      case ORACLE_GATE_NOOP: break;
      case ORACLE_GATE_CONSTANT: /* local_constants[i] - wires[i] */
        for (unsigned i = 0; i < d->num_constants; ++i) c[nc++] = FE_SUB(gc[i], w[i]);
        break;
      case ORACLE_GATE_PUBLIC_INPUT: /* wires[i] - public_inputs_hash[i] */
        for (int i = 0; i < 4; ++i) c[nc++] = FE_SUBC(w[i], pi_hash[i]);
        break;
      case ORACLE_GATE_ARITHMETIC: /* output - (m0 m1 c0 + addend c1), num_routed / 4 operations per row */
        for (unsigned i = 0; i < d->num_routed / 4; ++i) {
          const FE prod = FE_MUL(FE_MUL(w[4 * i], w[4 * i + 1]), gc[0]);
          c[nc++] = FE_SUB(w[4 * i + 3], FE_ADD(prod, FE_MUL(w[4 * i + 2], gc[1])));
        }
        break;
      case ORACLE_GATE_POSEIDON:
        FN(poseidon_gate_eval)(w, c);
        nc = 123;
        break;
    }
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
