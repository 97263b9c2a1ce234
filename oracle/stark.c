/* ORACLE — TEST INFRASTRUCTURE ONLY (see bb31.h).
 * CPU restatement of the fib_air proving path the reference drives at native/src/fib_air.rs:27-75:
 *   p3_uni_stark::prove / verify  ->  TwoAdicFriPcs::{commit, open, verify}  ->  p3_fri::{prove, verify}
 * for the configuration north_star names (BabyBear, Poseidon2 MMCS, DuplexChallenger<_, Perm16, 16, 8>,
 * non-hiding), with FibonacciAir exactly as native/src/fib_air.rs:224-264 and the trace of :266-284.
 * Every protocol step lives in the ABSENT crates p3-uni-stark / p3-fri / p3-challenger / p3-commit 0.4.2
 * [UPSTREAM-RECALL]; there is no fixture for it anywhere in the reference (it only checks
 * prove -> verify, fib_air.rs:70-72).  PARITY UNPINNED against upstream; what IS checked:
 *   - p3o_verify_fib_air below is an independent statement of the verifier equations, and accepts;
 *   - the HIP prover must reproduce these proof bytes exactly.
 * Wire format (ours; the reference never serialises a proof): u32 little-endian Montgomery words,
 * vectors prefixed by a u32 count, fields in the order of Plonky3's Proof / FriProof structs. */
#include "stark_common.h"

/* p3_uni_stark::prove for FibonacciAir.  Returns malloc'd proof bytes. */
int p3o_prove_fib_air_hash(int hash, uint64_t a, uint64_t b, unsigned log_n, unsigned log_blowup, unsigned log_final_poly_len,
                           unsigned num_queries, unsigned pow_bits, uint8_t **out, size_t *out_len) {
    if (hash != 0 && hash != 1) return -1;
    if (log_n < 1 || log_blowup < 1 || log_n + log_blowup > BB_TWO_ADICITY || log_final_poly_len > log_n) return -1;
    /* p3_fri::prover::prove: if log_final_poly_len > 0, log_min_height > log_final_poly_len + log_blowup */
    if (log_final_poly_len > 0 && log_final_poly_len >= log_n) return -1;
    const size_t n = (size_t)1 << log_n, big = n << log_blowup;
    const unsigned log_big = log_n + log_blowup;
    const uint32_t gen = bb_to_monty(BB_GENERATOR_CANON);
    buf_t pf = {0};
    /* trace: fib_air.rs:266-284 */
    uint32_t *trace = malloc(n * 2 * 4);
    { uint32_t l = bb_to_monty((uint32_t)(a % BB_P)), r = bb_to_monty((uint32_t)(b % BB_P));
      for (size_t i = 0; i < n; i++) { trace[2 * i] = l; trace[2 * i + 1] = r; uint32_t t = bb_add(l, r); l = r; r = t; } }
    uint32_t pis[3] = {trace[0], trace[1], trace[2 * (n - 1) + 1]};
    /* pcs.commit(trace): bit-reversed coset LDE with shift GENERATOR/1, then mmcs.commit */
    uint32_t *lde_t = malloc(big * 2 * 4);
    p3o_coset_lde_batch(trace, lde_t, n, 2, log_blowup, gen, 1);
    uint32_t root_t[8];
    const uint32_t *mp[1] = {lde_t}; size_t hh[1] = {big}, ww[1] = {2};
    p3o_tree_t *tree_t = p3o_mmcs_commit_kind(hash, mp, hh, ww, 1, root_t);
    chal_t ch; chal_init(&ch, hash);
    chal_observe(&ch, bb_to_monty(log_n)); /* log_ext_degree */
    chal_observe(&ch, bb_to_monty(log_n)); /* log_degree */
    chal_observe_digest(&ch, root_t);
    chal_observe_n(&ch, pis, 3);
    bb4_t alpha = chal_sample_ext(&ch);
    /* quotient_values on the quotient domain GENERATOR*<g_n> (quotient degree 1) */
    bb4_t apow[FIB_NCONS]; apow[0] = bb4_one();
    for (int k = 1; k < FIB_NCONS; k++) apow[k] = bb4_mul(apow[k - 1], alpha);
    uint32_t *qflat = malloc(n * 4 * 4);
    { uint32_t g = bb_two_adic_generator(log_n), ginv = bb_inv(g);
      uint32_t zh = bb_sub(bb_pow(gen, n), BB_ONE), zh_inv = bb_inv(zh);
      uint32_t *xq = malloc(n * 4);
      power_table(xq, n, gen, g);
      #pragma omp parallel for schedule(static)
      for (size_t i = 0; i < n; i++) {
          uint32_t x = xq[i];
          const uint32_t *loc = lde_t + rev_bits(i, log_n) * 2, *nxt = lde_t + rev_bits((i + 1) & (n - 1), log_n) * 2;
          uint32_t first = bb_mul(zh, bb_inv(bb_sub(x, BB_ONE)));
          uint32_t last = bb_mul(zh, bb_inv(bb_sub(x, ginv)));
          uint32_t trans = bb_sub(x, ginv);
          bb4_t q = bb4_scale(fib_fold_base(loc, nxt, pis, first, last, trans, apow), zh_inv);
          memcpy(qflat + 4 * i, q.c, 16);
      }
      free(xq); }
    /* commit quotient chunk: domain shift = GENERATOR so the LDE shift is GENERATOR/GENERATOR = 1 */
    uint32_t *lde_q = malloc(big * 4 * 4);
    p3o_coset_lde_batch(qflat, lde_q, n, 4, log_blowup, BB_ONE, 1);
    uint32_t root_q[8];
    mp[0] = lde_q; ww[0] = 4;
    p3o_tree_t *tree_q = p3o_mmcs_commit_kind(hash, mp, hh, ww, 1, root_q);
    chal_observe_digest(&ch, root_q);
    bb4_t zeta = chal_sample_ext(&ch);
    bb4_t zeta_next = bb4_scale(zeta, bb_two_adic_generator(log_n));
    /* pcs.open: opened values (observed), then the batching challenge */
    bb4_t t_loc[2], t_nxt[2], q_z[4];
    interpolate_low_coset(lde_t, n, 2, gen, zeta, t_loc);
    interpolate_low_coset(lde_t, n, 2, gen, zeta_next, t_nxt);
    interpolate_low_coset(lde_q, n, 4, gen, zeta, q_z);
    for (int i = 0; i < 2; i++) chal_observe_ext(&ch, t_loc[i]);
    for (int i = 0; i < 2; i++) chal_observe_ext(&ch, t_nxt[i]);
    for (int i = 0; i < 4; i++) chal_observe_ext(&ch, q_z[i]);
    bb4_t al = chal_sample_ext(&ch);
    bb4_t alp[8]; alp[0] = bb4_one(); for (int k = 1; k < 8; k++) alp[k] = bb4_mul(alp[k - 1], al);
    /* reduced openings over the LDE domain GENERATOR*<g_big>, bit-reversed order */
    bb4_t *ro = malloc(big * sizeof(bb4_t));
    { bb4_t ry0 = bb4_zero(), ry1 = bb4_zero(), ry2 = bb4_zero();
      for (int j = 0; j < 2; j++) { ry0 = bb4_add(ry0, bb4_mul(alp[j], t_loc[j])); ry1 = bb4_add(ry1, bb4_mul(alp[j], t_nxt[j])); }
      for (int j = 0; j < 4; j++) ry2 = bb4_add(ry2, bb4_mul(alp[j], q_z[j]));
      uint32_t g = bb_two_adic_generator(log_big);
      uint32_t *xs = malloc(big * 4);
      power_table(xs, big, gen, g);
      #pragma omp parallel for schedule(static)
      for (size_t i = 0; i < big; i++) {
          uint32_t xi = xs[rev_bits(i, log_big)];
          bb4_t rt = bb4_zero(), rq = bb4_zero();
          for (int j = 0; j < 2; j++) rt = bb4_add(rt, bb4_scale(alp[j], lde_t[2 * i + j]));
          for (int j = 0; j < 4; j++) rq = bb4_add(rq, bb4_scale(alp[j], lde_q[4 * i + j]));
          bb4_t d0 = bb4_inv(bb4_sub(zeta, bb4_from_base(xi))), d1 = bb4_inv(bb4_sub(zeta_next, bb4_from_base(xi)));
          bb4_t r = bb4_mul(bb4_sub(ry0, rt), d0);
          r = bb4_add(r, bb4_mul(alp[2], bb4_mul(bb4_sub(ry1, rt), d1)));
          r = bb4_add(r, bb4_mul(alp[4], bb4_mul(bb4_sub(ry2, rq), d0)));
          ro[i] = r;
      }
      free(xs); }
    /* FRI commit phase */
    size_t final_len = ((size_t)1 << log_blowup) << log_final_poly_len;
    unsigned n_rounds = 0;
    for (size_t l = big; l > final_len; l >>= 1) n_rounds++;
    p3o_tree_t **ftrees = malloc((n_rounds + 1) * sizeof *ftrees);
    bb4_t **flayers = malloc((n_rounds + 1) * sizeof *flayers);
    uint32_t (*froots)[8] = malloc((n_rounds + 1) * 32);
    bb4_t *folded = ro; size_t flen = big;
    for (unsigned r = 0; r < n_rounds; r++) {
        flayers[r] = folded;
        mp[0] = (const uint32_t *)folded; hh[0] = flen / 2; ww[0] = 8; /* ExtensionMmcs: width-2 ext rows flattened */
        ftrees[r] = p3o_mmcs_commit_kind(hash, mp, hh, ww, 1, froots[r]);
        chal_observe_digest(&ch, froots[r]);
        bb4_t beta = chal_sample_ext(&ch);
        bb4_t *next = malloc((flen / 2) * sizeof(bb4_t));
        fold_matrix(folded, flen, beta, next);
        folded = next; flen /= 2;
    }
    /* final polynomial: truncate, un-bit-reverse, idft (coefficients); observed */
    size_t fpl = (size_t)1 << log_final_poly_len;
    bb4_t *fpoly = malloc(fpl * sizeof(bb4_t));
    { uint32_t *ev = malloc(fpl * 4 * 4), *co = malloc(fpl * 4 * 4);
      for (size_t i = 0; i < fpl; i++) memcpy(ev + 4 * i, folded[rev_bits(i, log_final_poly_len)].c, 16);
      p3o_idft_batch(ev, co, fpl, 4);
      for (size_t i = 0; i < fpl; i++) { memcpy(fpoly[i].c, co + 4 * i, 16); chal_observe_ext(&ch, fpoly[i]); }
      free(ev); free(co); }
    uint32_t witness = chal_grind(&ch, pow_bits);
    /* ---- serialise ---- */
    put_u32(&pf, 0x42463350u); put_u32(&pf, 1); put_u32(&pf, log_n);
    put_words(&pf, root_t, 8); put_words(&pf, root_q, 8);
    put_u32(&pf, 2); for (int i = 0; i < 2; i++) put_words(&pf, t_loc[i].c, 4);
    put_u32(&pf, 2); for (int i = 0; i < 2; i++) put_words(&pf, t_nxt[i].c, 4);
    put_u32(&pf, 1); put_u32(&pf, 4); for (int i = 0; i < 4; i++) put_words(&pf, q_z[i].c, 4);
    put_u32(&pf, n_rounds); for (unsigned r = 0; r < n_rounds; r++) put_words(&pf, froots[r], 8);
    put_u32(&pf, num_queries);
    uint32_t *path = malloc((log_big + 1) * 32), rowbuf[8];
    for (unsigned q = 0; q < num_queries; q++) {
        size_t index = chal_sample_bits(&ch, log_big);
        put_u32(&pf, 2); /* input_proof: one BatchOpening per commitment round */
        p3o_mmcs_open_batch(tree_t, index, rowbuf, path);
        put_u32(&pf, 1); put_u32(&pf, 2); put_words(&pf, rowbuf, 2); put_path(&pf, path, log_big);
        p3o_mmcs_open_batch(tree_q, index, rowbuf, path);
        put_u32(&pf, 1); put_u32(&pf, 4); put_words(&pf, rowbuf, 4); put_path(&pf, path, log_big);
        put_u32(&pf, n_rounds);
        for (unsigned r = 0; r < n_rounds; r++) {
            size_t idx = index >> r, pair = idx >> 1;
            p3o_mmcs_open_batch(ftrees[r], pair, rowbuf, path);
            put_words(&pf, rowbuf + 4 * ((idx ^ 1) & 1), 4); /* sibling_value */
            put_path(&pf, path, log_big - 1 - r);
        }
    }
    put_u32(&pf, (uint32_t)fpl); for (size_t i = 0; i < fpl; i++) put_words(&pf, fpoly[i].c, 4);
    put_u32(&pf, witness);
    /* cleanup */
    for (unsigned r = 0; r < n_rounds; r++) { p3o_mmcs_free(ftrees[r]); if (r) free(flayers[r]); }
    if (n_rounds) free(folded);
    free(ro); free(ftrees); free(flayers); free(froots); free(fpoly); free(path);
    p3o_mmcs_free(tree_t); p3o_mmcs_free(tree_q); chal_free(&ch);
    free(trace); free(lde_t); free(qflat); free(lde_q);
    *out = pf.p; *out_len = pf.len;
    return 0;
}
int p3o_prove_fib_air(uint64_t a, uint64_t b, unsigned log_n, unsigned log_blowup, unsigned log_final_poly_len,
                      unsigned num_queries, unsigned pow_bits, uint8_t **out, size_t *out_len) {
    return p3o_prove_fib_air_hash(0, a, b, log_n, log_blowup, log_final_poly_len, num_queries, pow_bits, out, out_len);
}
void p3o_free(void *p) { free(p); }

/* p3_uni_stark::verify + TwoAdicFriPcs::verify + p3_fri::verifier for FibonacciAir.
 * 0 = accept; positive codes name the failed check. */
int p3o_verify_fib_air_hash(int hash, const uint8_t *proof, size_t len, uint64_t a, uint64_t b, uint64_t x_pub, unsigned log_n,
                            unsigned log_blowup, unsigned log_final_poly_len, unsigned num_queries, unsigned pow_bits) {
    if (hash != 0 && hash != 1) return -1;
    rd_t rd = {proof, len, 0, 0};
    const unsigned log_big = log_n + log_blowup;
    const size_t n = (size_t)1 << log_n;
    const uint32_t gen = bb_to_monty(BB_GENERATOR_CANON);
    if (get_u32(&rd) != 0x42463350u || get_u32(&rd) != 1) return 1;
    if (get_u32(&rd) != log_n) return 2;
    uint32_t root_t[8], root_q[8];
    get_digests(&rd, hash, root_t, 1); get_digests(&rd, hash, root_q, 1);
    bb4_t t_loc[2], t_nxt[2], q_z[4];
    if (get_u32(&rd) != 2) return 3;
    for (int i = 0; i < 2; i++) t_loc[i] = get_ext(&rd);
    if (get_u32(&rd) != 2) return 3;
    for (int i = 0; i < 2; i++) t_nxt[i] = get_ext(&rd);
    if (get_u32(&rd) != 1 || get_u32(&rd) != 4) return 3;
    for (int i = 0; i < 4; i++) q_z[i] = get_ext(&rd);
    if (rd.bad) return 4;
    uint32_t pis[3] = {bb_to_monty((uint32_t)(a % BB_P)), bb_to_monty((uint32_t)(b % BB_P)), bb_to_monty((uint32_t)(x_pub % BB_P))};
    chal_t ch; chal_init(&ch, hash);
    chal_observe(&ch, bb_to_monty(log_n)); chal_observe(&ch, bb_to_monty(log_n));
    chal_observe_digest(&ch, root_t); chal_observe_n(&ch, pis, 3);
    bb4_t alpha = chal_sample_ext(&ch);
    chal_observe_digest(&ch, root_q);
    bb4_t zeta = chal_sample_ext(&ch);
    uint32_t g_n = bb_two_adic_generator(log_n);
    bb4_t zeta_next = bb4_scale(zeta, g_n);
    /* constraints at zeta: selectors_at_point on the trace domain (shift 1) */
    { bb4_t zh = bb4_sub(bb4_pow(zeta, n), bb4_one());
      bb4_t ginv = bb4_from_base(bb_inv(g_n));
      bb4_t first = bb4_mul(zh, bb4_inv(bb4_sub(zeta, bb4_one())));
      bb4_t last = bb4_mul(zh, bb4_inv(bb4_sub(zeta, ginv)));
      bb4_t trans = bb4_sub(zeta, ginv);
      bb4_t c[FIB_NCONS] = {
          bb4_mul(first, bb4_sub(t_loc[0], bb4_from_base(pis[0]))), bb4_mul(first, bb4_sub(t_loc[1], bb4_from_base(pis[1]))),
          bb4_mul(trans, bb4_sub(t_loc[1], t_nxt[0])), bb4_mul(trans, bb4_sub(bb4_add(t_loc[0], t_loc[1]), t_nxt[1])),
          bb4_mul(last, bb4_sub(t_loc[1], bb4_from_base(pis[2])))};
      bb4_t folded = bb4_zero();
      for (int k = 0; k < FIB_NCONS; k++) folded = bb4_add(bb4_mul(folded, alpha), c[k]); /* VerifierConstraintFolder: Horner */
      /* quotient(zeta) = sum_e basis_e * chunk[e]; basis_e = x^e */
      bb4_t quot = bb4_zero();
      for (int e = 0; e < 4; e++) { bb4_t be = bb4_zero(); be.c[e] = BB_ONE; quot = bb4_add(quot, bb4_mul(be, q_z[e])); }
      if (!bb4_eq(bb4_mul(folded, bb4_inv(zh)), quot)) { chal_free(&ch); return 10; } /* OodEvaluationMismatch */ }
    /* pcs.verify */
    for (int i = 0; i < 2; i++) chal_observe_ext(&ch, t_loc[i]);
    for (int i = 0; i < 2; i++) chal_observe_ext(&ch, t_nxt[i]);
    for (int i = 0; i < 4; i++) chal_observe_ext(&ch, q_z[i]);
    bb4_t al = chal_sample_ext(&ch);
    bb4_t alp[8]; alp[0] = bb4_one(); for (int k = 1; k < 8; k++) alp[k] = bb4_mul(alp[k - 1], al);
    unsigned n_rounds = get_u32(&rd);
    if (rd.bad || n_rounds != log_big - log_blowup - log_final_poly_len) { chal_free(&ch); return 5; }
    uint32_t (*froots)[8] = malloc((n_rounds + 1) * 32);
    bb4_t *betas = malloc((n_rounds + 1) * sizeof(bb4_t));
    for (unsigned r = 0; r < n_rounds; r++) get_digests(&rd, hash, froots[r], 1);
    for (unsigned r = 0; r < n_rounds; r++) { chal_observe_digest(&ch, froots[r]); betas[r] = chal_sample_ext(&ch); }
    if (get_u32(&rd) != num_queries) { free(froots); free(betas); chal_free(&ch); return 6; }
    /* the final polynomial and the witness sit after the queries: find them first */
    size_t qstart = rd.pos;
    for (unsigned q = 0; q < num_queries && !rd.bad; q++) {
        if (get_u32(&rd) != 2) rd.bad = 1;
        for (int m = 0; m < 2 && !rd.bad; m++) { get_u32(&rd); uint32_t w = get_u32(&rd); rd.pos += 4 * (size_t)w; uint32_t pl = get_u32(&rd); rd.pos += 32 * (size_t)pl; }
        uint32_t nr = get_u32(&rd);
        for (uint32_t r = 0; r < nr && !rd.bad; r++) { rd.pos += 16; uint32_t pl = get_u32(&rd); rd.pos += 32 * (size_t)pl; }
    }
    uint32_t fpl = get_u32(&rd);
    if (rd.bad || fpl != (1u << log_final_poly_len)) { free(froots); free(betas); chal_free(&ch); return 7; }
    bb4_t *fpoly = malloc(fpl * sizeof(bb4_t));
    for (uint32_t i = 0; i < fpl; i++) { fpoly[i] = get_ext(&rd); chal_observe_ext(&ch, fpoly[i]); }
    uint32_t witness = get_u32(&rd);
    int rc = 0;
    if (rd.bad || rd.pos != len) rc = 8;
    if (!rc && !chal_check_witness(&ch, pow_bits, witness)) rc = 11; /* InvalidPowWitness */
    rd.pos = qstart;
    uint32_t *path = malloc((log_big + 1) * 32);
    for (unsigned q = 0; q < num_queries && !rc; q++) {
        size_t index = chal_sample_bits(&ch, log_big);
        uint32_t trow[2], qrow[4];
        size_t dims_h[1] = {(size_t)1 << log_big}, dims_w[1];
        if (get_u32(&rd) != 2) { rc = 12; break; }
        if (get_u32(&rd) != 1 || get_u32(&rd) != 2) { rc = 12; break; }
        get_words(&rd, trow, 2);
        if (get_u32(&rd) != log_big) { rc = 12; break; } get_digests(&rd, hash, path, log_big);
        dims_w[0] = 2;
        if (p3o_mmcs_verify_batch_kind(hash, root_t, dims_h, dims_w, 1, index, trow, path, log_big)) { rc = 13; break; }
        if (get_u32(&rd) != 1 || get_u32(&rd) != 4) { rc = 12; break; }
        get_words(&rd, qrow, 4);
        if (get_u32(&rd) != log_big) { rc = 12; break; } get_digests(&rd, hash, path, log_big);
        dims_w[0] = 4;
        if (p3o_mmcs_verify_batch_kind(hash, root_q, dims_h, dims_w, 1, index, qrow, path, log_big)) { rc = 13; break; }
        /* reduced opening at the queried point x = GENERATOR * g_big^bitrev(index) */
        uint32_t xi = bb_mul(gen, bb_pow(bb_two_adic_generator(log_big), rev_bits(index, log_big)));
        bb4_t d0 = bb4_inv(bb4_sub(zeta, bb4_from_base(xi))), d1 = bb4_inv(bb4_sub(zeta_next, bb4_from_base(xi)));
        bb4_t ro = bb4_zero(); int k = 0;
        for (int j = 0; j < 2; j++, k++) ro = bb4_add(ro, bb4_mul(alp[k], bb4_mul(bb4_sub(t_loc[j], bb4_from_base(trow[j])), d0)));
        for (int j = 0; j < 2; j++, k++) ro = bb4_add(ro, bb4_mul(alp[k], bb4_mul(bb4_sub(t_nxt[j], bb4_from_base(trow[j])), d1)));
        for (int j = 0; j < 4; j++, k++) ro = bb4_add(ro, bb4_mul(alp[k], bb4_mul(bb4_sub(q_z[j], bb4_from_base(qrow[j])), d0)));
        /* fold chain */
        if (get_u32(&rd) != n_rounds) { rc = 12; break; }
        bb4_t folded = ro; size_t idx = index;
        for (unsigned r = 0; r < n_rounds; r++) {
            unsigned log_folded_height = log_big - 1 - r;
            bb4_t sib = get_ext(&rd);
            if (get_u32(&rd) != log_folded_height) { rc = 12; break; }
            get_digests(&rd, hash, path, log_folded_height);
            bb4_t ev[2]; ev[idx & 1] = folded; ev[(idx & 1) ^ 1] = sib;
            size_t pair = idx >> 1;
            size_t dh[1] = {(size_t)1 << log_folded_height}, dw[1] = {8};
            if (p3o_mmcs_verify_batch_kind(hash, froots[r], dh, dw, 1, pair, (const uint32_t *)ev, path, log_folded_height)) { rc = 14; break; }
            /* fold_row: interpolate (s, e0), (-s, e1) at beta, s = g_{h+1}^bitrev(pair) */
            uint32_t s = bb_pow(bb_two_adic_generator(log_folded_height + 1), rev_bits(pair, log_folded_height));
            bb4_t num = bb4_mul(bb4_sub(betas[r], bb4_from_base(s)), bb4_sub(ev[1], ev[0]));
            folded = bb4_add(ev[0], bb4_scale(num, bb_inv(bb_sub(bb_neg(s), s))));
            idx = pair;
        }
        if (rc) break;
        /* final polynomial (coefficients) at the point of the folded index */
        unsigned log_final_height = log_blowup + log_final_poly_len;
        uint32_t xf = bb_pow(bb_two_adic_generator(log_final_height), rev_bits(idx, log_final_height));
        bb4_t ev = bb4_zero();
        for (uint32_t i = fpl; i-- > 0;) ev = bb4_add(bb4_scale(ev, xf), fpoly[i]);
        if (!bb4_eq(ev, folded)) rc = 15; /* FinalPolyMismatch */
    }
    if (rd.bad && !rc) rc = 9;
    free(path); free(fpoly); free(froots); free(betas); chal_free(&ch);
    return rc;
}

int p3o_verify_fib_air(const uint8_t *proof, size_t len, uint64_t a, uint64_t b, uint64_t x_pub, unsigned log_n,
                       unsigned log_blowup, unsigned log_final_poly_len, unsigned num_queries, unsigned pow_bits) {
    return p3o_verify_fib_air_hash(0, proof, len, a, b, x_pub, log_n, log_blowup, log_final_poly_len, num_queries, pow_bits);
}
