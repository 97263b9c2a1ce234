/* ORACLE — TEST INFRASTRUCTURE ONLY (see bb31.h).
 * CPU restatement of the HIDING half of the reference's configuration (native/src/fib_air.rs:40-65):
 *   MerkleTreeHidingMmcs<.., SmallRng, DIGEST 4 (u64) / 8 (field), SALT_ELEMS 4> with SmallRng::seed_from_u64(1)
 *   HidingFriPcs::new(dft, val_mmcs, fri_params, num_random_codewords = 4, SmallRng::seed_from_u64(1))
 *   p3_uni_stark::prove / verify with SC::Pcs::ZK = true
 * for either hash configuration (hash 0: Poseidon2 + DuplexChallenger; 1: the reference's Keccak hashes +
 * SerializingChallenger32).  All of it lives in the ABSENT crates p3-fri / p3-merkle-tree / p3-uni-stark 0.4.2 and
 * rand 0.9.2: [UPSTREAM-RECALL] in STRUCTURE, PARITY UNPINNED — in particular the draw order of the three random
 * streams and the exact form of the quotient-chunk blinding are this file's statement, not a checked copy of upstream.
 *
 * The protocol as stated here (h = 2^log_n trace rows, w = 2, NRC = 4 random codewords, SALT = 4, D = 4):
 *   streams   three xoshiro256++ streams, each SmallRng::seed_from_u64(seed): `mmcs` (input MMCS salts), `fri` (the FRI
 *             MMCS is built from a CLONE of the input MMCS, fib_air.rs:59, so its salts restart the stream) and `pcs`.
 *   trace     HidingFriPcs::commit: the h x w trace becomes a 2h x (w + NRC) matrix — per trace row, w + 2 NRC draws
 *             from `pcs`: the row keeps its w values followed by the first NRC draws, the next row is the remaining
 *             w + NRC draws — committed over the domain of size 2h (shift 1): the interpolant agrees with the trace on
 *             the original domain (the even points).  Bit-reversed coset LDE (shift GENERATOR), then the hiding MMCS:
 *             every matrix gets a height x SALT matrix of `mmcs` draws hashed into its leaf rows.
 *   quotient  on the disjoint coset GENERATOR <g_4h> (log_quotient_degree 1 + 1 for zk): 4 chunks q_c on the cosets
 *             D_c = s_c <g_h>, s_c = GENERATOR g_4h^c; chunk c is blinded to q_c + Z_{D_c} t_c (degree < 2h) with t_c a
 *             random polynomial of degree < h (coefficients: h x D draws from `pcs`, c = 0, 1, 2) and
 *             t_3 = -k_3 sum_{c<3} t_c / k_c, k_c = prod_{j != c} Z_{D_j}(s_c), so that the blinding cancels in the
 *             verifier's recomposition sum_c zps_c(zeta) q_c(zeta).  Four matrices of width D in one hiding commitment.
 *   random    get_opt_randomization_poly_commitment: a 2h x (NRC + D) matrix of `pcs` draws, committed like the trace.
 *   opening   rounds [random @ zeta], [trace @ zeta, zeta g_h], [chunks @ zeta]; every column of every round enters the
 *             FRI batch, so the random columns mask the batched quotient.
 * Wire format: version 2 of stark.c's ("P3FB", 2, log_n, three roots, opened values per round, FRI commit phase,
 * queries with salts next to every opened row, final polynomial, witness). */
#include "stark_common.h"

#define HID_W 2
#define HID_NRC 4
#define HID_SALT 4
#define HID_D 4
#define HID_TW (HID_W + HID_NRC)   /* randomized trace width */
#define HID_RW (HID_NRC + HID_D)   /* randomization-polynomial matrix width */
#define HID_CHUNKS 4

typedef struct { uint64_t s[4]; } rng_t;

static uint32_t *rand_matrix(rng_t *r, size_t h, size_t w) {
    uint32_t *m = malloc((h * w + 1) * 4);
    p3o_rng_fill_field(r->s, m, h * w);
    return m;
}

/* commitment over (matrix, salt) pairs: leaf row = m0 || s0 || m1 || s1 ... (MerkleTreeHidingMmcs::commit wraps every
 * input in a HorizontalPair with its salt matrix) */
typedef struct { p3o_tree_t *tree; uint32_t **salts; size_t n; const uint32_t **mats; size_t *widths; size_t height; } hcommit_t;
static hcommit_t hiding_commit(int hash, rng_t *rng, const uint32_t *const *mats, const size_t *widths, size_t n, size_t height, uint32_t root[8]) {
    hcommit_t c; c.n = n; c.height = height;
    c.salts = malloc(n * sizeof *c.salts); c.mats = malloc(n * sizeof *c.mats); c.widths = malloc(n * sizeof *c.widths);
    const uint32_t **mp = malloc(2 * n * sizeof *mp); size_t *hh = malloc(2 * n * sizeof *hh), *ww = malloc(2 * n * sizeof *ww);
    for (size_t i = 0; i < n; i++) {
        c.salts[i] = rand_matrix(rng, height, HID_SALT);
        c.mats[i] = mats[i]; c.widths[i] = widths[i];
        mp[2 * i] = mats[i]; hh[2 * i] = height; ww[2 * i] = widths[i];
        mp[2 * i + 1] = c.salts[i]; hh[2 * i + 1] = height; ww[2 * i + 1] = HID_SALT;
    }
    c.tree = p3o_mmcs_commit_kind(hash, mp, hh, ww, 2 * n, root);
    free(mp); free(hh); free(ww);
    return c;
}
static void hcommit_free(hcommit_t *c) {
    p3o_mmcs_free(c->tree);
    for (size_t i = 0; i < c->n; i++) free(c->salts[i]);
    free(c->salts); free(c->mats); free(c->widths);
}
/* BatchOpening of a hiding commitment: values per matrix, then the salts per matrix, then the sibling path */
static void put_hiding_opening(buf_t *pf, const hcommit_t *c, size_t index, unsigned depth) {
    put_u32(pf, (uint32_t)c->n);
    for (size_t m = 0; m < c->n; m++) { put_u32(pf, (uint32_t)c->widths[m]); put_words(pf, c->mats[m] + index * c->widths[m], c->widths[m]); }
    for (size_t m = 0; m < c->n; m++) { put_u32(pf, HID_SALT); put_words(pf, c->salts[m] + index * HID_SALT, HID_SALT); }
    size_t tot = 0; for (size_t m = 0; m < c->n; m++) tot += c->widths[m] + HID_SALT;
    uint32_t *rows = malloc(tot * 4), *path = malloc((depth + 1) * 32);
    p3o_mmcs_open_batch(c->tree, index, rows, path);
    put_path(pf, path, depth);
    free(rows); free(path);
}

/* the four chunk cosets of the quotient domain: s_c^h = GENERATOR^h w4^c, k_c = prod_{j != c} (s_c^h - s_j^h) */
static void chunk_constants(unsigned log_n, uint32_t sh[HID_CHUNKS], uint32_t kc[HID_CHUNKS]) {
    const size_t h = (size_t)1 << log_n;
    const uint32_t gh = bb_pow(bb_to_monty(BB_GENERATOR_CANON), h), w4 = bb_two_adic_generator(2);
    uint32_t p = BB_ONE;
    for (int c = 0; c < HID_CHUNKS; c++) { sh[c] = bb_mul(gh, p); p = bb_mul(p, w4); }
    for (int c = 0; c < HID_CHUNKS; c++) {
        uint32_t k = BB_ONE;
        for (int j = 0; j < HID_CHUNKS; j++) if (j != c) k = bb_mul(k, bb_sub(sh[c], sh[j]));
        kc[c] = k;
    }
}

int p3o_prove_fib_air_hiding(int hash, uint64_t a, uint64_t b, unsigned log_n, unsigned log_blowup, unsigned log_final_poly_len,
                             unsigned num_queries, unsigned pow_bits, uint64_t seed, uint8_t **out, size_t *out_len) {
    if (hash != 0 && hash != 1) return -1;
    const unsigned log_ext = log_n + 1, log_big = log_ext + log_blowup;
    if (log_n < 1 || log_blowup < 1 || log_big > BB_TWO_ADICITY) return -1;
    if (log_final_poly_len >= log_ext) return -1;
    const size_t h = (size_t)1 << log_n, h2 = 2 * h, big = h2 << log_blowup, qn = 4 * h;
    const uint32_t gen = bb_to_monty(BB_GENERATOR_CANON);
    rng_t rng_mmcs, rng_fri, rng_pcs;
    p3o_rng_seed_from_u64(rng_mmcs.s, seed); p3o_rng_seed_from_u64(rng_fri.s, seed); p3o_rng_seed_from_u64(rng_pcs.s, seed);
    buf_t pf = {0};
    /* trace (fib_air.rs:266-284) and its randomization */
    uint32_t *rt = malloc(h2 * HID_TW * 4);
    uint32_t pis[3];
    { uint32_t l = bb_to_monty((uint32_t)(a % BB_P)), r = bb_to_monty((uint32_t)(b % BB_P));
      pis[0] = l; pis[1] = r;
      for (size_t i = 0; i < h; i++) {
          uint32_t d[HID_W + 2 * HID_NRC];
          p3o_rng_fill_field(rng_pcs.s, d, HID_W + 2 * HID_NRC);
          uint32_t *even = rt + (2 * i) * HID_TW, *odd = even + HID_TW;
          even[0] = l; even[1] = r;
          memcpy(even + HID_W, d, HID_NRC * 4);
          memcpy(odd, d + HID_NRC, HID_TW * 4);
          pis[2] = r;
          uint32_t t = bb_add(l, r); l = r; r = t;
      } }
    uint32_t *lde_t = malloc(big * HID_TW * 4);
    p3o_coset_lde_batch(rt, lde_t, h2, HID_TW, log_blowup, gen, 1);
    uint32_t root_t[8], root_q[8], root_r[8];
    const uint32_t *mp1[1] = {lde_t}; size_t ww1[1] = {HID_TW};
    hcommit_t ct = hiding_commit(hash, &rng_mmcs, mp1, ww1, 1, big, root_t);
    chal_t ch; chal_init(&ch, hash);
    chal_observe(&ch, bb_to_monty(log_ext)); /* log_ext_degree = log_degree + is_zk */
    chal_observe(&ch, bb_to_monty(log_n));   /* log_degree */
    chal_observe_digest(&ch, root_t);
    chal_observe_n(&ch, pis, 3);
    bb4_t alpha = chal_sample_ext(&ch);
    /* quotient on GENERATOR <g_4h> (the first 4h rows of the bit-reversed LDE), constraints vanish on the ORIGINAL domain */
    bb4_t apow[FIB_NCONS]; apow[0] = bb4_one();
    for (int k = 1; k < FIB_NCONS; k++) apow[k] = bb4_mul(apow[k - 1], alpha);
    const unsigned log_q = log_n + 2;
    uint32_t *qchunk[HID_CHUNKS];
    for (int c = 0; c < HID_CHUNKS; c++) qchunk[c] = malloc(h * HID_D * 4);
    { uint32_t g4 = bb_two_adic_generator(log_q), gh_inv = bb_inv(bb_two_adic_generator(log_n));
      uint32_t *xq = malloc(qn * 4);
      power_table(xq, qn, gen, g4);
      #pragma omp parallel for schedule(static)
      for (size_t i = 0; i < qn; i++) {
          uint32_t x = xq[i];
          const uint32_t *loc = lde_t + rev_bits(i, log_q) * HID_TW, *nxt = lde_t + rev_bits((i + 4) & (qn - 1), log_q) * HID_TW;
          uint32_t zh = bb_sub(bb_pow(x, h), BB_ONE);
          uint32_t first = bb_mul(zh, bb_inv(bb_sub(x, BB_ONE)));
          uint32_t last = bb_mul(zh, bb_inv(bb_sub(x, gh_inv)));
          uint32_t trans = bb_sub(x, gh_inv);
          bb4_t q = bb4_scale(fib_fold_base(loc, nxt, pis, first, last, trans, apow), bb_inv(zh));
          memcpy(qchunk[i & 3] + HID_D * (i >> 2), q.c, 16); /* split_evals: chunk c takes rows c, c + 4, ... */
      }
      free(xq); }
    /* blinded chunk polynomials q_c + (X^h - s_c^h) t_c as coefficient vectors of length 2h, then evaluated on the LDE coset */
    uint32_t sh[HID_CHUNKS], kc[HID_CHUNKS];
    chunk_constants(log_n, sh, kc);
    uint32_t *tcoef[HID_CHUNKS];
    for (int c = 0; c < HID_CHUNKS - 1; c++) tcoef[c] = rand_matrix(&rng_pcs, h, HID_D);
    tcoef[HID_CHUNKS - 1] = malloc(h * HID_D * 4);
    { uint32_t kinv[HID_CHUNKS]; for (int c = 0; c < HID_CHUNKS; c++) kinv[c] = bb_inv(kc[c]);
      #pragma omp parallel for schedule(static) if (h >= 4096)
      for (size_t i = 0; i < h * HID_D; i++) {
          uint32_t s = 0;
          for (int c = 0; c < HID_CHUNKS - 1; c++) s = bb_add(s, bb_mul(tcoef[c][i], kinv[c]));
          tcoef[HID_CHUNKS - 1][i] = bb_neg(bb_mul(kc[HID_CHUNKS - 1], s));
      } }
    uint32_t *lde_q[HID_CHUNKS];
    for (int c = 0; c < HID_CHUNKS; c++) {
        uint32_t *co = malloc(h * HID_D * 4), *ext = calloc(big * HID_D, 4), *nat = malloc(big * HID_D * 4);
        p3o_idft_batch(qchunk[c], co, h, HID_D);  /* coefficients of q_c(s_c X) */
        uint32_t s_c = bb_mul(gen, bb_pow(bb_two_adic_generator(log_q), (uint64_t)c)), sinv = bb_inv(s_c);
        uint32_t *spw = malloc(h * 4);
        power_table(spw, h, BB_ONE, sinv);
        #pragma omp parallel for schedule(static) if (h >= 4096)
        for (size_t k = 0; k < h; k++) {
            for (int j = 0; j < HID_D; j++) {
                uint32_t ak = bb_mul(co[k * HID_D + j], spw[k]);
                ext[k * HID_D + j] = bb_sub(ak, bb_mul(sh[c], tcoef[c][k * HID_D + j]));
                ext[(h + k) * HID_D + j] = tcoef[c][k * HID_D + j];
            }
        }
        free(spw);
        p3o_coset_dft_batch(ext, nat, big, HID_D, gen);
        lde_q[c] = malloc(big * HID_D * 4);
        p3o_bit_reverse_rows(lde_q[c], nat, big, HID_D);
        free(co); free(ext); free(nat);
    }
    const uint32_t *mpq[HID_CHUNKS]; size_t wwq[HID_CHUNKS];
    for (int c = 0; c < HID_CHUNKS; c++) { mpq[c] = lde_q[c]; wwq[c] = HID_D; }
    hcommit_t cq = hiding_commit(hash, &rng_mmcs, mpq, wwq, HID_CHUNKS, big, root_q);
    chal_observe_digest(&ch, root_q);
    /* randomization polynomial commitment */
    uint32_t *rm = rand_matrix(&rng_pcs, h2, HID_RW), *lde_r = malloc(big * HID_RW * 4);
    p3o_coset_lde_batch(rm, lde_r, h2, HID_RW, log_blowup, gen, 1);
    const uint32_t *mpr[1] = {lde_r}; size_t wwr[1] = {HID_RW};
    hcommit_t cr = hiding_commit(hash, &rng_mmcs, mpr, wwr, 1, big, root_r);
    chal_observe_digest(&ch, root_r);
    bb4_t zeta = chal_sample_ext(&ch);
    bb4_t zeta_next = bb4_scale(zeta, bb_two_adic_generator(log_n));
    /* opened values: every committed column is a polynomial of degree < 2h, interpolated from the low coset */
    bb4_t r_z[HID_RW], t_z[HID_TW], t_zn[HID_TW], q_z[HID_CHUNKS][HID_D];
    interpolate_low_coset(lde_r, h2, HID_RW, gen, zeta, r_z);
    interpolate_low_coset(lde_t, h2, HID_TW, gen, zeta, t_z);
    interpolate_low_coset(lde_t, h2, HID_TW, gen, zeta_next, t_zn);
    for (int c = 0; c < HID_CHUNKS; c++) interpolate_low_coset(lde_q[c], h2, HID_D, gen, zeta, q_z[c]);
    for (int i = 0; i < HID_RW; i++) chal_observe_ext(&ch, r_z[i]);
    for (int i = 0; i < HID_TW; i++) chal_observe_ext(&ch, t_z[i]);
    for (int i = 0; i < HID_TW; i++) chal_observe_ext(&ch, t_zn[i]);
    for (int c = 0; c < HID_CHUNKS; c++) for (int i = 0; i < HID_D; i++) chal_observe_ext(&ch, q_z[c][i]);
    bb4_t al = chal_sample_ext(&ch);
    const int NPOW = HID_RW + 2 * HID_TW + HID_CHUNKS * HID_D;
    bb4_t alp[HID_RW + 2 * HID_TW + HID_CHUNKS * HID_D]; alp[0] = bb4_one();
    for (int k = 1; k < NPOW; k++) alp[k] = bb4_mul(alp[k - 1], al);
    /* reduced openings over the LDE domain, committed order */
    bb4_t *ro = malloc(big * sizeof(bb4_t));
    { uint32_t g = bb_two_adic_generator(log_big);
      uint32_t *xs = malloc(big * 4);
      power_table(xs, big, gen, g);
      #pragma omp parallel for schedule(static)
      for (size_t i = 0; i < big; i++) {
          uint32_t xi = xs[rev_bits(i, log_big)];
          bb4_t d0 = bb4_inv(bb4_sub(zeta, bb4_from_base(xi))), d1 = bb4_inv(bb4_sub(zeta_next, bb4_from_base(xi)));
          bb4_t s0 = bb4_zero(), s1 = bb4_zero(); int k = 0;
          for (int j = 0; j < HID_RW; j++, k++) s0 = bb4_add(s0, bb4_mul(alp[k], bb4_sub(r_z[j], bb4_from_base(lde_r[i * HID_RW + j]))));
          for (int j = 0; j < HID_TW; j++, k++) s0 = bb4_add(s0, bb4_mul(alp[k], bb4_sub(t_z[j], bb4_from_base(lde_t[i * HID_TW + j]))));
          for (int j = 0; j < HID_TW; j++, k++) s1 = bb4_add(s1, bb4_mul(alp[k], bb4_sub(t_zn[j], bb4_from_base(lde_t[i * HID_TW + j]))));
          for (int c = 0; c < HID_CHUNKS; c++)
              for (int j = 0; j < HID_D; j++, k++) s0 = bb4_add(s0, bb4_mul(alp[k], bb4_sub(q_z[c][j], bb4_from_base(lde_q[c][i * HID_D + j]))));
          ro[i] = bb4_add(bb4_mul(s0, d0), bb4_mul(s1, d1));
      }
      free(xs); }
    /* FRI commit phase: ExtensionMmcs over the hiding MMCS (salts from the `fri` stream) */
    size_t final_len = ((size_t)1 << log_blowup) << log_final_poly_len;
    unsigned n_rounds = 0;
    for (size_t l = big; l > final_len; l >>= 1) n_rounds++;
    hcommit_t *fc = malloc((n_rounds + 1) * sizeof *fc);
    bb4_t **flayers = malloc((n_rounds + 1) * sizeof *flayers);
    uint32_t (*froots)[8] = malloc((n_rounds + 1) * 32);
    bb4_t *folded = ro; size_t flen = big;
    for (unsigned r = 0; r < n_rounds; r++) {
        flayers[r] = folded;
        const uint32_t *mpf[1] = {(const uint32_t *)folded}; size_t wwf[1] = {8};
        fc[r] = hiding_commit(hash, &rng_fri, mpf, wwf, 1, flen / 2, froots[r]);
        chal_observe_digest(&ch, froots[r]);
        bb4_t beta = chal_sample_ext(&ch);
        bb4_t *next = malloc((flen / 2) * sizeof(bb4_t));
        fold_matrix(folded, flen, beta, next);
        folded = next; flen /= 2;
    }
    size_t fpl = (size_t)1 << log_final_poly_len;
    bb4_t *fpoly = malloc(fpl * sizeof(bb4_t));
    { uint32_t *ev = malloc(fpl * 4 * 4), *co = malloc(fpl * 4 * 4);
      for (size_t i = 0; i < fpl; i++) memcpy(ev + 4 * i, folded[rev_bits(i, log_final_poly_len)].c, 16);
      p3o_idft_batch(ev, co, fpl, 4);
      for (size_t i = 0; i < fpl; i++) { memcpy(fpoly[i].c, co + 4 * i, 16); chal_observe_ext(&ch, fpoly[i]); }
      free(ev); free(co); }
    uint32_t witness = chal_grind(&ch, pow_bits);
    /* ---- serialise ---- */
    put_u32(&pf, 0x42463350u); put_u32(&pf, 2); put_u32(&pf, log_n);
    put_words(&pf, root_t, 8); put_words(&pf, root_q, 8); put_words(&pf, root_r, 8);
    put_u32(&pf, HID_RW); for (int i = 0; i < HID_RW; i++) put_words(&pf, r_z[i].c, 4);
    put_u32(&pf, HID_TW); for (int i = 0; i < HID_TW; i++) put_words(&pf, t_z[i].c, 4);
    put_u32(&pf, HID_TW); for (int i = 0; i < HID_TW; i++) put_words(&pf, t_zn[i].c, 4);
    put_u32(&pf, HID_CHUNKS);
    for (int c = 0; c < HID_CHUNKS; c++) { put_u32(&pf, HID_D); for (int i = 0; i < HID_D; i++) put_words(&pf, q_z[c][i].c, 4); }
    put_u32(&pf, n_rounds); for (unsigned r = 0; r < n_rounds; r++) put_words(&pf, froots[r], 8);
    put_u32(&pf, num_queries);
    for (unsigned q = 0; q < num_queries; q++) {
        size_t index = chal_sample_bits(&ch, log_big);
        put_u32(&pf, 3); /* input_proof: one BatchOpening per commitment round, in opening order */
        put_hiding_opening(&pf, &cr, index, log_big);
        put_hiding_opening(&pf, &ct, index, log_big);
        put_hiding_opening(&pf, &cq, index, log_big);
        put_u32(&pf, n_rounds);
        for (unsigned r = 0; r < n_rounds; r++) {
            size_t idx = index >> r, pair = idx >> 1;
            const uint32_t *row = (const uint32_t *)flayers[r] + pair * 8;
            put_words(&pf, row + 4 * ((idx ^ 1) & 1), 4); /* sibling_value */
            put_u32(&pf, HID_SALT); put_words(&pf, fc[r].salts[0] + pair * HID_SALT, HID_SALT);
            uint32_t *rows = malloc((8 + HID_SALT) * 4), *path = malloc((log_big + 1) * 32);
            p3o_mmcs_open_batch(fc[r].tree, pair, rows, path);
            put_path(&pf, path, log_big - 1 - r);
            free(rows); free(path);
        }
    }
    put_u32(&pf, (uint32_t)fpl); for (size_t i = 0; i < fpl; i++) put_words(&pf, fpoly[i].c, 4);
    put_u32(&pf, witness);
    /* cleanup */
    for (unsigned r = 0; r < n_rounds; r++) { hcommit_free(&fc[r]); if (r) free(flayers[r]); }
    if (n_rounds) free(folded);
    free(ro); free(fc); free(flayers); free(froots); free(fpoly);
    hcommit_free(&ct); hcommit_free(&cq); hcommit_free(&cr); chal_free(&ch);
    for (int c = 0; c < HID_CHUNKS; c++) { free(qchunk[c]); free(tcoef[c]); free(lde_q[c]); }
    free(rt); free(lde_t); free(rm); free(lde_r);
    *out = pf.p; *out_len = pf.len;
    return 0;
}

/* ------------------------------------------------------------------ verifier (written independently of the prover above) */
typedef struct { uint32_t n; uint32_t width[HID_CHUNKS]; uint32_t vals[HID_CHUNKS * HID_D + HID_RW]; uint32_t salts[HID_CHUNKS * HID_SALT]; } hopen_t;
/* reads one hiding BatchOpening and checks it against `root`: leaf row = m0 || s0 || m1 || s1 ... */
static int read_check_opening(rd_t *rd, int hash, const uint32_t root[8], size_t index, unsigned depth, uint32_t n_mats,
                              const uint32_t *widths, hopen_t *o, uint32_t *path) {
    if (get_u32(rd) != n_mats) return 12;
    o->n = n_mats;
    size_t off = 0;
    for (uint32_t m = 0; m < n_mats; m++) {
        if (get_u32(rd) != widths[m]) return 12;
        o->width[m] = widths[m];
        get_words(rd, o->vals + off, widths[m]); off += widths[m];
    }
    for (uint32_t m = 0; m < n_mats; m++) { if (get_u32(rd) != HID_SALT) return 12; get_words(rd, o->salts + m * HID_SALT, HID_SALT); }
    if (get_u32(rd) != depth) return 12;
    get_digests(rd, hash, path, depth);
    if (rd->bad) return 9;
    uint32_t row[HID_CHUNKS * (HID_D + HID_SALT) + HID_RW + HID_SALT];
    size_t hh[2 * HID_CHUNKS], ww[2 * HID_CHUNKS], p = 0; off = 0;
    for (uint32_t m = 0; m < n_mats; m++) {
        memcpy(row + p, o->vals + off, widths[m] * 4); p += widths[m]; off += widths[m];
        memcpy(row + p, o->salts + m * HID_SALT, HID_SALT * 4); p += HID_SALT;
        hh[2 * m] = hh[2 * m + 1] = (size_t)1 << depth; ww[2 * m] = widths[m]; ww[2 * m + 1] = HID_SALT;
    }
    return p3o_mmcs_verify_batch_kind(hash, root, hh, ww, 2 * n_mats, index, row, path, depth) ? 13 : 0;
}

int p3o_verify_fib_air_hiding(int hash, const uint8_t *proof, size_t len, uint64_t a, uint64_t b, uint64_t x_pub, unsigned log_n,
                              unsigned log_blowup, unsigned log_final_poly_len, unsigned num_queries, unsigned pow_bits) {
    if (hash != 0 && hash != 1) return -1;
    rd_t rd = {proof, len, 0, 0};
    const unsigned log_ext = log_n + 1, log_big = log_ext + log_blowup;
    if (log_n < 1 || log_blowup < 1 || log_big > BB_TWO_ADICITY || log_final_poly_len >= log_ext) return -1;
    const size_t h = (size_t)1 << log_n;
    const uint32_t gen = bb_to_monty(BB_GENERATOR_CANON);
    if (get_u32(&rd) != 0x42463350u || get_u32(&rd) != 2) return 1;
    if (get_u32(&rd) != log_n) return 2;
    uint32_t root_t[8], root_q[8], root_r[8];
    get_digests(&rd, hash, root_t, 1); get_digests(&rd, hash, root_q, 1); get_digests(&rd, hash, root_r, 1);
    bb4_t r_z[HID_RW], t_z[HID_TW], t_zn[HID_TW], q_z[HID_CHUNKS][HID_D];
    if (get_u32(&rd) != HID_RW) return 3;
    for (int i = 0; i < HID_RW; i++) r_z[i] = get_ext(&rd);
    if (get_u32(&rd) != HID_TW) return 3;
    for (int i = 0; i < HID_TW; i++) t_z[i] = get_ext(&rd);
    if (get_u32(&rd) != HID_TW) return 3;
    for (int i = 0; i < HID_TW; i++) t_zn[i] = get_ext(&rd);
    if (get_u32(&rd) != HID_CHUNKS) return 3;
    for (int c = 0; c < HID_CHUNKS; c++) { if (get_u32(&rd) != HID_D) return 3; for (int i = 0; i < HID_D; i++) q_z[c][i] = get_ext(&rd); }
    if (rd.bad) return 4;
    uint32_t pis[3] = {bb_to_monty((uint32_t)(a % BB_P)), bb_to_monty((uint32_t)(b % BB_P)), bb_to_monty((uint32_t)(x_pub % BB_P))};
    chal_t ch; chal_init(&ch, hash);
    chal_observe(&ch, bb_to_monty(log_ext)); chal_observe(&ch, bb_to_monty(log_n));
    chal_observe_digest(&ch, root_t); chal_observe_n(&ch, pis, 3);
    bb4_t alpha = chal_sample_ext(&ch);
    chal_observe_digest(&ch, root_q);
    chal_observe_digest(&ch, root_r);
    bb4_t zeta = chal_sample_ext(&ch);
    uint32_t g_h = bb_two_adic_generator(log_n);
    bb4_t zeta_next = bb4_scale(zeta, g_h);
    /* constraints at zeta against the recomposed quotient: sum_c zps_c(zeta) * chunk_c(zeta) */
    { bb4_t zh_pow = bb4_pow(zeta, h);
      bb4_t zh = bb4_sub(zh_pow, bb4_one());
      bb4_t ginv = bb4_from_base(bb_inv(g_h));
      bb4_t first = bb4_mul(zh, bb4_inv(bb4_sub(zeta, bb4_one())));
      bb4_t last = bb4_mul(zh, bb4_inv(bb4_sub(zeta, ginv)));
      bb4_t trans = bb4_sub(zeta, ginv);
      bb4_t c[FIB_NCONS] = {
          bb4_mul(first, bb4_sub(t_z[0], bb4_from_base(pis[0]))), bb4_mul(first, bb4_sub(t_z[1], bb4_from_base(pis[1]))),
          bb4_mul(trans, bb4_sub(t_z[1], t_zn[0])), bb4_mul(trans, bb4_sub(bb4_add(t_z[0], t_z[1]), t_zn[1])),
          bb4_mul(last, bb4_sub(t_z[1], bb4_from_base(pis[2])))};
      bb4_t folded = bb4_zero();
      for (int k = 0; k < FIB_NCONS; k++) folded = bb4_add(bb4_mul(folded, alpha), c[k]);
      uint32_t sh[HID_CHUNKS], kc[HID_CHUNKS];
      chunk_constants(log_n, sh, kc);
      bb4_t quot = bb4_zero();
      for (int ci = 0; ci < HID_CHUNKS; ci++) {
          bb4_t zp = bb4_from_base(bb_inv(kc[ci]));
          for (int j = 0; j < HID_CHUNKS; j++) if (j != ci) zp = bb4_mul(zp, bb4_sub(zh_pow, bb4_from_base(sh[j])));
          bb4_t v = bb4_zero();
          for (int e = 0; e < HID_D; e++) { bb4_t be = bb4_zero(); be.c[e] = BB_ONE; v = bb4_add(v, bb4_mul(be, q_z[ci][e])); }
          quot = bb4_add(quot, bb4_mul(zp, v));
      }
      if (!bb4_eq(bb4_mul(folded, bb4_inv(zh)), quot)) { chal_free(&ch); return 10; } /* OodEvaluationMismatch */ }
    for (int i = 0; i < HID_RW; i++) chal_observe_ext(&ch, r_z[i]);
    for (int i = 0; i < HID_TW; i++) chal_observe_ext(&ch, t_z[i]);
    for (int i = 0; i < HID_TW; i++) chal_observe_ext(&ch, t_zn[i]);
    for (int c = 0; c < HID_CHUNKS; c++) for (int i = 0; i < HID_D; i++) chal_observe_ext(&ch, q_z[c][i]);
    bb4_t al = chal_sample_ext(&ch);
    const int NPOW = HID_RW + 2 * HID_TW + HID_CHUNKS * HID_D;
    bb4_t alp[HID_RW + 2 * HID_TW + HID_CHUNKS * HID_D]; alp[0] = bb4_one();
    for (int k = 1; k < NPOW; k++) alp[k] = bb4_mul(alp[k - 1], al);
    unsigned n_rounds = get_u32(&rd);
    if (rd.bad || n_rounds != log_big - log_blowup - log_final_poly_len) { chal_free(&ch); return 5; }
    uint32_t (*froots)[8] = malloc((n_rounds + 1) * 32);
    bb4_t *betas = malloc((n_rounds + 1) * sizeof(bb4_t));
    for (unsigned r = 0; r < n_rounds; r++) get_digests(&rd, hash, froots[r], 1);
    for (unsigned r = 0; r < n_rounds; r++) { chal_observe_digest(&ch, froots[r]); betas[r] = chal_sample_ext(&ch); }
    if (get_u32(&rd) != num_queries) { free(froots); free(betas); chal_free(&ch); return 6; }
    /* the final polynomial and the witness sit after the queries: every query has the same length */
    size_t qstart = rd.pos;
    { size_t per_open[3] = {0, 0, 0}; const uint32_t nm[3] = {1, 1, HID_CHUNKS}, wsum[3] = {HID_RW, HID_TW, HID_CHUNKS * HID_D};
      size_t qlen = 4;
      for (int k = 0; k < 3; k++) { per_open[k] = 4 + 4 * (nm[k] + wsum[k]) + 4 * nm[k] * (1 + HID_SALT) + 4 + 32 * (size_t)log_big; qlen += per_open[k]; }
      qlen += 4;
      for (unsigned r = 0; r < n_rounds; r++) qlen += 16 + 4 + 4 * HID_SALT + 4 + 32 * (size_t)(log_big - 1 - r);
      rd.pos += qlen * num_queries;
      if (rd.pos > len) rd.bad = 1; }
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
    const uint32_t w_r[1] = {HID_RW}, w_t[1] = {HID_TW}, w_q[HID_CHUNKS] = {HID_D, HID_D, HID_D, HID_D};
    for (unsigned q = 0; q < num_queries && !rc; q++) {
        size_t index = chal_sample_bits(&ch, log_big);
        hopen_t orr, ot, oq;
        if (get_u32(&rd) != 3) { rc = 12; break; }
        if ((rc = read_check_opening(&rd, hash, root_r, index, log_big, 1, w_r, &orr, path))) break;
        if ((rc = read_check_opening(&rd, hash, root_t, index, log_big, 1, w_t, &ot, path))) break;
        if ((rc = read_check_opening(&rd, hash, root_q, index, log_big, HID_CHUNKS, w_q, &oq, path))) break;
        uint32_t xi = bb_mul(gen, bb_pow(bb_two_adic_generator(log_big), rev_bits(index, log_big)));
        bb4_t d0 = bb4_inv(bb4_sub(zeta, bb4_from_base(xi))), d1 = bb4_inv(bb4_sub(zeta_next, bb4_from_base(xi)));
        bb4_t ro = bb4_zero(); int k = 0;
        for (int j = 0; j < HID_RW; j++, k++) ro = bb4_add(ro, bb4_mul(alp[k], bb4_mul(bb4_sub(r_z[j], bb4_from_base(orr.vals[j])), d0)));
        for (int j = 0; j < HID_TW; j++, k++) ro = bb4_add(ro, bb4_mul(alp[k], bb4_mul(bb4_sub(t_z[j], bb4_from_base(ot.vals[j])), d0)));
        for (int j = 0; j < HID_TW; j++, k++) ro = bb4_add(ro, bb4_mul(alp[k], bb4_mul(bb4_sub(t_zn[j], bb4_from_base(ot.vals[j])), d1)));
        for (int c = 0; c < HID_CHUNKS; c++)
            for (int j = 0; j < HID_D; j++, k++) ro = bb4_add(ro, bb4_mul(alp[k], bb4_mul(bb4_sub(q_z[c][j], bb4_from_base(oq.vals[c * HID_D + j])), d0)));
        if (get_u32(&rd) != n_rounds) { rc = 12; break; }
        bb4_t folded = ro; size_t idx = index;
        for (unsigned r = 0; r < n_rounds; r++) {
            unsigned lfh = log_big - 1 - r;
            bb4_t sib = get_ext(&rd);
            uint32_t salt[HID_SALT];
            if (get_u32(&rd) != HID_SALT) { rc = 12; break; }
            get_words(&rd, salt, HID_SALT);
            if (get_u32(&rd) != lfh) { rc = 12; break; }
            get_digests(&rd, hash, path, lfh);
            bb4_t ev[2]; ev[idx & 1] = folded; ev[(idx & 1) ^ 1] = sib;
            size_t pair = idx >> 1;
            uint32_t row[8 + HID_SALT];
            memcpy(row, ev[0].c, 16); memcpy(row + 4, ev[1].c, 16); memcpy(row + 8, salt, HID_SALT * 4);
            size_t dh[2] = {(size_t)1 << lfh, (size_t)1 << lfh}, dw[2] = {8, HID_SALT};
            if (p3o_mmcs_verify_batch_kind(hash, froots[r], dh, dw, 2, pair, row, path, lfh)) { rc = 14; break; }
            uint32_t s = bb_pow(bb_two_adic_generator(lfh + 1), rev_bits(pair, lfh));
            bb4_t num = bb4_mul(bb4_sub(betas[r], bb4_from_base(s)), bb4_sub(ev[1], ev[0]));
            folded = bb4_add(ev[0], bb4_scale(num, bb_inv(bb_sub(bb_neg(s), s))));
            idx = pair;
        }
        if (rc) break;
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
