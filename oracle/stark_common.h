/* ORACLE — TEST INFRASTRUCTURE ONLY (see bb31.h).
 * Pieces shared by the two restatements of the fib_air proving path: stark.c (non-hiding: TwoAdicFriPcs +
 * MerkleTreeMmcs) and stark_hiding.c (the reference's hiding configuration: HidingFriPcs + MerkleTreeHidingMmcs,
 * native/src/fib_air.rs:40-65).  Challengers, byte buffer / reader, FibonacciAir constraint folding, barycentric
 * interpolation, FRI fold.  All [UPSTREAM-RECALL] of the absent crates p3-challenger / p3-fri / p3-uni-stark 0.4.2. */
#ifndef P3O_STARK_COMMON_H
#define P3O_STARK_COMMON_H
#include "p3_oracle.h"
#include "bb31.h"
#include <stdlib.h>
#include <string.h>

#if defined(__GNUC__)
#define P3O_UNUSED __attribute__((unused))
#else
#define P3O_UNUSED
#endif

/* ------------------------------------------------------------------ challenger */
/* kind 0: DuplexChallenger<F, Perm, WIDTH 16, RATE 8>: observe buffers up to RATE inputs then duplexes
 * (overwrite, permute, refill output with state[0..8]); sample pops from the BACK of the output.
 * kind 1: SerializingChallenger32<BabyBear, HashChallenger<u8, Keccak256Hash, 32>> (fib_air.rs:53,66)
 * [UPSTREAM-RECALL, p3-challenger 0.4.2 absent]: HashChallenger keeps an input byte buffer and a 32-byte output
 * buffer; observe clears the output and appends; sample pops output bytes from the BACK, flushing first when it
 * is empty (output = Keccak256(input), input := output as the chaining value).  The serialising wrapper observes a
 * field element as the 4 little-endian bytes of its unique u32 (the Montgomery word), a [u64; 4] digest as its 32
 * little-endian bytes, and samples a base element by rejection: u32 from 4 sampled bytes, masked to 31 bits,
 * accepted when below P. */
typedef struct {
    int kind;
    uint32_t state[16], in[8], out[8];
    int n_in, n_out;
    uint8_t *ibuf; size_t ilen, icap;
    uint8_t obuf[32]; int n_obuf;
} chal_t;
static P3O_UNUSED void chal_init(chal_t *c, int kind) { memset(c, 0, sizeof *c); c->kind = kind; }
static P3O_UNUSED void chal_free(chal_t *c) { free(c->ibuf); c->ibuf = NULL; }
static P3O_UNUSED void chal_duplex(chal_t *c) {
    for (int i = 0; i < c->n_in; i++) c->state[i] = c->in[i];
    c->n_in = 0;
    p3o_poseidon2_permute(c->state);
    memcpy(c->out, c->state, 32);
    c->n_out = 8;
}
static P3O_UNUSED void hc_observe_bytes(chal_t *c, const uint8_t *p, size_t n) {
    c->n_obuf = 0;
    if (c->ilen + n > c->icap) { c->icap = (c->ilen + n) * 2 + 64; c->ibuf = realloc(c->ibuf, c->icap); }
    memcpy(c->ibuf + c->ilen, p, n);
    c->ilen += n;
}
static P3O_UNUSED void hc_flush(chal_t *c) {
    p3o_keccak256(c->ibuf, c->ilen, c->obuf);
    c->n_obuf = 32;
    c->ilen = 0;
    hc_observe_bytes(c, c->obuf, 32);  /* chaining value */
    c->n_obuf = 32;                    /* (the append above is not an observation: the output stays valid) */
}
static P3O_UNUSED uint8_t hc_sample_byte(chal_t *c) {
    if (!c->n_obuf) hc_flush(c);
    return c->obuf[--c->n_obuf];
}
static P3O_UNUSED void chal_observe(chal_t *c, uint32_t v) {
    if (c->kind) { uint8_t le[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; hc_observe_bytes(c, le, 4); return; }
    c->n_out = 0;
    c->in[c->n_in++] = v;
    if (c->n_in == 8) chal_duplex(c);
}
static P3O_UNUSED void chal_observe_n(chal_t *c, const uint32_t *v, size_t n) { for (size_t i = 0; i < n; i++) chal_observe(c, v[i]); }
/* a commitment: 8 field elements (kind 0) or [u64; 4] = the same 32 little-endian bytes (kind 1) */
static P3O_UNUSED void chal_observe_digest(chal_t *c, const uint32_t d[8]) { chal_observe_n(c, d, 8); }
static P3O_UNUSED void chal_observe_ext(chal_t *c, bb4_t v) { chal_observe_n(c, v.c, 4); }
static P3O_UNUSED uint32_t chal_sample(chal_t *c) {
    if (c->kind) {
        for (;;) {
            uint32_t v = 0;
            for (int i = 0; i < 4; i++) v |= (uint32_t)hc_sample_byte(c) << (8 * i);
            v &= 0x7fffffffu;
            if (v < BB_P) return bb_to_monty(v);
        }
    }
    if (c->n_in || !c->n_out) chal_duplex(c);
    return c->out[--c->n_out];
}
static P3O_UNUSED bb4_t chal_sample_ext(chal_t *c) { bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = chal_sample(c); return r; }
static P3O_UNUSED size_t chal_sample_bits(chal_t *c, unsigned bits) {
    return (size_t)bb_from_monty(chal_sample(c)) & (((size_t)1 << bits) - 1);
}
static P3O_UNUSED int chal_check_witness(chal_t *c, unsigned bits, uint32_t w) { chal_observe(c, w); return chal_sample_bits(c, bits) == 0; }
/* GrindingChallenger::grind, serial build: the smallest canonical witness (find_any == find without rayon). */
static P3O_UNUSED uint32_t chal_grind(chal_t *c, unsigned bits) {
    for (uint32_t i = 0; i < BB_P; i++) {
        chal_t t = *c;
        if (c->kind) { t.ibuf = malloc(c->ilen + 64); t.icap = c->ilen + 64; memcpy(t.ibuf, c->ibuf, c->ilen); }
        int ok = chal_check_witness(&t, bits, bb_to_monty(i));
        if (c->kind) free(t.ibuf);
        if (ok) { chal_check_witness(c, bits, bb_to_monty(i)); return bb_to_monty(i); }
    }
    return 0;
}

/* ------------------------------------------------------------------ byte buffer */
typedef struct { uint8_t *p; size_t len, cap; } buf_t;
static P3O_UNUSED void put_u32(buf_t *b, uint32_t v) {
    if (b->len + 4 > b->cap) { b->cap = b->cap ? b->cap * 2 : 4096; b->p = realloc(b->p, b->cap); }
    memcpy(b->p + b->len, &v, 4); b->len += 4;
}
static P3O_UNUSED void put_words(buf_t *b, const uint32_t *w, size_t n) { for (size_t i = 0; i < n; i++) put_u32(b, w[i]); }

static P3O_UNUSED size_t rev_bits(size_t x, unsigned bits) { size_t y = 0; for (unsigned i = 0; i < bits; i++) { y = (y << 1) | (x & 1); x >>= 1; } return y; }

/* ------------------------------------------------------------------ FibonacciAir */
/* fib_air.rs:232-264: constraints in builder order, selectors multiplied in by when_*:
 *   first*(left-a), first*(right-b), trans*(right-next.left), trans*(left+right-next.right), last*(right-x);
 * folded as sum_k alpha^(4-k) C_k (ProverConstraintFolder: first constraint gets the highest power). */
#define FIB_NCONS 5
static P3O_UNUSED bb4_t fib_fold_base(const uint32_t loc[2], const uint32_t nxt[2], const uint32_t pis[3], uint32_t first,
                           uint32_t last, uint32_t trans, const bb4_t apow[FIB_NCONS]) {
    uint32_t c[FIB_NCONS] = {
        bb_mul(first, bb_sub(loc[0], pis[0])), bb_mul(first, bb_sub(loc[1], pis[1])),
        bb_mul(trans, bb_sub(loc[1], nxt[0])), bb_mul(trans, bb_sub(bb_add(loc[0], loc[1]), nxt[1])),
        bb_mul(last, bb_sub(loc[1], pis[2]))};
    bb4_t acc = bb4_zero();
    for (int k = 0; k < FIB_NCONS; k++) acc = bb4_add(acc, bb4_scale(apow[FIB_NCONS - 1 - k], c[k]));
    return acc;
}

/* interpolate_coset: value at `z` of the degree<h interpolant of column evaluations given on shift*<g_h>;
 * rows arrive in bit-reversed order (the committed LDE's first h rows).  Barycentric:
 *   p(z) = (z^h - s^h)/(h s^h) * sum_i x_i y_i / (z - x_i). */
static P3O_UNUSED void interpolate_low_coset(const uint32_t *lde_br, size_t h, size_t w, uint32_t shift, bb4_t z, bb4_t *ys) {
    unsigned lh = 0; while (((size_t)1 << lh) < h) lh++;
    uint32_t g = bb_two_adic_generator(lh);
    /* blocks of consecutive points, one partial sum per block and column, added up in block order afterwards: field
     * addition is exact, so the words do not depend on the thread count (the multi-core leg of bench.py's cpu_baseline) */
    const size_t BLK = 1024, nblk = (h + BLK - 1) / BLK;
    bb4_t *part = malloc(nblk * w * sizeof(bb4_t));
    #pragma omp parallel for schedule(static) if (nblk > 1)
    for (size_t b = 0; b < nblk; b++) {
        size_t lo = b * BLK, hi = lo + BLK < h ? lo + BLK : h;
        bb4_t *acc = part + b * w;
        for (size_t c = 0; c < w; c++) acc[c] = bb4_zero();
        uint32_t x = bb_mul(shift, bb_pow(g, lo));
        for (size_t i = lo; i < hi; i++) {
            bb4_t d = bb4_inv(bb4_sub(z, bb4_from_base(x)));
            const uint32_t *row = lde_br + rev_bits(i, lh) * w;
            bb4_t dx = bb4_scale(d, x);
            for (size_t c = 0; c < w; c++) acc[c] = bb4_add(acc[c], bb4_scale(dx, row[c]));
            x = bb_mul(x, g);
        }
    }
    for (size_t c = 0; c < w; c++) ys[c] = bb4_zero();
    for (size_t b = 0; b < nblk; b++)
        for (size_t c = 0; c < w; c++) ys[c] = bb4_add(ys[c], part[b * w + c]);
    free(part);
    uint32_t sh = bb_pow(shift, h);
    bb4_t zh = bb4_pow(z, h);
    bb4_t f = bb4_scale(bb4_sub(zh, bb4_from_base(sh)), bb_inv(bb_mul(bb_to_monty((uint32_t)h), sh)));
    for (size_t c = 0; c < w; c++) ys[c] = bb4_mul(ys[c], f);
}

/* out[i] = first * step^i for i < n, in blocks (threads; the same words as the serial recurrence) */
static P3O_UNUSED void power_table(uint32_t *out, size_t n, uint32_t first, uint32_t step) {
    const size_t BLK = 4096, nblk = (n + BLK - 1) / BLK;
    #pragma omp parallel for schedule(static) if (nblk > 1)
    for (size_t b = 0; b < nblk; b++) {
        size_t lo = b * BLK, hi = lo + BLK < n ? lo + BLK : n;
        uint32_t acc = bb_mul(first, bb_pow(step, lo));
        for (size_t i = lo; i < hi; i++) { out[i] = acc; acc = bb_mul(acc, step); }
    }
}

typedef struct { unsigned log_blowup, log_final_poly_len, num_queries, pow_bits; } fri_params_t;

/* TwoAdicFriFolding::fold_matrix: pairs (lo, hi) = (f(x), f(-x)) at x = g^bitrev(i) (subgroup, no shift):
 *   out[i] = (1/2 + beta/(2x)) lo + (1/2 - beta/(2x)) hi. */
static P3O_UNUSED void fold_matrix(const bb4_t *in, size_t len, bb4_t beta, bb4_t *out) {
    size_t half = len / 2;
    unsigned lh = 0; while (((size_t)1 << lh) < half) lh++;
    uint32_t ginv = bb_inv(bb_two_adic_generator(lh + 1));
    uint32_t one_half = bb_inv(bb_to_monty(2));
    bb4_t hb = bb4_scale(beta, one_half);
    uint32_t *pw = malloc((half ? half : 1) * 4);
    power_table(pw, half, BB_ONE, ginv);
    #pragma omp parallel for schedule(static) if (half >= 4096)
    for (size_t i = 0; i < half; i++) {
        bb4_t power = bb4_scale(hb, pw[rev_bits(i, lh)]);
        bb4_t oh = bb4_from_base(one_half);
        out[i] = bb4_add(bb4_mul(bb4_add(oh, power), in[2 * i]), bb4_mul(bb4_sub(oh, power), in[2 * i + 1]));
    }
    free(pw);
}

static P3O_UNUSED void put_path(buf_t *b, const uint32_t *path, size_t n) { put_u32(b, (uint32_t)n); put_words(b, path, n * 8); }

/* ------------------------------------------------------------------ verifier */
typedef struct { const uint8_t *p; size_t len, pos; int bad; } rd_t;
static P3O_UNUSED uint32_t get_u32(rd_t *r) { uint32_t v = 0; if (r->pos + 4 > r->len) { r->bad = 1; return 0; } memcpy(&v, r->p + r->pos, 4); r->pos += 4; return v; }
static P3O_UNUSED void get_words(rd_t *r, uint32_t *w, size_t n) { for (size_t i = 0; i < n; i++) { w[i] = get_u32(r); if (w[i] >= BB_P) r->bad = 1; } }
static P3O_UNUSED bb4_t get_ext(rd_t *r) { bb4_t v; get_words(r, v.c, 4); return v; }
/* n digests: field elements below P (Poseidon2) or raw [u64; 4] bytes (Keccak) */
static P3O_UNUSED void get_digests(rd_t *r, int hash, uint32_t *w, size_t n) {
    if (!hash) { get_words(r, w, 8 * n); return; }
    for (size_t i = 0; i < 8 * n; i++) w[i] = get_u32(r);
}


#endif
