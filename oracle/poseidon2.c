/* ORACLE — TEST INFRASTRUCTURE ONLY (see bb31.h).
 * Poseidon2 over BabyBear, width 16, x^7, R_F = 8 (4+4), R_P = 13: the permutation the
 * reference names at native/src/poseidon_cpu.rs:17-18 (default_babybear_poseidon2_16()).
 * The implementation lives in the ABSENT crates p3-poseidon2 / p3-baby-bear 0.4.2; this file
 * restates the published Poseidon2 algorithm (Grassi-Khovratovich-Schofnegger 2023):
 *   external linear layer  circ(2*M4, M4, M4, M4), M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
 *   internal linear layer  1 + diag(V),
 *     V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/4, 1/8, 1/2^27, -1/2^8, -1/16, -1/2^27]
 * PINNING: tests/golden/poseidon2_bb16_kat.json holds Plonky3's own known-answer vector
 * (test_poseidon2_width_16_random: constants from Xoroshiro128Plus seed 1); this code reproduces
 * it bit-for-bit through p3o_poseidon2_permute_rc.  Default round constants come from the
 * published Grain-LFSR procedure (tools/gen_poseidon2_rc.py).  Provenance: DESIGN.md. */
#include "p3_oracle.h"
#include "bb31.h"
#include "poseidon2_rc16.h"
#include <string.h>

static uint32_t DIAG[16];
static int diag_ready = 0;

static void init_diag(void) {
    if (diag_ready) return;
    uint32_t two = bb_to_monty(2);
    uint32_t i2 = bb_inv(two);
    uint32_t i2_8 = bb_pow(i2, 8), i2_27 = bb_pow(i2, 27);
    uint32_t v[16] = {
        bb_neg(two), BB_ONE, two, i2, bb_to_monty(3), bb_to_monty(4), bb_neg(i2),
        bb_neg(bb_to_monty(3)), bb_neg(bb_to_monty(4)), i2_8, bb_pow(i2, 2), bb_pow(i2, 3), i2_27,
        bb_neg(i2_8), bb_neg(bb_pow(i2, 4)), bb_neg(i2_27)};
    memcpy(DIAG, v, sizeof v);
    diag_ready = 1;
}

static inline uint32_t sbox7(uint32_t x) {
    uint32_t x2 = bb_mul(x, x), x3 = bb_mul(x2, x), x4 = bb_mul(x2, x2);
    return bb_mul(x3, x4);
}
static inline void mat4(uint32_t *x) {
    uint32_t a = x[0], b = x[1], c = x[2], d = x[3];
    uint32_t s = bb_add(bb_add(a, b), bb_add(c, d));
    /* row i = s + x_i + 2*x_{i+1}: [2 3 1 1] etc. */
    x[0] = bb_add(bb_add(s, a), bb_add(b, b));
    x[1] = bb_add(bb_add(s, b), bb_add(c, c));
    x[2] = bb_add(bb_add(s, c), bb_add(d, d));
    x[3] = bb_add(bb_add(s, d), bb_add(a, a));
}
static void external_linear(uint32_t s[16]) {
    for (int i = 0; i < 16; i += 4) mat4(s + i);
    uint32_t sums[4];
    for (int k = 0; k < 4; k++)
        sums[k] = bb_add(bb_add(s[k], s[k + 4]), bb_add(s[k + 8], s[k + 12]));
    for (int i = 0; i < 16; i++) s[i] = bb_add(s[i], sums[i & 3]);
}
static void internal_linear(uint32_t s[16]) {
    uint32_t tot = 0;
    for (int i = 0; i < 16; i++) tot = bb_add(tot, s[i]);
    for (int i = 0; i < 16; i++) s[i] = bb_add(bb_mul(s[i], DIAG[i]), tot);
}

void p3o_poseidon2_permute_rc(uint32_t s[16], const uint32_t ei[4][16], const uint32_t in[13],
                              const uint32_t ef[4][16]) {
    init_diag();
    external_linear(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = sbox7(bb_add(s[i], ei[r][i]));
        external_linear(s);
    }
    for (int r = 0; r < 13; r++) {
        s[0] = sbox7(bb_add(s[0], in[r]));
        internal_linear(s);
    }
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 16; i++) s[i] = sbox7(bb_add(s[i], ef[r][i]));
        external_linear(s);
    }
}

void p3o_poseidon2_permute(uint32_t s[16]) {
    p3o_poseidon2_permute_rc(s, P3_RC16_EXT_INIT_MONTY, P3_RC16_INTERNAL_MONTY,
                             P3_RC16_EXT_FINAL_MONTY);
}
