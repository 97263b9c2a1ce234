/* ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * BabyBear (P = 2^31 - 2^27 + 1) in 32-bit Montgomery form, restating
 *   reference native/src/backend_vulkan.rs:882-917 (add_mod / sub_mod /
 *   monty_reduce / mul_mod inside cpu_stage_u32_in_place) and :944-957
 *   (monty_to_canonical); same arithmetic device-side in
 *   native/shaders/fft_stage.wgsl:36-70.
 * Values are u32 Montgomery residues x*2^32 mod P, always reduced to [0,P).
 *
 * Field constants not present in the reference tree (they live in the absent
 * p3-baby-bear 0.4.2 crate) are derived/checked in tests/test_oracle_field.py:
 *   multiplicative generator 31, TWO_ADICITY 27,
 *   two_adic_generator(27) = 31^15 = 0x1a427a41,
 *   quartic extension F[x]/(x^4 - 11)  [UPSTREAM-RECALL: BinomialExtensionField<BabyBear,4>, W = 11].
 */
#ifndef P3_ORACLE_BB31_H
#define P3_ORACLE_BB31_H
#include <stdint.h>
#include <stddef.h>

#define BB_P 0x78000001u
#define BB_MONTY_MU 0x88000001u /* P^-1 mod 2^32 (backend_vulkan.rs:884) */
#define BB_TWO_ADICITY 27
#define BB_GENERATOR_CANON 31u
#define BB_EXT_W_CANON 11u

static inline uint32_t bb_add(uint32_t a, uint32_t b) { /* backend_vulkan.rs:887-894 */
    uint32_t s = a + b;
    return s >= BB_P ? s - BB_P : s;
}
static inline uint32_t bb_sub(uint32_t a, uint32_t b) { /* backend_vulkan.rs:896-902 */
    return a >= b ? a - b : a + BB_P - b;
}
static inline uint32_t bb_neg(uint32_t a) { return a ? BB_P - a : 0; }
static inline uint32_t bb_monty_reduce(uint64_t x) { /* backend_vulkan.rs:904-913 */
    uint64_t t = (x * (uint64_t)BB_MONTY_MU) & 0xffffffffull;
    uint64_t u = t * (uint64_t)BB_P;
    uint64_t d = x - u;
    uint32_t hi = (uint32_t)(d >> 32);
    return x < u ? hi + BB_P : hi;
}
static inline uint32_t bb_mul(uint32_t a, uint32_t b) { /* backend_vulkan.rs:915-917 */
    return bb_monty_reduce((uint64_t)a * (uint64_t)b);
}
static inline uint32_t bb_from_monty(uint32_t x) { /* backend_vulkan.rs:944-957 */
    return bb_monty_reduce((uint64_t)x);
}
static inline uint32_t bb_to_monty(uint32_t canon) { /* inverse of the above: x*2^32 mod P */
    return (uint32_t)((((uint64_t)(canon % BB_P)) << 32) % BB_P);
}
#define BB_ONE 0x0ffffffeu /* 2^32 mod P */
#define BB_ZERO 0u

static inline uint32_t bb_pow(uint32_t base, uint64_t e) {
    uint32_t r = BB_ONE;
    while (e) {
        if (e & 1) r = bb_mul(r, base);
        base = bb_mul(base, base);
        e >>= 1;
    }
    return r;
}
static inline uint32_t bb_inv(uint32_t a) { return bb_pow(a, (uint64_t)BB_P - 2); }
static inline uint32_t bb_halve(uint32_t a) { return bb_mul(a, bb_inv(bb_to_monty(2))); }

/* BabyBear::two_adic_generator(bits) = (31^15)^(2^(27-bits)); used at
 * backend_vulkan.rs:982 (twiddles_for_stage). */
static inline uint32_t bb_two_adic_generator(unsigned bits) {
    uint32_t g = bb_pow(bb_to_monty(BB_GENERATOR_CANON), 15); /* order 2^27 */
    for (unsigned i = bits; i < BB_TWO_ADICITY; i++) g = bb_mul(g, g);
    return g;
}

/* ---- quartic extension F[x]/(x^4 - 11), coefficients low-degree first ---- */
typedef struct { uint32_t c[4]; } bb4_t;

static inline bb4_t bb4_from_base(uint32_t a) { bb4_t r = {{a, 0, 0, 0}}; return r; }
static inline bb4_t bb4_zero(void) { bb4_t r = {{0, 0, 0, 0}}; return r; }
static inline bb4_t bb4_one(void) { return bb4_from_base(BB_ONE); }
static inline bb4_t bb4_add(bb4_t a, bb4_t b) {
    bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_add(a.c[i], b.c[i]); return r;
}
static inline bb4_t bb4_sub(bb4_t a, bb4_t b) {
    bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_sub(a.c[i], b.c[i]); return r;
}
static inline bb4_t bb4_neg(bb4_t a) {
    bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_neg(a.c[i]); return r;
}
static inline bb4_t bb4_scale(bb4_t a, uint32_t s) {
    bb4_t r; for (int i = 0; i < 4; i++) r.c[i] = bb_mul(a.c[i], s); return r;
}
static inline bb4_t bb4_mul(bb4_t a, bb4_t b) {
    const uint32_t w = bb_to_monty(BB_EXT_W_CANON);
    uint32_t t[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) t[i + j] = bb_add(t[i + j], bb_mul(a.c[i], b.c[j]));
    bb4_t r;
    for (int i = 0; i < 4; i++) r.c[i] = i < 3 ? bb_add(t[i], bb_mul(w, t[i + 4])) : t[i];
    return r;
}
static inline bb4_t bb4_square(bb4_t a) { return bb4_mul(a, a); }
static inline int bb4_eq(bb4_t a, bb4_t b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3];
}
static inline bb4_t bb4_pow(bb4_t base, uint64_t e) {
    bb4_t r = bb4_one();
    while (e) {
        if (e & 1) r = bb4_mul(r, base);
        base = bb4_mul(base, base);
        e >>= 1;
    }
    return r;
}
/* Inverse through the tower F < F[y]/(y^2-11) < F[x]/(x^2-y):
 * conj(a)(x) = a(-x); a*conj(a) = c + d*y with y = x^2, then invert in the quadratic field. */
static inline bb4_t bb4_inv(bb4_t a) {
    const uint32_t w = bb_to_monty(BB_EXT_W_CANON);
    bb4_t conj = {{a.c[0], bb_neg(a.c[1]), a.c[2], bb_neg(a.c[3])}};
    bb4_t n = bb4_mul(a, conj); /* = c + d x^2, odd coefficients vanish */
    uint32_t c = n.c[0], d = n.c[2];
    uint32_t den = bb_sub(bb_mul(c, c), bb_mul(w, bb_mul(d, d)));
    uint32_t di = bb_inv(den);
    bb4_t q = {{bb_mul(c, di), 0, bb_neg(bb_mul(d, di)), 0}};
    return bb4_mul(conj, q);
}
#endif
