// BabyBear (P = 2^31 - 2^27 + 1) Montgomery arithmetic for gfx950 device code and the C++ host side.
// Same residue system as the reference's shaders (native/shaders/fft_stage.wgsl:36-70,
// native/src/backend_vulkan.rs:882-917): u32 words x*2^32 mod P kept in [0, P).
// The reduction is re-derived for CDNA4: two v_mad_u64_u32 and one v_mul_lo_u32 (see mul below) and branch-free
// min() corrections instead of the reference's compare-and-branch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bb {

constexpr uint32_t P = 0x78000001u;
constexpr uint32_t MU = 0x88000001u;   // P^-1 mod 2^32
constexpr uint32_t ONE = 0x0ffffffeu;  // 2^32 mod P
constexpr uint32_t R2 = 0x45dddde3u;   // 2^64 mod P
constexpr uint32_t GEN = 31u;          // multiplicative generator (canonical)
constexpr uint32_t TWO_ADICITY = 27;

#define BB_HD __host__ __device__ __forceinline__

BB_HD uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

BB_HD uint32_t add(uint32_t a, uint32_t b) {
    uint32_t s = a + b;  // < 2P < 2^32
    return umin32(s, s - P);
}
BB_HD uint32_t sub(uint32_t a, uint32_t b) {
    uint32_t d = a - b;
    return umin32(d, d + P);
}
BB_HD uint32_t neg(uint32_t a) { return a ? P - a : 0u; }
BB_HD uint32_t dbl(uint32_t a) { return add(a, a); }

BB_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// x in [0, 2^32 * P)  ->  x * 2^-32 mod P in [0, P)
BB_HD uint32_t monty_reduce(uint64_t x) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    uint32_t t = lo * MU;
    uint32_t u = mulhi32(t, P);  // low words of x and t*P agree, so no borrow crosses bit 32
    uint32_t r = hi - u;
    return umin32(r, r + P);
}
// Montgomery product, FIVE instructions on gfx950: v_mad_u64_u32 (x = a b), v_mul_lo_u32 (t = lo(x) * -P^-1), v_mad_u64_u32
// (x + t P: the low word cancels, the high word is the result in [0, 2P)), v_add, v_min.  Until round 3 this was the seven-
// instruction form (v_mul_lo, v_mul_hi, v_mul_lo, v_mul_hi, v_sub, v_add, v_min) on the strength of round 1's first issue-rate
// probe, which had v_mad_u64_u32 at a third of the multiplies' rate; the second probe (profiles/r01_microbench2_valu_issue_rates.txt:
// 557 against 566 G wave-instructions/s at eight waves per SIMD) and a butterfly loop timed both ways
// (tools/mulform_bench.hip, profiles/r04_mulform_bench.txt: +9 % butterflies/s) say otherwise.  Same value for every input.
// BB_MUL_SPLIT=1 (compile time) restores the seven-instruction form.
#ifndef BB_MUL_SPLIT
#define BB_MUL_SPLIT 0
#endif
constexpr uint32_t NMU = 0u - MU;  // -P^-1 mod 2^32
BB_HD uint32_t mul(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && BB_MUL_SPLIT
    uint32_t lo = a * b, hi = __umulhi(a, b);
    uint32_t t = lo * MU;
    uint32_t u = __umulhi(t, P);
    uint32_t r = hi - u;
    return umin32(r, r + P);
#elif defined(__HIP_DEVICE_COMPILE__)
    const uint64_t x = (uint64_t)a * b;
    const uint32_t t = (uint32_t)x * NMU;
    const uint32_t r = (uint32_t)((x + (uint64_t)t * P) >> 32);  // x + t P < P^2 + 2^32 P < 2^64
    return umin32(r, r - P);
#else
    return monty_reduce((uint64_t)a * b);
#endif
}
// sum of two products under ONE reduction: a0 b0 + a1 b1 < 2 P^2 < 2^32 P, so the 64-bit sum is a valid Montgomery input
// (two v_mad_u64_u32 + the three-instruction tail instead of two products and a modular addition: 6 instead of 13)
BB_HD uint32_t dot2(uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
#if defined(__HIP_DEVICE_COMPILE__) && !BB_MUL_SPLIT
    const uint64_t x = (uint64_t)a0 * b0 + (uint64_t)a1 * b1;
    const uint32_t t = (uint32_t)x * NMU;
    const uint32_t r = (uint32_t)((x + (uint64_t)t * P) >> 32);  // < (2 P^2 + 2^32 P) / 2^32 < 2P
    return umin32(r, r - P);
#else
    return monty_reduce((uint64_t)a0 * b0 + (uint64_t)a1 * b1);
#endif
}
BB_HD uint32_t sqr(uint32_t a) { return mul(a, a); }
BB_HD uint32_t to_monty(uint32_t canon) { return mul(canon, R2); }
BB_HD uint32_t from_monty(uint32_t m) { return monty_reduce((uint64_t)m); }

BB_HD uint32_t pow(uint32_t base, uint64_t e) {
    uint32_t r = ONE;
    while (e) {
        if (e & 1) r = mul(r, base);
        base = sqr(base);
        e >>= 1;
    }
    return r;
}
BB_HD uint32_t inv(uint32_t a) { return pow(a, (uint64_t)P - 2); }

// two_adic_generator(bits) = (31^15)^(2^(27-bits))  (backend_vulkan.rs:982)
BB_HD uint32_t two_adic_generator(uint32_t bits) {
    uint32_t g = pow(to_monty(GEN), 15);
    for (uint32_t i = bits; i < TWO_ADICITY; i++) g = sqr(g);
    return g;
}

// x^7
BB_HD uint32_t pow7(uint32_t x) {
    uint32_t x2 = sqr(x), x3 = mul(x2, x), x4 = sqr(x2);
    return mul(x3, x4);
}

// Signed Montgomery product (Seiler): inputs in (-P, P) as int32, output in (-P, P), FIVE integer ops and
// no correction: r = hi(a*b) - hi(t*P), t = lo(a*b) * P^-1 (signed).  |r| < (P^2 + 2^31 P)/2^32 < P.
BB_HD int32_t smul(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t lo = (int32_t)((uint32_t)a * (uint32_t)b), hi = __mulhi(a, b);
    int32_t t = (int32_t)((uint32_t)lo * MU);
    return hi - __mulhi(t, (int32_t)P);
#else
    int64_t x = (int64_t)a * b;
    int32_t t = (int32_t)((uint32_t)x * MU);
    return (int32_t)((x - (int64_t)t * (int64_t)P) >> 32);
#endif
}
// (s + rc)^7 for s, rc in [0, P): the sum enters the chain as s + (rc - P) in [-P, P) (one add), the four
// products stay signed, one final correction brings the result back to [0, P).
BB_HD uint32_t sbox7_add(uint32_t s, uint32_t rc) {
    int32_t x = (int32_t)(s + (rc - P));
    int32_t x2 = smul(x, x), x3 = smul(x2, x), x4 = smul(x2, x2);
    uint32_t r = (uint32_t)smul(x3, x4);
    return umin32(r, r + P);
}

// ---- quartic extension F[x]/(x^4 - 11) (Challenge = BinomialExtensionField<BabyBear,4>,
//      reference native/src/fib_air.rs:23) ----
struct Ext {
    uint32_t c[4];
};
constexpr uint32_t W_CANON = 11u;
constexpr uint32_t W_MONTY = 0x37ffffe9u;  // 11 * 2^32 mod P

BB_HD Ext ext_zero() { return Ext{{0, 0, 0, 0}}; }
BB_HD Ext ext_from_base(uint32_t a) { return Ext{{a, 0, 0, 0}}; }
BB_HD Ext ext_one() { return ext_from_base(ONE); }
BB_HD Ext add(const Ext& a, const Ext& b) {
    return Ext{{add(a.c[0], b.c[0]), add(a.c[1], b.c[1]), add(a.c[2], b.c[2]), add(a.c[3], b.c[3])}};
}
BB_HD Ext sub(const Ext& a, const Ext& b) {
    return Ext{{sub(a.c[0], b.c[0]), sub(a.c[1], b.c[1]), sub(a.c[2], b.c[2]), sub(a.c[3], b.c[3])}};
}
BB_HD Ext neg(const Ext& a) { return Ext{{neg(a.c[0]), neg(a.c[1]), neg(a.c[2]), neg(a.c[3])}}; }
BB_HD Ext scale(const Ext& a, uint32_t s) {
    return Ext{{mul(a.c[0], s), mul(a.c[1], s), mul(a.c[2], s), mul(a.c[3], s)}};
}
BB_HD bool eq(const Ext& a, const Ext& b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3];
}
// Schoolbook product modulo x^4 - 11: every output coefficient is a sum of four products (b pre-multiplied by 11 where the
// index wraps), taken two products per Montgomery reduction (dot2): 4 x (two dot2 + one addition) + three products by 11 =
// 75 instructions instead of the 16 products, 3 products by 11 and 12 additions (169) of the first form.  Exact field
// arithmetic: the value does not depend on the grouping.
BB_HD Ext mul(const Ext& a, const Ext& b) {
    const uint32_t w1 = mul(b.c[1], W_MONTY), w2 = mul(b.c[2], W_MONTY), w3 = mul(b.c[3], W_MONTY);
    return Ext{{add(dot2(a.c[0], b.c[0], a.c[1], w3), dot2(a.c[2], w2, a.c[3], w1)),
                add(dot2(a.c[0], b.c[1], a.c[1], b.c[0]), dot2(a.c[2], w3, a.c[3], w2)),
                add(dot2(a.c[0], b.c[2], a.c[1], b.c[1]), dot2(a.c[2], b.c[0], a.c[3], w3)),
                add(dot2(a.c[0], b.c[3], a.c[1], b.c[2]), dot2(a.c[2], b.c[1], a.c[3], b.c[0]))}};
}
BB_HD Ext sqr(const Ext& a) { return mul(a, a); }
BB_HD Ext pow(Ext base, uint64_t e) {
    Ext r = ext_one();
    while (e) {
        if (e & 1) r = mul(r, base);
        base = sqr(base);
        e >>= 1;
    }
    return r;
}
BB_HD Ext inv(const Ext& a) {
    Ext conj{{a.c[0], neg(a.c[1]), a.c[2], neg(a.c[3])}};
    Ext n = mul(a, conj);  // c + d x^2
    uint32_t c = n.c[0], d = n.c[2];
    uint32_t den = sub(sqr(c), mul(W_MONTY, sqr(d)));
    uint32_t di = inv(den);
    Ext q{{mul(c, di), 0, neg(mul(d, di)), 0}};
    return mul(conj, q);
}

}  // namespace bb
