// Poseidon2 (same permutation as poseidon2.hip.h) evaluated in DOUBLE PRECISION integer arithmetic.
// On gfx950 int32 and fp64 VALU ops issue at the same rate, but in fp64 a modular addition is ONE exact
// v_add_f64 (no reduction while |v| < 2^53) instead of add/sub/min, which removes ~25% of the permutation's
// instructions.  State elements are doubles holding integers congruent (mod P) to CANONICAL field values
// (not Montgomery): the kernel converts at load/store.
//   mulmod(a, b): h = a*b; l = fma(a, b, -h) (exact error); q = rint(h / P); r = fma(-q, P, h) + l
//     exact for |a*b| < 2^84: h - qP is an integer below 2^32, l an integer below 2^31; result |r| < 1.1 P.
#pragma once
#include "poseidon2.hip.h"

namespace p2f {

constexpr double PD = 2013265921.0;
constexpr double PINV = 1.0 / 2013265921.0;

__device__ __forceinline__ double mulmod(double a, double b) {
    double h = a * b;
    double l = __fma_rn(a, b, -h);
    double q = __builtin_rint(h * PINV);
    double r = __fma_rn(-q, PD, h);
    return r + l;
}
// a * b (mod P) when aP = a / P is already known: q = rint(b * aP) is within 1 of a*b/P; q * (P - 1) =
// q * 15 * 2^27 is exact, so t = fma(a, b, -q(P-1)) = (ab - qP) + q is an integer below 2^33, hence exact.
// Five ops, |result| <= P/2 + 1 + |ab| 2^-52.  Exact while |ab| < 2^76.
constexpr double PM1 = 2013265920.0;
__device__ __forceinline__ double mulmod_p(double a, double b, double aP) {
    double q = __builtin_rint(b * aP);
    double t = __fma_rn(a, b, -(q * PM1));
    return t - q;
}
__device__ __forceinline__ double reduce(double x) {  // |result| <= P/2 + eps
    return __fma_rn(-__builtin_rint(x * PINV), PD, x);
}
// tot + x * 2^-K (mod P), K <= 8, for integers |x| < 2^44, |tot| < 2^40: xt = tot + x / 2^K is exact (K fractional
// bits), fract(xt) = (x mod 2^K) / 2^K because tot is an integer, and xt - fract * P = tot + (x - (x mod 2^K) P) / 2^K
// is an integer (P = 1 mod 2^K) below 2^45: three exact ops.  inv = +-2^-K.
__device__ __forceinline__ double add_scaled_pow2(double tot, double x, double inv) {
    double xt = __fma_rn(x, inv, tot);
    double fr = __builtin_amdgcn_fract(xt);
    return __fma_rn(fr, -PD, xt);
}
// The same product in FOUR ops, none of them a rounding instruction: with M = 1.5 * 2^52 (ulp 1),
//   qb = fma(b, aP, M)           = M + rint(b * aP) = M + q                  (the classic magic-number rounding)
//   c  = fma(-qb, P - 1, M * P)  = M - q (P - 1)                             (exact: a multiple of 2^27 below 2^71 plus M; M * P is a double)
//   t  = fma(a, b, c)            = M + (ab - qP) + q                         (exact: M plus an integer below 2^40)
//   t - qb                       = ab - qP                                   (exact difference of two integers below 2^53)
// Valid for |b * aP| < 2^51 and |ab| < 2^76, i.e. for every product of the permutation (the largest S-box input is
// 35 (P/2 + 1) < 2^36).  One op less than mulmod_p, two less than mulmod: the S-box drops from 23 to 19 fp64 ops.
constexpr double MAGIC = 6755399441055744.0;                 // 1.5 * 2^52
constexpr double MAGIC_P = 6755399441055744.0 * 2013265921.0;  // 45 * 2^78 + 3 * 2^51: exactly representable
// Operand placement matters: a VOP3 fp64 op reads at most ONE scalar / literal operand on gfx950, and hipcc turns
// fma(x, literal, constant) into v_fmac + a v_mov_b64 copy of the constant per use.  So M and M * P live in VGPR pairs
// the compiler cannot rematerialise (MagicRegs, pinned once per kernel), -(P - 1) comes from an SGPR pair: every line
// below is ONE v_fma_f64 / v_add_f64.
struct MagicRegs { double m, mp; };
__device__ __forceinline__ MagicRegs magic_regs() {
    MagicRegs r{MAGIC, MAGIC_P};
    asm volatile("" : "+v"(r.m), "+v"(r.mp));
    return r;
}
__device__ __forceinline__ double mulmod_m(double a, double b, double aP, const MagicRegs& k, double neg_pm1) {
    const double qb = __fma_rn(b, aP, k.m);
    const double c = __fma_rn(qb, neg_pm1, k.mp);
    const double t = __fma_rn(a, b, c);
    return t - qb;
}
// x^7: x2 = x*x, x3 = x2*x, x4 = x3*x share xP = x / P; the last product x3 * x4 uses x3 / P.
__device__ __forceinline__ double sbox7(double x, const MagicRegs& k, double neg_pm1, double pinv) {
    const double xP = x * pinv;
    const double x2 = mulmod_m(x, x, xP, k, neg_pm1), x3 = mulmod_m(x, x2, xP, k, neg_pm1), x4 = mulmod_m(x, x3, xP, k, neg_pm1);
    return mulmod_m(x3, x4, x3 * pinv, k, neg_pm1);
}

struct ConstsF64 {
    double ext[8][16];
    double in[13];
    // multipliers of the internal diagonal that are not fp64 inline constants, kept in SGPRs (as literals the
    // compiler would pick v_fmac + a copy of the addend): 3, 15, 2^-2, 2^-3, 2^-4, 2^-8
    double k3, k15, i4, i8, i16, i256;
    double neg_pm1, pinv;  // -(P - 1), 1 / P
};
constexpr uint32_t c_from_monty(uint32_t m) {  // m * 2^-32 mod P
    uint32_t t = m * bb::MU;
    uint64_t u = (uint64_t)t * bb::P;
    uint32_t uh = (uint32_t)(u >> 32);
    return uh ? bb::P - uh : 0;  // (m - t*P) >> 32 with hi(m) = 0
}
constexpr uint64_t c_powmod(uint64_t b, uint64_t e) { uint64_t r = 1; b %= bb::P; while (e) { if (e & 1) r = r * b % bb::P; b = b * b % bb::P; e >>= 1; } return r; }
constexpr ConstsF64 make_consts() {
    ConstsF64 c{};
    for (int r = 0; r < 4; r++)
        for (int i = 0; i < 16; i++) {
            c.ext[r][i] = (double)c_from_monty(P3_RC16_EXT_INIT_MONTY[r][i]);
            c.ext[4 + r][i] = (double)c_from_monty(P3_RC16_EXT_FINAL_MONTY[r][i]);
        }
    for (int r = 0; r < 13; r++) c.in[r] = (double)c_from_monty(P3_RC16_INTERNAL_MONTY[r]);
    c.k3 = 3.0; c.k15 = 15.0; c.i4 = 0.25; c.i8 = 0.125; c.i16 = 0.0625; c.i256 = 0.00390625;
    c.neg_pm1 = -2013265920.0; c.pinv = 1.0 / 2013265921.0;
    return c;
}
static __device__ __constant__ ConstsF64 d_c = make_consts();

__device__ __forceinline__ void mat4(double& a, double& b, double& c, double& d) {
    double t01 = a + b, t23 = c + d, t0123 = t01 + t23;
    double t01123 = t0123 + b, t01233 = t0123 + d;
    double nd = __fma_rn(a, 2.0, t01233), nb = __fma_rn(c, 2.0, t01123);
    double na = t01123 + t01, nc = t01233 + t23;
    a = na; b = nb; c = nc; d = nd;
}
__device__ __forceinline__ void external_linear(double (&s)[16]) {
#pragma unroll
    for (int i = 0; i < 16; i += 4) mat4(s[i], s[i + 1], s[i + 2], s[i + 3]);
    double t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = (s[k] + s[k + 4]) + (s[k + 8] + s[k + 12]);
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] += t[i & 3];
}

// The 13 internal rounds.  Entry: integers |s_i| <= 2^37 (what the fourth external round leaves: 35 (P/2 + 1) < 2^36).
// V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/4, 1/8, 2^-27, -2^-8, -1/16, -2^-27]; 2^-27 = -15 (mod P)
// because P - 1 = 15 * 2^27.  The integer-multiplier lanes grow up to 15x per round: fold them back after
// every fourth round (15^4 * 2^37 < 2^53: still exact integers); the fractional lanes stay below their entry bound.
__device__ __forceinline__ void internal_rounds(double (&s)[16], const MagicRegs& mk) {
    const double npm1 = d_c.neg_pm1, pinv = d_c.pinv;
    const double k3 = d_c.k3, k15 = d_c.k15, i4 = d_c.i4, i8 = d_c.i8, i16 = d_c.i16, i256 = d_c.i256;
    auto internal_round = [&](int r) {
        s[0] = sbox7(s[0] + d_c.in[r], mk, npm1, pinv);
        double tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])) +
                     (((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15])));
        tot = reduce(tot);
        s[0] = __fma_rn(s[0], -2.0, tot);
        s[1] = tot + s[1];
        s[2] = __fma_rn(s[2], 2.0, tot);
        s[3] = add_scaled_pow2(tot, s[3], 0.5);
        s[4] = __fma_rn(s[4], k3, tot);
        s[5] = __fma_rn(s[5], 4.0, tot);
        s[6] = add_scaled_pow2(tot, s[6], -0.5);
        s[7] = __fma_rn(s[7], -k3, tot);
        s[8] = __fma_rn(s[8], -4.0, tot);
        s[9] = add_scaled_pow2(tot, s[9], i256);
        s[10] = add_scaled_pow2(tot, s[10], i4);
        s[11] = add_scaled_pow2(tot, s[11], i8);
        s[12] = __fma_rn(s[12], -k15, tot);
        s[13] = add_scaled_pow2(tot, s[13], -i256);
        s[14] = add_scaled_pow2(tot, s[14], -i16);
        s[15] = __fma_rn(s[15], k15, tot);
    };
    auto fold_integer_lanes = [&]() {
        s[1] = reduce(s[1]); s[2] = reduce(s[2]); s[4] = reduce(s[4]); s[5] = reduce(s[5]);
        s[7] = reduce(s[7]); s[8] = reduce(s[8]); s[12] = reduce(s[12]); s[15] = reduce(s[15]);
    };
    _Pragma("clang loop unroll(disable)")
    for (int blk = 0; blk < 3; blk++) {
        _Pragma("clang loop unroll(disable)")
        for (int r = 0; r < 4; r++) internal_round(4 * blk + r);
        fold_integer_lanes();
    }
    internal_round(12);
    fold_integer_lanes();
}

// |s_i| <= 2^33 on entry (canonical inputs or compress inputs < P); every intermediate stays below 2^53.
__device__ __forceinline__ void permute(double (&s)[16]) {
    const MagicRegs mk = magic_regs();
    const double npm1 = d_c.neg_pm1, pinv = d_c.pinv;
    external_linear(s);  // <= 35 * 2^33 < 2^39
    _Pragma("clang loop unroll(disable)")
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7(s[i] + d_c.ext[r][i], mk, npm1, pinv);  // |.| <= P/2 + 1
        external_linear(s);                                                // <= 35 (P/2 + 1) < 2^36
    }
    internal_rounds(s, mk);
    _Pragma("clang loop unroll(disable)")
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7(s[i] + d_c.ext[r][i], mk, npm1, pinv);
        external_linear(s);
    }
}

// Montgomery word <-> canonical double
__device__ __forceinline__ double load_elem(uint32_t monty) { return (double)bb::from_monty(monty); }
// The Montgomery word of an (unreduced) result, v * 2^32 mod P, straight from the four-op product with the constant
// 2^32 mod P: |v| < 2^37 here, so the signed remainder lies within (-P/2 - 2^12, P/2 + 2^12); one conversion, and
// min(w, w + P) on the unsigned bit pattern adds P exactly to the negative ones: 7 instructions, where reduce + sign fix +
// convert + the integer to_monty took 15.
constexpr double R_MOD_P = 268435454.0;                        // 2^32 mod P
constexpr double R_MOD_P_OVER_P = 268435454.0 / 2013265921.0;
__device__ __forceinline__ uint32_t store_elem(double v, const MagicRegs& k) {
    const double r = mulmod_m(R_MOD_P, v, R_MOD_P_OVER_P, k, d_c.neg_pm1);
    const uint32_t w = (uint32_t)(int32_t)r;
    return min(w, w + bb::P);
}
__device__ __forceinline__ uint32_t store_elem(double v) { return store_elem(v, magic_regs()); }

}  // namespace p2f
