// Poseidon2 (same permutation as poseidon2.cuh) evaluated in DOUBLE PRECISION integer arithmetic.
// On gfx950 int32 and fp64 VALU ops issue at the same rate, but in fp64 a modular addition is ONE exact
// v_add_f64 (no reduction while |v| < 2^53) instead of add/sub/min, which removes ~25% of the permutation's
// instructions.  State elements are doubles holding integers congruent (mod P) to CANONICAL field values
// (not Montgomery): the kernel converts at load/store.
//   mulmod(a, b): h = a*b; l = fma(a, b, -h) (exact error); q = rint(h / P); r = fma(-q, P, h) + l
//     exact for |a*b| < 2^84: h - qP is an integer below 2^32, l an integer below 2^31; result |r| < 1.1 P.
#pragma once
#include "poseidon2.cuh"

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
__device__ __forceinline__ double reduce(double x) {  // |result| <= P/2 + eps
    return __fma_rn(-__builtin_rint(x * PINV), PD, x);
}
__device__ __forceinline__ double sbox7(double x) {
    double x2 = mulmod(x, x), x3 = mulmod(x2, x), x4 = mulmod(x2, x2);
    return mulmod(x3, x4);
}

struct ConstsF64 {
    double ext[8][16];
    double in[13];
    double frac[16];  // canonical value of V[i] for the fractional diagonal entries, 0 elsewhere
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
    uint64_t i2 = c_powmod(2, bb::P - 2);
    // V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/4, 1/8, 2^-27, -2^-8, -1/16, -2^-27]
    c.frac[3] = (double)i2; c.frac[6] = (double)(bb::P - i2);
    c.frac[9] = (double)c_powmod(i2, 8); c.frac[10] = (double)c_powmod(i2, 2); c.frac[11] = (double)c_powmod(i2, 3);
    c.frac[12] = (double)c_powmod(i2, 27); c.frac[13] = (double)(bb::P - c_powmod(i2, 8));
    c.frac[14] = (double)(bb::P - c_powmod(i2, 4)); c.frac[15] = (double)(bb::P - c_powmod(i2, 27));
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

// |s_i| <= 2^33 on entry (canonical inputs or compress inputs < P); every intermediate stays below 2^50.
__device__ __forceinline__ void permute(double (&s)[16]) {
    external_linear(s);  // <= 35 * 2^33 < 2^39
    _Pragma("clang loop unroll(disable)")
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7(s[i] + d_c.ext[r][i]);  // |.| < 1.1 P
        external_linear(s);                                                // < 39 P < 2^37
    }
    _Pragma("clang loop unroll(disable)")
    for (int r = 0; r < 13; r++) {
        s[0] = sbox7(s[0] + d_c.in[r]);
        // the integer-multiplier lanes grow ~4x per round: fold them back every round pair (cheap: 3 ops each)
        if (r & 1) {
            s[1] = reduce(s[1]); s[2] = reduce(s[2]); s[4] = reduce(s[4]); s[5] = reduce(s[5]);
            s[7] = reduce(s[7]); s[8] = reduce(s[8]); s[12] = reduce(s[12]); s[15] = reduce(s[15]);
        }
        double tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])) +
                     (((s[8] + s[9]) + (s[10] + s[11])) + ((s[12] + s[13]) + (s[14] + s[15])));
        tot = reduce(tot);
        s[0] = __fma_rn(s[0], -2.0, tot);
        s[1] = tot + s[1];
        s[2] = __fma_rn(s[2], 2.0, tot);
        s[3] = tot + mulmod(s[3], d_c.frac[3]);
        s[4] = __fma_rn(s[4], 3.0, tot);
        s[5] = __fma_rn(s[5], 4.0, tot);
        s[6] = tot + mulmod(s[6], d_c.frac[6]);
        s[7] = __fma_rn(s[7], -3.0, tot);
        s[8] = __fma_rn(s[8], -4.0, tot);
        s[9] = tot + mulmod(s[9], d_c.frac[9]);
        s[10] = tot + mulmod(s[10], d_c.frac[10]);
        s[11] = tot + mulmod(s[11], d_c.frac[11]);
        s[12] = __fma_rn(s[12], -15.0, tot);  // 2^-27 = -15 (mod P) because P - 1 = 15 * 2^27
        s[13] = tot + mulmod(s[13], d_c.frac[13]);
        s[14] = tot + mulmod(s[14], d_c.frac[14]);
        s[15] = __fma_rn(s[15], 15.0, tot);   // -2^-27 = 15
    }
    _Pragma("clang loop unroll(disable)")
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = sbox7(s[i] + d_c.ext[r][i]);
        external_linear(s);
    }
}

// Montgomery word <-> canonical double
__device__ __forceinline__ double load_elem(uint32_t monty) { return (double)bb::from_monty(monty); }
__device__ __forceinline__ uint32_t store_elem(double v) {
    double r = reduce(v);                 // (-P/2 - eps, P/2 + eps)
    r = r < 0.0 ? r + PD : r;             // [0, P)
    return bb::to_monty((uint32_t)r);
}

}  // namespace p2f
