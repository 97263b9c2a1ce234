// Poseidon2 (width 16) with ONE state per DPP QUAD: lane q of the quad holds elements 4q .. 4q+3 as doubles (the exact-integer
// fp64 arithmetic of poseidon2_f64.hip.h).  For the Merkle layers of 2^12 .. 2^15 digests: with one state per lane such a layer
// is at most one wave per SIMD, so its time is one permutation's issue time (~4160 dependent fp64 instructions: 11.5-12.4 us a
// launch whatever the layer's size, 45 of them in a 2^20 proof = 0.53 ms, profiles/r03_single_proof_latency_breakdown.txt);
// the 16-lanes-per-state form (poseidon2_coop.hip.h, ~1.1k instructions) costs 4.2x the lane-instructions and is kept for layers
// below 2^12.  Here a wave carries 16 states through ~1.6k instructions: 1.6x the lane-instructions of the per-lane form.
//   external layer: x^7 on the lane's four elements, M4 of circ(2 M4, M4, M4, M4) inside the lane (a block of the matrix IS a
//                   lane), the four-block column sums by two quad_perm exchanges per element;
//   internal layer: x^7 on element 0 computed by every lane and kept by lane 0, the sum of the sixteen elements by an in-lane
//                   sum and two exchanges, the diagonal as `tot + x * V` in the three-operation form that is exact for the
//                   integer multipliers and for the 2^-k ones alike (p2f::add_scaled_pow2; V per lane in registers).
// Same values as p2f::permute for every input (exact integer arithmetic, only the grouping of the additions differs).
#pragma once
#include "poseidon2_f64.hip.h"

namespace p2q {

template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
constexpr int QP_XOR1 = 1 | (0 << 2) | (3 << 4) | (2 << 6);  // lane ^ 1 inside the quad
constexpr int QP_XOR2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);  // lane ^ 2
__device__ __forceinline__ double quad_sum(double v) {  // exact integers: every lane of the quad ends with the same value
    v += dpp_d<QP_XOR1>(v);
    return v + dpp_d<QP_XOR2>(v);
}

struct LaneConsts {
    double ext[8][4];  // external round constants of this lane's four elements
    double diag[4];    // V[4q + i] as a double: an integer, or +-2^-k (2^-27 = -15 mod P)
};
__device__ __forceinline__ LaneConsts lane_consts(uint32_t q) {
    // V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/4, 1/8, 2^-27 = -15, -2^-8, -1/16, -2^-27 = 15]
    constexpr double V[16] = {-2.0, 1.0, 2.0, 0.5, 3.0, 4.0, -0.5, -3.0, -4.0, 0.00390625, 0.25, 0.125, -15.0, -0.00390625, -0.0625, 15.0};
    LaneConsts c;
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 4; i++) c.ext[r][i] = p2f::d_c.ext[r][4 * q + i];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double v = V[i];
#pragma unroll
        for (int k = 1; k < 4; k++) v = q == (uint32_t)k ? V[4 * k + i] : v;
        c.diag[i] = v;
    }
    return c;
}

__device__ __forceinline__ void external_linear(double (&s)[4]) {
    p2f::mat4(s[0], s[1], s[2], s[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] += quad_sum(s[i]);
}

// s: this lane's four elements (integers |s_i| <= 2^33 on entry, as p2f::permute); every lane of the quad must be active.
__device__ __forceinline__ void permute(double (&s)[4], const LaneConsts& c, bool lane0) {
    const p2f::MagicRegs mk = p2f::magic_regs();
    const double npm1 = p2f::d_c.neg_pm1, pinv = p2f::d_c.pinv;
    external_linear(s);
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) s[i] = p2f::sbox7(s[i] + c.ext[r][i], mk, npm1, pinv);
        external_linear(s);
    }
    // internal rounds: the integer-multiplier elements grow up to 15x per round and are folded back after every fourth round
    // (15^4 2^37 < 2^53), exactly as in p2f::internal_rounds; folding the fractional-multiplier elements too is harmless
    _Pragma("clang loop unroll(disable)")
    for (int r = 0; r < 13; r++) {
        const double sb = p2f::sbox7(s[0] + p2f::d_c.in[r], mk, npm1, pinv);
        s[0] = lane0 ? sb : s[0];
        const double tot = p2f::reduce(quad_sum((s[0] + s[1]) + (s[2] + s[3])));
#pragma unroll
        for (int i = 0; i < 4; i++) s[i] = p2f::add_scaled_pow2(tot, s[i], c.diag[i]);
        if ((r & 3) == 3 || r == 12) {
#pragma unroll
            for (int i = 0; i < 4; i++) s[i] = p2f::reduce(s[i]);
        }
    }
#pragma unroll
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) s[i] = p2f::sbox7(s[i] + c.ext[r][i], mk, npm1, pinv);
        external_linear(s);
    }
}

}  // namespace p2q
