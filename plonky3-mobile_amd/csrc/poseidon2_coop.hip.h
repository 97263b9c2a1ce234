// Lane-cooperative Poseidon2 (width 16): ONE state spread over 16 consecutive lanes (= one DPP row), one
// element per lane.  Same permutation as poseidon2.hip.h, restructured for LATENCY: ~1.1k wave-instructions per
// permutation instead of ~7.3k, so the small Merkle layers (tree tops, FRI tail) whose time is one
// permutation latency per level finish ~6x sooner.  It spends ~2.5x more lane-ops per permutation, so the
// large layers keep the one-state-per-lane kernels.
//   external layer: M4 inside each quad via DPP quad_perm, the 4-quad column sum via DPP row_ror:4/8/12
//   internal layer: row sum by rotate-and-add (row_ror 8,4,2,1), diagonal as one Montgomery product per lane
#pragma once
#include "poseidon2.hip.h"

namespace p2c {

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
constexpr int QP(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int ROW_ROR(int n) { return 0x120 + n; }

// out_p = 2 x_p + 3 x_{p+1} + x_{p+2} + x_{p+3} (indices cyclic in the quad) = S + x_p + 2 x_{p+1}, then add the
// same row of the other three quads.
__device__ __forceinline__ uint32_t external_linear(uint32_t x) {
    uint32_t n1 = dpp<QP(1, 2, 3, 0)>(x), n2 = dpp<QP(2, 3, 0, 1)>(x), n3 = dpp<QP(3, 0, 1, 2)>(x);
    uint32_t s = bb::add(bb::add(x, n1), bb::add(n2, n3));
    uint32_t y = bb::add(bb::add(s, x), bb::dbl(n1));
    // lanes l, l+4, l+8, l+12 of the row hold the same matrix row of the four blocks
    uint32_t t = bb::add(bb::add(y, dpp<ROW_ROR(4)>(y)), bb::add(dpp<ROW_ROR(8)>(y), dpp<ROW_ROR(12)>(y)));
    return bb::add(y, t);
}
__device__ __forceinline__ uint32_t row_sum(uint32_t v) {
    v = bb::add(v, dpp<ROW_ROR(8)>(v));
    v = bb::add(v, dpp<ROW_ROR(4)>(v));
    v = bb::add(v, dpp<ROW_ROR(2)>(v));
    return bb::add(v, dpp<ROW_ROR(1)>(v));
}

struct LaneConst {
    uint32_t rc[8];  // external round constants of this lane's state element
    uint32_t diag;   // Montgomery form of V[lane]
    uint32_t lane0;  // 1 on the lane holding element 0
};

// V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/4, 1/8, 2^-27, -2^-8, -1/16, -2^-27] in Montgomery form
struct DiagTable { uint32_t v[16]; };
constexpr uint32_t cmul(uint32_t a, uint32_t b) {  // constexpr Montgomery product
    uint64_t x = (uint64_t)a * b;
    uint32_t t = (uint32_t)x * bb::MU;
    uint64_t u = (uint64_t)t * bb::P;
    uint32_t hi = (uint32_t)(x >> 32), uh = (uint32_t)(u >> 32);
    return hi >= uh ? hi - uh : hi - uh + bb::P;
}
constexpr uint32_t cpow(uint32_t b, uint64_t e) { uint32_t r = bb::ONE; while (e) { if (e & 1) r = cmul(r, b); b = cmul(b, b); e >>= 1; } return r; }
constexpr uint32_t cneg(uint32_t a) { return a ? bb::P - a : 0; }
constexpr uint32_t cmont(uint32_t c) { return cmul(c, bb::R2); }
constexpr DiagTable make_diag() {
    DiagTable d{};
    uint32_t i2 = cpow(cmont(2), bb::P - 2);
    uint32_t i2_8 = cpow(i2, 8), i2_27 = cpow(i2, 27);
    d.v[0] = cneg(cmont(2)); d.v[1] = bb::ONE; d.v[2] = cmont(2); d.v[3] = i2; d.v[4] = cmont(3); d.v[5] = cmont(4);
    d.v[6] = cneg(i2); d.v[7] = cneg(cmont(3)); d.v[8] = cneg(cmont(4)); d.v[9] = i2_8; d.v[10] = cpow(i2, 2);
    d.v[11] = cpow(i2, 3); d.v[12] = i2_27; d.v[13] = cneg(i2_8); d.v[14] = cneg(cpow(i2, 4)); d.v[15] = cneg(i2_27);
    return d;
}
static __device__ __constant__ DiagTable d_diag = make_diag();

__device__ __forceinline__ LaneConst lane_constants(uint32_t lane16) {
    LaneConst c;
#pragma unroll
    for (int r = 0; r < 8; r++) c.rc[r] = p2::P2_RC.ext[r][lane16];
    c.diag = d_diag.v[lane16];
    c.lane0 = lane16 == 0;
    return c;
}

// v: this lane's element of the state; all 16 lanes of the row must be active.
__device__ __forceinline__ uint32_t permute(uint32_t v, const LaneConst& c) {
    v = external_linear(v);
#pragma unroll
    for (int r = 0; r < 4; r++) v = external_linear(bb::sbox7_add(v, c.rc[r]));
    _Pragma("clang loop unroll(disable)")
    for (int r = 0; r < 13; r++) {
        uint32_t sb = bb::sbox7_add(v, p2::P2_RC.in[r]);
        v = c.lane0 ? sb : v;
        uint32_t tot = row_sum(v);
        v = bb::add(tot, bb::mul(v, c.diag));
    }
#pragma unroll
    for (int r = 4; r < 8; r++) v = external_linear(bb::sbox7_add(v, c.rc[r]));
    return v;
}

}  // namespace p2c
