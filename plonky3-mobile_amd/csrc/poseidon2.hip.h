// Poseidon2 over BabyBear, width 16, x^7, 4 + 13 + 4 rounds — the permutation the reference names
// at native/src/poseidon_cpu.rs:17-18 (default_babybear_poseidon2_16()).  One state per lane, all 16
// words in VGPRs, every loop unrolled so the round constants fold into instruction literals.
// Integer-VALU bound (about 560 Montgomery products per permutation); no LDS, no MFMA.
//   external layer: circ(2*M4, M4, M4, M4), M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
//   internal layer: 1 + diag(-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 2^-8, 1/4, 1/8, 2^-27, -2^-8, -1/16, -2^-27)
// The diagonal is applied with shifts/adds: division by 2^k is exact Montgomery-style halving, using
// P = 1 (mod 2^27): x/2^k = ((x + m) >> k) + 15*m << (27-k),  m = (-x) mod 2^k.
#pragma once
#include "bb31.hip.h"
#include "poseidon2_rc16.h"

namespace p2 {

template <int K>
BB_HD uint32_t div2k(uint32_t x) {
    static_assert(K >= 1 && K <= 27, "");
    if constexpr (K == 1) {
        // x/2 = (x >> 1) + (x odd ? (P+1)/2 : 0): four ops
        return (x >> 1) + ((0u - (x & 1u)) & ((bb::P + 1u) >> 1));
    }
    uint32_t m = (0u - x) & ((1u << K) - 1u);
    return ((x + m) >> K) + ((m * 15u) << (27 - K));
}

// [2 3 1 1; 1 2 3 1; 1 1 2 3; 3 1 1 2] with 7 additions and 2 doublings
BB_HD void mat4(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    uint32_t t01 = bb::add(a, b), t23 = bb::add(c, d);
    uint32_t t0123 = bb::add(t01, t23);
    uint32_t t01123 = bb::add(t0123, b), t01233 = bb::add(t0123, d);
    uint32_t nd = bb::add(t01233, bb::dbl(a));  // 3a + b + c + 2d
    uint32_t nb = bb::add(t01123, bb::dbl(c));  // a + 2b + 3c + d
    uint32_t na = bb::add(t01123, t01);         // 2a + 3b + c + d
    uint32_t nc = bb::add(t01233, t23);         // a + b + 2c + 3d
    a = na; b = nb; c = nc; d = nd;
}

BB_HD void external_linear(uint32_t (&s)[16]) {
#pragma unroll
    for (int i = 0; i < 16; i += 4) mat4(s[i], s[i + 1], s[i + 2], s[i + 3]);
    uint32_t t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = bb::add(bb::add(s[k], s[k + 4]), bb::add(s[k + 8], s[k + 12]));
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = bb::add(s[i], t[i & 3]);
}

BB_HD void internal_linear(uint32_t (&s)[16]) {
    uint32_t p0 = bb::add(bb::add(s[1], s[2]), bb::add(s[3], s[4]));
    uint32_t p1 = bb::add(bb::add(s[5], s[6]), bb::add(s[7], s[8]));
    uint32_t p2 = bb::add(bb::add(s[9], s[10]), bb::add(s[11], s[12]));
    uint32_t p3 = bb::add(bb::add(s[13], s[14]), s[15]);
    uint32_t part = bb::add(bb::add(p0, p1), bb::add(p2, p3));
    uint32_t tot = bb::add(part, s[0]);
    s[0] = bb::sub(part, s[0]);                       // -2
    s[1] = bb::add(tot, s[1]);                        // 1
    s[2] = bb::add(tot, bb::dbl(s[2]));               // 2
    s[3] = bb::add(tot, div2k<1>(s[3]));              // 1/2
    s[4] = bb::add(tot, bb::add(bb::dbl(s[4]), s[4]));  // 3
    s[5] = bb::add(tot, bb::dbl(bb::dbl(s[5])));      // 4
    s[6] = bb::sub(tot, div2k<1>(s[6]));              // -1/2
    s[7] = bb::sub(tot, bb::add(bb::dbl(s[7]), s[7]));  // -3
    s[8] = bb::sub(tot, bb::dbl(bb::dbl(s[8])));      // -4
    s[9] = bb::add(tot, div2k<8>(s[9]));              // 2^-8
    s[10] = bb::add(tot, div2k<2>(s[10]));            // 1/4
    s[11] = bb::add(tot, div2k<3>(s[11]));            // 1/8
    s[12] = bb::add(tot, div2k<27>(s[12]));           // 2^-27
    s[13] = bb::sub(tot, div2k<8>(s[13]));            // -2^-8
    s[14] = bb::sub(tot, div2k<4>(s[14]));            // -1/16
    s[15] = bb::sub(tot, div2k<27>(s[15]));           // -2^-27
}

// Round constants for device code live in __constant__ memory and the round loops stay ROLLED: fully
// unrolled, the permutation is ~55 KB of straight-line code and every wave streams it through the 64 KB
// instruction cache at its own position — the kernel then runs at instruction-fetch speed, not VALU speed
// (measured: 13% fewer VALU ops changed nothing while unrolled).  Rolled, the loop bodies are ~6 KB.
struct RoundConstants {
    uint32_t ext[8][16];
    uint32_t in[13];
};
constexpr RoundConstants make_round_constants() {
    RoundConstants rc{};
    for (int r = 0; r < 4; r++)
        for (int i = 0; i < 16; i++) {
            rc.ext[r][i] = P3_RC16_EXT_INIT_MONTY[r][i];
            rc.ext[4 + r][i] = P3_RC16_EXT_FINAL_MONTY[r][i];
        }
    for (int r = 0; r < 13; r++) rc.in[r] = P3_RC16_INTERNAL_MONTY[r];
    return rc;
}
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __constant__ RoundConstants d_rc = make_round_constants();
#define P2_RC d_rc
#define P2_ROLLED _Pragma("clang loop unroll(disable)")
#else
static constexpr RoundConstants h_rc = make_round_constants();
#define P2_RC h_rc
#define P2_ROLLED
#endif

BB_HD void permute(uint32_t (&s)[16]) {
    external_linear(s);
    P2_ROLLED
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = bb::sbox7_add(s[i], P2_RC.ext[r][i]);
        external_linear(s);
    }
    P2_ROLLED
    for (int r = 0; r < 13; r++) {
        s[0] = bb::sbox7_add(s[0], P2_RC.in[r]);
        internal_linear(s);
    }
    P2_ROLLED
    for (int r = 4; r < 8; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) s[i] = bb::sbox7_add(s[i], P2_RC.ext[r][i]);
        external_linear(s);
    }
}


}  // namespace p2
