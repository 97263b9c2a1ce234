// Keccak-f[1600], one state per lane (25 x u64 = 50 VGPRs), for the hash configuration the reference itself wires
// into its MMCS (native/src/fib_air.rs:28-38: PaddingFreeSponge<KeccakF, 25, 17, 4>, SerializingHasher,
// CompressionFunctionFromHasher<_, 2, 4>; p3-keccak 0.4.2 -> tiny-keccak 2.0.2, both absent: FIPS 202 restated).
// One round is fully unrolled (rho/pi are register renames, rotations are constant v_alignbit_b32 pairs, chi is
// v_bfi/v_xor), the 24 rounds stay ROLLED so the kernel's code stays inside the instruction cache — the same
// lesson as the Poseidon2 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kk {

static __device__ __constant__ uint64_t d_rc[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

__device__ __forceinline__ constexpr unsigned rho(int i) {
    constexpr unsigned R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    return R[i];
}
// 64-bit rotate by a constant as two v_alignbit_b32 (hipcc otherwise emits 64-bit shifts + or: twice the issue time)
template <unsigned N>
__device__ __forceinline__ uint64_t rotl(uint64_t v) {
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    if constexpr (N == 0) return v;
    else if constexpr (N == 32) return ((uint64_t)lo << 32) | hi;
    else if constexpr (N < 32) {
        const uint32_t nh = __builtin_amdgcn_alignbit(hi, lo, 32 - N), nl = __builtin_amdgcn_alignbit(lo, hi, 32 - N);
        return ((uint64_t)nh << 32) | nl;
    } else {
        constexpr unsigned M = N - 32;
        const uint32_t nh = __builtin_amdgcn_alignbit(lo, hi, 32 - M), nl = __builtin_amdgcn_alignbit(hi, lo, 32 - M);
        return ((uint64_t)nh << 32) | nl;
    }
}

// gfx950's three-input bit operation (v_bitop3_b32: any boolean function of three words, truth table in the immediate,
// evaluated with a = 0xF0, b = 0xCC, c = 0xAA) on both halves of a 64-bit word.  hipcc does not form it from the 64-bit
// source expressions of the round below (it emits v_xor + v_bfi pairs): spelled out, a round is 180 VALU instructions
// instead of 260 (theta 20 + 50 three-way XORs instead of 40 + 10 + 50 two-way ones, chi 50 instead of 100).
template <int TT>
__device__ __forceinline__ uint64_t bitop3_64(uint64_t a, uint64_t b, uint64_t c) {
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, TT);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), TT);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t xor3(uint64_t a, uint64_t b, uint64_t c) { return bitop3_64<0x96>(a, b, c); }          // a ^ b ^ c
__device__ __forceinline__ uint64_t chi3(uint64_t a, uint64_t b, uint64_t c) { return bitop3_64<0xd2>(a, b, c); }          // a ^ (~b & c)

__device__ __forceinline__ void round(uint64_t (&a)[25], uint64_t rc) {
    uint64_t c[5], b[25];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = xor3(xor3(a[x], a[x + 5], a[x + 10]), a[x + 15], a[x + 20]);
#pragma unroll
    for (int x = 0; x < 5; x++) {
        const uint64_t cm = c[(x + 4) % 5], cr = rotl<1>(c[(x + 1) % 5]);
#pragma unroll
        for (int y = 0; y < 5; y++) a[x + 5 * y] = xor3(a[x + 5 * y], cm, cr);
    }
    // rho + pi: B[y, 2x + 3y] = rot(A[x, y], r[x, y]); written out so that every rotation count is a constant
#define KK_RP(x, y) b[(y) + 5 * ((2 * (x) + 3 * (y)) % 5)] = rotl<rho((x) + 5 * (y))>(a[(x) + 5 * (y)]);
#define KK_ROW(y) KK_RP(0, y) KK_RP(1, y) KK_RP(2, y) KK_RP(3, y) KK_RP(4, y)
    KK_ROW(0) KK_ROW(1) KK_ROW(2) KK_ROW(3) KK_ROW(4)
#undef KK_ROW
#undef KK_RP
#pragma unroll
    for (int y = 0; y < 5; y++)
#pragma unroll
        for (int x = 0; x < 5; x++) a[x + 5 * y] = chi3(b[x + 5 * y], b[(x + 1) % 5 + 5 * y], b[(x + 2) % 5 + 5 * y]);
    a[0] ^= rc;
}

__device__ __forceinline__ void permute(uint64_t (&a)[25]) {
    _Pragma("clang loop unroll(disable)")
    for (int r = 0; r < 24; r++) round(a, d_rc[r]);
}

// Keccak-f when only the first four words of the result are wanted (a digest: every MMCS hash and compression): the
// last round computes just row y = 0, which needs theta on the diagonal a[0], a[6], a[12], a[18], a[24], four rotations
// and four chi words — 58 VALU instructions instead of 180.  Words 4..24 of `a` are left unspecified.
// The FIRST round is peeled out of the rolled loop: every caller absorbs one block into a zeroed state, so at least eight of
// the 25 words entering it are compile-time zeros (17 for a compression, 19-20 for a salted leaf) and theta folds — column
// parities of one or two words, a[x, y] ^ D[x] = D[x] for the zero words — 142 instead of 180 instructions for a compression.
__device__ __forceinline__ void permute_digest(uint64_t (&a)[25]) {
    round(a, 0x0000000000000001ULL);
    _Pragma("clang loop unroll(disable)")
    for (int r = 1; r < 23; r++) round(a, d_rc[r]);
    uint64_t c[5], b[5];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = xor3(xor3(a[x], a[x + 5], a[x + 10]), a[x + 15], a[x + 20]);
    // B[x, 0] = rot(A[x, x] ^ D[x], rho[x, x]): the row of B that chi turns into the first five words
#define KK_DIAG(x) b[x] = rotl<rho(6 * (x))>(xor3(a[6 * (x)], c[((x) + 4) % 5], rotl<1>(c[((x) + 1) % 5])));
    KK_DIAG(0) KK_DIAG(1) KK_DIAG(2) KK_DIAG(3) KK_DIAG(4)
#undef KK_DIAG
#pragma unroll
    for (int x = 0; x < 4; x++) a[x] = chi3(b[x], b[(x + 1) % 5], b[(x + 2) % 5]);
    a[0] ^= d_rc[23];
}

// Lane-cooperative form, ONE state per wave: state word A[x + 5y] lives in lane x + 8y (x, y < 5; the other 39 lanes are
// padding and hold zero between the steps).  With rows of eight, everything that moves along x (theta's D, chi) is a
// DPP row shift inside a 16-lane row — the wrap-around x = 4 -> 0 is a second shift restricted to half the banks, and
// the zero padding makes the two shifted copies simply XOR together; the column parity is one row_ror:8 plus the
// gfx950 v_permlane32_swap / v_permlane16_swap pair; rho is a per-lane v_alignbit pair; only pi crosses lanes
// arbitrarily (two ds_bpermute).  68 wave-instructions and ONE LDS round trip per round (the first version:
// shuffles for everything, 18 ds_bpermute in five dependent stages) — a permutation costs ~5 us instead of ~9, the
// same ~9 us as one lane holding all 25 words (but a lane-instruction cost ~13x higher: small layers only).  Every lane of the wave must call it.  Not inlined: one copy per kernel.
__device__ __forceinline__ uint32_t coop_index() {  // which state word this lane holds; 25 = padding lane
    const uint32_t lane = threadIdx.x & 63u, x = lane & 7u, y = lane >> 3;
    return (x < 5u && y < 5u) ? x + 5u * y : 25u;
}
template <int CTRL, int BANKS>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {  // lanes without a source (other row, disabled bank) read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, BANKS, true);
}
constexpr int ROW_SHL = 0x100, ROW_SHR = 0x110, ROW_ROR = 0x120;
__device__ __forceinline__ uint32_t wave_column_xor(uint32_t v) {  // XOR over the eight lanes x + 8y of a column, in every lane
    const uint32_t p = v ^ dpp0<ROW_ROR + 8, 0xf>(v);
    const auto s = __builtin_amdgcn_permlane32_swap(p, p, false, false);
    const uint32_t u = s[0] ^ s[1];
    const auto t = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return t[0] ^ t[1];
}
__device__ __noinline__ uint64_t f_coop(uint64_t a) {
    const uint32_t lane = threadIdx.x & 63u, x = lane & 7u, y = lane >> 3, idx = coop_index();
    const uint32_t M = idx < 25u ? 0xffffffffu : 0u, first = lane == 0 ? 0xffffffffu : 0u;
    // pi: B[y' + 5 ((2x' + 3y') % 5)] = rot(A[x' + 5y']); the lane at (X, Y) therefore reads from x' = (X + 3Y) % 5, y' = X
    const int pi_addr = (int)(4u * (idx < 25u ? (x + 3u * y) % 5u + 8u * x : lane));
    constexpr uint8_t RHO[26] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14, 0};
    uint32_t rho = 0;
#pragma unroll
    for (uint32_t i = 0; i < 25; i++) rho = idx == i ? RHO[i] : rho;
    // rotl64 by rho as (optional swap of the halves) + two v_alignbit by sh = (32 - rho % 32) % 32; rho = 0 is "swap, then rotate by 32"
    const bool swap = rho >= 32u || rho == 0u;
    const uint32_t sh = (32u - (rho & 31u)) & 31u;
    const uint64_t rcv = d_rc[lane < 24u ? lane : 0u];  // lane r keeps RC[r]: read with v_readlane, no scalar load in the loop
    const uint32_t rc_lo = (uint32_t)rcv, rc_hi = (uint32_t)(rcv >> 32);
    uint32_t lo = (uint32_t)a & M, hi = (uint32_t)(a >> 32) & M;
    _Pragma("clang loop unroll(disable)")
    for (int r = 0; r < 24; r++) {
        // theta
        const uint32_t c_lo = wave_column_xor(lo), c_hi = wave_column_xor(hi);
        const uint32_t m_lo = dpp0<ROW_SHR + 1, 0xf>(c_lo) ^ dpp0<ROW_SHL + 4, 0x5>(c_lo);  // C[x - 1]
        const uint32_t m_hi = dpp0<ROW_SHR + 1, 0xf>(c_hi) ^ dpp0<ROW_SHL + 4, 0x5>(c_hi);
        const uint32_t p_lo = dpp0<ROW_SHL + 1, 0xf>(c_lo) ^ dpp0<ROW_SHR + 4, 0xa>(c_lo);  // C[x + 1]
        const uint32_t p_hi = dpp0<ROW_SHL + 1, 0xf>(c_hi) ^ dpp0<ROW_SHR + 4, 0xa>(c_hi);
        lo = (lo ^ m_lo ^ __builtin_amdgcn_alignbit(p_lo, p_hi, 31)) & M;
        hi = (hi ^ m_hi ^ __builtin_amdgcn_alignbit(p_hi, p_lo, 31)) & M;
        // rho, then pi
        const uint32_t l2 = swap ? hi : lo, h2 = swap ? lo : hi;
        const uint32_t b_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(pi_addr, (int)__builtin_amdgcn_alignbit(l2, h2, sh));
        const uint32_t b_hi = (uint32_t)__builtin_amdgcn_ds_bpermute(pi_addr, (int)__builtin_amdgcn_alignbit(h2, l2, sh));
        // chi
        const uint32_t b1_lo = dpp0<ROW_SHL + 1, 0xf>(b_lo) ^ dpp0<ROW_SHR + 4, 0xa>(b_lo), b1_hi = dpp0<ROW_SHL + 1, 0xf>(b_hi) ^ dpp0<ROW_SHR + 4, 0xa>(b_hi);
        const uint32_t b2_lo = dpp0<ROW_SHL + 2, 0xf>(b_lo) ^ dpp0<ROW_SHR + 3, 0xf>(b_lo), b2_hi = dpp0<ROW_SHL + 2, 0xf>(b_hi) ^ dpp0<ROW_SHR + 3, 0xf>(b_hi);
        // iota: RC[r] into lane 0
        const uint32_t k_lo = (uint32_t)__builtin_amdgcn_readlane((int)rc_lo, r) & first, k_hi = (uint32_t)__builtin_amdgcn_readlane((int)rc_hi, r) & first;
        lo = ((b_lo ^ (~b1_lo & b2_lo)) & M) ^ k_lo;
        hi = ((b_hi ^ (~b1_hi & b2_hi)) & M) ^ k_hi;
    }
    return ((uint64_t)hi << 32) | lo;
}

}  // namespace kk
