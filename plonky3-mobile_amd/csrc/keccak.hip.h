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

__device__ __forceinline__ void round(uint64_t (&a)[25], uint64_t rc) {
    uint64_t c[5], b[25];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; x++) {
        const uint64_t d = c[(x + 4) % 5] ^ rotl<1>(c[(x + 1) % 5]);
#pragma unroll
        for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
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
        for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= rc;
}

__device__ __forceinline__ void permute(uint64_t (&a)[25]) {
    _Pragma("clang loop unroll(disable)")
    for (int r = 0; r < 24; r++) round(a, d_rc[r]);
}

// Lane-cooperative form: state word i = x + 5y of one permutation lives in lane i of a HALF-wave (lanes 0..24 and
// 32..56 of a wave carry two independent states; the other lanes carry don't-care values); theta / pi / chi move data
// with wave shuffles inside the half.  About 50 wave-instructions per round instead of ~260 for one lane holding all 25
// words: one permutation costs ~4 us of latency instead of ~13.  Every lane of the wave must call it.  Not inlined:
// one copy per kernel.
__device__ __forceinline__ uint64_t shfl64(uint64_t v, uint32_t src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, (int)src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __noinline__ uint64_t f_coop(uint64_t a) {
    const uint32_t lane = threadIdx.x & 63u, base = lane & 32u, sub = lane & 31u, l = sub < 25u ? sub : 0u, x = l % 5u, y = l / 5u;
    const uint32_t col1 = base + x + 5u * ((y + 1u) % 5u), col2 = base + x + 5u * ((y + 2u) % 5u),
                   col3 = base + x + 5u * ((y + 3u) % 5u), col4 = base + x + 5u * ((y + 4u) % 5u);
    const uint32_t xm1 = base + (x + 4u) % 5u + 5u * y, xp1 = base + (x + 1u) % 5u + 5u * y, xp2 = base + (x + 2u) % 5u + 5u * y;
    // pi: B[y' + 5 ((2x' + 3y') % 5)] = rot(A[x' + 5y']); the lane at (X, Y) therefore reads from x' = (X + 3Y) % 5, y' = X
    const uint32_t pi_src = base + (x + 3u * y) % 5u + 5u * x;
    constexpr uint8_t RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    uint32_t rho = 0;
#pragma unroll
    for (uint32_t i = 0; i < 25; i++) rho = l == i ? RHO[i] : rho;
    // round constants: lane r of the wave keeps RC[r]; the round reads it with v_readlane (no scalar load in the loop)
    const uint64_t rcv = d_rc[lane < 24u ? lane : 0u];
    const uint32_t rc_lo = (uint32_t)rcv, rc_hi = (uint32_t)(rcv >> 32), first = sub == 0 ? 0xffffffffu : 0u;
    _Pragma("clang loop unroll(disable)")
    for (int r = 0; r < 24; r++) {
        const uint64_t c = a ^ shfl64(a, col1) ^ shfl64(a, col2) ^ shfl64(a, col3) ^ shfl64(a, col4);  // column parity
        const uint64_t cp = shfl64(c, xp1);
        a ^= shfl64(c, xm1) ^ ((cp << 1) | (cp >> 63));
        const uint64_t rot = (a << rho) | (a >> ((64u - rho) & 63u));
        const uint64_t bb = shfl64(rot, pi_src);
        a = bb ^ (~shfl64(bb, xp1) & shfl64(bb, xp2));
        const uint32_t k_lo = (uint32_t)__builtin_amdgcn_readlane((int)rc_lo, r) & first, k_hi = (uint32_t)__builtin_amdgcn_readlane((int)rc_hi, r) & first;
        a ^= ((uint64_t)k_hi << 32) | k_lo;
    }
    return a;
}

}  // namespace kk
