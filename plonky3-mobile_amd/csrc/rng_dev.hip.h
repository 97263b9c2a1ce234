// Device-side pieces of the SmallRng streams shared by rng.hip (the fills) and prover.hip (the one-launch prover of tiny
// instances, prover_tiny.hip.inc): xoshiro256++ itself and the GF(2) jump by ONE WAVE in the lane-interleaved basis.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bb31.hip.h"

namespace p3 {

constexpr uint32_t RNG_CHUNK_LOG = 8, RNG_CHUNK = 1u << RNG_CHUNK_LOG;  // raw draws per lane
constexpr uint32_t RNG_MAX_JUMP = 22;                                    // up to 2^22 chunks per fill
constexpr uint32_t RNG_TINY_CHUNK_LOG = 6;                               // the one-launch prover's short chunks (jump slot RNG_MAX_JUMP = T^64)

__host__ __device__ __forceinline__ uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
__host__ __device__ __forceinline__ uint64_t xoshiro_next(uint64_t (&s)[4]) {
    const uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}


// ---- device ----
// GF(2) matrix-vector product by ONE WAVE in the lane-interleaved basis (see jump_matrices): the state is wave-uniform
// (four 64-bit words in SGPRs), lane l holds the four rows that produce its bits, an output bit is the parity of
// row & state, and the new state words are the four BALLOTS of those bits — no shuffle, no LDS, ~80 instructions.
struct LaneRows { uint64_t r[4][4]; };
__device__ __forceinline__ LaneRows load_rows(const uint64_t* __restrict__ m) {
    LaneRows k;
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const ulonglong2* p = reinterpret_cast<const ulonglong2*>(m + ((size_t)i * 64 + lane) * 4);
        const ulonglong2 a = p[0], b = p[1];
        k.r[i][0] = a.x; k.r[i][1] = a.y; k.r[i][2] = b.x; k.r[i][3] = b.y;
    }
    return k;
}
__device__ __forceinline__ void wave_matvec(const LaneRows& k, uint64_t (&s)[4]) {
    uint64_t out[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t x = (k.r[i][0] & s[0]) ^ (k.r[i][1] & s[1]) ^ (k.r[i][2] & s[2]) ^ (k.r[i][3] & s[3]);
        const uint32_t f = (uint32_t)x ^ (uint32_t)(x >> 32);
        out[i] = __ballot((__builtin_popcount(f) & 1) != 0);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = out[i];
}
// original basis -> lane-interleaved basis of a wave-uniform state: word i, bit l = original bit 4l + i
__device__ __forceinline__ void to_interleaved(uint64_t (&s)[4]) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t word = lane < 32u ? (lane < 16u ? s[0] : s[1]) : (lane < 48u ? s[2] : s[3]);
    const uint32_t bit0 = (lane & 15u) * 4u;
    uint64_t out[4];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) out[i] = __ballot(((word >> (bit0 + i)) & 1ull) != 0);
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = out[i];
}
// lane-interleaved -> original basis, per lane (each lane its own state): original word W = bits 4l + i of the
// interleaved words for l = 16W .. 16W + 15, i.e. a 4-way bit interleave of four 16-bit pieces
__device__ __forceinline__ uint64_t spread16(uint64_t x) {  // bit k of x (k < 16) -> bit 4k
    x &= 0xffffull;
    x = (x | (x << 24)) & 0x000000ff000000ffull;
    x = (x | (x << 12)) & 0x000f000f000f000full;
    x = (x | (x << 6)) & 0x0303030303030303ull;
    x = (x | (x << 3)) & 0x1111111111111111ull;
    return x;
}
__device__ __forceinline__ void from_interleaved(uint64_t (&s)[4]) {
    uint64_t out[4];
#pragma unroll
    for (int w = 0; w < 4; w++)
        out[w] = spread16(s[0] >> (16 * w)) | (spread16(s[1] >> (16 * w)) << 1) | (spread16(s[2] >> (16 * w)) << 2) |
                 (spread16(s[3] >> (16 * w)) << 3);
#pragma unroll
    for (int w = 0; w < 4; w++) s[w] = out[w];
}
// SMALL fills (up to 32 chunks = 8192 raw draws) by ONE wave: lane l walks to chunk l (only the chunks the fill can need), generates
// its 256 candidates into `raw` (LDS, RNG_SMALL_CHUNKS x RNG_SMALL_STRIDE words), the wave scans the counts in registers and compacts
// chunk by chunk with ballot ranks.  Returns the number of accepted draws among the chunks generated (the caller raises its
// shortage flag when that is below n) and leaves, in `s_start` / `my_base` / `cnt`, what the lane owning the n-th element needs to
// replay its chunk for the generator state.  Every lane of the wave must call it; `raw` is this wave's own.
constexpr uint32_t RNG_SMALL_CHUNKS = 32, RNG_SMALL_STRIDE = RNG_CHUNK + 1;
// CHUNK_LOG = 8: chunks of 256 raw draws (jump = J_0 = T^256), up to RNG_SMALL_CHUNKS of them; CHUNK_LOG = 6: chunks of 64 (jump = T^64, slot
// RNG_MAX_JUMP of the table), up to 64 of them — a lane's 64 sequential draws instead of 256 are what a fill of a few hundred elements waits for.
// `raw`: n_chunks x (2^CHUNK_LOG + 1) words.
template <uint32_t CHUNK_LOG>
__device__ __forceinline__ uint32_t rng_small_fill_wave(const uint64_t (&seed)[4], const uint64_t* __restrict__ jump, uint32_t n_chunks, uint32_t* raw,
                                                        uint32_t* out, uint32_t n, uint64_t (&s_start)[4], uint32_t& my_base, uint32_t& cnt) {
    constexpr uint32_t CHUNK = 1u << CHUNK_LOG, STRIDE = CHUNK + 1, PER = CHUNK / 64;
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t cur[4] = {seed[0], seed[1], seed[2], seed[3]};
    to_interleaved(cur);
    const LaneRows j0 = load_rows(jump);
    uint64_t s[4] = {cur[0], cur[1], cur[2], cur[3]};
    for (uint32_t i = 1; i < n_chunks; i++) {
        wave_matvec(j0, cur);
        if (lane == i) { s[0] = cur[0]; s[1] = cur[1]; s[2] = cur[2]; s[3] = cur[3]; }
    }
    from_interleaved(s);
#pragma unroll
    for (int w = 0; w < 4; w++) s_start[w] = s[w];
    cnt = 0;
    if (lane < n_chunks) {
        for (uint32_t i = 0; i < CHUNK; i++) {
            const uint32_t v = (uint32_t)(xoshiro_next(s) >> 32) >> 1;
            cnt += v < bb::P ? 1u : 0u;
            raw[lane * STRIDE + i] = v;
        }
    }
    uint32_t inc = cnt;  // inclusive scan of the chunk counts over the wave
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)inc, off, 64);
        if (lane >= off) inc += u;
    }
    my_base = inc - cnt;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // one wave: orders the LDS writes before the reads below
    __builtin_amdgcn_wave_barrier();
    for (uint32_t c = 0; c < n_chunks; c++) {  // wave-uniform
        const uint32_t base = (uint32_t)__shfl((int)my_base, (int)c, 64);
        if (base >= n) break;
        uint32_t v[PER], below = 0;
        bool acc[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            v[k] = raw[c * STRIDE + PER * lane + k];
            acc[k] = v[k] < bb::P;
            const uint64_t b = __ballot(acc[k]);
            below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, below));
        }
        uint32_t rank = base + below;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++)
            if (acc[k]) { if (rank < n) out[rank] = v[k]; rank++; }
    }
    return (uint32_t)__shfl((int)inc, 63, 64);
}
// chunks a small fill of n elements generates (margin n / 8 + 1024 raw draws: > 50 standard deviations of the rejections), or 0
// when it does not fit one wave's small fill
__host__ __device__ __forceinline__ uint32_t rng_small_chunks(uint64_t n) {
    const uint64_t raw_small = n + n / 8 + 1024;
    const uint64_t chunks_small = (raw_small + RNG_CHUNK - 1) / RNG_CHUNK;
    return chunks_small <= RNG_SMALL_CHUNKS ? (uint32_t)chunks_small : 0u;
}

// the same with chunks of 64 raw draws, up to 64 chunks; margin n / 8 + 256 (> 20 standard deviations of the rejections for n <= 3400)
__host__ __device__ __forceinline__ uint32_t rng_tiny_chunks(uint64_t n) {
    const uint64_t raw = n + n / 8 + 256;
    const uint64_t chunks = (raw + (1u << RNG_TINY_CHUNK_LOG) - 1) >> RNG_TINY_CHUNK_LOG;
    return chunks <= 64 ? (uint32_t)chunks : 0u;
}

}  // namespace p3
