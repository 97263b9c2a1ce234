// The random streams of the reference's hiding configuration, generated ON THE DEVICE with the exact sequential
// semantics of the host generator: `SmallRng::seed_from_u64(1)` (native/src/fib_air.rs:50,65) = xoshiro256++ seeded
// through SplitMix64 (rand 0.9.2, absent: restated from the published algorithms), BabyBear elements drawn by
// rejection of 31-bit words (p3-monty-31's StandardUniform: v = next_u32() >> 1, accepted when v < P, used as the
// Montgomery word) [UPSTREAM-RECALL for the sampling convention].
//
// A sequential generator with data-dependent rejections, in parallel:
//   - the state transition of xoshiro256++ is linear over GF(2), so "advance by m steps" is a 256 x 256 bit matrix;
//     the host builds J_k = T^(CHUNK * 2^k) once; a wave reaches its first chunk of CHUNK raw draws with the matrices
//     of that chunk index's set bits and walks J_0 to the next 63, each product done by the whole wave in a
//     lane-interleaved basis (lane l owns four rows; the four ballots of the parity bits ARE the product: ~80
//     instructions, no shuffle, instead of ~3000 for one lane);
//   - pass 1 generates every chunk ONCE: it counts the accepted draws and keeps all 256 raw candidates of the chunk, written
//     through a 64 x 32 LDS tile so that each chunk's candidates leave in whole 128-byte pieces; a two-level scan turns the
//     counts into output offsets; pass 2 (one wave per chunk, 16-byte loads) compacts the accepted candidates to their places
//     with a wave scan — neighbours write neighbours; the lane that holds the n-th value replays its chunk from the saved state
//     up to that draw and writes the generator state back to the stream, so the next fill continues exactly where a host loop
//     would.  (The first form — pass 1 only counted and pass 2 generated every draw again, each lane writing 4-byte words 960
//     bytes from its neighbour's — cost 50 instead of 31 instructions per draw and 3.4 % of the hiding bench; retired in round 5.)
// Nothing synchronises with the host.  A fill that runs out of raw draws (probability far below 2^-100 with the margin
// used) raises the error word instead of producing a short stream.
#include "bb31.hip.h"
#include "common.h"
#include "rng.h"
#include "rng_dev.hip.h"

#include <algorithm>
#include <memory>
#include <mutex>

namespace p3 {

// ---- host: SplitMix64 seeding and the GF(2) jump matrices ----
void rng_seed_from_u64(uint64_t s[4], uint64_t state) {
    for (int i = 0; i < 4; i++) {
        state += 0x9e3779b97f4a7c15ULL;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        s[i] = z ^ (z >> 31);
    }
}
namespace {
struct Bits256 { uint64_t w[4]; };
struct Mat256 { Bits256 col[256]; };  // M v = xor of col[j] over the set bits j of v
Bits256 matvec(const Mat256& m, const Bits256& v) {
    Bits256 r{{0, 0, 0, 0}};
    for (int j = 0; j < 256; j++)
        if ((v.w[j >> 6] >> (j & 63)) & 1)
            for (int k = 0; k < 4; k++) r.w[k] ^= m.col[j].w[k];
    return r;
}
void matsquare(const Mat256& m, Mat256* out) {
    for (int j = 0; j < 256; j++) out->col[j] = matvec(m, m.col[j]);
}
// J[k] = T^(RNG_CHUNK * 2^k), T = one state transition; built once per process
const std::vector<uint64_t>& jump_matrices() {
    static std::vector<uint64_t> flat;
    static std::once_flag once;
    std::call_once(once, [] {
        std::unique_ptr<Mat256> a(new Mat256()), b(new Mat256());
        for (int j = 0; j < 256; j++) {
            uint64_t s[4] = {0, 0, 0, 0};
            s[j >> 6] = 1ull << (j & 63);
            (void)xoshiro_next(s);
            for (int k = 0; k < 4; k++) a->col[j].w[k] = s[k];
        }
        // Stored ROW-major in the lane-interleaved basis the device works in: permuted index p = 64 i + l stands for
        // state bit 4 l + i (lane l of a wave owns bits 4l .. 4l+3), so that the four ballots of a wave ARE the four
        // words of the product.  Row p_out = mask over permuted input indices; rows 64 i + l, i = 0..3, belong to lane l.
        auto orig = [](int p) { return 4 * (p & 63) + (p >> 6); };
        flat.resize((size_t)(RNG_MAX_JUMP + 1) * 256 * 4);
        auto store = [&](uint32_t k) {
            for (int po = 0; po < 256; po++) {
                uint64_t row[4] = {0, 0, 0, 0};
                const int bo = orig(po);
                for (int pi = 0; pi < 256; pi++)
                    if ((a->col[orig(pi)].w[bo >> 6] >> (bo & 63)) & 1) row[pi >> 6] |= 1ull << (pi & 63);
                for (int w = 0; w < 4; w++) flat[((size_t)k * 256 + po) * 4 + w] = row[w];
            }
        };
        for (uint32_t i = 0; i < RNG_CHUNK_LOG; i++) {
            if (i == RNG_TINY_CHUNK_LOG) store(RNG_MAX_JUMP);  // slot RNG_MAX_JUMP: T^64, the short chunks of the one-launch prover's fills
            matsquare(*a, b.get());
            a.swap(b);
        }
        for (uint32_t k = 0; k < RNG_MAX_JUMP; k++) {
            store(k);
            matsquare(*a, b.get());
            a.swap(b);
        }
    });
    return flat;
}
}  // namespace

// exclusive scan of counts[0..n) in two levels: every workgroup scans its 1024 counts in place and leaves their total in
// bsum[block]; one workgroup then scans the block totals; pass 2 adds bsum[chunk / 1024] to the in-block offset.
// (One workgroup walking the whole array took 183 us for the 2^17.8 chunks of the prover's largest fill.)
constexpr uint32_t SCAN_TILE = 1024;
__global__ void __launch_bounds__(256) rng_scan_tiles_kernel(uint32_t* counts, uint32_t n, uint32_t* bsum) {
    P3_LATENCY_BOUND_KERNEL();
    __shared__ uint32_t wsum[4];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * SCAN_TILE + tid * 4, lane = tid & 63u, wv = tid >> 6;
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = base + k < n ? counts[base + k] : 0u;
    const uint32_t mine = c[0] + c[1] + c[2] + c[3];
    uint32_t inc = mine;  // inclusive scan over the wave
#pragma unroll
    for (uint32_t off = 1; off < 64; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, off, 64);
        if (lane >= off) inc += v;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t run = inc - mine;
    for (uint32_t w = 0; w < wv; w++) run += wsum[w];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (base + k < n) counts[base + k] = run;
        run += c[k];
    }
    if (tid == 255) bsum[blockIdx.x] = run;
}
__global__ void __launch_bounds__(1024) rng_scan_kernel(uint32_t* counts, uint32_t n) {  // n <= 4096 block totals
    P3_LATENCY_BOUND_KERNEL();
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += counts[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
    for (uint32_t i = lo; i < hi; i++) { uint32_t c = counts[i]; counts[i] = run; run += c; }
}
// ---- one generation per draw: pass 1 keeps the RAW 31-bit candidates, pass 2 only compacts them ----
// Pass 1: one wave per 64 x SUB consecutive chunks — the wave jumps to its first chunk with the matrices of the set bits of that chunk
// index, then walks J_0 = T^CHUNK to the next ones, lane i keeping the i-th state — and every candidate is kept: the wave's 64 lanes (= 64 chunks) produce one candidate each
// per step; 32 steps fill a 64 x 32 tile in LDS, which goes out transposed — lanes 0..31 write 128 contiguous bytes of one
// chunk, lanes 32..63 of the next — so raw[chunk][0..256) is written in whole 128-byte pieces instead of 4-byte words 1 KB apart.
// SUB consecutive chunks per lane (template): the walk that hands every lane its first state costs ~80 wave-instructions per lane
// (63 products per wave) — 20 per generated draw-step at one chunk per lane, as much as the draws themselves (25).  A lane that
// simply keeps generating into its next chunk needs no product for it: at four chunks per lane the walk (over T^(4 CHUNK)) is 5 per
// step.  More chunks per lane mean fewer waves, so the host picks the largest SUB that still leaves >= 1024 waves.
template <uint32_t SUB_LOG>
__global__ void __launch_bounds__(64) rng_gen_kernel(const DevRng* st, const uint64_t* __restrict__ jump, uint32_t n_chunks, uint32_t n_bits,
                                                     uint64_t* states, uint32_t* counts, uint32_t* raw) {
    constexpr uint32_t SUB = 1u << SUB_LOG;
    __shared__ uint32_t tile[64 * 33];
    const uint32_t lane = threadIdx.x, first = blockIdx.x * (64u * SUB);  // the wave's first chunk; lane l owns first + l SUB ..+ SUB
    uint64_t cur[4] = {st->s[0], st->s[1], st->s[2], st->s[3]};
    to_interleaved(cur);
    const LaneRows j0 = load_rows(jump + (size_t)SUB_LOG * 256 * 4);  // T^(CHUNK * SUB)
    for (uint32_t k = 6 + SUB_LOG; k < n_bits; k++)
        if ((first >> k) & 1u) wave_matvec(load_rows(jump + (size_t)k * 256 * 4), cur);  // uniform branch
    uint64_t s[4] = {cur[0], cur[1], cur[2], cur[3]};
    const uint32_t left = n_chunks - first;
    const uint32_t last = left < 64u * SUB ? (left + SUB - 1) / SUB : 64u;  // lanes of this wave that own a chunk
    for (uint32_t i = 1; i < last; i++) {
        wave_matvec(j0, cur);
        if (lane == i) { s[0] = cur[0]; s[1] = cur[1]; s[2] = cur[2]; s[3] = cur[3]; }
    }
    from_interleaved(s);
    const uint32_t half = lane >> 5, col = lane & 31u;
    for (uint32_t sc = 0; sc < SUB; sc++) {
        const uint32_t t = first + lane * SUB + sc;
        if (t < n_chunks) {
#pragma unroll
            for (int w = 0; w < 4; w++) states[(size_t)t * 4 + w] = s[w];
        }
        uint32_t cnt = 0;
        for (uint32_t round = 0; round < RNG_CHUNK / 32; round++) {
#pragma unroll 4
            for (uint32_t i = 0; i < 32; i++) {
                const uint32_t v = (uint32_t)(xoshiro_next(s) >> 32) >> 1;
                cnt += v < bb::P ? 1u : 0u;
                tile[lane * 33 + i] = v;
            }
            __syncthreads();  // one wave per workgroup: orders the tile's writes before its transposed reads
            for (uint32_t c = 0; c < 64; c += 2) {  // lane c's chunk on lanes 0..31, lane c + 1's on lanes 32..63
                const uint32_t tc = first + (c + half) * SUB + sc;
                if (tc < n_chunks) raw[((size_t)tc << RNG_CHUNK_LOG) + round * 32 + col] = tile[(c + half) * 33 + col];
            }
            __syncthreads();
        }
        if (t < n_chunks) counts[t] = cnt;
    }
}
// Pass 2: one wave per chunk.  Lane l holds candidates 4l .. 4l+3 (one 16-byte load), a wave scan of the acceptance counts gives
// every accepted value its place, neighbours write neighbours.  The lane that holds the n-th element of the fill replays its chunk
// from the saved state up to that draw and leaves the generator there: the next fill continues where a host loop would.
__global__ void __launch_bounds__(256) rng_compact_kernel(DevRng* st, const uint64_t* states, const uint32_t* offsets, const uint32_t* bsum,
                                                          uint32_t n_chunks, const uint32_t* raw, uint32_t* out, uint64_t n, uint32_t* err) {
    const uint32_t lane = threadIdx.x & 63u, c = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const uint64_t base = (uint64_t)offsets[c] + bsum[c / SCAN_TILE];
    if (base >= n) return;  // the stream was complete before this chunk
    const uint4 v4 = reinterpret_cast<const uint4*>(raw + ((size_t)c << RNG_CHUNK_LOG))[lane];
    const uint32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
    // rank of candidate 4 lane + k among the chunk's accepted ones = accepted candidates of lower lanes (the four acceptance BALLOTS,
    // counted below this lane by v_mbcnt: eight instructions, no shuffle scan) + accepted ones of this lane before k
    uint32_t below = 0, total = 0;
    bool acc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        acc[k] = v[k] < bb::P;
        const uint64_t b = __ballot(acc[k]);
        below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, below));
        total += (uint32_t)__builtin_popcountll(b);  // wave-uniform: scalar
    }
    // accepted values to their rank in the wave's 256-word LDS row, then out in rank order: lane l writes ranks l, l + 64, ... so
    // every store instruction of the wave covers 256 contiguous bytes
    __shared__ uint32_t packed[4][RNG_CHUNK];
    uint32_t* row = packed[threadIdx.x >> 6];
    uint32_t rank = below;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (acc[k]) row[rank++] = v[k];
    }
    if (n - base <= total) {  // wave-uniform, true for ONE wave of the fill: the n-th element lies in this chunk
        const uint32_t want = (uint32_t)(n - base) - 1u;  // its rank
        uint32_t r = below;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            if (acc[k]) {
                if (r == want) {  // replay this chunk up to raw draw 4 lane + k: the stream continues right after it
                    uint64_t s[4] = {states[(size_t)c * 4], states[(size_t)c * 4 + 1], states[(size_t)c * 4 + 2], states[(size_t)c * 4 + 3]};
                    for (uint32_t i = 0; i <= 4 * lane + k; i++) (void)xoshiro_next(s);
#pragma unroll
                    for (int w = 0; w < 4; w++) st->s[w] = s[w];
                }
                r++;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();  // the row belongs to this wave alone: LDS operations of one wave complete in order
#pragma unroll
    for (uint32_t i = 0; i < RNG_CHUNK; i += 64) {
        const uint32_t idx = i + lane;
        if (idx < total && base + idx < n) out[base + idx] = row[idx];
    }
    if (c + 1 == n_chunks && lane == 0 && base + total < n) atomicOr(err, 1u);  // ran out of raw draws (never, with the margin used)
}

// SMALL fills (up to 32 chunks = 8192 raw draws) in ONE launch of one wave: the reference's own instance (n = 8, fib_air.rs:56-57)
// draws 304, 1536 and ~200 elements per stream, and the four-launch path above costs such a fill ~50 us whatever its size (two waves
// walking 63 jump products each for chunks nobody needs, two scan launches over a few counters).  Lane l walks to chunk l (only the
// chunks the fill can need), generates its 256 candidates into LDS, the wave scans the 32 counts in registers and compacts chunk by
// chunk with ballot ranks; the lane that owns the chunk of the n-th element replays it and leaves the generator state.
__global__ void __launch_bounds__(64) rng_small_fill_kernel(DevRng* st, const uint64_t* __restrict__ jump, uint32_t n_chunks, uint32_t* out, uint32_t n,
                                                            uint32_t* err) {
    __shared__ uint32_t raw[RNG_SMALL_CHUNKS * RNG_SMALL_STRIDE];
    const uint32_t lane = threadIdx.x;
    const uint64_t seed[4] = {st->s[0], st->s[1], st->s[2], st->s[3]};
    uint64_t s_start[4];
    uint32_t my_base, cnt;
    const uint32_t total = rng_small_fill_wave<RNG_CHUNK_LOG>(seed, jump, n_chunks, raw, out, n, s_start, my_base, cnt);  // rng_dev.hip.h
    if (total < n) { if (lane == 0) atomicOr(err, 1u); return; }  // ran out of raw draws (never, with the margin used)
    // generator state right after the n-th accepted draw: the lane whose chunk holds it replays that chunk
    if (lane < n_chunks && my_base < n && n <= my_base + cnt) {
        uint64_t r[4] = {s_start[0], s_start[1], s_start[2], s_start[3]};
        uint32_t need = n - my_base;  // accepted draws of this chunk up to and including the n-th element
        while (need) {
            const uint32_t v = (uint32_t)(xoshiro_next(r) >> 32) >> 1;
            need -= v < bb::P ? 1u : 0u;
        }
#pragma unroll
        for (int w = 0; w < 4; w++) st->s[w] = r[w];
    }
}

__global__ void rng_set_kernel(DevRng* st, uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint32_t* clear) {
    DevRng* r = st + threadIdx.x;  // `count` consecutive streams, all seeded alike (the hiding prover's three: one launch)
    r->s[0] = a; r->s[1] = b; r->s[2] = c; r->s[3] = d;
    if (clear && threadIdx.x == 0) *clear = 0;  // the caller's shortage flag, reset with the streams
}

int rng_seed(hipStream_t stream, DevRng* st, uint64_t seed, uint32_t count, uint32_t* clear) {
    uint64_t s[4];
    rng_seed_from_u64(s, seed);
    hipLaunchKernelGGL(rng_set_kernel, dim3(1), dim3(count), 0, stream, st, s[0], s[1], s[2], s[3], clear);
    P3_HIP(hipGetLastError());
    return OK;
}

static uint64_t fill_chunks(uint64_t n) {
    const uint64_t raw = n + n / 8 + 64 * (uint64_t)RNG_CHUNK;
    return (raw + RNG_CHUNK - 1) / RNG_CHUNK;
}
bool rng_fill_supported(uint64_t n) { return (fill_chunks(n) >> RNG_MAX_JUMP) == 0; }
int rng_workspace_words(uint64_t n_max, size_t* words) {
    const uint64_t chunks = fill_chunks(n_max);
    if (chunks >> RNG_MAX_JUMP) return fail(ERR_BAD_ARG, "rng: fill too large");
    // states (4 x u64) + counts + block totals + the raw candidates of every chunk (one-generation fill)
    *words = (size_t)chunks * 8 + (size_t)chunks + (size_t)(chunks / SCAN_TILE + 1) + 16 + ((size_t)chunks << RNG_CHUNK_LOG) + 4;
    return OK;
}

// jump matrices: one copy per (thread, device) context, uploaded at first use
int rng_jump_table(Context& cx, const uint64_t** out) {
    if (!cx.rng_jump) {
        const std::vector<uint64_t>& j = jump_matrices();
        P3_HIP(hipMalloc(reinterpret_cast<void**>(&cx.rng_jump), j.size() * 8));
        P3_HIP(hipMemcpy(cx.rng_jump, j.data(), j.size() * 8, hipMemcpyHostToDevice));
    }
    if (out) *out = cx.rng_jump;
    return OK;
}

int rng_fill_field(Context& cx, hipStream_t stream, DevRng* st, uint32_t* out, uint64_t n, uint32_t* workspace, uint32_t* err) {
    if (!n) return OK;
    { int rc = rng_jump_table(cx, nullptr); if (rc) return rc; }
    // raw draws: n / (P / 2^31) = n * 1.0667 expected; 12.5 % + 64 chunks of margin is > 50 standard deviations
    {   // small fills: one launch (rng_small_fill_kernel).  Margin: n / 8 + 1024 raw draws beyond n is > 50 standard deviations of the
        // rejections for every n (expected n / 15, deviation 0.27 sqrt(n))
        const uint32_t chunks_small = rng_small_chunks(n);
        if (chunks_small) {
            hipLaunchKernelGGL(rng_small_fill_kernel, dim3(1), dim3(64), 0, stream, st, cx.rng_jump, chunks_small, out, (uint32_t)n, err);
            P3_HIP(hipGetLastError());
            return OK;
        }
    }
    const uint64_t raw = n + n / 8 + 64 * (uint64_t)RNG_CHUNK;
    const uint64_t chunks64 = (raw + RNG_CHUNK - 1) / RNG_CHUNK;
    if (chunks64 >> RNG_MAX_JUMP) return fail(ERR_BAD_ARG, "rng: fill too large");
    const uint32_t chunks = (uint32_t)chunks64;
    uint32_t n_bits = 0;
    while ((1u << n_bits) < chunks) n_bits++;
    uint64_t* states = reinterpret_cast<uint64_t*>(workspace);
    uint32_t* counts = workspace + (size_t)chunks * 8;
    const uint32_t tiles = (chunks + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t* bsum = counts + chunks;
    // raw candidates behind the block totals, 16-byte aligned
    uint32_t* rawc = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(bsum + tiles + 1) + 15u) & ~(uintptr_t)15u);
    // chunks per lane: the most that still leaves the chip >= 1024 waves
    const uint32_t sub_log = chunks >= (1024u * 64u * 4u) ? 2u : chunks >= (1024u * 64u * 2u) ? 1u : 0u;
    const dim3 grid((chunks + (64u << sub_log) - 1) / (64u << sub_log));
    if (sub_log == 2) hipLaunchKernelGGL(rng_gen_kernel<2>, grid, dim3(64), 0, stream, st, cx.rng_jump, chunks, n_bits, states, counts, rawc);
    else if (sub_log == 1) hipLaunchKernelGGL(rng_gen_kernel<1>, grid, dim3(64), 0, stream, st, cx.rng_jump, chunks, n_bits, states, counts, rawc);
    else hipLaunchKernelGGL(rng_gen_kernel<0>, grid, dim3(64), 0, stream, st, cx.rng_jump, chunks, n_bits, states, counts, rawc);
    P3_HIP(hipGetLastError());
    hipLaunchKernelGGL(rng_scan_tiles_kernel, dim3(tiles), dim3(256), 0, stream, counts, chunks, bsum);
    P3_HIP(hipGetLastError());
    hipLaunchKernelGGL(rng_scan_kernel, dim3(1), dim3(1024), 0, stream, bsum, tiles);
    P3_HIP(hipGetLastError());
    hipLaunchKernelGGL(rng_compact_kernel, dim3((chunks + 3) / 4), dim3(256), 0, stream, st, states, counts, bsum, chunks, rawc, out, n, err);
    P3_HIP(hipGetLastError());
    return OK;
}

}  // namespace p3
