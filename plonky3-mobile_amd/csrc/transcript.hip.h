// Device-resident Fiat-Shamir transcript of the fib_air prover: the challengers of challenger.h restated for ONE
// WAVEFRONT, so that observe-root -> sample-challenge -> next kernel needs no host round trip.
//   HASH_POSEIDON2  DuplexChallenger<BabyBear, Poseidon2-16, 16, 8>: the sponge state is spread over the 16 lanes of a
//                   DPP row (poseidon2_coop.hip.h: ~1.1k wave-instructions of latency per permutation instead of ~7k for a
//                   single lane); every row of the wave carries the same state, so every lane sees the same results.
//   HASH_KECCAK     SerializingChallenger32<BabyBear, HashChallenger<u8, Keccak256Hash, 32>> (native/src/fib_air.rs:53):
//                   kept as a STREAMING sponge — the chaining digest and the observed bytes are absorbed as they arrive,
//                   so at any time the state is "full blocks absorbed + a partial block", which is also exactly what
//                   the proof-of-work search needs.  The bookkeeping is wave-uniform, the permutation lane-cooperative
//                   (one state word per lane, shuffles for theta / pi / chi).
// Same conventions as the host classes (p3-challenger 0.4.2 is absent: [UPSTREAM-RECALL], see challenger.h); the proofs
// these produce are compared byte for byte with the host transcript of the CPU restatement in tests/.
#pragma once
#include "bb31.hip.h"
#include "keccak.hip.h"
#include "mmcs.h"
#include "poseidon2_coop.hip.h"

namespace p3 {

// 1/(z - x) through the quadratic tower needs these constants of the point z (prover.hip inv_denoms_kernel)
struct DenConsts {
    uint32_t z0, z1, z2, z3, k0, k1, c1, z2w, z3w;
};
BB_HD DenConsts den_consts(const bb::Ext& z) {
    DenConsts k{};
    k.z0 = z.c[0]; k.z1 = z.c[1]; k.z2 = z.c[2]; k.z3 = z.c[3];
    const uint32_t W = bb::W_MONTY;
    // B^2 = (z1^2 + W z3^2) + 2 z1 z3 Y;  Y B^2 = W * 2 z1 z3 + (z1^2 + W z3^2) Y = c0 + c1 Y
    uint32_t b0 = bb::add(bb::sqr(z.c[1]), bb::mul(W, bb::sqr(z.c[3]))), b1 = bb::dbl(bb::mul(z.c[1], z.c[3]));
    uint32_t c0 = bb::mul(W, b1);
    k.c1 = b0;
    k.k0 = bb::sub(bb::mul(W, bb::sqr(z.c[2])), c0);  // A^2 = (a0^2 + W z2^2) + 2 a0 z2 Y
    k.k1 = bb::dbl(z.c[2]);
    k.z2w = bb::mul(W, z.c[2]);
    k.z3w = bb::mul(W, z.c[3]);
    return k;
}

// The Keccak-256 proof-of-work search starts from the sponge state after the complete blocks plus up to two
// template blocks (pending tail bytes, zeroed witness bytes, 0x01 / 0x80 padding).
struct KeccakGrindArgs {
    uint64_t state[25];
    uint64_t block[2][17];
    uint32_t n_blocks, wpos;  // witness byte offset within the template (may straddle lanes and blocks)
    uint32_t mask, base;
};

constexpr uint32_t MAX_FRI_ROUNDS = 32;
constexpr uint32_t ST_GRIND_MISS = 1u;  // the first search range held no witness: the host continues the search

// HashChallenger<u8, Keccak256, 32> as a streaming sponge.  Lives in HBM between kernels (DevState::kc) and in LDS
// while a transcript kernel runs (byte-wise updates of HBM would cost a memory round trip each).
struct KState {
    uint64_t st[25];
    uint32_t blen, n_obuf;
    alignas(8) uint8_t blk[136];  // word-aligned: k_observe_word stores 32 bits at a time
    alignas(8) uint8_t obuf[32];  // word-aligned: k_sample pops 32 bits at a time
};
static_assert(sizeof(KState) % 8 == 0, "KState is copied as 64-bit words");

// Everything the transcript produces and the later kernels consume; one per prover, in HBM.
struct DevState {
    // DuplexChallenger
    uint32_t st[16], inb[8], outb[8], n_in, n_out;
    KState kc;
    // challenges and what is derived from them
    uint32_t pis[4];
    bb::Ext apow[5];                  // alpha^0..4
    bb::Ext zeta, zeta_next;
    DenConsts k0, k1;
    bb::Ext alp[8], y02, y1;          // batching challenge powers, ry0 + alpha^4 ry2, alpha^2 ry1
    bb::Ext half_beta[MAX_FRI_ROUNDS];
    uint32_t grind_result;            // atomicMin target: smallest canonical witness found so far
    uint32_t status;
    KeccakGrindArgs kgrind;
};

// ---- Keccak streaming sponge on the LDS copy.  EVERY lane of the wave runs these functions with the same (uniform)
// control flow: the scalar bookkeeping is read by all lanes and written by lane 0, and the permutation is
// lane-cooperative — state lane i = x + 5y lives in wave lane i, theta / pi / chi move data with wave shuffles: about
// 50 wave-instructions per round instead of ~260 for one lane holding all 25 words, i.e. a flush costs ~4 us of
// latency instead of ~27.  Not inlined: one copy of the permutation per kernel.
__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// the lane-cooperative permutation itself lives in keccak.hip.h (kk::f_coop: one state per wave, word x + 5y in lane x + 8y)
__device__ __noinline__ void k_absorb_block(KState* k) {
    const uint32_t lane = threadIdx.x & 63u, idx = kk::coop_index();
    lds_wave_sync();
    uint64_t a = idx < 25u ? k->st[idx] : 0ull;
    if (idx < 17u) {
        uint64_t w = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) w |= (uint64_t)k->blk[8 * idx + b] << (8 * b);
        a ^= w;
    }
    a = kk::f_coop(a);
    if (idx < 25u) k->st[idx] = a;
    if (lane == 0) k->blen = 0;
    lds_wave_sync();
}
__device__ __forceinline__ void k_observe_byte(KState* k, uint8_t b) {
    const uint32_t bl = k->blen;  // the same word for every lane
    if ((threadIdx.x & 63u) == 0) { k->n_obuf = 0; k->blk[bl] = b; k->blen = bl + 1; }
    lds_wave_sync();
    if (bl + 1 == 136) k_absorb_block(k);
}
// A field element's four bytes (little endian) in ONE step: every observation of these transcripts is a 32-bit word and the
// sponge's fill level starts at 0 or 32 (the chained digest), so the level is always a multiple of four and a word never straddles
// the 136-byte block.  (The byte-wise form costs an LDS round trip and a wave fence per BYTE: the 576 bytes of the hiding prover's 36
// opened values were 60 of ts_open_h_kernel's 88 us.)
__device__ __forceinline__ void k_observe_word(KState* k, uint32_t v) {
    const uint32_t bl = k->blen;  // the same word for every lane
    if (bl & 3u) {  // never with these transcripts; kept exact for any other caller
        for (int i = 0; i < 4; i++) k_observe_byte(k, (uint8_t)(v >> (8 * i)));
        return;
    }
    if ((threadIdx.x & 63u) == 0) { k->n_obuf = 0; *reinterpret_cast<uint32_t*>(k->blk + bl) = v; k->blen = bl + 4; }
    lds_wave_sync();
    if (bl + 4 == 136) k_absorb_block(k);
}
__device__ __noinline__ void k_flush(KState* k) {  // output = Keccak256(input); the digest also starts the next input
    const uint32_t lane = threadIdx.x & 63u, n = k->blen;
    for (uint32_t i = n + lane; i < 136; i += 64) k->blk[i] = 0;
    lds_wave_sync();
    if (lane == 0) { k->blk[n] ^= 0x01; k->blk[135] ^= 0x80; }
    k_absorb_block(k);
    uint8_t byte = 0;
    if (lane < 32u) byte = (uint8_t)(k->st[lane >> 3] >> (8 * (lane & 7u)));
    lds_wave_sync();
    if (lane < 25u) k->st[lane] = 0;
    if (lane < 32u) { k->obuf[lane] = byte; k->blk[lane] = byte; }
    if (lane == 0) { k->blen = 32; k->n_obuf = 32; }
    lds_wave_sync();
}
__device__ __noinline__ uint32_t k_sample(KState* k) {
    for (;;) {  // rejection sampling of a 31-bit value below P; every lane follows the same path
        uint32_t v = 0;
        // the output buffer is popped from its END, least significant byte first: four pops are the byte-reversed word at n - 4.
        // It holds 32 bytes after a flush and loses four per sample, so it is always a whole number of words here: ONE step per
        // sample instead of four byte pops with a wave fence each (the 100 query indices of a proof were ~60 us of byte pops)
        if (k->n_obuf == 0) k_flush(k);
        if ((k->n_obuf & 3u) == 0) {
            const uint32_t nb = k->n_obuf - 4;
            const uint32_t w = *reinterpret_cast<const uint32_t*>(k->obuf + nb);
            v = __builtin_bswap32(w);
            if ((threadIdx.x & 63u) == 0) k->n_obuf = nb;
            lds_wave_sync();
            v &= 0x7fffffffu;
            if (v < bb::P) return bb::to_monty(v);
            continue;
        }
        for (int i = 0; i < 4; i++) {
            if (k->n_obuf == 0) k_flush(k);
            const uint32_t nb = k->n_obuf - 1;
            v |= (uint32_t)k->obuf[nb] << (8 * i);
            if ((threadIdx.x & 63u) == 0) k->n_obuf = nb;
            lds_wave_sync();
        }
        v &= 0x7fffffffu;
        if (v < bb::P) return bb::to_monty(v);
    }
}

// One copy of the lane-cooperative permutation per kernel: every observe / sample site would otherwise inline it.
__device__ __noinline__ uint32_t coop_permute_call(uint32_t v) {
    const p2c::LaneConst lc = p2c::lane_constants(threadIdx.x & 15u);
    return p2c::permute(v, lc);
}

// ---- the wave-wide challenger ---------------------------------------------------------------------------------
struct DevChal {
    int kind;
    DevState* ds;
    KState* k;  // Keccak configuration: the sponge, in LDS (one KState of __shared__ memory per transcript kernel)
    uint32_t lane16;
    uint32_t st, inb, outb, n_in, n_out;  // lane i of each row holds state[i], in[i], out[i]

    // called by the first wave of the workgroup (threadIdx.x < 64); lds: one KState of shared memory
    __device__ __forceinline__ void begin(int kind_, DevState* ds_, KState* lds, bool fresh) {
        kind = kind_; ds = ds_; k = lds;
        lane16 = threadIdx.x & 15u;
        if (kind == HASH_KECCAK) {
            uint64_t* d = reinterpret_cast<uint64_t*>(lds);
            const uint64_t* s = reinterpret_cast<const uint64_t*>(&ds->kc);
            for (uint32_t i = threadIdx.x; i < sizeof(KState) / 8; i += 64) d[i] = fresh ? 0ull : s[i];
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            st = inb = outb = n_in = n_out = 0;
            return;
        }
        if (fresh) { st = 0; inb = 0; outb = 0; n_in = 0; n_out = 0; return; }
        st = ds->st[lane16];
        inb = ds->inb[lane16 & 7u];
        outb = ds->outb[lane16 & 7u];
        n_in = ds->n_in; n_out = ds->n_out;
    }
    __device__ __forceinline__ void end() {
        if (kind == HASH_KECCAK) {
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            const uint64_t* s = reinterpret_cast<const uint64_t*>(k);
            uint64_t* d = reinterpret_cast<uint64_t*>(&ds->kc);
            for (uint32_t i = threadIdx.x; i < sizeof(KState) / 8; i += 64) d[i] = s[i];
            return;
        }
        if (threadIdx.x < 16) ds->st[lane16] = st;
        if (threadIdx.x < 8) { ds->inb[lane16] = inb; ds->outb[lane16] = outb; }
        if (threadIdx.x == 0) { ds->n_in = n_in; ds->n_out = n_out; }
    }
    __device__ __forceinline__ void duplex() {
        const uint32_t v = lane16 < n_in ? inb : st;
        st = coop_permute_call(v);
        outb = st;
        n_in = 0; n_out = 8;
    }
    // ---- Keccak streaming sponge, lane 0 only (free functions on the LDS copy: see below the struct) ----
    // ---- the challenger interface: every lane of the wave calls these with the same arguments ----
    __device__ __forceinline__ void observe(uint32_t v) {  // a field element (Montgomery word)
        if (kind == HASH_KECCAK) {
            k_observe_word(k, v);
            return;
        }
        n_out = 0;
        if (lane16 == n_in) inb = v;
        n_in++;
        if (n_in == 8) duplex();
    }
    __device__ __forceinline__ void observe_n(const uint32_t* v, uint32_t n) { for (uint32_t i = 0; i < n; i++) observe(v[i]); }
    __device__ __forceinline__ void observe_ext(const bb::Ext& e) { observe_n(e.c, 4); }
    __device__ __forceinline__ uint32_t sample() {
        if (kind == HASH_KECCAK) return k_sample(k);
        if (n_in || !n_out) duplex();
        n_out--;
        return (uint32_t)__shfl((int)outb, (int)n_out, 16);
    }
    __device__ __forceinline__ bb::Ext sample_ext() { bb::Ext r; for (int i = 0; i < 4; i++) r.c[i] = sample(); return r; }
    __device__ __forceinline__ uint32_t sample_bits(uint32_t bits) { return bb::from_monty(sample()) & ((1u << bits) - 1u); }
};

}  // namespace p3
