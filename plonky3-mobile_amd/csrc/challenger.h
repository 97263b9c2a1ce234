// Host-side Fiat-Shamir transcripts of the fib_air prover / verifier (a few dozen hash calls per proof: host work,
// exactly as in the reference, native/src/fib_air.rs:53,66):
//   HASH_POSEIDON2  DuplexChallenger<BabyBear, Poseidon2-16, WIDTH 16, RATE 8>           (north_star's configuration)
//   HASH_KECCAK     SerializingChallenger32<BabyBear, HashChallenger<u8, Keccak256Hash, 32>>  (fib_air.rs:53 itself)
// plus the host forms of the Keccak MMCS hashes the verifier needs.  p3-challenger / p3-keccak 0.4.2 are absent:
// conventions [UPSTREAM-RECALL]; Keccak-f / Keccak-256 themselves are pinned against hashlib in tests/.
#pragma once
#include <cstring>
#include <vector>

#include "bb31.hip.h"
#include "mmcs.h"
#include "poseidon2.hip.h"

namespace p3 {

// ---- Keccak on the host --------------------------------------------------------------------------------------
inline void keccak_f_host(uint64_t a[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const unsigned RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    auto rotl = [](uint64_t v, unsigned n) { return n ? (v << n) | (v >> (64 - n)) : v; };
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) {
            uint64_t d = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], RHO[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
}
// absorbs the complete 136-byte blocks of `in` into st; returns the number of bytes consumed
inline size_t keccak256_absorb_full(uint64_t st[25], const uint8_t* in, size_t n) {
    size_t off = 0;
    while (n - off >= 136) {
        for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, in + off + 8 * i, 8); st[i] ^= w; }
        keccak_f_host(st);
        off += 136;
    }
    return off;
}
// Keccak256Hash (tiny-keccak Keccak::v256: original 0x01 padding, rate 136)
inline void keccak256_host(const uint8_t* in, size_t n, uint8_t out[32]) {
    uint64_t st[25] = {0};
    size_t off = keccak256_absorb_full(st, in, n);
    uint8_t blk[136] = {0};
    memcpy(blk, in + off, n - off);
    blk[n - off] ^= 0x01;
    blk[135] ^= 0x80;
    for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, blk + 8 * i, 8); st[i] ^= w; }
    keccak_f_host(st);
    memcpy(out, st, 32);
}
// SerializingHasher<PaddingFreeSponge<KeccakF, 25, 17, 4>> over a row of Montgomery words
inline void keccak_hash_row_host(const uint32_t* items, size_t n, uint32_t out[8]) {
    uint64_t st[25] = {0};
    const size_t n64 = (n + 1) / 2;
    for (size_t i = 0; i < n64; i += 17) {
        size_t take = n64 - i < 17 ? n64 - i : 17;
        for (size_t k = 0; k < take; k++) {
            size_t e = 2 * (i + k);
            st[k] = (uint64_t)items[e] | (e + 1 < n ? (uint64_t)items[e + 1] << 32 : 0);
        }
        keccak_f_host(st);
    }
    memcpy(out, st, 32);
}
inline void keccak_compress_host(const uint32_t* l, const uint32_t* r, uint32_t out[8]) {
    uint64_t st[25] = {0};
    memcpy(st, l, 32);
    memcpy(st + 4, r, 32);
    keccak_f_host(st);
    memcpy(out, st, 32);
}

// ---- challengers ---------------------------------------------------------------------------------------------
struct Challenger {
    int kind = HASH_POSEIDON2;
    // duplex (Poseidon2)
    uint32_t state[16] = {0}, in[8] = {0}, out[8] = {0};
    int n_in = 0, n_out = 0;
    // hash challenger (Keccak-256): input bytes, output bytes popped from the back
    std::vector<uint8_t> ibuf;
    uint8_t obuf[32] = {0};
    int n_obuf = 0;

    explicit Challenger(int k = HASH_POSEIDON2) : kind(k) {}
    void duplex() {
        for (int i = 0; i < n_in; i++) state[i] = in[i];
        n_in = 0;
        p2::permute(state);
        memcpy(out, state, 32);
        n_out = 8;
    }
    void flush() {  // output = H(input); the digest also becomes the next input's prefix (chaining value)
        keccak256_host(ibuf.data(), ibuf.size(), obuf);
        n_obuf = 32;
        ibuf.assign(obuf, obuf + 32);
    }
    void observe(uint32_t v) {  // a field element: its Montgomery word (to_unique_u32), little endian
        if (kind == HASH_KECCAK) {
            n_obuf = 0;
            for (int i = 0; i < 4; i++) ibuf.push_back((uint8_t)(v >> (8 * i)));
            return;
        }
        n_out = 0;
        in[n_in++] = v;
        if (n_in == 8) duplex();
    }
    void observe_n(const uint32_t* v, size_t n) { for (size_t i = 0; i < n; i++) observe(v[i]); }
    // a commitment: 8 field elements, or [u64; 4] = the same 32 little-endian bytes
    void observe_digest(const uint32_t* d) { observe_n(d, 8); }
    void observe_ext(const bb::Ext& e) { observe_n(e.c, 4); }
    uint32_t sample() {
        if (kind == HASH_KECCAK) {
            for (;;) {  // rejection sampling of a 31-bit value below P
                uint32_t v = 0;
                for (int i = 0; i < 4; i++) {
                    if (!n_obuf) flush();
                    v |= (uint32_t)obuf[--n_obuf] << (8 * i);
                }
                v &= 0x7fffffffu;
                if (v < bb::P) return bb::to_monty(v);
            }
        }
        if (n_in || !n_out) duplex();
        return out[--n_out];
    }
    bb::Ext sample_ext() { bb::Ext r; for (int i = 0; i < 4; i++) r.c[i] = sample(); return r; }
    size_t sample_bits(unsigned bits) { return (size_t)bb::from_monty(sample()) & (((size_t)1 << bits) - 1); }
    bool check_witness(unsigned bits, uint32_t w) { observe(w); return sample_bits(bits) == 0; }
};

}  // namespace p3
