/* ORACLE — TEST INFRASTRUCTURE ONLY (see bb31.h).
 * The hash configuration the reference ACTUALLY wires into its MMCS (native/src/fib_air.rs:28-38):
 *   U64Hash    = PaddingFreeSponge<KeccakF, WIDTH 25, RATE 17, OUT 4>         (:31-32)
 *   FieldHash  = SerializingHasher<U64Hash>                                    (:34-35)
 *   MyCompress = CompressionFunctionFromHasher<U64Hash, 2, 4>                  (:37-38)
 * KeccakF is Keccak-f[1600] (FIPS 202 permutation, crate p3-keccak 0.4.2 -> tiny-keccak 2.0.2, both absent:
 * the permutation is restated from the public specification and PINNED by SHA3-256 / Keccak-256 digests of
 * arbitrary messages against python's hashlib in tests/test_oracle_keccak.py).
 * [UPSTREAM-RECALL, parity unpinned] SerializingHasher feeds the inner u64 hasher with
 * BabyBear::into_u64_stream: two field elements per u64, first in the low half, each as its unique u32
 * (the Montgomery word, the same words the reference uploads, backend_vulkan.rs:2002-2005); a trailing odd
 * element fills the low half of a last u64.  PaddingFreeSponge overwrites RATE lanes per block and permutes
 * after every full block and after a non-empty partial one; CompressionFunctionFromHasher hashes the 8 lanes
 * of the two digests (one block). */
#include "p3_oracle.h"
#include <string.h>

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
/* rotation offsets r[x][y] of the rho step, lane index x + 5y */
static const unsigned RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39,
                                 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static uint64_t rotl(uint64_t v, unsigned n) { return n ? (v << n) | (v >> (64 - n)) : v; }

void p3o_keccak_f(uint64_t a[25]) {
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) {
            uint64_t d = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
        /* rho + pi: B[y, 2x + 3y] = rot(A[x, y], r[x, y]) */
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], RHO[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
}

/* PaddingFreeSponge<KeccakF, 25, 17, 4>::hash_iter over a u64 stream */
void p3o_keccak_sponge_u64(const uint64_t *items, size_t n, uint64_t out[4]) {
    uint64_t st[25] = {0};
    size_t i = 0;
    while (i < n) {
        size_t take = n - i < 17 ? n - i : 17;
        memcpy(st, items + i, take * 8);
        p3o_keccak_f(st);
        i += take;
    }
    memcpy(out, st, 32);
}
/* SerializingHasher<U64Hash>::hash_iter over a row of field elements (Montgomery words) */
void p3o_keccak_hash_row(const uint32_t *items, size_t n, uint32_t out[8]) {
    uint64_t st[25] = {0}, dig[4];
    size_t n64 = (n + 1) / 2, i = 0;
    while (i < n64) {
        size_t take = n64 - i < 17 ? n64 - i : 17;
        for (size_t k = 0; k < take; k++) {
            size_t e = 2 * (i + k);
            st[k] = (uint64_t)items[e] | (e + 1 < n ? (uint64_t)items[e + 1] << 32 : 0);
        }
        p3o_keccak_f(st);
        i += take;
    }
    memcpy(dig, st, 32);
    for (int k = 0; k < 4; k++) { out[2 * k] = (uint32_t)dig[k]; out[2 * k + 1] = (uint32_t)(dig[k] >> 32); }
}
/* CompressionFunctionFromHasher<U64Hash, 2, 4>: digests as 8 u32 words = 4 little-endian u64 lanes */
void p3o_keccak_compress(const uint32_t l[8], const uint32_t r[8], uint32_t out[8]) {
    uint64_t st[25] = {0};
    for (int k = 0; k < 4; k++) {
        st[k] = (uint64_t)l[2 * k] | (uint64_t)l[2 * k + 1] << 32;
        st[4 + k] = (uint64_t)r[2 * k] | (uint64_t)r[2 * k + 1] << 32;
    }
    p3o_keccak_f(st);
    for (int k = 0; k < 4; k++) { out[2 * k] = (uint32_t)st[k]; out[2 * k + 1] = (uint32_t)(st[k] >> 32); }
}

/* Keccak256Hash::hash_iter over bytes (p3-keccak: tiny-keccak's Keccak::v256 = ORIGINAL Keccak padding 0x01,
 * rate 136, 32-byte digest) — the byte hash of the reference's challenger (fib_air.rs:29-30,53). */
void p3o_keccak256(const uint8_t *in, size_t n, uint8_t out[32]) {
    uint64_t st[25] = {0};
    uint8_t blk[136];
    size_t off = 0;
    for (;;) {
        size_t take = n - off < 136 ? n - off : 136;
        memset(blk, 0, 136);
        memcpy(blk, in + off, take);
        int last = take < 136;
        if (last) { blk[take] ^= 0x01; blk[135] ^= 0x80; }
        for (int i = 0; i < 17; i++) { uint64_t w; memcpy(&w, blk + 8 * i, 8); st[i] ^= w; }
        p3o_keccak_f(st);
        off += take;
        if (last) break;
    }
    memcpy(out, st, 32);
}
