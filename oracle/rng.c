/* ORACLE — TEST INFRASTRUCTURE ONLY (see bb31.h).
 * The random stream of the reference's hiding configuration: `SmallRng::seed_from_u64(1)` (native/src/fib_air.rs:50,65)
 * drawing BabyBear elements.  rand 0.9.2 (pinned in native/Cargo.lock) is absent; restated from its published
 * algorithms [UPSTREAM-RECALL]:
 *   SmallRng on 64-bit targets = Xoshiro256PlusPlus; seed_from_u64 expands the seed with SplitMix64 (4 outputs);
 *   next_u32 = upper half of next_u64;
 *   StandardUniform for MontyField31 (p3-monty-31): loop { v = next_u32() >> 1; if v < P { return new_monty(v) } } —
 *   the accepted 31-bit value IS the Montgomery word.
 * Pinned: xoshiro256++ against the generator's published reference vector (state 1,2,3,4), SplitMix64(1) against its
 * published first output (tests/test_oracle_rng.py).  The field-sampling convention is recalled, unpinned. */
#include "p3_oracle.h"
#include "bb31.h"

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void p3o_rng_seed_from_u64(uint64_t s[4], uint64_t state) {
    for (int i = 0; i < 4; i++) {
        state += 0x9e3779b97f4a7c15ULL;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        s[i] = z ^ (z >> 31);
    }
}
uint64_t p3o_rng_next_u64(uint64_t s[4]) {
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
/* n BabyBear elements (Montgomery words) in stream order */
void p3o_rng_fill_field(uint64_t s[4], uint32_t *out, size_t n) {
    for (size_t i = 0; i < n;) {
        uint32_t v = (uint32_t)(p3o_rng_next_u64(s) >> 32) >> 1;
        if (v < BB_P) out[i++] = v;
    }
}
