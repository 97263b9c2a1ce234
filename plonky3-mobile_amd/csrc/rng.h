// Device-resident SmallRng streams (rng.hip): xoshiro256++ state in HBM, BabyBear elements drawn in stream order.
#pragma once
#include "common.h"

namespace p3 {

struct DevRng {
    uint64_t s[4];
};

void rng_seed_from_u64(uint64_t s[4], uint64_t seed);  // SmallRng::seed_from_u64 (SplitMix64 expansion), host side
// enqueue: st[0..count) = SmallRng::seed_from_u64(seed)
int rng_seed(hipStream_t stream, DevRng* st, uint64_t seed, uint32_t count = 1, uint32_t* clear = nullptr);  // count consecutive streams, the same seed; *clear = 0
// workspace (32-bit words, 8-byte aligned) a fill of up to n_max elements needs
int rng_workspace_words(uint64_t n_max, size_t* words);
// whether ONE fill may produce n elements (the jump matrices cover 2^22 chunks of 256 raw draws)
bool rng_fill_supported(uint64_t n);
// enqueue: out[0..n) = the next n field elements of the stream (Montgomery words), *st advanced exactly as a host loop
// would leave it; *err |= 1 if the fill ran out of raw draws (does not happen with the margin used)
int rng_fill_field(Context& cx, hipStream_t stream, DevRng* st, uint32_t* out, uint64_t n, uint32_t* workspace, uint32_t* err);

// the context's device copy of the generator's GF(2) jump matrices (J_k = T^(256 * 2^k), lane-interleaved basis), uploaded at first use
int rng_jump_table(Context& cx, const uint64_t** out);

}  // namespace p3
