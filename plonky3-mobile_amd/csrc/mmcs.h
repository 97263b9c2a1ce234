// Merkle-tree prover data kept in HBM (Plonky3 MerkleTree<..>: leaves + digest_layers).
#pragma once
#include <memory>
#include <cstring>
#include "common.h"

namespace p3 {

enum HashKind : int { HASH_POSEIDON2 = 0, HASH_KECCAK = 1 };

struct Tree {
    int kind = HASH_POSEIDON2;
    std::vector<const uint32_t*> mats;  // borrowed device pointers (or entries of `owned`)
    std::vector<size_t> heights, widths, strides;  // strides: words between rows (= widths for dense matrices)
    std::vector<void*> owned;           // device copies made by the host-pointer commit
    uint32_t* layers = nullptr;         // all digest layers, leaf layer first, 8 words per digest
    bool root_copied = false;           // the top kernel also wrote the root to the caller's host-mapped buffer
    bool owns_layers = true;            // false: caller-provided arena (prover), nothing to free
    std::vector<size_t> layer_off, layer_len;  // offsets in words / lengths in digests
    uint32_t log_max_height = 0;
    uint32_t* staging = nullptr;        // open_batch gather buffer
    size_t staging_words = 0;
    ~Tree();
};

// ext_layers: optional caller-owned storage of (2*max_height - 1) * 8 words for the digest layers.
// strides: optional words between two rows per matrix (a matrix may be a column group of a wider one); default = widths.
// profile (common.h): how layers of 2^10 .. 2^15 digests are hashed — the latency forms (a lone tree / proof) or the per-lane
// forms (several provers sharing the chip).  Digests are the same either way.
int mmcs_commit(hipStream_t stream, const uint32_t* const* d_mats, const size_t* heights, const size_t* widths,
                size_t n_mats, Tree** out, uint32_t* ext_layers = nullptr, uint32_t* root_copy = nullptr,
                int kind = HASH_POSEIDON2, const size_t* strides = nullptr, int profile = PROFILE_LATENCY);
inline size_t mmcs_layer_words(uint64_t max_height) { return (size_t)(2 * max_height - 1) * 8; }
int mmcs_root(hipStream_t stream, const Tree& t, uint32_t root_out[8]);
int mmcs_open(hipStream_t stream, const Tree& t, uint64_t index, uint32_t* rows_out, uint32_t* path_out);
int poseidon2_permute_states(hipStream_t stream, uint32_t* d_states, uint64_t n);
int poseidon2_f64_probe(hipStream_t stream, const double* d_in, uint32_t* d_out, uint64_t n, int mode);
int poseidon2_permute_states_variant(hipStream_t stream, uint32_t* d_states, uint64_t n, int variant);
int keccak_f_states(hipStream_t stream, uint64_t* d_states, uint64_t n);

}  // namespace p3
