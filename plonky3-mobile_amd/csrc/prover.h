// fib_air prover object: one instance owns its HBM arena and stream, reusable across proofs.
#pragma once
#include <chrono>
#include <string>
#include <vector>

#include "common.h"

namespace p3 {

// p3_fri::FriParameters (the mmcs field is implied: ExtensionMmcs over the Poseidon2 tree)
struct FriParams {
    uint32_t log_blowup, log_final_poly_len, num_queries, proof_of_work_bits;
};

// Largest LDE domain (log2 of its points) a prover object admits: BASELINE configs[2] (2^24 rows at blowup 4) — the largest
// size any test or bench has run; the field's two-adicity (27) would admit twice that, which nothing has ever exercised.
constexpr uint32_t MAX_LOG_DOMAIN = 26;
// the hiding prover's (log_n + 1 + log_blowup): 2^22 trace rows at blowup 2, exercised by tests/test_gpu_hiding.py
constexpr uint32_t MAX_LOG_DOMAIN_HIDING = 24;

struct StageTimes {  // host wall clock per stage, accumulated over `proofs`
    double trace_commit_ms = 0, quotient_commit_ms = 0, open_ms = 0, fri_commit_ms = 0, grind_ms = 0, query_ms = 0;
    uint64_t proofs = 0;
};

class FibProver {
  public:
    FibProver();
    ~FibProver();
    FibProver(const FibProver&) = delete;
    // hash: mmcs.h HashKind; profile: common.h Profile (a lone prover: latency; provers sharing the chip: throughput)
    int init(uint32_t log_n, const FriParams& fp, hipStream_t stream, bool own_stream, int hash = 0, int profile = PROFILE_LATENCY);
    // proves the FibonacciAir instance with first row (a, b); public values [a, b, last right value]
    int prove(uint64_t a, uint64_t b, std::vector<uint8_t>* proof);
    // the same in two halves: enqueue returns once the proof's launches are queued (at most two proofs in flight, the second
    // behind the first on the prover's stream); finish waits for the oldest one and serialises it
    int enqueue(uint64_t a, uint64_t b);
    int finish(std::vector<uint8_t>* proof);
    const StageTimes& times() const;
    void reset_times();
    // proofs so far whose first proof-of-work range held no witness; last_indices = the device's query-index buffer right
    // after the last such miss, before the search was continued (all zero by construction, tests/test_gpu_prover.py)
    uint64_t grind_misses(std::vector<uint32_t>* last_indices) const;

  private:
    struct Impl;
    Impl* im;
    int run(uint64_t a, uint64_t b, int slot, int phase, std::vector<uint8_t>* proof, void* pending_rec);
};

// The same prover for the HIDING half of the reference's configuration (native/src/fib_air.rs:40-65:
// MerkleTreeHidingMmcs + HidingFriPcs, SmallRng::seed_from_u64(seed)); wire format version 2 (prover_hiding.hip.inc).
class FibHidingProver {
  public:
    FibHidingProver();
    ~FibHidingProver();
    FibHidingProver(const FibHidingProver&) = delete;
    int init(uint32_t log_n, const FriParams& fp, hipStream_t stream, bool own_stream, int hash, uint64_t seed, int profile = PROFILE_LATENCY);
    int prove(uint64_t a, uint64_t b, std::vector<uint8_t>* proof);

  private:
    struct Impl;
    Impl* im;
};

// verifier.hip: p3_uni_stark::verify for FibonacciAir on the host (0 = accept, else the failed check's code)
int verify_fib_air(const uint8_t* proof, size_t len, uint64_t a_pub, uint64_t b_pub, uint64_t x_pub, uint32_t log_n,
                   const FriParams& fp, std::string* why, int hash = 0);
// the verifier of hiding proofs (wire format version 2)
int verify_fib_air_hiding(const uint8_t* proof, size_t len, uint64_t a_pub, uint64_t b_pub, uint64_t x_pub, uint32_t log_n,
                          const FriParams& fp, std::string* why, int hash);

}  // namespace p3
