// C++ host-side mirror of the reference's Rust interfaces for the hip backend, header-only over the C ABI
// (include/p3hip.h).  Names, argument meaning and error behaviour follow the reference:
//   BackendKind / set_backend_kind[_from_str] / get_backend_kind / take_last_error   native/src/gpu_dft.rs:14-68
//   GpuDft::{default, with_backend, dft_batch} + Plonky3's provided idft/coset methods     native/src/gpu_dft.rs:70-115
//   RowMajorMatrix (p3_matrix::dense): row-major values + width
//   benchmark_input / percentile_ms / generate_trace_rows                                   native/src/fib_air.rs:77-96,266-284
//   MerkleTreeMmcs (Mmcs::commit / open_batch) and FibAirProver (prove)                      native/src/fib_air.rs:40-70
// Rust's `Result<_, String>` becomes p3hip::Error (thrown); there is NO CPU fallback here — the reference's
// GpuDft falls back to Radix2DitParallel on Err (gpu_dft.rs:100-112); a C++ caller catches and decides.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "p3hip.h"

namespace p3hip {

constexpr uint32_t P = 0x78000001u;
constexpr uint32_t MONTY_ONE = 0x0ffffffeu;
constexpr uint32_t GENERATOR_MONTY = (uint32_t)((31ull << 32) % P);  // Val::GENERATOR

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline std::string take_last_error() {  // gpu_dft.rs:65-68
    const char* m = p3hip_take_last_error();
    return m ? std::string(m) : std::string();
}
inline void check(int rc) {
    if (rc != 0) throw Error(rc, take_last_error());
}

enum class BackendKind : int { Cpu = 0, Vulkan = 1, Metal = 2, WebGpu = 3, Hip = 4 };  // gpu_dft.rs:14-40
inline void set_backend_kind_from_str(const std::string& v) {                             // gpu_dft.rs:53-63
    if (p3hip_set_backend(v.c_str()) != 0) throw Error(P3HIP_ERR_BACKEND, take_last_error());
}
inline BackendKind get_backend_kind() { return (BackendKind)p3hip_get_backend(); }       // gpu_dft.rs:49-51
inline std::pair<bool, std::string> is_available() {                                      // lib.rs:167-179
    char buf[256];
    int rc = p3hip_is_available(buf, sizeof buf);
    if (rc != 0) (void)p3hip_take_last_error();
    return {rc == 0, std::string(buf)};
}
inline uint32_t to_monty(uint64_t canon) { return (uint32_t)(((canon % P) << 32) % P); }

struct RowMajorMatrix {  // p3_matrix::dense::RowMajorMatrix<BabyBear>, values are Montgomery words
    std::vector<uint32_t> values;
    size_t width = 0;
    RowMajorMatrix() = default;
    RowMajorMatrix(std::vector<uint32_t> v, size_t w) : values(std::move(v)), width(w) {}
    size_t height() const { return width ? values.size() / width : 0; }
};

class GpuDft {  // gpu_dft.rs:70-115
  public:
    GpuDft() : backend_(get_backend_kind()) {}  // Default (gpu_dft.rs:76-83)
    static GpuDft with_backend(BackendKind b) { GpuDft d; d.backend_ = b; return d; }  // gpu_dft.rs:86-92
    BackendKind backend() const { return backend_; }
    RowMajorMatrix dft_batch(const RowMajorMatrix& m) const {
        require_hip();
        RowMajorMatrix out(std::vector<uint32_t>(m.values.size()), m.width);
        check(p3hip_dft_batch_bb31(m.values.data(), out.values.data(), m.height(), m.width));
        return out;
    }
    RowMajorMatrix idft_batch(const RowMajorMatrix& m) const {
        require_hip();
        RowMajorMatrix out(std::vector<uint32_t>(m.values.size()), m.width);
        check(p3hip_idft_batch_bb31(m.values.data(), out.values.data(), m.height(), m.width));
        return out;
    }
    RowMajorMatrix coset_dft_batch(const RowMajorMatrix& m, uint32_t shift_monty) const {
        require_hip();
        RowMajorMatrix out(std::vector<uint32_t>(m.values.size()), m.width);
        check(p3hip_coset_dft_batch_bb31(m.values.data(), out.values.data(), m.height(), m.width, shift_monty));
        return out;
    }
    RowMajorMatrix coset_lde_batch(const RowMajorMatrix& m, unsigned added_bits, uint32_t shift_monty,
                                   bool bit_reversed_out = false) const {
        require_hip();
        RowMajorMatrix out(std::vector<uint32_t>(m.values.size() << added_bits), m.width);
        check(p3hip_coset_lde_batch_bb31(m.values.data(), out.values.data(), m.height(), m.width, added_bits,
                                         shift_monty, bit_reversed_out ? 1 : 0));
        return out;
    }

  private:
    void require_hip() const {
        if (backend_ != BackendKind::Hip)
            throw Error(P3HIP_ERR_BACKEND, "only the hip backend runs here (the CPU path is the caller's Radix2DitParallel)");
    }
    BackendKind backend_;
};

// fib_air.rs:77-86
inline RowMajorMatrix benchmark_input(size_t height, size_t width) {
    std::vector<uint32_t> v(height * width);
    for (size_t i = 0; i < v.size(); i++) v[i] = to_monty(((uint64_t)i * 17 + 3) % P);
    return RowMajorMatrix(std::move(v), width);
}
// fib_air.rs:88-96 (nearest rank)
inline double percentile_ms(std::vector<double> s, double q) {
    if (s.empty()) return 0.0;
    std::sort(s.begin(), s.end());
    size_t idx = (size_t)std::ceil(q * (double)s.size());
    idx = idx ? idx - 1 : 0;
    return s[std::min(idx, s.size() - 1)];
}

class MerkleTree {  // prover data: device matrices + digest layers owned by the library
  public:
    MerkleTree() = default;
    MerkleTree(const MerkleTree&) = delete;
    MerkleTree(MerkleTree&& o) noexcept : h_(o.h_), widths_(std::move(o.widths_)) { o.h_ = nullptr; }
    ~MerkleTree() { if (h_) p3hip_mmcs_free(h_); }
    size_t log_max_height() const { return p3hip_mmcs_log_max_height(h_); }
    p3hip_tree_t* h_ = nullptr;
    std::vector<size_t> widths_;
};
class MerkleTreeMmcs {  // Mmcs<BabyBear>: Poseidon2 hashes (north_star) or the Keccak ones fib_air.rs:28-51 wires
  public:
    explicit MerkleTreeMmcs(int hash = P3HIP_HASH_POSEIDON2) : hash_(hash) {}
    std::pair<std::vector<uint32_t>, MerkleTree> commit(const std::vector<RowMajorMatrix>& mats) const {
        std::vector<const uint32_t*> ptrs;
        std::vector<size_t> hs, ws;
        for (auto& m : mats) { ptrs.push_back(m.values.data()); hs.push_back(m.height()); ws.push_back(m.width); }
        std::vector<uint32_t> root(8);
        MerkleTree t;
        check(p3hip_mmcs_commit_hash(hash_, ptrs.data(), hs.data(), ws.data(), mats.size(), root.data(), &t.h_));
        t.widths_ = ws;
        return {std::move(root), std::move(t)};
    }
    // -> (opened rows per matrix, sibling digests)
    std::pair<std::vector<std::vector<uint32_t>>, std::vector<uint32_t>> open_batch(size_t index, const MerkleTree& t) const {
        size_t tot = 0;
        for (size_t w : t.widths_) tot += w;
        std::vector<uint32_t> rows(tot ? tot : 1), path(t.log_max_height() * 8 + 8);
        check(p3hip_mmcs_open_batch(t.h_, index, rows.data(), path.data(), nullptr));
        std::vector<std::vector<uint32_t>> out;
        size_t off = 0;
        for (size_t w : t.widths_) { out.emplace_back(rows.begin() + off, rows.begin() + off + w); off += w; }
        path.resize(t.log_max_height() * 8);
        return {std::move(out), std::move(path)};
    }

  private:
    int hash_;
};

struct FriParameters {  // p3_fri::FriParameters; defaults = create_benchmark_fri_params
    uint32_t log_blowup = 1, log_final_poly_len = 0, num_queries = 100, proof_of_work_bits = 16;
};
class FibAirProver {  // prove(&config, &FibonacciAir{}, generate_trace_rows(a, b, n), &pis), fib_air.rs:61-70
  public:
    // hash: P3HIP_HASH_POSEIDON2 (north_star) or P3HIP_HASH_KECCAK (the reference's own hashes, fib_air.rs:28-53)
    // profile (include/p3hip.h PROFILES): fixed at creation, like the reference's backend (native/src/gpu_dft.rs:85-92).  A prover made on its
    // own proves one proof at a time: the latency profile; provers that share the chip take P3HIP_PROFILE_THROUGHPUT.
    FibAirProver(unsigned log_n, FriParameters fp = FriParameters(), int hash = P3HIP_HASH_POSEIDON2, int profile = P3HIP_PROFILE_LATENCY,
                 bool hiding = false, uint64_t seed = 1) {
        p3hip_fri_params_t c{fp.log_blowup, fp.log_final_poly_len, fp.num_queries, fp.proof_of_work_bits};
        check(p3hip_fib_prover_create_profile(profile, hash, hiding ? 1 : 0, hiding ? seed : 0, log_n, &c, nullptr, 1, &h_));
    }
    FibAirProver(const FibAirProver&) = delete;
    ~FibAirProver() { if (h_) p3hip_fib_prover_destroy(h_); }
    std::vector<uint8_t> prove(uint64_t a, uint64_t b) {
        const uint8_t* p = nullptr;
        size_t n = 0;
        check(p3hip_fib_prover_prove(h_, a, b, &p, &n));
        return std::vector<uint8_t>(p, p + n);
    }

  private:
    p3hip_fib_prover_t* h_ = nullptr;
};

// A pool of provers (one host thread + stream each) for batches of independent instances (BASELINE configs[3]).
class FibAirBatchProver {
  public:
    FibAirBatchProver(unsigned log_n, unsigned n_provers = 8, FriParameters fp = FriParameters(), int hash = P3HIP_HASH_POSEIDON2) {
        p3hip_fri_params_t c{fp.log_blowup, fp.log_final_poly_len, fp.num_queries, fp.proof_of_work_bits};
        check(p3hip_fib_batch_create_hash(hash, log_n, &c, n_provers, &h_));
    }
    FibAirBatchProver(const FibAirBatchProver&) = delete;
    ~FibAirBatchProver() { if (h_) p3hip_fib_batch_destroy(h_); }
    std::vector<std::vector<uint8_t>> prove(const std::vector<std::pair<uint64_t, uint64_t>>& instances) {
        size_t n = instances.size();
        std::vector<uint64_t> a(n), b(n);
        for (size_t i = 0; i < n; i++) { a[i] = instances[i].first; b[i] = instances[i].second; }
        std::vector<const uint8_t*> ptrs(n);
        std::vector<size_t> lens(n);
        check(p3hip_fib_batch_prove(h_, n, a.data(), b.data(), ptrs.data(), lens.data()));
        std::vector<std::vector<uint8_t>> out(n);
        for (size_t i = 0; i < n; i++) out[i].assign(ptrs[i], ptrs[i] + lens[i]);
        return out;
    }

  private:
    p3hip_fib_batch_t* h_ = nullptr;
};

// generate_trace_rows' last right value = the public value x (fib_air.rs:57,68)
inline uint64_t fib_public_x(uint64_t a, uint64_t b, uint64_t n) {
    uint64_t l = a % P, r = b % P;
    for (uint64_t i = 1; i < n; i++) { uint64_t t = (l + r) % P; l = r; r = t; }
    return r;
}
// verify(&config, &FibonacciAir{}, &proof, &pis) (fib_air.rs:71-72); throws Error("fib_air verification failed: ...")
inline void verify_fib_air(const std::vector<uint8_t>& proof, uint64_t a, uint64_t b, uint64_t x, unsigned log_n,
                           FriParameters fp = FriParameters(), int hash = P3HIP_HASH_POSEIDON2) {
    p3hip_fri_params_t c{fp.log_blowup, fp.log_final_poly_len, fp.num_queries, fp.proof_of_work_bits};
    check(p3hip_verify_fib_air_hash(hash, proof.data(), proof.size(), a, b, x, log_n, &c));
}
// run_fib_air_zk (fib_air.rs:27-75) on the hip backend (non-hiding; either hash configuration): "fib_air ok (n=8, x=21)"
inline std::string run_fib_air(unsigned log_n = 3, uint64_t a = 0, uint64_t b = 1, FriParameters fp = FriParameters(),
                               int hash = P3HIP_HASH_POSEIDON2) {
    uint64_t n = 1ull << log_n, x = fib_public_x(a, b, n);
    FibAirProver prover(log_n, fp, hash);
    verify_fib_air(prover.prove(a, b), a, b, x, log_n, fp, hash);
    return "fib_air ok (n=" + std::to_string(n) + ", x=" + std::to_string(x) + ")";
}

// The String of the reference's runFibAirZk() (lib.rs:37-83) from the library: its own instance and configuration
// (n = 8, x = 21, Keccak hashes, hiding MMCS + PCS, seed 1) on the backend the selector names.  Never throws: failures are text.
inline std::string run_fib_air_zk_report() {
    std::string buf(1024, '\0');
    p3hip_run_fib_air_zk(&buf[0], buf.size());
    return std::string(buf.c_str());
}
// The String of runDftBenchmark() (lib.rs:86-131); cpu_dft supplies the CPU column (Radix2DitParallel in the Rust shim).
inline std::string run_dft_benchmark_report(p3hip_cpu_dft_fn cpu_dft = nullptr, void* user = nullptr) {
    std::string buf(1 << 14, '\0');
    p3hip_run_dft_benchmark(cpu_dft, user, &buf[0], buf.size());
    return std::string(buf.c_str());
}

}  // namespace p3hip
