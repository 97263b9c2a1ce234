// C++ host program over include/p3hip.hpp — the compiled-language counterpart of the reference's harness
// (native/src/fib_air.rs:98-222 run_dft_benchmark + :27-75 run_fib_air_zk): runs the reference's benchmark
// shapes through GpuDft, checks the round trip idft(dft(x)) == x, commits and opens a tree, proves fib_air.
// Build: g++ -std=c++17 -O2 -Iinclude tools/host_demo.cpp -Lplonky3-mobile_amd -lp3hip -Wl,-rpath,... -o tools/_bin/host_demo
#include <chrono>
#include <cstdio>

#include "p3hip.hpp"

using namespace p3hip;

int main(int argc, char** argv) {
    try {
        // selector semantics (gpu_dft.rs:53-63)
        set_backend_kind_from_str("HIP");
        bool threw = false;
        try { set_backend_kind_from_str("cuda"); } catch (const Error& e) { threw = std::string(e.what()) == "unknown backend 'cuda'"; }
        if (!threw || get_backend_kind() != BackendKind::Hip) { std::printf("FAIL selector\n"); return 1; }
        auto avail = is_available();
        std::printf("%s\n", avail.second.c_str());
        if (!avail.first) return 2;
        GpuDft dft;
        const size_t cases[][2] = {{256, 8}, {1024, 8}, {4096, 32}, {16384, 8}, {256, 1000}};  // from fib_air.rs:103-117
        std::printf("dft benchmark (repeats=5, warmup=1, stats=avg/median/p95)\n");
        for (auto& c : cases) {
            RowMajorMatrix x = benchmark_input(c[0], c[1]);
            dft.dft_batch(x);
            std::vector<double> ms;
            RowMajorMatrix y;
            for (int r = 0; r < 5; r++) {
                auto t0 = std::chrono::steady_clock::now();
                y = dft.dft_batch(x);
                ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
            if (dft.idft_batch(y).values != x.values) { std::printf("FAIL round trip h=%zu w=%zu\n", c[0], c[1]); return 3; }
            double avg = 0; for (double v : ms) avg += v; avg /= ms.size();
            std::printf("h=%zu, w=%zu: hip_e2e(avg=%.3f med=%.3f p95=%.3f)ms\n", c[0], c[1], avg, percentile_ms(ms, 0.5), percentile_ms(ms, 0.95));
        }
        // non power-of-two height is an error value, not a crash (backend_vulkan.rs:1992-1995)
        try { dft.dft_batch(RowMajorMatrix(std::vector<uint32_t>(24), 2)); std::printf("FAIL no error\n"); return 4; }
        catch (const Error& e) { std::printf("expected error: %s\n", e.what()); }
        // Mmcs
        RowMajorMatrix lde = dft.coset_lde_batch(benchmark_input(1024, 2), 1, GENERATOR_MONTY, true);
        MerkleTreeMmcs mmcs;
        auto ct = mmcs.commit({lde});
        auto op = mmcs.open_batch(5, ct.second);
        if (op.first[0] != std::vector<uint32_t>(lde.values.begin() + 10, lde.values.begin() + 12) || op.second.size() != 11 * 8) { std::printf("FAIL open\n"); return 5; }
        std::printf("root[0]=%08x path=%zu digests\n", ct.first[0], op.second.size() / 8);
        // the same commitment under the reference's own Keccak hashes (fib_air.rs:28-38)
        MerkleTreeMmcs kmmcs(P3HIP_HASH_KECCAK);
        auto kt = kmmcs.commit({lde});
        auto kop = kmmcs.open_batch(5, kt.second);
        if (kt.first == ct.first || kop.first[0] != op.first[0] || kop.second.size() != 11 * 8) { std::printf("FAIL keccak mmcs\n"); return 10; }
        std::printf("keccak root[0]=%08x\n", kt.first[0]);
        // prover
        FibAirProver prover(argc > 1 ? std::atoi(argv[1]) : 12);
        auto proof = prover.prove(0, 1);
        auto again = prover.prove(0, 1);
        if (proof != again || proof.size() < 1000) { std::printf("FAIL prove\n"); return 6; }
        std::printf("fib_air ok: proof %zu bytes, first word %08x\n", proof.size(), *(const uint32_t*)proof.data());
        std::string rep = run_fib_air();  // the reference's instance: n = 8, x = 21 (fib_air.rs:56-57)
        std::printf("%s\n", rep.c_str());
        if (rep != "fib_air ok (n=8, x=21)") { std::printf("FAIL run_fib_air\n"); return 7; }
        // the same instance under the reference's own hashes (Keccak MMCS + Keccak-256 hash challenger)
        if (run_fib_air(3, 0, 1, FriParameters(), P3HIP_HASH_KECCAK) != rep) { std::printf("FAIL run_fib_air keccak\n"); return 11; }
        try { verify_fib_air(proof, 0, 1, 5, argc > 1 ? std::atoi(argv[1]) : 12); std::printf("FAIL verify accepted a wrong x\n"); return 8; }
        catch (const Error& e) { std::printf("expected error: %s\n", e.what()); }
        {   // batch of independent proofs through the library's prover pool
            unsigned ln = argc > 1 ? std::atoi(argv[1]) : 12;
            FibAirBatchProver pool(ln, 4);
            std::vector<std::pair<uint64_t, uint64_t>> inst;
            for (uint64_t i = 0; i < 16; i++) inst.push_back({i, i + 1});
            auto t0 = std::chrono::steady_clock::now();
            auto proofs = pool.prove(inst);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            for (size_t i = 0; i < inst.size(); i++) verify_fib_air(proofs[i], inst[i].first, inst[i].second, fib_public_x(inst[i].first, inst[i].second, 1ull << ln), ln);
            if (proofs[0] != proof) { std::printf("FAIL batch proof 0 differs\n"); return 9; }
            std::printf("batch of %zu proofs (2^%u rows): %.1f ms, all verified\n", inst.size(), ln, ms);
        }
        {   // the reference's two report strings from the library itself (lib.rs:37-131 -> fib_air::run_fib_air_zk / run_dft_benchmark)
            std::string zk = run_fib_air_zk_report();  // its own instance and configuration: Keccak hashes, hiding MMCS + PCS, seed 1
            std::printf("%s\n", zk.c_str());
            if (zk != "fib_air zk ok (n=8, x=21)") { std::printf("FAIL run_fib_air_zk_report\n"); return 12; }
            set_backend_kind_from_str("vulkan");
            std::string refused = run_fib_air_zk_report();
            set_backend_kind_from_str("hip");
            if (refused.find("failed") == std::string::npos) { std::printf("FAIL the report ignored the selector\n"); return 13; }
            std::string bm = run_dft_benchmark_report();  // no CPU column: libp3hip has no CPU transform to time
            size_t lines = 1;
            for (char ch : bm) lines += ch == '\n';
            std::printf("%s\n", bm.substr(0, bm.find('\n')).c_str());
            if (lines != 12 || bm.find("failed") != std::string::npos) { std::printf("FAIL run_dft_benchmark_report\n"); return 14; }
        }
        std::printf("OK\n");
        return 0;
    } catch (const Error& e) {
        std::printf("p3hip error %d: %s\n", e.code, e.what());
        return 10;
    }
}
