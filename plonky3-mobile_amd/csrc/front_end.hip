// The reference's two report-returning entry points behind the C ABI, for native/src/lib.rs:37-131 to call:
//   run_fib_air_zk()      native/src/fib_air.rs:27-75   "fib_air zk ok (n=8, x=21)" / "fib_air zk failed: ..."
//   run_dft_benchmark()   native/src/fib_air.rs:98-222  one header line + one line per shape
// Text conventions of the JNI wrappers (lib.rs:45-67,100-115): never abort, failures are TEXT containing "failed"; a backend
// message waiting in the one-slot mailbox is appended as "\nHIP error: ..." (lib.rs:62-65 appends "\nVulkan error: ").
// Built on the public C ABI only (include/p3hip.h): what a host written against the header could do itself.
#include "../../include/p3hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

namespace {

int emit(const std::string& text, char* out, size_t cap) {
    if (out && cap) snprintf(out, cap, "%s", text.c_str());
    return (int)text.size();
}
std::string fmt(const char* f, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}
std::string take_error_text() {
    const char* e = p3hip_take_last_error();
    return e ? std::string(e) : std::string();
}
const char* backend_name(int k) {
    switch (k) {
        case P3HIP_BACKEND_CPU: return "cpu";
        case P3HIP_BACKEND_VULKAN: return "vulkan";
        case P3HIP_BACKEND_METAL: return "metal";
        case P3HIP_BACKEND_WEBGPU: return "webgpu";
        default: return "hip";
    }
}
// fib_air.rs:88-96
double percentile_ms(std::vector<double> s, double q) {
    if (s.empty()) return 0.0;
    std::sort(s.begin(), s.end());
    const size_t n = s.size();
    size_t idx = (size_t)std::ceil(q * (double)n);
    idx = idx ? idx - 1 : 0;
    return s[std::min(idx, n - 1)];
}
struct Stats { double avg, med, p95; };
Stats stats(const std::vector<double>& v) {
    double sum = 0;
    for (double x : v) sum += x;
    return {v.empty() ? 0.0 : sum / (double)v.size(), percentile_ms(v, 0.50), percentile_ms(v, 0.95)};
}
double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

int p3hip_run_fib_air_zk(char* out, size_t cap) {
    std::string msg;
    const int backend = p3hip_get_backend();
    if (backend != P3HIP_BACKEND_HIP) {
        // the selector is honoured, not overridden (fib_air.rs:60 hard-codes Vulkan; the patched shim asks GpuDft::default())
        msg = fmt("fib_air zk failed: backend '%s' is selected and libp3hip serves only 'hip' (setBackend(\"hip\"))", backend_name(backend));
        return emit(msg, out, cap);
    }
    // fib_air.rs:56-57,62: n = 8, x = 21, create_test_fri_params(challenge_mmcs, 2) = {log_blowup 2, log_final_poly_len 2,
    // num_queries 2, proof_of_work_bits 1}; Keccak hashes (hash kind 1), hiding MMCS + PCS, SmallRng::seed_from_u64(1)
    const unsigned log_n = 3;
    const uint64_t a = 0, b = 1, x = 21;
    const p3hip_fri_params_t fp{2, 2, 2, 1};
    p3hip_fib_prover_t* prover = nullptr;
    int rc = p3hip_fib_prover_create_hiding(1, log_n, &fp, 1, nullptr, 1, &prover);
    const uint8_t* proof = nullptr;
    size_t len = 0;
    if (rc == 0) rc = p3hip_fib_prover_prove(prover, a, b, &proof, &len);
    if (rc != 0) {
        msg = "fib_air zk failed: " + take_error_text();
    } else {
        rc = p3hip_verify_fib_air_hiding(1, proof, len, a, b, x, log_n, &fp);
        if (rc != 0) {
            std::string why = take_error_text();  // "fib_air verification failed: <check>"; the reference prints format!("{err:?}")
            const std::string prefix = "fib_air verification failed: ";
            if (why.compare(0, prefix.size(), prefix) == 0) why = why.substr(prefix.size());
            msg = "fib_air zk failed: " + why;
        } else {
            msg = fmt("fib_air zk ok (n=%u, x=%llu)", 1u << log_n, (unsigned long long)x);
        }
    }
    if (prover) p3hip_fib_prover_destroy(prover);
    const std::string pending = take_error_text();
    if (!pending.empty()) msg += "\nHIP error: " + pending;
    return emit(msg, out, cap);
}

int p3hip_run_dft_benchmark(p3hip_cpu_dft_fn cpu_dft, void* user, char* out, size_t cap) {
    std::string err;
    {
        char avail[256];
        if (p3hip_is_available(avail, sizeof avail) != 0) {  // fib_air.rs:99: is_vulkan_available()?
            (void)take_error_text();
            return emit(std::string("dft benchmark failed: ") + avail, out, cap);
        }
    }
    static const size_t cases[][2] = {{256, 8}, {1024, 8}, {4096, 8}, {16384, 8}, {4096, 32}, {16384, 32}, {4096, 64},
                                      {4096, 128}, {16384, 64}, {16384, 128}, {256, 16000}};  // fib_air.rs:103-117
    const size_t warmup = 1, repeats = 10, e2e_batch = 4;
    std::vector<std::string> lines{fmt("dft benchmark (repeats=%zu, warmup=%zu, stats=avg/median/p95)", repeats, warmup)};
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return emit("dft benchmark failed: cannot create a HIP stream / events", out, cap);
    auto fail_with = [&](const std::string& m) { err = m; };
    for (const auto& c : cases) {
        const size_t h = c[0], w = c[1], n = h * w;
        std::vector<uint32_t> input(n), gpu_out(n), cpu_out;
        for (size_t i = 0; i < n; i++) {  // benchmark_input (fib_air.rs:77-86): canonical (17 i + 3) mod P -> Montgomery word
            const uint64_t v = ((uint64_t)i * 17u + 3u) % 0x78000001ull;
            input[i] = (uint32_t)((v << 32) % 0x78000001ull);
        }
        (void)take_error_text();
        uint32_t *pin_in = nullptr, *pin_out = nullptr, *d_in = nullptr, *d_out = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&pin_in), n * 4) != hipSuccess || hipHostMalloc(reinterpret_cast<void**>(&pin_out), n * 4 * e2e_batch) != hipSuccess ||
            p3hip_malloc(reinterpret_cast<void**>(&d_in), n * 4 * e2e_batch) != 0 || p3hip_malloc(reinterpret_cast<void**>(&d_out), n * 4 * e2e_batch) != 0) {
            fail_with(fmt("allocation failed at h=%zu, w=%zu", h, w));
        }
        Stats cpu{0, 0, 0}, e2e{}, batched{}, kern{};
        if (err.empty()) {
            memcpy(pin_in, input.data(), n * 4);
            for (size_t i = 0; i < warmup && err.empty(); i++)
                if (p3hip_dft_batch_bb31(input.data(), gpu_out.data(), h, w) != 0) fail_with(take_error_text());
        }
        if (err.empty() && cpu_dft) {
            cpu_out.resize(n);
            std::vector<double> s;
            for (size_t i = 0; i < warmup; i++) (void)cpu_dft(user, input.data(), cpu_out.data(), h, w);
            for (size_t r = 0; r < repeats && err.empty(); r++) {
                const double t = now_ms();
                if (cpu_dft(user, input.data(), cpu_out.data(), h, w) != 0) fail_with("cpu benchmark output missing");
                s.push_back(now_ms() - t);
            }
            cpu = stats(s);
        }
        if (err.empty()) {  // e2e: host matrix in, host matrix out (the dft_batch boundary, backend_vulkan.rs:1988-2063)
            std::vector<double> s;
            for (size_t r = 0; r < repeats && err.empty(); r++) {
                const double t = now_ms();
                if (p3hip_dft_batch_bb31(input.data(), gpu_out.data(), h, w) != 0) fail_with(take_error_text());
                s.push_back(now_ms() - t);
            }
            e2e = stats(s);
        }
        if (err.empty()) {  // e2e batched: e2e_batch transforms per synchronisation (backend_vulkan.rs:1695-1987)
            std::vector<double> s;
            for (size_t r = 0; r < warmup + repeats && err.empty(); r++) {
                const double t = now_ms();
                for (size_t k = 0; k < e2e_batch && err.empty(); k++) {
                    if (hipMemcpyAsync(d_in + k * n, pin_in, n * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
                        p3hip_dft_batch_bb31_dev(d_in + k * n, d_out + k * n, h, w, st) != 0 ||
                        hipMemcpyAsync(pin_out + k * n, d_out + k * n, n * 4, hipMemcpyDeviceToHost, st) != hipSuccess)
                        fail_with("batched transform failed: " + take_error_text());
                }
                if (hipStreamSynchronize(st) != hipSuccess) fail_with("stream synchronisation failed");
                if (r >= warmup) s.push_back((now_ms() - t) / (double)e2e_batch);
            }
            batched = stats(s);
        }
        if (err.empty()) {  // kernel only: device resident, HIP events (backend_vulkan.rs:1428-1693 uses GPU timestamps)
            std::vector<double> s;
            for (size_t r = 0; r < warmup + repeats && err.empty(); r++) {
                (void)hipEventRecord(e0, st);
                if (p3hip_dft_batch_bb31_dev(d_in, d_out, h, w, st) != 0) fail_with(take_error_text());
                (void)hipEventRecord(e1, st);
                (void)hipEventSynchronize(e1);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= warmup) s.push_back(ms);
            }
            kern = stats(s);
        }
        if (err.empty()) {
            const std::string pending = take_error_text();  // fib_air.rs:182-186: any backend error during the runs is fatal
            if (!pending.empty()) fail_with(fmt("hip benchmark error at h=%zu, w=%zu: ", h, w) + pending);
        }
        if (err.empty() && cpu_dft) {  // fib_air.rs:190-196: one API-path transform, compared with the CPU's
            if (p3hip_dft_batch_bb31(input.data(), gpu_out.data(), h, w) != 0) fail_with(take_error_text());
            else if (gpu_out != cpu_out || memcmp(pin_out, cpu_out.data(), n * 4) != 0) fail_with(fmt("dft benchmark mismatch at h=%zu, w=%zu", h, w));
        }
        if (pin_in) (void)hipHostFree(pin_in);
        if (pin_out) (void)hipHostFree(pin_out);
        if (d_in) (void)p3hip_free(d_in);
        if (d_out) (void)p3hip_free(d_out);
        if (!err.empty()) break;
        std::string line = fmt("h=%zu, w=%zu:", h, w);
        if (cpu_dft) line += fmt(" cpu(avg=%.3f med=%.3f p95=%.3f)ms", cpu.avg, cpu.med, cpu.p95);
        const struct { const char* name; const char* sp; Stats s; } cols[] = {{"hip_e2e", "e2e", e2e}, {"hip_e2e_batched", "e2e_batched", batched}, {"hip_kernel", "kernel", kern}};
        for (const auto& col : cols) {
            line += fmt(" %s(avg=%.3f med=%.3f p95=%.3f)ms", col.name, col.s.avg, col.s.med, col.s.p95);
            if (cpu_dft) line += fmt(" speedup_%s(avg)=%.2fx", col.sp, col.s.avg > 0 ? cpu.avg / col.s.avg : 0.0);
        }
        lines.push_back(line);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(st);
    if (!err.empty()) return emit("dft benchmark failed: " + err, out, cap);
    std::string text;
    for (size_t i = 0; i < lines.size(); i++) text += (i ? "\n" : "") + lines[i];
    return emit(text, out, cap);
}

}  // extern "C"
