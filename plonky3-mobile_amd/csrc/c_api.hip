// extern "C" surface of libp3hip (include/p3hip.h).  Thin: argument checks, context lookup, host<->device
// staging for the host-pointer entry points.  Never throws across the boundary (JNI wrappers in the
// reference catch panics, native/src/lib.rs:45-59; here every path returns a status code).
#include "../../include/p3hip.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <deque>
#include <thread>
#include <cctype>
#include <cstdio>
#include <cstring>

#include "bb31.hip.h"
#include "common.h"
#include "mmcs.h"
#include "prover.h"
#include "rng.h"

namespace p3 {
bool take_error(std::string* out);
}
using namespace p3;

// gpu_dft.rs:41: `static BACKEND_KIND: AtomicU8` — default is the GPU backend.
static std::atomic<uint8_t> g_backend{P3HIP_BACKEND_HIP};

template <class F>
static int guarded(F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        return fail(ERR_INTERNAL, std::string("exception: ") + e.what());
    } catch (...) {
        return fail(ERR_INTERNAL, "unknown exception");
    }
}

extern "C" {

int p3hip_set_backend(const char* name) {
    if (!name) return fail(ERR_BACKEND, "unknown backend '<null>'");
    std::string v;
    for (const char* p = name; *p; ++p) v.push_back((char)std::tolower((unsigned char)*p));
    uint8_t kind;
    if (v == "cpu") kind = P3HIP_BACKEND_CPU;
    else if (v == "vulkan") kind = P3HIP_BACKEND_VULKAN;
    else if (v == "metal") kind = P3HIP_BACKEND_METAL;
    else if (v == "webgpu") kind = P3HIP_BACKEND_WEBGPU;
    else if (v == "hip") kind = P3HIP_BACKEND_HIP;
    else return fail(ERR_BACKEND, "unknown backend '" + v + "'");
    g_backend.store(kind, std::memory_order_relaxed);
    return OK;
}

int p3hip_get_backend(void) { return g_backend.load(std::memory_order_relaxed); }

int p3hip_is_available(char* msg, size_t cap) {
    return guarded([&]() -> int {
        Context* cx = nullptr;
        int rc = get_context(&cx);
        std::string text;
        if (rc == OK) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, cx->device) == hipSuccess)
                text = std::string("HIP available: ") + prop.name + " (" + prop.gcnArchName + ")";
            else
                text = "HIP available";
        } else {
            std::string err;
            take_error(&err);
            text = "HIP unavailable: " + err;
            set_error(err);
        }
        if (msg && cap) snprintf(msg, cap, "%s", text.c_str());
        return rc;
    });
}

const char* p3hip_take_last_error(void) {
    static thread_local std::string held;
    if (!take_error(&held)) return nullptr;
    return held.c_str();
}

int p3hip_malloc(void** dev_ptr, size_t bytes) {
    if (!dev_ptr) return fail(ERR_BAD_ARG, "p3hip_malloc: null out pointer");
    Context* cx;
    int rc = get_context(&cx);
    if (rc) return rc;
    P3_HIP(hipMalloc(dev_ptr, bytes ? bytes : 4));
    return OK;
}
int p3hip_free(void* dev_ptr) {
    if (dev_ptr) P3_HIP(hipFree(dev_ptr));
    return OK;
}
int p3hip_upload(void* dev_dst, const void* host_src, size_t bytes) {
    if (bytes && (!dev_dst || !host_src)) return fail(ERR_BAD_ARG, "p3hip_upload: null pointer");
    if (bytes) P3_HIP(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return OK;
}
int p3hip_download(void* host_dst, const void* dev_src, size_t bytes) {
    if (bytes && (!host_dst || !dev_src)) return fail(ERR_BAD_ARG, "p3hip_download: null pointer");
    if (bytes) P3_HIP(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return OK;
}
int p3hip_sync(void* stream) {
    P3_HIP(hipStreamSynchronize((hipStream_t)stream));
    return OK;
}
void p3hip_release_thread_context(void) { release_thread_contexts(); }

}  // extern "C"

// ---- TwoAdicSubgroupDft ------------------------------------------------------------------------
enum DftOp { OP_DFT, OP_IDFT, OP_COSET_DFT, OP_COSET_LDE };

static int dft_dev(DftOp op, const uint32_t* d_in, uint32_t* d_out, size_t h, size_t w, unsigned added_bits,
                   uint32_t shift, int br_out, hipStream_t stream) {
    return guarded([&]() -> int {
        if (h == 0 || w == 0) return OK;
        if (!d_in || !d_out) return fail(ERR_BAD_ARG, "null matrix pointer");
        if (w > 0xffffffffull) return fail(ERR_BAD_ARG, "width too large");
        if (shift >= bb::P) return fail(ERR_BAD_ARG, "shift is not a reduced Montgomery word");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        switch (op) {
            case OP_DFT: return ntt_dft(*cx, stream, d_in, d_out, h, (uint32_t)w, false);
            case OP_IDFT: return ntt_dft(*cx, stream, d_in, d_out, h, (uint32_t)w, true);
            case OP_COSET_DFT: return ntt_coset_dft(*cx, stream, d_in, d_out, h, (uint32_t)w, shift);
            default: return ntt_coset_lde(*cx, stream, d_in, d_out, h, (uint32_t)w, added_bits, shift, br_out != 0);
        }
    });
}

// Per-call timing line of the reference's backend (backend_vulkan.rs:1385-1423, log_vulkan_timing :23-39):
//   "vulkan dft: h=.. w=.. stages=.. upload=..ms stages=..ms readback=..ms total=..ms gpu(stage=.. copy_back=.. total=..)"
// Here: "hip dft: ..." with the gpu(...) part from HIP events on the stream.  Written to stderr when
// P3HIP_LOG_TIMING=1 (the reference always logs; a library should not) and always kept as the thread's last line
// (p3hip_last_timing_line), so a host can surface it like the reference's logcat line.
static thread_local std::string g_last_timing;
static bool log_timing_enabled() {
    static const bool on = [] { const char* e = getenv("P3HIP_LOG_TIMING"); return e && atoi(e) != 0; }();
    return on;
}

// host-pointer path: H2D, kernels, D2H, sync (the reference's e2e shape, backend_vulkan.rs:1107-1394)
static int dft_host(DftOp op, const uint32_t* in, uint32_t* out, size_t h, size_t w, unsigned added_bits,
                    uint32_t shift, int br_out) {
    return guarded([&]() -> int {
        if (h == 0 || w == 0) return OK;
        if (!in || !out) return fail(ERR_BAD_ARG, "null matrix pointer");
        if (!is_pow2(h)) return fail(ERR_BAD_ARG, "hip backend requires power-of-two height, got " + std::to_string(h));
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        using clk = std::chrono::steady_clock;
        auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
        const auto t_total = clk::now();
        size_t in_bytes = h * w * 4;
        size_t out_rows = op == OP_COSET_LDE ? (h << added_bits) : h;
        size_t out_bytes = out_rows * w * 4;
        hipStream_t stream = nullptr;
        rc = cx->ws(stream, 2).reserve(in_bytes);
        if (rc) return rc;
        rc = cx->ws(stream, 3).reserve(out_bytes);
        if (rc) return rc;
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        for (auto& e : ev) P3_HIP(hipEventCreate(&e));
        auto drop_events = [&] { for (auto& e : ev) if (e) (void)hipEventDestroy(e); };
        const auto t_upload = clk::now();
        if (hipMemcpyAsync(cx->ws(stream, 2).ptr, in, in_bytes, hipMemcpyHostToDevice, stream) != hipSuccess) { drop_events(); return fail(ERR_HIP, "upload failed"); }
        (void)hipEventRecord(ev[0], stream);
        const double upload_ms = ms_since(t_upload);
        const auto t_stages = clk::now();
        rc = dft_dev(op, cx->ws(stream, 2).as<uint32_t>(), cx->ws(stream, 3).as<uint32_t>(), h, w, added_bits, shift, br_out, stream);
        if (rc) { drop_events(); return rc; }
        (void)hipEventRecord(ev[1], stream);
        const double stages_ms = ms_since(t_stages);
        const auto t_read = clk::now();
        if (hipMemcpyAsync(out, cx->ws(stream, 3).ptr, out_bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) { drop_events(); return fail(ERR_HIP, "readback failed"); }
        (void)hipEventRecord(ev[2], stream);
        if (hipStreamSynchronize(stream) != hipSuccess) { drop_events(); return fail(ERR_HIP, "synchronise failed"); }
        const double readback_ms = ms_since(t_read), total_ms = ms_since(t_total);
        float g_stage = 0, g_copy = 0, g_total = 0;
        (void)hipEventElapsedTime(&g_stage, ev[0], ev[1]);
        (void)hipEventElapsedTime(&g_copy, ev[1], ev[2]);
        (void)hipEventElapsedTime(&g_total, ev[0], ev[2]);
        drop_events();
        char line[320];
        snprintf(line, sizeof line,
                 "hip dft: op=%s h=%zu w=%zu stages=%u upload=%.3fms stages=%.3fms readback=%.3fms total=%.3fms gpu(stage=%.3fms copy_back=%.3fms total=%.3fms)",
                 op == OP_DFT ? "dft" : op == OP_IDFT ? "idft" : op == OP_COSET_DFT ? "coset_dft" : "coset_lde", h, w,
                 log2u(out_rows), upload_ms, stages_ms, readback_ms, total_ms, g_stage, g_copy, g_total);
        g_last_timing = line;
        if (log_timing_enabled()) fprintf(stderr, "%s\n", line);
        return OK;
    });
}

extern "C" {

const char* p3hip_last_timing_line(void) { return g_last_timing.empty() ? nullptr : g_last_timing.c_str(); }

int p3hip_dft_batch_bb31(const uint32_t* in, uint32_t* out, size_t h, size_t w) {
    return dft_host(OP_DFT, in, out, h, w, 0, bb::ONE, 0);
}
int p3hip_idft_batch_bb31(const uint32_t* in, uint32_t* out, size_t h, size_t w) {
    return dft_host(OP_IDFT, in, out, h, w, 0, bb::ONE, 0);
}
int p3hip_coset_dft_batch_bb31(const uint32_t* in, uint32_t* out, size_t h, size_t w, uint32_t shift) {
    return dft_host(OP_COSET_DFT, in, out, h, w, 0, shift, 0);
}
int p3hip_coset_lde_batch_bb31(const uint32_t* in, uint32_t* out, size_t h, size_t w, unsigned added_bits,
                               uint32_t shift, int br_out) {
    return dft_host(OP_COSET_LDE, in, out, h, w, added_bits, shift, br_out);
}
int p3hip_dft_batch_bb31_dev(const uint32_t* in, uint32_t* out, size_t h, size_t w, void* stream) {
    return dft_dev(OP_DFT, in, out, h, w, 0, bb::ONE, 0, (hipStream_t)stream);
}
int p3hip_idft_batch_bb31_dev(const uint32_t* in, uint32_t* out, size_t h, size_t w, void* stream) {
    return dft_dev(OP_IDFT, in, out, h, w, 0, bb::ONE, 0, (hipStream_t)stream);
}
int p3hip_coset_dft_batch_bb31_dev(const uint32_t* in, uint32_t* out, size_t h, size_t w, uint32_t shift, void* stream) {
    return dft_dev(OP_COSET_DFT, in, out, h, w, 0, shift, 0, (hipStream_t)stream);
}
int p3hip_coset_lde_batch_bb31_dev(const uint32_t* in, uint32_t* out, size_t h, size_t w, unsigned added_bits,
                                   uint32_t shift, int br_out, void* stream) {
    return dft_dev(OP_COSET_LDE, in, out, h, w, added_bits, shift, br_out, (hipStream_t)stream);
}
int p3hip_coset_lde_from_coeffs_bb31_dev(const uint32_t* coeffs, uint32_t* out, size_t h, size_t w, unsigned added_bits,
                                         uint32_t shift, void* stream) {
    return guarded([&]() -> int {
        if (h == 0 || w == 0) return OK;
        if (!coeffs || !out || coeffs == out) return fail(ERR_BAD_ARG, "coset_lde_from_coeffs: null or aliased pointers");
        if (w > 0xffffffffull) return fail(ERR_BAD_ARG, "width too large");
        if (shift >= bb::P) return fail(ERR_BAD_ARG, "shift is not a reduced Montgomery word");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        hipStream_t st = (hipStream_t)stream;
        if ((rc = cx->ws(st, 2).reserve(h * w * 4))) return rc;
        return ntt_coset_lde_from_coeffs(*cx, st, coeffs, out, cx->ws(st, 2).as<uint32_t>(), h, (uint32_t)w, added_bits, shift);
    });
}
int p3hip_dft_plan_bb31(size_t height, size_t width, uint32_t* stages_per_pass, size_t cap, size_t* n_passes) {
    return guarded([&]() -> int {
        if (!n_passes) return fail(ERR_BAD_ARG, "dft_plan: null argument");
        *n_passes = 0;
        if (height == 0 || width == 0) return OK;
        if (!is_pow2(height)) return fail(ERR_BAD_ARG, "hip backend requires power-of-two height, got " + std::to_string(height));
        const uint32_t n = log2u(height);
        if (n > bb::TWO_ADICITY) return fail(ERR_BAD_ARG, "height exceeds BabyBear two-adicity");
        const std::vector<uint32_t> d = ntt_dft_plan(n);
        *n_passes = d.size();
        for (size_t i = 0; i < d.size() && i < cap; i++) if (stages_per_pass) stages_per_pass[i] = d[i];
        return OK;
    });
}

int p3hip_bit_reverse_rows_dev(const uint32_t* in, uint32_t* out, size_t h, size_t w, void* stream) {
    return guarded([&]() -> int {
        if (h == 0 || w == 0) return OK;
        if (!in || !out || in == out) return fail(ERR_BAD_ARG, "bit_reverse_rows: null or aliased pointers");
        return bit_reverse_rows((hipStream_t)stream, in, out, h, (uint32_t)w);
    });
}

// ---- FibonacciAir workload ----------------------------------------------------------------------
int p3hip_fib_trace_dev(uint64_t a, uint64_t b, size_t n, uint32_t* d_out, void* stream) {
    return guarded([&]() -> int {
        if (!n) return OK;
        if (!d_out) return fail(ERR_BAD_ARG, "fib_trace: null output");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        return fib_trace((hipStream_t)stream, a, b, n, d_out);
    });
}

// ---- Poseidon2 ----------------------------------------------------------------------------------
int p3hip_poseidon2_permute_dev(uint32_t* d_states, size_t n, void* stream) {
    return guarded([&]() -> int {
        if (n && !d_states) return fail(ERR_BAD_ARG, "null state pointer");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        return poseidon2_permute_states((hipStream_t)stream, d_states, n);
    });
}
int p3hip_poseidon2_permute(uint32_t* states, size_t n) {
    return guarded([&]() -> int {
        if (!n) return OK;
        if (!states) return fail(ERR_BAD_ARG, "null state pointer");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        hipStream_t stream = nullptr;
        rc = cx->ws(stream, 2).reserve(n * 64);
        if (rc) return rc;
        P3_HIP(hipMemcpy(cx->ws(stream, 2).ptr, states, n * 64, hipMemcpyHostToDevice));
        rc = poseidon2_permute_states(stream, cx->ws(stream, 2).as<uint32_t>(), n);
        if (rc) return rc;
        P3_HIP(hipMemcpy(states, cx->ws(stream, 2).ptr, n * 64, hipMemcpyDeviceToHost));
        return OK;
    });
}

// test / diagnostics entries of the two arithmetic forms of the permutation
int p3hip_poseidon2_permute_variant_dev(uint32_t* d_states, size_t n, int variant, void* stream) {
    return guarded([&]() -> int {
        if (n && !d_states) return fail(ERR_BAD_ARG, "null state pointer");
        if (variant != 0 && variant != 1) return fail(ERR_BAD_ARG, "poseidon2 variant: 0 = int32, 1 = fp64");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        return poseidon2_permute_states_variant((hipStream_t)stream, d_states, n, variant);
    });
}
int p3hip_poseidon2_f64_probe_dev(const double* d_in, uint32_t* d_out, size_t n, int mode, void* stream) {
    return guarded([&]() -> int {
        if (n && (!d_in || !d_out)) return fail(ERR_BAD_ARG, "null pointer");
        if (mode < 0 || mode > 2) return fail(ERR_BAD_ARG, "probe mode: 0 = permutation, 1 = internal rounds, 2 = reduce");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        return poseidon2_f64_probe((hipStream_t)stream, d_in, d_out, n, mode);
    });
}

// ---- Mmcs ---------------------------------------------------------------------------------------
}  // extern "C"
struct p3hip_tree {
    Tree* t;
};
extern "C" {

static int commit_async_kind(int kind, const uint32_t* const* d_mats, const size_t* heights, const size_t* widths,
                             size_t n_mats, p3hip_tree_t** tree_out, void* stream) {
    return guarded([&]() -> int {
        if (!tree_out) return fail(ERR_BAD_ARG, "mmcs_commit: null tree_out");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        Tree* t = nullptr;
        rc = mmcs_commit((hipStream_t)stream, d_mats, heights, widths, n_mats, &t, nullptr, nullptr, kind, nullptr, cx->profile);
        if (rc) return rc;
        *tree_out = new p3hip_tree{t};
        return OK;
    });
}
int p3hip_mmcs_commit_async_dev(const uint32_t* const* d_mats, const size_t* heights, const size_t* widths,
                                size_t n_mats, p3hip_tree_t** tree_out, void* stream) {
    return commit_async_kind(HASH_POSEIDON2, d_mats, heights, widths, n_mats, tree_out, stream);
}
int p3hip_mmcs_commit_hash_dev(int hash, const uint32_t* const* d_mats, const size_t* heights, const size_t* widths,
                               size_t n_mats, uint32_t root_out[8], p3hip_tree_t** tree_out, void* stream) {
    if (!root_out) return fail(ERR_BAD_ARG, "mmcs_commit: null root_out");
    int rc = commit_async_kind(hash, d_mats, heights, widths, n_mats, tree_out, stream);
    if (rc) return rc;
    rc = p3hip_mmcs_root(*tree_out, root_out, stream);
    if (rc) { p3hip_mmcs_free(*tree_out); *tree_out = nullptr; }
    return rc;
}
// commit with CALLER-PROVIDED digest-layer storage (p3hip_mmcs_layer_words(max height) words): nothing is allocated, so
// the call only enqueues (the fib_air prover commits this way into its arena)
size_t p3hip_mmcs_layer_words(size_t max_height) { return mmcs_layer_words(max_height); }
int p3hip_mmcs_commit_into_dev(int hash, const uint32_t* const* d_mats, const size_t* heights, const size_t* widths, size_t n_mats,
                               uint32_t* d_layers, p3hip_tree_t** tree_out, void* stream) {
    return guarded([&]() -> int {
        if (!tree_out || !d_layers) return fail(ERR_BAD_ARG, "mmcs_commit_into: null argument");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        Tree* t = nullptr;
        rc = mmcs_commit((hipStream_t)stream, d_mats, heights, widths, n_mats, &t, d_layers, nullptr, hash, nullptr, cx->profile);
        if (rc) return rc;
        *tree_out = new p3hip_tree{t};
        return OK;
    });
}
int p3hip_keccak_f_dev(uint64_t* d_states, size_t n, void* stream) {
    return guarded([&]() -> int {
        if (!d_states && n) return fail(ERR_BAD_ARG, "keccak_f: null states");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        return keccak_f_states((hipStream_t)stream, d_states, n);
    });
}
int p3hip_mmcs_root(const p3hip_tree_t* tree, uint32_t root_out[8], void* stream) {
    return guarded([&]() -> int {
        if (!tree || !root_out) return fail(ERR_BAD_ARG, "mmcs_root: null argument");
        return mmcs_root((hipStream_t)stream, *tree->t, root_out);
    });
}
int p3hip_mmcs_commit_dev(const uint32_t* const* d_mats, const size_t* heights, const size_t* widths,
                          size_t n_mats, uint32_t root_out[8], p3hip_tree_t** tree_out, void* stream) {
    if (!root_out) return fail(ERR_BAD_ARG, "mmcs_commit: null root_out");
    int rc = p3hip_mmcs_commit_async_dev(d_mats, heights, widths, n_mats, tree_out, stream);
    if (rc) return rc;
    rc = p3hip_mmcs_root(*tree_out, root_out, stream);
    if (rc) { p3hip_mmcs_free(*tree_out); *tree_out = nullptr; }
    return rc;
}
size_t p3hip_mmcs_log_max_height(const p3hip_tree_t* tree) { return tree ? tree->t->log_max_height : 0; }
size_t p3hip_mmcs_num_layers(const p3hip_tree_t* tree) { return tree ? tree->t->layer_len.size() : 0; }
const uint32_t* p3hip_mmcs_layer_dev(const p3hip_tree_t* tree, size_t layer, size_t* len_out) {
    if (!tree || layer >= tree->t->layer_len.size()) return nullptr;
    if (len_out) *len_out = tree->t->layer_len[layer];
    return tree->t->layers + tree->t->layer_off[layer];
}
int p3hip_mmcs_open_batch(const p3hip_tree_t* tree, size_t index, uint32_t* rows_out, uint32_t* path_out,
                          void* stream) {
    return guarded([&]() -> int {
        if (!tree) return fail(ERR_BAD_ARG, "mmcs_open_batch: null tree");
        size_t row_words = 0;
        for (size_t w : tree->t->widths) row_words += w;
        if ((row_words && !rows_out) || (tree->t->log_max_height && !path_out))
            return fail(ERR_BAD_ARG, "mmcs_open_batch: null output buffer");
        return mmcs_open((hipStream_t)stream, *tree->t, index, rows_out, path_out);
    });
}
void p3hip_mmcs_free(p3hip_tree_t* tree) {
    if (!tree) return;
    delete tree->t;
    delete tree;
}
int p3hip_mmcs_commit(const uint32_t* const* mats, const size_t* heights, const size_t* widths, size_t n_mats,
                      uint32_t root_out[8], p3hip_tree_t** tree_out) {
    return p3hip_mmcs_commit_hash(HASH_POSEIDON2, mats, heights, widths, n_mats, root_out, tree_out);
}
int p3hip_mmcs_commit_hash(int hash, const uint32_t* const* mats, const size_t* heights, const size_t* widths, size_t n_mats,
                           uint32_t root_out[8], p3hip_tree_t** tree_out) {
    return guarded([&]() -> int {
        if (!mats || !heights || !widths || !n_mats || !root_out || !tree_out)
            return fail(ERR_BAD_ARG, "mmcs_commit: null/empty argument");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        std::vector<void*> owned;
        std::vector<const uint32_t*> dptr;
        auto cleanup = [&]() { for (void* p : owned) (void)hipFree(p); };
        for (size_t i = 0; i < n_mats; i++) {
            size_t bytes = heights[i] * widths[i] * 4;
            void* d = nullptr;
            if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) { cleanup(); return fail(ERR_HIP, "hipMalloc failed"); }
            owned.push_back(d);
            if (bytes && hipMemcpy(d, mats[i], bytes, hipMemcpyHostToDevice) != hipSuccess) { cleanup(); return fail(ERR_HIP, "hipMemcpy failed"); }
            dptr.push_back((const uint32_t*)d);
        }
        rc = p3hip_mmcs_commit_hash_dev(hash, dptr.data(), heights, widths, n_mats, root_out, tree_out, nullptr);
        if (rc) { cleanup(); return rc; }
        (*tree_out)->t->owned = owned;
        return OK;
    });
}

}  // extern "C"

// ---- fib_air prover (p3_uni_stark::prove as driven by native/src/fib_air.rs:60-70) ---------------
struct p3hip_fib_prover {
    std::unique_ptr<FibProver> plain;         // TwoAdicFriPcs + MerkleTreeMmcs
    std::unique_ptr<FibHidingProver> hiding;  // HidingFriPcs + MerkleTreeHidingMmcs (fib_air.rs:40-65)
    std::vector<uint8_t> last;
    size_t proof_len = 0;  // length of this prover's proofs, known once one has SUCCEEDED (0 before); never read from `last`
    int prove(uint64_t a, uint64_t b) {
        int rc = hiding ? hiding->prove(a, b, &last) : plain->prove(a, b, &last);
        if (rc) last.clear();  // a failed proof leaves nothing behind that could be mistaken for one
        else proof_len = last.size();
        return rc;
    }
};

extern "C" {

int p3hip_set_thread_profile(int profile) {
    return guarded([&]() -> int {
        if (profile != P3HIP_PROFILE_THROUGHPUT && profile != P3HIP_PROFILE_LATENCY) return fail(ERR_BAD_ARG, "set_thread_profile: unknown profile");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        cx->profile = profile == P3HIP_PROFILE_THROUGHPUT ? PROFILE_THROUGHPUT : PROFILE_LATENCY;
        return OK;
    });
}
int p3hip_get_thread_profile(void) {
    Context* cx;
    if (get_context(&cx)) return -1;
    return cx->profile == PROFILE_THROUGHPUT ? P3HIP_PROFILE_THROUGHPUT : P3HIP_PROFILE_LATENCY;
}

int p3hip_fib_prover_create_profile(int profile, int hash, int hiding, uint64_t seed, unsigned log_n, const p3hip_fri_params_t* params,
                                    void* stream, int own_stream, p3hip_fib_prover_t** out) {
    return guarded([&]() -> int {
        if (!params || !out) return fail(ERR_BAD_ARG, "fib_prover_create: null argument");
        if (profile != P3HIP_PROFILE_THROUGHPUT && profile != P3HIP_PROFILE_LATENCY) return fail(ERR_BAD_ARG, "fib_prover_create: unknown profile");
        const int prof = profile == P3HIP_PROFILE_THROUGHPUT ? PROFILE_THROUGHPUT : PROFILE_LATENCY;
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        hipStream_t st = (hipStream_t)stream;
        bool own = false;
        if (own_stream) {
            P3_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            own = true;
        }
        std::unique_ptr<p3hip_fib_prover> p(new p3hip_fib_prover());
        FriParams fp{params->log_blowup, params->log_final_poly_len, params->num_queries, params->proof_of_work_bits};
        // on failure the prover's destructor destroys the owned stream
        if (hiding) {
            p->hiding.reset(new FibHidingProver());
            rc = p->hiding->init(log_n, fp, st, own, hash, seed, prof);
        } else {
            p->plain.reset(new FibProver());
            rc = p->plain->init(log_n, fp, st, own, hash, prof);
        }
        if (rc) return rc;
        *out = p.release();
        return OK;
    });
}
// a prover created on its own proves one proof at a time (what the reference does, fib_air.rs:56-72): the latency profile
int p3hip_fib_prover_create(unsigned log_n, const p3hip_fri_params_t* params, void* stream, int own_stream,
                            p3hip_fib_prover_t** out) {
    return p3hip_fib_prover_create_profile(P3HIP_PROFILE_LATENCY, HASH_POSEIDON2, 0, 0, log_n, params, stream, own_stream, out);
}
int p3hip_fib_prover_create_hash(int hash, unsigned log_n, const p3hip_fri_params_t* params, void* stream, int own_stream,
                                 p3hip_fib_prover_t** out) {
    return p3hip_fib_prover_create_profile(P3HIP_PROFILE_LATENCY, hash, 0, 0, log_n, params, stream, own_stream, out);
}

int p3hip_fib_prover_prove(p3hip_fib_prover_t* prover, uint64_t a, uint64_t b, const uint8_t** proof_out,
                           size_t* proof_len) {
    return guarded([&]() -> int {
        if (!prover || !proof_out || !proof_len) return fail(ERR_BAD_ARG, "fib_prover_prove: null argument");
        int rc = prover->prove(a, b);
        if (rc) return rc;
        *proof_out = prover->last.data();
        *proof_len = prover->last.size();
        return OK;
    });
}

int p3hip_fib_prover_prove_into(p3hip_fib_prover_t* prover, uint64_t a, uint64_t b, uint8_t* out, size_t cap, size_t* proof_len) {
    return guarded([&]() -> int {
        if (!prover || !out || !proof_len) return fail(ERR_BAD_ARG, "fib_prover_prove_into: null argument");
        // proofs of one prover have one length: once it is known, a buffer that cannot hold it fails BEFORE any work is queued
        // (the length is a field of its own, set by a proof that succeeded: `last` may be empty or stale after a failure)
        if (prover->proof_len && prover->proof_len > cap) {
            *proof_len = prover->proof_len;
            return fail(ERR_BAD_ARG, "fib_prover_prove_into: the proof needs " + std::to_string(prover->proof_len) + " bytes");
        }
        int rc = prover->prove(a, b);
        if (rc) return rc;
        *proof_len = prover->last.size();
        if (prover->last.size() > cap) return fail(ERR_BAD_ARG, "fib_prover_prove_into: the proof needs " + std::to_string(prover->last.size()) + " bytes");
        memcpy(out, prover->last.data(), prover->last.size());
        return OK;
    });
}

int p3hip_fib_prover_enqueue(p3hip_fib_prover_t* prover, uint64_t a, uint64_t b) {
    return guarded([&]() -> int {
        if (!prover) return fail(ERR_BAD_ARG, "fib_prover_enqueue: null argument");
        if (!prover->plain) return fail(ERR_BAD_ARG, "fib_prover_enqueue: the hiding prover proves one proof at a time");
        return prover->plain->enqueue(a, b);
    });
}
int p3hip_fib_prover_finish(p3hip_fib_prover_t* prover, const uint8_t** proof_out, size_t* proof_len) {
    return guarded([&]() -> int {
        if (!prover || !proof_out || !proof_len) return fail(ERR_BAD_ARG, "fib_prover_finish: null argument");
        if (!prover->plain) return fail(ERR_BAD_ARG, "fib_prover_finish: the hiding prover proves one proof at a time");
        int rc = prover->plain->finish(&prover->last);
        if (rc) { prover->last.clear(); return rc; }
        prover->proof_len = prover->last.size();
        *proof_out = prover->last.data();
        *proof_len = prover->last.size();
        return OK;
    });
}

int p3hip_fib_prover_stage_times(p3hip_fib_prover_t* prover, double out_ms[6], uint64_t* proofs, int reset) {
    if (!prover || !out_ms) return fail(ERR_BAD_ARG, "fib_prover_stage_times: null argument");
    if (!prover->plain) return fail(ERR_BAD_ARG, "fib_prover_stage_times: not kept by the hiding prover");
    const StageTimes& t = prover->plain->times();
    out_ms[0] = t.trace_commit_ms; out_ms[1] = t.quotient_commit_ms; out_ms[2] = t.open_ms;
    out_ms[3] = t.fri_commit_ms; out_ms[4] = t.grind_ms; out_ms[5] = t.query_ms;
    if (proofs) *proofs = t.proofs;
    if (reset) prover->plain->reset_times();
    return OK;
}

int p3hip_fib_prover_grind_miss_probe(p3hip_fib_prover_t* prover, uint64_t* misses, uint32_t* indices_out, size_t cap, size_t* n_out) {
    if (!prover || !misses) return fail(ERR_BAD_ARG, "fib_prover_grind_miss_probe: null argument");
    if (!prover->plain) return fail(ERR_BAD_ARG, "fib_prover_grind_miss_probe: not kept by the hiding prover");
    std::vector<uint32_t> idx;
    *misses = prover->plain->grind_misses(&idx);
    if (n_out) *n_out = idx.size();
    if (indices_out) for (size_t i = 0; i < idx.size() && i < cap; i++) indices_out[i] = idx[i];
    return OK;
}

void p3hip_fib_prover_destroy(p3hip_fib_prover_t* prover) { delete prover; }

int p3hip_fib_prover_create_hiding(int hash, unsigned log_n, const p3hip_fri_params_t* params, uint64_t seed, void* stream,
                                   int own_stream, p3hip_fib_prover_t** out) {
    return p3hip_fib_prover_create_profile(P3HIP_PROFILE_LATENCY, hash, 1, seed, log_n, params, stream, own_stream, out);
}
int p3hip_verify_fib_air_hiding(int hash, const uint8_t* proof, size_t len, uint64_t a, uint64_t b, uint64_t x, unsigned log_n,
                                const p3hip_fri_params_t* params) {
    return guarded([&]() -> int {
        if (!proof || !params) return fail(ERR_BAD_ARG, "verify_fib_air_hiding: null argument");
        FriParams fp{params->log_blowup, params->log_final_poly_len, params->num_queries, params->proof_of_work_bits};
        std::string why;
        int rc = verify_fib_air_hiding(proof, len, a, b, x, log_n, fp, &why, hash);
        if (rc != 0) set_error("fib_air verification failed: " + why);
        return rc;
    });
}

// ---- SmallRng::seed_from_u64 streams on the device (rng.hip) ----
}  // extern "C"
struct p3hip_rng {
    DevRng* d = nullptr;
    uint32_t* err = nullptr;
    DevBuf ws;
    ~p3hip_rng() { if (d) (void)hipFree(d); }
};
extern "C" {
int p3hip_rng_create(uint64_t seed, p3hip_rng_t** out) {
    return guarded([&]() -> int {
        if (!out) return fail(ERR_BAD_ARG, "rng_create: null argument");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        std::unique_ptr<p3hip_rng> r(new p3hip_rng());
        P3_HIP(hipMalloc(reinterpret_cast<void**>(&r->d), sizeof(DevRng) + 16));
        r->err = reinterpret_cast<uint32_t*>(r->d + 1);
        uint64_t s[6] = {0, 0, 0, 0, 0, 0};
        rng_seed_from_u64(s, seed);
        P3_HIP(hipMemcpy(r->d, s, sizeof(DevRng) + 16, hipMemcpyHostToDevice));
        *out = r.release();
        return OK;
    });
}
int p3hip_rng_fill_field_dev(p3hip_rng_t* rng, uint32_t* d_out, size_t n, void* stream) {
    return guarded([&]() -> int {
        if (!rng || (n && !d_out)) return fail(ERR_BAD_ARG, "rng_fill_field: null argument");
        Context* cx;
        int rc = get_context(&cx);
        if (rc) return rc;
        size_t words = 0;
        if ((rc = rng_workspace_words(n, &words))) return rc;
        if ((rc = rng->ws.reserve(words * 4))) return rc;
        return rng_fill_field(*cx, (hipStream_t)stream, rng->d, d_out, n, rng->ws.as<uint32_t>(), rng->err);
    });
}
int p3hip_rng_state(p3hip_rng_t* rng, uint64_t state_out[4], void* stream) {
    return guarded([&]() -> int {
        if (!rng || !state_out) return fail(ERR_BAD_ARG, "rng_state: null argument");
        uint64_t s[6];
        P3_HIP(hipMemcpyAsync(s, rng->d, sizeof(DevRng) + 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
        P3_HIP(hipStreamSynchronize((hipStream_t)stream));
        memcpy(state_out, s, 32);
        if ((uint32_t)s[4] != 0) return fail(ERR_INTERNAL, "rng: a fill ran out of raw draws");
        return OK;
    });
}
void p3hip_rng_destroy(p3hip_rng_t* rng) { delete rng; }

// MerkleTreeHidingMmcs::commit (fib_air.rs:40-51): every matrix is paired with a height x SALT_ELEMS matrix of draws
// from the MMCS's rng; leaf rows are m0 || s0 || m1 || s1 ...  The salt matrices live in HBM and belong to the tree.
int p3hip_mmcs_commit_hiding_dev(int hash, const uint32_t* const* d_mats, const size_t* heights, const size_t* widths, size_t n_mats,
                                 p3hip_rng_t* rng, uint32_t root_out[8], p3hip_tree_t** tree_out, void* stream) {
    return guarded([&]() -> int {
        if (!d_mats || !heights || !widths || !n_mats || !rng || !root_out || !tree_out) return fail(ERR_BAD_ARG, "mmcs_commit_hiding: null/empty argument");
        if (n_mats > 32) return fail(ERR_BAD_ARG, "mmcs_commit_hiding: at most 32 matrices per commitment");
        constexpr size_t SALT = 4;
        std::vector<void*> owned;
        auto cleanup = [&]() { for (void* p : owned) (void)hipFree(p); };
        std::vector<const uint32_t*> mp;
        std::vector<size_t> hh, ww;
        for (size_t i = 0; i < n_mats; i++) {
            void* salt = nullptr;
            if (hipMalloc(&salt, heights[i] * SALT * 4 + 16) != hipSuccess) { cleanup(); return fail(ERR_HIP, "hipMalloc failed"); }
            owned.push_back(salt);
            int rc = p3hip_rng_fill_field_dev(rng, (uint32_t*)salt, heights[i] * SALT, stream);
            if (rc) { cleanup(); return rc; }
            mp.push_back(d_mats[i]); hh.push_back(heights[i]); ww.push_back(widths[i]);
            mp.push_back((const uint32_t*)salt); hh.push_back(heights[i]); ww.push_back(SALT);
        }
        int rc = p3hip_mmcs_commit_hash_dev(hash, mp.data(), hh.data(), ww.data(), mp.size(), root_out, tree_out, stream);
        if (rc) { cleanup(); return rc; }
        (*tree_out)->t->owned = owned;
        return OK;
    });
}

int p3hip_verify_fib_air(const uint8_t* proof, size_t len, uint64_t a, uint64_t b, uint64_t x, unsigned log_n,
                         const p3hip_fri_params_t* params) {
    return p3hip_verify_fib_air_hash(HASH_POSEIDON2, proof, len, a, b, x, log_n, params);
}
int p3hip_verify_fib_air_hash(int hash, const uint8_t* proof, size_t len, uint64_t a, uint64_t b, uint64_t x, unsigned log_n,
                              const p3hip_fri_params_t* params) {
    return guarded([&]() -> int {
        if (!proof || !params) return fail(ERR_BAD_ARG, "verify_fib_air: null argument");
        FriParams fp{params->log_blowup, params->log_final_poly_len, params->num_queries, params->proof_of_work_bits};
        std::string why;
        int rc = verify_fib_air(proof, len, a, b, x, log_n, fp, &why, hash);
        if (rc != 0) set_error("fib_air verification failed: " + why);
        return rc;
    });
}

}  // extern "C"

// ---- batches of independent proofs (BASELINE configs[3]): a pool of provers, one host thread + stream each ----
struct p3hip_fib_batch {
    unsigned log_n = 0;
    int hash = HASH_POSEIDON2;
    int device = 0;  // the creator's current device: HIP's current device is per thread and defaults to 0
    FriParams fp{};
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    // Batches in submit order.  A worker takes the next unproved instance of the OLDEST batch that still has one, so a
    // prover that has finished its share of one batch starts on the next at once (no join between batches).
    struct Job {
        uint64_t ticket = 0;
        std::vector<uint64_t> a, b;
        size_t next = 0, done = 0;
        std::vector<std::vector<uint8_t>> proofs;
        int first_error = 0;
        std::string error_text;
    };
    std::deque<std::shared_ptr<Job>> jobs;   // submitted, not yet collected
    std::shared_ptr<Job> collected;          // the proofs handed out by the last collect stay alive until the next one
    uint64_t next_ticket = 1;
    bool stop = false;
    size_t ready = 0;
    int first_error = 0;  // of the workers' start-up
    std::string error_text;
    static constexpr size_t MAX_INFLIGHT = 8;

    bool hiding = false;  // provers of the reference's hiding configuration (every one seeded with `seed`, as a fresh
    uint64_t seed = 1;    // config per proof would be: fib_air.rs:50,65)
    int profile = PROFILE_THROUGHPUT;  // several provers share the chip; a pool of ONE prover is a lone prover: latency
    void worker_main() {
        if (hiding) {
            FibHidingProver prover;
            worker_loop(prover, [&](hipStream_t st) { return prover.init(log_n, fp, st, true, hash, seed, profile); });
        } else {
            FibProver prover;
            worker_loop(prover, [&](hipStream_t st) { return prover.init(log_n, fp, st, true, hash, profile); });
        }  // the prover (arena, stream) is gone before the thread's tables and scratch are freed
        release_thread_contexts();
    }
    // a std::thread whose function throws calls std::terminate: every failure becomes first_error instead
    template <class F>
    static int no_throw(F&& f) {
        try { return f(); }
        catch (const std::exception& e) { return fail(ERR_INTERNAL, std::string("exception: ") + e.what()); }
        catch (...) { return fail(ERR_INTERNAL, "unknown exception"); }
    }
    std::shared_ptr<Job> pick(size_t* index) {  // under mu
        for (auto& j : jobs)
            if (j->next < j->a.size()) { *index = j->next++; return j; }
        return nullptr;
    }
    template <class Prover, class Init>
    void worker_loop(Prover& prover, Init&& init) {
        int rc = no_throw([&]() -> int {
            if (hipSetDevice(device) != hipSuccess) return fail(ERR_HIP, "hipSetDevice failed in a batch worker");
            int r = get_context_status();
            if (r) return r;
            hipStream_t st = nullptr;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return fail(ERR_HIP, "hipStreamCreateWithFlags failed");
            return init(st);
        });
        std::string start_text;
        if (rc != OK) take_error(&start_text);
        {
            std::unique_lock<std::mutex> lk(mu);
            if (rc != OK && first_error == 0) { first_error = rc; error_text = start_text; }
            ready++;
            cv_done.notify_all();
        }
        for (;;) {
            std::shared_ptr<Job> job;
            size_t i = 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || (job = pick(&i)) != nullptr; });
                if (!job) return;  // stop
            }
            int prc = rc;
            std::string text = start_text;
            if (prc == OK) {
                prc = no_throw([&]() -> int { return prover.prove(job->a[i], job->b[i], &job->proofs[i]); });
                if (prc != OK) take_error(&text);
            }
            {
                std::unique_lock<std::mutex> lk(mu);
                if (prc != OK && job->first_error == 0) { job->first_error = prc; job->error_text = text; }
                if (++job->done == job->a.size()) cv_done.notify_all();
            }
        }
    }
    static int get_context_status() { Context* cx; return get_context(&cx); }
};

extern "C" {

int p3hip_fib_batch_create(unsigned log_n, const p3hip_fri_params_t* params, unsigned n_provers, p3hip_fib_batch_t** out) {
    return p3hip_fib_batch_create_hash(HASH_POSEIDON2, log_n, params, n_provers, out);
}
static int fib_batch_create(int hash, unsigned log_n, const p3hip_fri_params_t* params, unsigned n_provers, bool hiding,
                            uint64_t seed, p3hip_fib_batch_t** out);
int p3hip_fib_batch_create_hash(int hash, unsigned log_n, const p3hip_fri_params_t* params, unsigned n_provers,
                                p3hip_fib_batch_t** out) {
    return fib_batch_create(hash, log_n, params, n_provers, false, 1, out);
}
int p3hip_fib_batch_create_hiding(int hash, unsigned log_n, const p3hip_fri_params_t* params, uint64_t seed, unsigned n_provers,
                                  p3hip_fib_batch_t** out) {
    return fib_batch_create(hash, log_n, params, n_provers, true, seed, out);
}
static int fib_batch_create(int hash, unsigned log_n, const p3hip_fri_params_t* params, unsigned n_provers, bool hiding,
                            uint64_t seed, p3hip_fib_batch_t** out) {
    return guarded([&]() -> int {
        if (!params || !out || n_provers == 0 || n_provers > 64) return fail(ERR_BAD_ARG, "fib_batch_create: bad argument");
        if (hash != HASH_POSEIDON2 && hash != HASH_KECCAK) return fail(ERR_BAD_ARG, "fib_batch_create: unknown hash configuration");
        std::unique_ptr<p3hip_fib_batch> bt(new p3hip_fib_batch());
        bt->log_n = log_n;
        bt->hash = hash;
        bt->hiding = hiding; bt->seed = seed;
        bt->profile = n_provers > 1 ? PROFILE_THROUGHPUT : PROFILE_LATENCY;
        P3_HIP(hipGetDevice(&bt->device));
        bt->fp = FriParams{params->log_blowup, params->log_final_poly_len, params->num_queries, params->proof_of_work_bits};
        for (unsigned t = 0; t < n_provers; t++) bt->workers.emplace_back([p = bt.get()] { p->worker_main(); });
        {
            std::unique_lock<std::mutex> lk(bt->mu);
            bt->cv_done.wait(lk, [&] { return bt->ready == n_provers; });
        }
        if (bt->first_error) {
            int rc = bt->first_error;
            std::string text = bt->error_text;
            p3hip_fib_batch_destroy(bt.release());
            return fail(rc, text);
        }
        *out = bt.release();
        return OK;
    });
}

int p3hip_fib_batch_submit(p3hip_fib_batch_t* bt, size_t n, const uint64_t* a, const uint64_t* b, uint64_t* ticket_out) {
    return guarded([&]() -> int {
        if (!bt || !ticket_out || (n && (!a || !b))) return fail(ERR_BAD_ARG, "fib_batch_submit: null argument");
        auto job = std::make_shared<p3hip_fib_batch::Job>();
        job->a.assign(a, a + n);
        job->b.assign(b, b + n);
        job->proofs.resize(n);
        std::unique_lock<std::mutex> lk(bt->mu);
        if (bt->jobs.size() >= p3hip_fib_batch::MAX_INFLIGHT)
            return fail(ERR_BAD_ARG, "fib_batch_submit: too many batches in flight (collect one first)");
        job->ticket = bt->next_ticket++;
        *ticket_out = job->ticket;
        bt->jobs.push_back(job);
        bt->cv_work.notify_all();
        return OK;
    });
}

int p3hip_fib_batch_collect(p3hip_fib_batch_t* bt, uint64_t ticket, const uint8_t** proofs_out, size_t* lens_out) {
    return guarded([&]() -> int {
        if (!bt) return fail(ERR_BAD_ARG, "fib_batch_collect: null argument");
        std::shared_ptr<p3hip_fib_batch::Job> job;
        {
            std::unique_lock<std::mutex> lk(bt->mu);
            auto it = std::find_if(bt->jobs.begin(), bt->jobs.end(), [&](const std::shared_ptr<p3hip_fib_batch::Job>& j) { return j->ticket == ticket; });
            if (it == bt->jobs.end()) return fail(ERR_BAD_ARG, "fib_batch_collect: unknown ticket");
            job = *it;
            if (job->a.size() && (!proofs_out || !lens_out)) return fail(ERR_BAD_ARG, "fib_batch_collect: null argument");
            bt->cv_done.wait(lk, [&] { return job->done == job->a.size(); });
            bt->jobs.erase(std::find(bt->jobs.begin(), bt->jobs.end(), job));
            bt->collected = job;  // keeps the proof bytes alive until the next collect / prove / destroy
        }
        if (job->first_error) return fail(job->first_error, job->error_text);
        for (size_t i = 0; i < job->a.size(); i++) { proofs_out[i] = job->proofs[i].data(); lens_out[i] = job->proofs[i].size(); }
        return OK;
    });
}

int p3hip_fib_batch_prove(p3hip_fib_batch_t* bt, size_t n, const uint64_t* a, const uint64_t* b, const uint8_t** proofs_out,
                          size_t* lens_out) {
    if (!bt || (n && (!a || !b || !proofs_out || !lens_out))) return fail(ERR_BAD_ARG, "fib_batch_prove: null argument");
    if (!n) return OK;
    uint64_t ticket = 0;
    int rc = p3hip_fib_batch_submit(bt, n, a, b, &ticket);
    if (rc) return rc;
    return p3hip_fib_batch_collect(bt, ticket, proofs_out, lens_out);
}

void p3hip_fib_batch_destroy(p3hip_fib_batch_t* bt) {
    if (!bt) return;
    {
        std::unique_lock<std::mutex> lk(bt->mu);
        bt->stop = true;
        bt->cv_work.notify_all();
    }
    for (auto& t : bt->workers) t.join();
    delete bt;
}

}  // extern "C"
