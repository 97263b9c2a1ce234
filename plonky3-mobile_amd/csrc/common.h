// Shared host-side plumbing for libp3hip: error mailbox, per-thread device context, cached tables.
// Conventions carried over from the reference: errors are values (status code +
// take-and-clear message, native/src/gpu_dft.rs:42,65-68), device state is per calling thread
// (native/src/backend_vulkan.rs:100-124), nothing aborts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>

namespace p3 {

// First statement of every LATENCY-bound kernel (single-wave transcript steps, lane-cooperative tree levels, the small
// FRI folds, query gathers): raise the wave's issue priority.  When several provers share the chip such a kernel's
// few waves sit on SIMDs beside up to eight waves of another prover's hash layer, and with equal priority the arbiter
// gives them an eighth of the issue slots: its dependent instruction chain then runs 2-3x slower (tree_levels_coop:
// 20 us alone, 51 us under four provers) while the large kernel would not notice the difference.
#ifndef P3_NO_SETPRIO
#define P3_LATENCY_BOUND_KERNEL() __builtin_amdgcn_s_setprio(3)
#else
#define P3_LATENCY_BOUND_KERNEL() ((void)0)
#endif

// Chosen when an object is CREATED (as the reference picks its backend at construction, native/src/gpu_dft.rs:85-92), never read
// from the environment.  LATENCY: one proof at a time is what the reference does (MainActivity.kt:29-33, fib_air.rs:56-72) — forms
// that shorten a lone proof's chain of dependent launches at the price of extra lane-instructions (quad Poseidon2 layers,
// cooperative Keccak up to 2^12 digests, the randomization commitment on a second side stream, the single-workgroup FRI tail).
// THROUGHPUT: several provers share the chip and VALU issue is what is short — the per-lane forms.  Results never differ.
enum Profile : int { PROFILE_THROUGHPUT = 0, PROFILE_LATENCY = 1 };

enum Status : int {
    OK = 0,
    ERR_BAD_ARG = -1,   // null pointer / non power-of-two height / bad width
    ERR_HIP = -2,       // HIP runtime failure (message in take_last_error)
    ERR_BACKEND = -3,   // unknown backend name, or hip backend not selected
    ERR_INTERNAL = -4,
};

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define P3_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess)                                                             \
            return p3::fail(p3::ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// Growable device scratch buffer (never shrinks; freed with the context).
struct DevBuf {
    void* ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    template <class T>
    T* as() const { return reinterpret_cast<T*>(ptr); }
    ~DevBuf();
};

struct TwoLevelTable {  // value(e) = lo[e & (2^T - 1)] * hi[e >> T]
    uint32_t* lo = nullptr;
    uint32_t* hi = nullptr;
    uint32_t T = 0;
};

struct CachedTable {  // a table built by a kernel on `built_on`; other streams wait for `ready` until it has completed
    TwoLevelTable t;
    hipEvent_t ready = nullptr;
    hipStream_t built_on = nullptr;
    bool complete = false;
};

// Device state of ONE (host thread, device) pair: cached tables and scratch.  Scratch slabs are keyed by the stream
// the work is enqueued on, so calls issued from one thread on different streams never share an intermediate.
struct Context {
    int device = -1;
    int profile = PROFILE_LATENCY;  // of the free functions called on this thread (p3hip_set_thread_profile); objects carry their own
    uint32_t* tile_tw[2] = {nullptr, nullptr};  // [inverse]: reference-layout stage tables, 2^12-1 words
    double2* tile_twd[2] = {nullptr, nullptr};  // the same tables as {w, w / P} doubles of canonical values (fp64 kernels)
    std::map<std::pair<uint32_t, int>, CachedTable> root_tables;                   // (q, inverse) -> w_{2^q}^e
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, CachedTable> scale_tables;  // (base, log_n, mult)
    std::map<std::pair<hipStream_t, int>, DevBuf> scratch;                         // (stream, slot)
    std::set<const void*> attr_done;  // kernels whose dynamic-LDS limit has been raised on this device
    uint64_t* rng_jump = nullptr;  // rng.hip: GF(2) jump matrices of the xoshiro256++ state transition
    std::map<uint32_t, CachedTable> selector_tables;  // prover.hip: log_n -> selectors on the quotient coset (t.lo)
    int init(int dev);
    // scratch slab `slot` of the calling thread for work enqueued on `stream`
    DevBuf& ws(hipStream_t stream, int slot);
    // enqueue-only: the table is built by a kernel on `stream` at first use and cached
    int get_root_table(hipStream_t stream, uint32_t q, bool inverse, TwoLevelTable* out);
    // value(j) = mult * base^j for j < 2^log_n
    int get_scale_table(hipStream_t stream, uint32_t base, uint32_t log_n, uint32_t mult, TwoLevelTable* out);
    // Call at the TOP of an entry point that will fetch up to `k` scale tables: if the bounded cache cannot take
    // them it is emptied HERE (device synchronised first), never between two fetches of one call.
    int reserve_scale_slots(size_t k);
    // records `ready` after the kernel that fills a freshly built entry / makes `stream` wait for an entry built elsewhere
    int mark_built(hipStream_t stream, CachedTable& e);
    int wait_ready(hipStream_t stream, CachedTable& e);
    // hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel on this context's device
    int ensure_dynamic_lds(const void* kernel, int bytes);
    ~Context();
};

// Context of the calling thread for the CURRENT device (hipGetDevice), created on first use.
int get_context(Context** out);
// Frees every context of the calling thread (tables, scratch).  Worker threads call it before they exit.
void release_thread_contexts();

// ---- ntt.hip ----
// All pointers are device pointers; launches are enqueued on `stream` and not synchronised.
// dft: natural order in, natural order out (TwoAdicSubgroupDft::dft_batch).  inverse=true gives idft_batch.
int ntt_dft(Context& cx, hipStream_t stream, const uint32_t* src, uint32_t* dst, uint64_t height,
            uint32_t width, bool inverse);
std::vector<uint32_t> ntt_dft_plan(uint32_t n);  // stages per pass (= per launch) of dft / idft over 2^n rows; host-only
// coset_lde: src = evaluations over the order-`height` subgroup (natural order), dst =
// (height << added_bits) x width evaluations over shift*<g>, natural or bit-reversed row order.
int ntt_coset_lde(Context& cx, hipStream_t stream, const uint32_t* src, uint32_t* dst, uint64_t height,
                  uint32_t width, uint32_t added_bits, uint32_t shift_monty, bool bit_reversed_out);
// The same LDE when the caller already holds the columns' COEFFICIENTS (natural order, `height` rows = degree bound):
// evaluations over shift*<g_{height << added_bits}> in bit-reversed row order, without the inverse transform.  `scratch`
// (height x width words) is used only by shapes outside the narrow plan.
int ntt_coset_lde_from_coeffs(Context& cx, hipStream_t stream, const uint32_t* coeffs, uint32_t* dst, uint32_t* scratch,
                              uint64_t height, uint32_t width, uint32_t added_bits, uint32_t shift_monty);
// coset_dft: coefficients (natural order) -> evaluations over shift*<g> (natural order).
int ntt_coset_dft(Context& cx, hipStream_t stream, const uint32_t* src, uint32_t* dst, uint64_t height,
                  uint32_t width, uint32_t shift_monty);
int bit_reverse_rows(hipStream_t stream, const uint32_t* src, uint32_t* dst, uint64_t height, uint32_t width);

// ---- fib_air.hip ----
int fib_trace(hipStream_t stream, uint64_t a, uint64_t b, uint64_t n, uint32_t* d_out);

inline bool is_pow2(uint64_t v) { return v && !(v & (v - 1)); }
inline uint32_t log2u(uint64_t v) {
    uint32_t l = 0;
    while ((1ull << l) < v) l++;
    return l;
}

}  // namespace p3
