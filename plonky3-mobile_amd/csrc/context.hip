// Per-(thread, device) context: twiddle / scale tables (built by a kernel at first use, cached in HBM) and scratch.
// Mirrors the reference's thread-local cached Vulkan runtime (native/src/backend_vulkan.rs:100-124) and
// its "rebuild the twiddle table only when log_n changes" policy (:1082,1088-1098).
//
// Contract of the `_dev` entry points (include/p3hip.h): enqueue on the caller's stream and return.  Hence
//   - tables are filled by a kernel on that stream; a table built on one stream and fetched from another is
//     guarded by an event (hipStreamWaitEvent) until it is known complete;
//   - scratch slabs are keyed by (stream, slot): two transforms in flight on two streams of one thread never share
//     an intermediate;
//   - the bounded scale-table cache is only ever emptied at the top of an entry point (reserve_scale_slots), never
//     between two fetches of one call.
#include "bb31.hip.h"
#include "common.h"

#include <memory>
#include <mutex>

namespace p3 {

static thread_local std::string g_last_error;
static thread_local bool g_has_error = false;

void set_error(const std::string& msg) {
    g_last_error = msg;
    g_has_error = true;
}
int fail(int code, const std::string& msg) {
    set_error(msg);
    return code;
}
// used by c_api.hip
bool take_error(std::string* out) {
    if (!g_has_error) return false;
    *out = g_last_error;
    g_has_error = false;
    g_last_error.clear();
    return true;
}

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return OK;
    if (ptr) {
        // the old slab may still be referenced by enqueued work
        P3_HIP(hipDeviceSynchronize());
        P3_HIP(hipFree(ptr));
        ptr = nullptr;
        cap = 0;
    }
    size_t want = bytes + (bytes >> 3);
    P3_HIP(hipMalloc(&ptr, want));
    cap = want;
    return OK;
}
DevBuf::~DevBuf() {
    if (ptr) (void)hipFree(ptr);
}

static int upload(const std::vector<uint32_t>& host, uint32_t** dev) {
    P3_HIP(hipMalloc(reinterpret_cast<void**>(dev), host.size() * 4));
    P3_HIP(hipMemcpy(*dev, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    return OK;
}

int Context::init(int dev) {
    if (device >= 0) return OK;
    // Stage tables in the reference's layout (backend_vulkan.rs:977-996): stage k at offset 2^k - 1 holds
    // w_{2^(k+1)}^e, e < 2^k.  One table of 12 stages serves every tile size (prefix property).  Built once per
    // context on the host: the first call of a thread on a device is its (blocking) initialisation.
    for (int invs = 0; invs < 2; invs++) {
        std::vector<uint32_t> t(1u << 12, 0);
        for (uint32_t k = 0; k < 12; k++) {
            uint32_t root = bb::two_adic_generator(k + 1);
            if (invs) root = bb::inv(root);
            uint32_t acc = bb::ONE;
            for (uint32_t e = 0; e < (1u << k); e++) {
                t[(1u << k) - 1 + e] = acc;
                acc = bb::mul(acc, root);
            }
        }
        int rc = upload(t, &tile_tw[invs]);
        if (rc) return rc;
        std::vector<double2> td(1u << 12, double2{0.0, 0.0});
        for (uint32_t i = 0; i + 1 < (1u << 12); i++) {
            const double w = (double)bb::from_monty(t[i]);
            td[i] = double2{w, w / 2013265921.0};
        }
        P3_HIP(hipMalloc(reinterpret_cast<void**>(&tile_twd[invs]), td.size() * sizeof(double2)));
        P3_HIP(hipMemcpy(tile_twd[invs], td.data(), td.size() * sizeof(double2), hipMemcpyHostToDevice));
    }
    device = dev;
    return OK;
}

DevBuf& Context::ws(hipStream_t stream, int slot) { return scratch[std::make_pair(stream, slot)]; }

// lo[i] = base^i (i < 2^T), hi[j] = mult * base^(j 2^T) (j < 2^hbits): log-time powers, one entry per lane
__global__ void two_level_build_kernel(uint32_t* lo, uint32_t* hi, uint32_t T, uint32_t hbits, uint32_t base, uint32_t mult) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nlo = 1u << T, nhi = 1u << hbits;
    if (i < nlo) lo[i] = bb::pow(base, i);
    else if (i < nlo + nhi) hi[i - nlo] = bb::mul(mult, bb::pow(base, (uint64_t)(i - nlo) << T));
}

int Context::mark_built(hipStream_t stream, CachedTable& e) {
    if (!e.ready) P3_HIP(hipEventCreateWithFlags(&e.ready, hipEventDisableTiming));
    P3_HIP(hipEventRecord(e.ready, stream));
    e.built_on = stream;
    e.complete = false;
    return OK;
}
int Context::wait_ready(hipStream_t stream, CachedTable& e) {
    if (e.complete || stream == e.built_on) return OK;  // same stream: ordered behind the build
    if (hipEventQuery(e.ready) == hipSuccess) { e.complete = true; return OK; }
    (void)hipGetLastError();  // hipErrorNotReady is not an error here
    P3_HIP(hipStreamWaitEvent(stream, e.ready, 0));
    return OK;
}

static int build_two_level(Context& cx, hipStream_t stream, uint32_t base, uint32_t log_n, uint32_t mult, CachedTable* e) {
    const uint32_t T = (log_n + 1) / 2, hbits = log_n - T;
    uint32_t* block = nullptr;
    P3_HIP(hipMalloc(reinterpret_cast<void**>(&block), (((size_t)1 << T) + ((size_t)1 << hbits)) * 4));
    e->t.lo = block;  // one allocation: lo, then hi
    e->t.hi = block + ((size_t)1 << T);
    e->t.T = T;
    const uint32_t total = (1u << T) + (1u << hbits);
    hipLaunchKernelGGL(two_level_build_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, e->t.lo, e->t.hi, T, hbits, base, mult);
    P3_HIP(hipGetLastError());
    return cx.mark_built(stream, *e);
}

static void free_table(CachedTable& e) {
    if (e.t.lo) (void)hipFree(e.t.lo);
    if (e.ready) (void)hipEventDestroy(e.ready);
    e = CachedTable{};
}

int Context::get_root_table(hipStream_t stream, uint32_t q, bool inverse, TwoLevelTable* out) {
    auto key = std::make_pair(q, inverse ? 1 : 0);
    auto it = root_tables.find(key);
    if (it == root_tables.end()) {
        uint32_t root = bb::two_adic_generator(q);
        if (inverse) root = bb::inv(root);
        CachedTable e;
        int rc = build_two_level(*this, stream, root, q, bb::ONE, &e);
        if (rc) return rc;
        it = root_tables.emplace(key, e).first;
    }
    int rc = wait_ready(stream, it->second);
    if (rc) return rc;
    *out = it->second.t;
    return OK;
}

constexpr size_t SCALE_TABLE_CAP = 256;  // entries; a table is at most 2 * 2^14 words

int Context::reserve_scale_slots(size_t k) {
    if (scale_tables.size() + k <= SCALE_TABLE_CAP) return OK;
    P3_HIP(hipDeviceSynchronize());  // the tables may be in use by enqueued work (rare: > 256 distinct keys)
    for (auto& kv : scale_tables) free_table(kv.second);
    scale_tables.clear();
    return OK;
}

int Context::get_scale_table(hipStream_t stream, uint32_t base, uint32_t log_n, uint32_t mult, TwoLevelTable* out) {
    auto key = std::make_tuple(base, log_n, mult);
    auto it = scale_tables.find(key);
    if (it == scale_tables.end()) {
        // no eviction here: the caller may hold pointers to other entries (reserve_scale_slots ran at its top)
        CachedTable e;
        int rc = build_two_level(*this, stream, base, log_n, mult, &e);
        if (rc) return rc;
        it = scale_tables.emplace(key, e).first;
    }
    int rc = wait_ready(stream, it->second);
    if (rc) return rc;
    *out = it->second.t;
    return OK;
}

int Context::ensure_dynamic_lds(const void* kernel, int bytes) {
    if (attr_done.count(kernel)) return OK;
    P3_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_done.insert(kernel);
    return OK;
}

Context::~Context() {
    for (auto& kv : root_tables) free_table(kv.second);
    for (auto& kv : scale_tables) free_table(kv.second);
    for (auto& kv : selector_tables) free_table(kv.second);
    for (int i = 0; i < 2; i++) if (tile_tw[i]) (void)hipFree(tile_tw[i]);
    for (int i = 0; i < 2; i++) if (tile_twd[i]) (void)hipFree(tile_twd[i]);
    if (rng_jump) (void)hipFree(rng_jump);
}

// One context per device the thread has used.  Leaked at thread exit on purpose (HIP may already be torn down when
// thread-local destructors run); threads that know they are done call release_thread_contexts().
static std::map<int, Context*>*& thread_contexts() {
    static thread_local std::map<int, Context*>* m = nullptr;
    return m;
}

int get_context(Context** out) {
    int count = 0;
    P3_HIP(hipGetDeviceCount(&count));
    if (count <= 0) return fail(ERR_HIP, "no HIP device visible");
    int dev = 0;
    P3_HIP(hipGetDevice(&dev));
    auto*& m = thread_contexts();
    if (!m) m = new std::map<int, Context*>();
    Context*& cx = (*m)[dev];
    if (!cx) cx = new Context();
    int rc = cx->init(dev);
    if (rc) return rc;
    *out = cx;
    return OK;
}

void release_thread_contexts() {
    auto*& m = thread_contexts();
    if (!m) return;
    int cur = 0;
    bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (auto& kv : *m) {
        if (hipSetDevice(kv.first) != hipSuccess) continue;
        (void)hipDeviceSynchronize();  // enqueued work may still use the tables / slabs
        delete kv.second;
    }
    if (have_cur) (void)hipSetDevice(cur);
    delete m;
    m = nullptr;
}

}  // namespace p3
