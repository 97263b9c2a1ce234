// Per-thread device context: twiddle / scale tables (built on the host once, cached in HBM) and scratch.
// Mirrors the reference's thread-local cached Vulkan runtime (native/src/backend_vulkan.rs:100-124) and
// its "rebuild the twiddle table only when log_n changes" policy (:1082,1088-1098).
#include "bb31.cuh"
#include "common.h"

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

int Context::init() {
    if (device >= 0) return OK;
    int count = 0;
    P3_HIP(hipGetDeviceCount(&count));
    if (count <= 0) return fail(ERR_HIP, "no HIP device visible");
    int dev = 0;
    P3_HIP(hipGetDevice(&dev));
    // Stage tables in the reference's layout (backend_vulkan.rs:977-996): stage k at offset 2^k - 1 holds
    // w_{2^(k+1)}^e, e < 2^k.  One table of 12 stages serves every tile size (prefix property).
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
    }
    device = dev;
    return OK;
}

static int build_two_level(uint32_t base, uint32_t log_n, uint32_t mult, TwoLevelTable* out) {
    uint32_t T = (log_n + 1) / 2;
    std::vector<uint32_t> lo(1u << T), hi(1u << (log_n - T));
    uint32_t acc = bb::ONE;
    for (auto& v : lo) { v = acc; acc = bb::mul(acc, base); }
    uint32_t step = acc;  // base^(2^T)
    acc = mult;
    for (auto& v : hi) { v = acc; acc = bb::mul(acc, step); }
    out->T = T;
    int rc = upload(lo, &out->lo);
    if (rc) return rc;
    return upload(hi, &out->hi);
}

int Context::get_root_table(uint32_t q, bool inverse, TwoLevelTable* out) {
    auto key = std::make_pair(q, inverse ? 1 : 0);
    auto it = root_tables.find(key);
    if (it != root_tables.end()) { *out = it->second; return OK; }
    uint32_t root = bb::two_adic_generator(q);
    if (inverse) root = bb::inv(root);
    TwoLevelTable t;
    int rc = build_two_level(root, q, bb::ONE, &t);
    if (rc) return rc;
    root_tables[key] = t;
    *out = t;
    return OK;
}

int Context::get_scale_table(uint32_t base, uint32_t log_n, uint32_t mult, TwoLevelTable* out) {
    auto key = std::make_tuple(base, log_n, mult);
    auto it = scale_tables.find(key);
    if (it != scale_tables.end()) { *out = it->second; return OK; }
    if (scale_tables.size() > 64) {  // bounded cache; tables may be in use by enqueued work
        P3_HIP(hipDeviceSynchronize());
        for (auto& kv : scale_tables) { (void)hipFree(kv.second.lo); (void)hipFree(kv.second.hi); }
        scale_tables.clear();
    }
    TwoLevelTable t;
    int rc = build_two_level(base, log_n, mult, &t);
    if (rc) return rc;
    scale_tables[key] = t;
    *out = t;
    return OK;
}

Context::~Context() {
    for (auto& kv : root_tables) { (void)hipFree(kv.second.lo); (void)hipFree(kv.second.hi); }
    for (auto& kv : scale_tables) { (void)hipFree(kv.second.lo); (void)hipFree(kv.second.hi); }
    for (int i = 0; i < 2; i++) if (tile_tw[i]) (void)hipFree(tile_tw[i]);
}

int get_context(Context** out) {
    static thread_local Context* cx = nullptr;  // leaked at thread exit on purpose: HIP may already be torn down
    if (!cx) cx = new Context();
    int rc = cx->init();
    if (rc) return rc;
    *out = cx;
    return OK;
}

}  // namespace p3
