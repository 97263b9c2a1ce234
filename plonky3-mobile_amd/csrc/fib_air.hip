// FibonacciAir workload pieces on the device (reference native/src/fib_air.rs:224-306).
//   generate_trace_rows (fib_air.rs:266-284): row 0 = (a, b), row i = (right_{i-1}, left_{i-1} + right_{i-1}).
// The recurrence is serial on the CPU; here each lane jumps to its chunk with a 2x2 matrix power
// (fast doubling over BabyBear) and then walks CHUNK rows, storing 8-byte rows coalesced per lane.
#include "bb31.hip.h"
#include "common.h"

namespace p3 {

constexpr uint32_t FIB_CHUNK = 16;

struct M2 { uint32_t a, b, c, d; };  // [[a b],[c d]]
__device__ __forceinline__ M2 m2mul(const M2& x, const M2& y) {
    return M2{bb::add(bb::mul(x.a, y.a), bb::mul(x.b, y.c)), bb::add(bb::mul(x.a, y.b), bb::mul(x.b, y.d)),
              bb::add(bb::mul(x.c, y.a), bb::mul(x.d, y.c)), bb::add(bb::mul(x.c, y.b), bb::mul(x.d, y.d))};
}

__global__ void fib_trace_kernel(uint32_t a0, uint32_t b0, uint64_t n, uint32_t* out) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t start = t * FIB_CHUNK;
    if (start >= n) return;
    // (left, right)_i = M^i (a0, b0),  M = [[0,1],[1,1]]
    M2 acc{bb::ONE, 0, 0, bb::ONE}, base{0, bb::ONE, bb::ONE, bb::ONE};
    for (uint64_t e = start; e; e >>= 1) {
        if (e & 1) acc = m2mul(acc, base);
        base = m2mul(base, base);
    }
    uint32_t l = bb::add(bb::mul(acc.a, a0), bb::mul(acc.b, b0));
    uint32_t r = bb::add(bb::mul(acc.c, a0), bb::mul(acc.d, b0));
    uint2* rows = reinterpret_cast<uint2*>(out);
    for (uint32_t i = 0; i < FIB_CHUNK && start + i < n; i++) {
        rows[start + i] = make_uint2(l, r);
        uint32_t nr = bb::add(l, r);
        l = r;
        r = nr;
    }
}

int fib_trace(hipStream_t stream, uint64_t a, uint64_t b, uint64_t n, uint32_t* d_out) {
    if (!n) return OK;
    if (!is_pow2(n)) return fail(ERR_BAD_ARG, "generate_trace_rows: n must be a power of two");  // fib_air.rs:267
    uint64_t threads = (n + FIB_CHUNK - 1) / FIB_CHUNK;
    uint32_t a0 = bb::to_monty((uint32_t)(a % bb::P)), b0 = bb::to_monty((uint32_t)(b % bb::P));
    hipLaunchKernelGGL(fib_trace_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, a0, b0, n, d_out);
    P3_HIP(hipGetLastError());
    return OK;
}

}  // namespace p3
