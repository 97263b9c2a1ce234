// Montgomery product forms on gfx950, timed in a butterfly-like loop (four independent chains per lane):
//   A: v_mul_lo + v_mul_hi + v_mul_lo + v_mul_hi + v_sub + v_add + v_min (7; bb31.hip.h until round 3)
//   B: v_mad_u64_u32 + v_mul_lo + v_mad_u64_u32 + v_add + v_min (5)
// build: hipcc -O3 --offload-arch=gfx950 tools/mulform_bench.hip -o tools/_bin/mulform_bench ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
constexpr uint32_t P = 0x78000001u, MU = 0x88000001u, NMU = 0u - MU;
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t mul_a(uint32_t a, uint32_t b) {
    uint32_t lo = a * b, hi = __umulhi(a, b);
    uint32_t t = lo * MU;
    uint32_t u = __umulhi(t, P);
    uint32_t r = hi - u;
    return umin32(r, r + P);
}
__device__ __forceinline__ uint32_t mul_b(uint32_t a, uint32_t b) {
    uint64_t x = (uint64_t)a * b;
    uint32_t t = (uint32_t)x * NMU;
    uint64_t y = x + (uint64_t)t * P;
    uint32_t r = (uint32_t)(y >> 32);
    return umin32(r, r - P);
}
__device__ __forceinline__ uint32_t addm(uint32_t a, uint32_t b) { uint32_t s = a + b; return umin32(s, s - P); }
__device__ __forceinline__ uint32_t subm(uint32_t a, uint32_t b) { uint32_t d = a - b; return umin32(d, d + P); }
template <int FORM>
__global__ void __launch_bounds__(256) bfly(uint32_t* p, int iters) {
    uint32_t v[8], w[4];
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 12;
    for (int k = 0; k < 8; k++) v[k] = p[i0 + k];
    for (int k = 0; k < 4; k++) w[k] = p[i0 + 8 + k];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t t = FORM ? mul_b(v[k + 4], w[k]) : mul_a(v[k + 4], w[k]);
            const uint32_t a = v[k];
            v[k] = addm(a, t);
            v[k + 4] = subm(a, t);
        }
        // rotate so that the chains mix like stages do
        const uint32_t t0 = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4]; v[4] = t0;
    }
    uint32_t s = 0;
    for (int k = 0; k < 8; k++) s ^= v[k];
    p[i0] = s;
}
int main() {
    const int blocks = 256 * 8 * 4, iters = 2000;
    uint32_t* d;
    const size_t n = (size_t)blocks * 256 * 12;
    hipMalloc(&d, n * 4);
    uint32_t* h = (uint32_t*)malloc(n * 4);
    for (size_t i = 0; i < n; i++) h[i] = (uint32_t)((i * 2654435761u) % P);
    uint32_t out[2];
    for (int form = 0; form < 2; form++) {
        hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        if (form) bfly<1><<<blocks, 256>>>(d, 10); else bfly<0><<<blocks, 256>>>(d, 10);
        hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
        hipEventRecord(a);
        if (form) bfly<1><<<blocks, 256>>>(d, iters); else bfly<0><<<blocks, 256>>>(d, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(&out[form], d, 4, hipMemcpyDeviceToHost);
        const double bf = (double)blocks * 256 * 4 * iters;
        printf("form %c: %.3f ms, %.2f G butterflies/s, checksum %08x\n", form ? 'B' : 'A', ms, bf / ms / 1e6, out[form]);
    }
    printf(out[0] == out[1] ? "results equal\n" : "RESULTS DIFFER\n");
    return out[0] != out[1];
}
