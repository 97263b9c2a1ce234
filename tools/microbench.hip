// Integer-VALU issue-rate probe for gfx950 ("measure it first, it sets the Poseidon2
// roofline").  Each kernel runs ITER x 8 independent dependency chains of one op per lane.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o gpurun_out/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../plonky3-mobile_amd/csrc/bb31.cuh"

#define ITER 4096
template <int OP>
__global__ void k(uint32_t* out, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 8 + i;
    uint32_t b = seed * 3 + 1;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = a[i] * b;                       // v_mul_lo_u32
            else if (OP == 1) a[i] = __umulhi(a[i], b);         // v_mul_hi_u32
            else if (OP == 2) { uint64_t p = (uint64_t)a[i] * b + a[i]; a[i] = (uint32_t)(p >> 32) ^ (uint32_t)p; } // mad_u64_u32
            else if (OP == 3) a[i] = __umul24(a[i], b) + 1;     // v_mul_u32_u24 / mad
            else if (OP == 4) a[i] = a[i] + b;                  // v_add_u32
            else if (OP == 5) a[i] = bb::mul(a[i] & 0x3fffffff, b & 0x3fffffff);  // full Montgomery product
            else if (OP == 6) a[i] = bb::add(a[i] & 0x3fffffff, b & 0x3fffffff);
            else if (OP == 7) a[i] = min(a[i], a[i] - b);       // v_sub + v_min
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* d) {
    int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 12345u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = 5.0 * blocks * threads * (double)ITER * 8;
    double gops = ops / (ms * 1e-3) / 1e9;
    // per-SIMD wave-instruction cost in cycles assuming 2.4 GHz, 1024 SIMDs, 64 lanes
    double cyc = 2.4e9 * 1024 * 64 / (gops * 1e9);
    printf("%-28s %9.1f Gop/s  ~%.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, gops, cyc);
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<4>("v_add_u32", d);
    run<7>("sub+min (2 ops)", d);
    run<0>("v_mul_lo_u32", d);
    run<1>("v_mul_hi_u32", d);
    run<2>("mad_u64_u32 (+xor)", d);
    run<3>("mul_u24 (+add)", d);
    run<6>("bb::add (and,and,add,sub,min)", d);
    run<5>("bb::mul (monty product)", d);
    return 0;
}
