// Effective shader clock under sustained VALU load (the Poseidon2 ceiling in DESIGN.md is quoted at 2.4 GHz):
// every wave runs a long dependent chain of v_fma_f64 (or v_mul_lo_u32) and reads s_memtime (shader cycles) and
// s_memrealtime (constant 100 MHz) before and after.   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/_bin/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void burn(double* out, unsigned long long* stamps, int iters) {
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c = 1e-9;
    double a2 = a + 1, a3 = a + 2, a4 = a + 3;
    unsigned x = threadIdx.x * 2654435761u + 1, x2 = x + 7, x3 = x + 11, x4 = x + 13;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { a = __fma_rn(a, b, c); a2 = __fma_rn(a2, b, c); a3 = __fma_rn(a3, b, c); a4 = __fma_rn(a4, b, c); }
        else { x = x * 2654435761u + 1; x2 = x2 * 2246822519u + 3; x3 = x3 * 3266489917u + 5; x4 = x4 * 668265263u + 7; }
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + a2 + a3 + a4 + (double)(x ^ x2 ^ x3 ^ x4);
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 200000;  // 8 waves per SIMD on every CU
    double* out; unsigned long long* st;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&st, sizeof(unsigned long long) * 2 * blocks);
    std::vector<unsigned long long> h(2 * blocks);
    for (int kind = 0; kind < 2; kind++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(burn<0>, dim3(blocks), dim3(threads), 0, 0, out, st, iters);
            else hipLaunchKernelGGL(burn<1>, dim3(blocks), dim3(threads), 0, 0, out, st, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
            double cyc = 0, wall = 0;
            for (int i = 0; i < blocks; i++) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
            // wall_clock64 ticks at 100 MHz
            double mhz = cyc / wall * 100.0;
            double instr = 4.0 * iters * (blocks * (threads / 64));  // wave-instructions
            printf("%s: kernel %.2f ms, shader clock %.0f MHz (clock64/wall_clock64), %.2f cycles per wave-instruction per SIMD at that clock, %.2f at 2400 MHz\n",
                   kind == 0 ? "v_fma_f64 x4 chains" : "v_mul_lo_u32 x4 chains", ms, mhz,
                   ms * 1e-3 * mhz * 1e6 * 1024 / instr, ms * 1e-3 * 2400e6 * 1024 / instr);
        }
    }
    return 0;
}
