// What shader clock does a ONE-workgroup kernel run at?  The one-launch prover of tiny instances (prover_tiny.hip.inc) is a chain of dependent
// instructions on one CU of an otherwise idle chip; its latency scales with the clock the power management grants such a load.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/clock_probe tools/clock_probe.hip && tools/_bin/clock_probe
// Prints shader cycles (s_memtime) per microsecond of the 100 MHz wall clock (s_memrealtime) for: a lone 64-thread kernel on an idle chip, the same
// kernel repeated back to back, and the same kernel right after / while a chip-filling kernel runs on another stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ void spin_kernel(uint64_t* out, uint32_t iters) {
    const uint64_t c0 = clock64(), w0 = wall_clock64();
    uint32_t x = threadIdx.x + 1;
    for (uint32_t i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;  // dependent chain
    const uint64_t c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = x; }
}
__global__ void busy_kernel(uint32_t* sink, uint32_t iters) {
    uint32_t x = threadIdx.x + blockIdx.x;
    for (uint32_t i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
    if (x == 0x12345678u) sink[0] = x;
}
static void report(const char* what, const uint64_t* h) {
    printf("%-70s %8.0f shader cycles / us (%.0f us)\n", what, (double)h[0] / ((double)h[1] / 100.0), (double)h[1] / 100.0);
}
int main() {
    uint64_t *d, h[3];
    uint32_t* sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 64);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const uint32_t iters = 60000;  // ~0.3-0.5 ms of dependent multiply-adds
    for (int rep = 0; rep < 3; rep++) {
        hipDeviceSynchronize();
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, d, iters);
        hipStreamSynchronize(s1);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        report(rep == 0 ? "lone 64-thread kernel, first launch on an idle chip" : "lone 64-thread kernel, again after a host round trip", h);
    }
    for (int rep = 0; rep < 20; rep++) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, d, iters);
    hipStreamSynchronize(s1);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    report("lone 64-thread kernel, the 20th of 20 back to back", h);
    hipLaunchKernelGGL(busy_kernel, dim3(256 * 8), dim3(256), 0, s2, sink, 400000);  // fills the chip for a few ms
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, d, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    report("64-thread kernel while a chip-filling kernel runs on another stream", h);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s1, d, iters);
    hipStreamSynchronize(s1);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    report("64-thread kernel right after the chip-filling kernel", h);
    return 0;
}
