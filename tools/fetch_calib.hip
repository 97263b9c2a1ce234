// FETCH_SIZE calibration for the access shape of the wide LDE's third kernel (narrow_fwd2_kernel<8, 5, 1> on 2^17 x 2633 words):
// 4 bytes per lane, 32 lanes on 128 contiguous bytes of one matrix row, sixteen rows per wave-instruction pair, rows 10532 bytes
// long (not a multiple of 128).  Kernels with KNOWN byte counts, to be run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE :
//   stream16      every lane 16 contiguous bytes, whole matrix          (the guide's calibrated case: FETCH_SIZE x 2 = bytes)
//   tiles4_w2633  K3's tile order, 4 B per lane, W = 2633               (the subject)
//   tiles4_w2688  the same with rows that ARE multiples of 128 bytes    (separates "4-byte lanes" from "misaligned rows")
//   tiles16_w2688 128-byte row segments read as 8 lanes x 16 B
// build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/_bin/fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void __launch_bounds__(256) stream16(const uint4* p, size_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
// tile = 256 consecutive rows x 32 consecutive words; block (512 threads): lane q = word of the 128-byte segment, t = tid >> 5, rows t + 16 j
__global__ void __launch_bounds__(512) tiles4(const uint32_t* p, uint32_t W, uint32_t tiles_per_rowblock, uint32_t* sink) {
    const uint32_t q = threadIdx.x & 31u, t = threadIdx.x >> 5;
    const uint32_t g0 = blockIdx.x / tiles_per_rowblock, s = blockIdx.x % tiles_per_rowblock;
    const uint32_t* base = p + ((size_t)g0 * 256 + t) * W + s * 32 + q;
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) acc ^= base[(size_t)j * 16 * W];
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(128) tiles16(const uint32_t* p, uint32_t W, uint32_t tiles_per_rowblock, uint32_t* sink) {
    const uint32_t q = threadIdx.x & 7u, t = threadIdx.x >> 3;  // 8 lanes x 16 B = one 128-byte segment, 16 rows per pass
    const uint32_t g0 = blockIdx.x / tiles_per_rowblock, s = blockIdx.x % tiles_per_rowblock;
    const uint32_t* base = p + ((size_t)g0 * 256 + t) * W + s * 32 + q * 4;
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; j++) { const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)j * 16 * W); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const uint32_t rows = 1u << 17;
    const size_t words = (size_t)rows * 2688;
    uint32_t *d, *sink;
    if (hipMalloc(&d, words * 4) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, words * 4);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        stream16<<<256 * 8, 256>>>(reinterpret_cast<const uint4*>(d), (size_t)rows * 2633 / 4, sink);
        tiles4<<<(rows / 256) * 82, 512>>>(d, 2633, 82, sink);
        tiles4<<<(rows / 256) * 84, 512>>>(d, 2688, 84, sink);
        tiles16<<<(rows / 256) * 84, 128>>>(d, 2688, 84, sink);
        hipDeviceSynchronize();
    }
    printf("bytes: stream16 %zu  tiles4_w2633 %zu  tiles4_w2688 %zu  tiles16_w2688 %zu\n", (size_t)rows * 2633 / 4 * 16, (size_t)rows * 82 * 32 * 4,
           (size_t)rows * 84 * 32 * 4, (size_t)rows * 84 * 32 * 4);
    return 0;
}
