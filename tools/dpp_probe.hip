// Prints what the gfx950 cross-lane primitives used by the cooperative Keccak-f actually do (lane i starts with value i):
//   hipcc --offload-arch=gfx950 -o tools/_bin/dpp_probe tools/dpp_probe.hip && tools/_bin/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
    const unsigned v = threadIdx.x, z = 1000u + threadIdx.x;
    out[0 * 64 + v] = __builtin_amdgcn_update_dpp(z, v, 0x101, 0xf, 0xf, false);  // row_shl:1, old = 1000 + lane
    out[1 * 64 + v] = __builtin_amdgcn_update_dpp(z, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    out[2 * 64 + v] = __builtin_amdgcn_update_dpp(z, v, 0x128, 0xf, 0xf, false);  // row_ror:8
    out[3 * 64 + v] = __builtin_amdgcn_update_dpp(z, v, 0x104, 0xf, 0x5, false);  // row_shl:4 bank_mask 0101
    out[4 * 64 + v] = __builtin_amdgcn_update_dpp(z, v, 0x114, 0xf, 0xa, false);  // row_shr:4 bank_mask 1010
    out[5 * 64 + v] = __builtin_amdgcn_update_dpp(z, v, 0x101, 0xf, 0xf, true);   // row_shl:1 bound_ctrl
    out[6 * 64 + v] = __builtin_amdgcn_update_dpp(z, v, 0x113, 0xf, 0xf, true);   // row_shr:3 bound_ctrl
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    out[7 * 64 + v] = r[0];
    out[8 * 64 + v] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[9 * 64 + v] = q[0];
    out[10 * 64 + v] = q[1];
}
int main() {
    unsigned* d;
    if (hipMalloc(&d, 11 * 64 * 4) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    unsigned h[11 * 64];
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char* names[11] = {"row_shl:1", "row_shr:1", "row_ror:8", "row_shl:4 bank 0101", "row_shr:4 bank 1010", "row_shl:1 bound_ctrl",
                             "row_shr:3 bound_ctrl", "permlane32_swap[0]", "permlane32_swap[1]", "permlane16_swap[0]", "permlane16_swap[1]"};
    for (int k = 0; k < 11; k++) {
        printf("%-22s:", names[k]);
        for (int i = 0; i < 64; i++) printf(" %u", h[k * 64 + i]);
        printf("\n");
    }
    return 0;
}
