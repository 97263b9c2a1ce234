// Instruction-issue microbenchmark for gfx950: N independent instructions of one opcode per loop iteration,
// written in inline asm so the compiler cannot fold them.  Reports cycles per wave-instruction per SIMD
// from s_memtime (shader clock), so DVFS does not distort the figure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2000
#define COMMA ,
#define REP8(X) X X X X X X X X

#define KERNEL(NAME, ASM, CLOB)                                                             \
    __global__ void NAME(uint64_t* out, uint32_t seed) {                                    \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed | 1; \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3, e = 1.000001;                             \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                          \
        for (int it = 0; it < ITERS; it++) {                                                 \
            REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(e) : CLOB);) \
        }                                                                                    \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                          \
        if (a0 + a1 + a2 + a3 + (uint32_t)(d0 + d1 + d2 + d3) == 0x12345) out[1] = 1;       \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                          \
    }

// each ASM string = 4 independent instructions
KERNEL(k_add, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8", "memory")
KERNEL(k_sub, "v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8", "memory")
KERNEL(k_min, "v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8", "memory")
KERNEL(k_and, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8", "memory")
KERNEL(k_add3, "v_add3_u32 %0, %0, %8, %1\n v_add3_u32 %1, %1, %8, %2\n v_add3_u32 %2, %2, %8, %3\n v_add3_u32 %3, %3, %8, %0", "memory")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8", "memory")
KERNEL(k_mulhi, "v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8", "memory")
KERNEL(k_mulhii, "v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %8\n v_mul_hi_i32 %3, %3, %8", "memory")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8", "memory")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %0", "memory")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3", "memory")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 3, %8\n v_lshl_add_u32 %1, %1, 3, %8\n v_lshl_add_u32 %2, %2, 3, %8\n v_lshl_add_u32 %3, %3, 3, %8", "memory")
KERNEL(k_fmaf32, "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8", "memory")
KERNEL(k_addf64, "v_add_f64 %4, %4, %9\n v_add_f64 %5, %5, %9\n v_add_f64 %6, %6, %9\n v_add_f64 %7, %7, %9", "memory")
KERNEL(k_mulf64, "v_mul_f64 %4, %4, %9\n v_mul_f64 %5, %5, %9\n v_mul_f64 %6, %6, %9\n v_mul_f64 %7, %7, %9", "memory")
KERNEL(k_fmaf64, "v_fma_f64 %4, %4, %9, %9\n v_fma_f64 %5, %5, %9, %9\n v_fma_f64 %6, %6, %9, %9\n v_fma_f64 %7, %7, %9, %9", "memory")
KERNEL(k_rndf64, "v_rndne_f64 %4, %4\n v_rndne_f64 %5, %5\n v_rndne_f64 %6, %6\n v_rndne_f64 %7, %7", "memory")
KERNEL(k_cvtf64u, "v_cvt_f64_u32 %4, %0\n v_cvt_f64_u32 %5, %1\n v_cvt_f64_u32 %6, %2\n v_cvt_f64_u32 %7, %3", "memory")
KERNEL(k_cvtuf64, "v_cvt_u32_f64 %0, %4\n v_cvt_u32_f64 %1, %5\n v_cvt_u32_f64 %2, %6\n v_cvt_u32_f64 %3, %7", "memory")
KERNEL(k_pkaddf32, "v_pk_add_f32 %4, %4, %9\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %9\n v_pk_add_f32 %7, %7, %9", "memory")
KERNEL(k_mad64, "v_mad_u64_u32 %4, vcc, %0, %8, %4\n v_mad_u64_u32 %5, vcc, %1, %8, %5\n v_mad_u64_u32 %6, vcc, %2, %8, %6\n v_mad_u64_u32 %7, vcc, %3, %8, %7", "memory" COMMA "vcc")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_ror:4 row_mask:0xf bank_mask:0xf", "memory")

KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc", "memory")
KERNEL(k_subco, "v_sub_co_u32 %0, vcc, %0, %8\n v_sub_co_u32 %1, vcc, %1, %8\n v_sub_co_u32 %2, vcc, %2, %8\n v_sub_co_u32 %3, vcc, %3, %8", "memory" COMMA "vcc")
KERNEL(k_addco, "v_add_co_u32 %0, vcc, %0, %8\n v_add_co_u32 %1, vcc, %1, %8\n v_add_co_u32 %2, vcc, %2, %8\n v_add_co_u32 %3, vcc, %3, %8", "memory" COMMA "vcc")
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8", "memory" COMMA "vcc")
KERNEL(k_max, "v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8", "memory")
KERNEL(k_xor, "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8", "memory")
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8", "memory")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 3, %0\n v_ashrrev_i32 %1, 3, %1\n v_ashrrev_i32 %2, 3, %2\n v_ashrrev_i32 %3, 3, %3", "memory")
KERNEL(k_subrev, "v_subrev_u32 %0, %8, %0\n v_subrev_u32 %1, %8, %1\n v_subrev_u32 %2, %8, %2\n v_subrev_u32 %3, %8, %3", "memory")
KERNEL(k_mini, "v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8", "memory")
KERNEL(k_addlit, "v_add_u32 %0, 0x87ffffff, %0\n v_add_u32 %1, 0x87ffffff, %1\n v_add_u32 %2, 0x87ffffff, %2\n v_add_u32 %3, 0x87ffffff, %3", "memory")

template <class K>
void run(const char* name, K kern, uint64_t* d, int waves_per_simd) {
    // blocks of 256 threads (4 waves = 1 per SIMD); waves_per_simd blocks per CU on all 256 CUs.  Wall clock
    // from HIP events around 5 launches: wave-instructions / time, expressed per SIMD at the nominal 2.4 GHz.
    int blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 777u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instrs = 5.0 * blocks * 4.0 * ITERS * 32.0;
    double per_simd_per_s = wave_instrs / (ms * 1e-3) / 1024.0;
    printf("%-14s waves/SIMD=%d: %7.1f G wave-instr/s chip-wide, %.2f cycles/wave-instr/SIMD at 2.4 GHz, %.1f T lane-ops/s\n", name,
           waves_per_simd, wave_instrs / (ms * 1e-3) / 1e9, 2.4e9 / per_simd_per_s, wave_instrs * 64 / (ms * 1e-3) / 1e12);
}

int main() {
    uint64_t* d;
    hipMalloc(&d, 64);
    for (int w : {8}) {
        run("v_add_u32", k_add, d, w); run("v_sub_u32", k_sub, d, w); run("v_min_u32", k_min, d, w); run("v_and_b32", k_and, d, w);
        run("v_add3_u32", k_add3, d, w); run("v_lshlrev", k_lshl, d, w); run("v_lshl_add", k_lshladd, d, w);
        run("v_mul_lo_u32", k_mullo, d, w); run("v_mul_hi_u32", k_mulhi, d, w); run("v_mul_hi_i32", k_mulhii, d, w);
        run("v_mul_u32_u24", k_mul24, d, w); run("v_mad_u32_u24", k_mad24, d, w); run("v_mad_u64_u32", k_mad64, d, w);
        run("v_fma_f32", k_fmaf32, d, w); run("v_pk_add_f32", k_pkaddf32, d, w);
        run("v_add_f64", k_addf64, d, w); run("v_mul_f64", k_mulf64, d, w); run("v_fma_f64", k_fmaf64, d, w);
        run("v_rndne_f64", k_rndf64, d, w); run("v_cvt_f64_u32", k_cvtf64u, d, w); run("v_cvt_u32_f64", k_cvtuf64, d, w);
        run("v_mov_dpp", k_dpp, d, w);
        run("v_cndmask_b32", k_cndmask, d, w); run("v_sub_co_u32", k_subco, d, w); run("v_add_co_u32", k_addco, d, w); run("v_cmp_lt_u32", k_cmp, d, w);
        run("v_max_u32", k_max, d, w); run("v_xor_b32", k_xor, d, w); run("v_mov_b32", k_mov, d, w); run("v_ashrrev_i32", k_ashr, d, w); run("v_subrev_u32", k_subrev, d, w); run("v_min_i32", k_mini, d, w); run("v_add_u32 literal", k_addlit, d, w);
        printf("\n");
    }
    return 0;
}
