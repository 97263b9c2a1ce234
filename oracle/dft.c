/* ORACLE — TEST INFRASTRUCTURE ONLY (see bb31.h).
 * Radix-2 DIT dft_batch over BabyBear restating the reference's CPU statement of
 * its own GPU algorithm, plus the (absent, upstream) coset-LDE wrapper. */
#include "p3_oracle.h"
#include "bb31.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* threads used by the OpenMP loops (1 = the reference build's single thread) */
void p3o_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
int p3o_max_threads(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}
uint32_t p3o_to_monty(uint32_t c) { return bb_to_monty(c); }
uint32_t p3o_from_monty(uint32_t m) { return bb_from_monty(m); }
uint32_t p3o_add(uint32_t a, uint32_t b) { return bb_add(a, b); }
uint32_t p3o_sub(uint32_t a, uint32_t b) { return bb_sub(a, b); }
uint32_t p3o_mul(uint32_t a, uint32_t b) { return bb_mul(a, b); }
uint32_t p3o_inv(uint32_t a) { return bb_inv(a); }
uint32_t p3o_pow(uint32_t a, uint64_t e) { return bb_pow(a, e); }
uint32_t p3o_two_adic_generator(unsigned bits) { return bb_two_adic_generator(bits); }
void p3o_ext_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]) {
    bb4_t x, y; memcpy(x.c, a, 16); memcpy(y.c, b, 16);
    bb4_t r = bb4_mul(x, y); memcpy(out, r.c, 16);
}
void p3o_ext_inv(const uint32_t a[4], uint32_t out[4]) {
    bb4_t x; memcpy(x.c, a, 16);
    bb4_t r = bb4_inv(x); memcpy(out, r.c, 16);
}

static unsigned log2_exact(size_t n) {
    unsigned l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}
static int is_pow2(size_t n) { return n && !(n & (n - 1)); }

/* backend_vulkan.rs:977-996 twiddles_for_stage / twiddle_table:
 * stage s holds step^0..step^(2^s-1), step = root^(2^(log_n-s-1)); stage s starts at 2^s-1. */
/* Rows of BLK consecutive powers are what the OpenMP loops below hand to a thread: a block starts from
 * bb_pow(base, first index) and multiplies on, so every entry is the same field element as in the serial
 * recurrence (field arithmetic is exact: same words whatever the thread count; p3o_set_threads(1) IS the serial loop). */
#define P3O_BLK ((size_t)4096)
void p3o_twiddle_table(unsigned log_n, uint32_t *out) {
    uint32_t root = bb_two_adic_generator(log_n);
    for (unsigned stage = 0; stage < log_n; stage++) {
        uint32_t step = root;
        for (unsigned i = 0; i < log_n - stage - 1; i++) step = bb_mul(step, step);
        size_t half = (size_t)1 << stage;
        uint32_t *dst = out + (half - 1);
        size_t nblk = (half + P3O_BLK - 1) / P3O_BLK;
        #pragma omp parallel for schedule(static) if (nblk > 1)
        for (size_t b = 0; b < nblk; b++) {
            size_t lo = b * P3O_BLK, hi = lo + P3O_BLK < half ? lo + P3O_BLK : half;
            uint32_t acc = bb_pow(step, lo);
            for (size_t i = lo; i < hi; i++) { dst[i] = acc; acc = bb_mul(acc, step); }
        }
    }
}

/* backend_vulkan.rs:998-1026 reverse_bits_len_usize / write_bit_reversed_rows_u32 */
static size_t reverse_bits_len(size_t x, unsigned bits) {
    size_t y = 0;
    for (unsigned i = 0; i < bits; i++) { y = (y << 1) | (x & 1); x >>= 1; }
    return y;
}
void p3o_bit_reverse_rows(uint32_t *dst, const uint32_t *src, size_t height, size_t width) {
    if (!width || !height) return;
    if (!is_pow2(height)) { memcpy(dst, src, height * width * 4); return; }
    unsigned bits = log2_exact(height);
    #pragma omp parallel for schedule(static) if (height >= 4096)
    for (size_t row = 0; row < height; row++)
        memcpy(dst + row * width, src + reverse_bits_len(row, bits) * width, width * 4);
}

/* backend_vulkan.rs:881-942 cpu_stage_u32_in_place (loop nest interchanged: butterfly
 * index outer, column inner — same butterflies, row-major friendly). */
void p3o_stage_in_place(uint32_t *data, size_t width, size_t height, unsigned stage,
                        const uint32_t *tw) {
    size_t m = (size_t)1 << (stage + 1), half = m >> 1;
    /* the butterflies of one stage touch disjoint row pairs: threads over j */
    #pragma omp parallel for schedule(static) if (height >= 8192)
    for (size_t j = 0; j < height / 2; j++) {
        size_t block = j / half, offset = j % half;
        size_t base = block * m + offset;
        if (base + half >= height) continue;
        uint32_t *r0 = data + base * width, *r1 = data + (base + half) * width;
        uint32_t w = tw[offset];
        for (size_t c = 0; c < width; c++) {
            uint32_t a = r0[c], t = bb_mul(r1[c], w);
            r0[c] = bb_add(a, t);
            r1[c] = bb_sub(a, t);
        }
    }
}

/* Definition the reference asserts equality with (fib_air.rs:193-196 vs Radix2DitParallel):
 * out[k][c] = sum_i in[i][c] * w^(ik), w = two_adic_generator(log2 h).  O(h^2 w): small h only. */
void p3o_naive_dft(const uint32_t *in, uint32_t *out, size_t height, size_t width) {
    unsigned log_n = log2_exact(height);
    uint32_t root = bb_two_adic_generator(log_n);
    uint32_t *pw = malloc(height * 4);
    pw[0] = BB_ONE;
    for (size_t i = 1; i < height; i++) pw[i] = bb_mul(pw[i - 1], root);
    for (size_t k = 0; k < height; k++)
        for (size_t c = 0; c < width; c++) {
            uint32_t acc = 0;
            for (size_t i = 0; i < height; i++)
                acc = bb_add(acc, bb_mul(in[i * width + c], pw[(i * k) & (height - 1)]));
            out[k * width + c] = acc;
        }
    free(pw);
}

/* backend_vulkan.rs:1988-2063 dft_batch -> :1028-1426 setup_vulkan_pipeline_plan:
 * bit-reverse rows (:1085), twiddle table (:1088), stages 0..log_n-1 with twiddle_base 2^s-1 (:1182-1294).
 * Natural order in, natural order out.  Returns -1 on non-power-of-two height (:1992-1995). */
int p3o_dft_batch(const uint32_t *in, uint32_t *out, size_t height, size_t width) {
    if (!height || !width) return 0;
    if (!is_pow2(height)) return -1;
    unsigned log_n = log2_exact(height);
    p3o_bit_reverse_rows(out, in, height, width);
    if (log_n == 0) return 0;
    uint32_t *tw = malloc(((size_t)1 << log_n) * 4);
    p3o_twiddle_table(log_n, tw);
    for (unsigned s = 0; s < log_n; s++)
        p3o_stage_in_place(out, width, height, s, tw + (((size_t)1 << s) - 1));
    free(tw);
    return 0;
}

/* [UPSTREAM p3-dft 0.4.2 TwoAdicSubgroupDft::idft_batch, source absent; SURVEY §8a R9]:
 * dft, divide_by_height, swap rows r <-> h-r for r in 1..h/2. */
int p3o_idft_batch(const uint32_t *in, uint32_t *out, size_t height, size_t width) {
    if (!height || !width) return 0;
    uint32_t *tmp = malloc(height * width * 4);
    int rc = p3o_dft_batch(in, tmp, height, width);
    if (rc) { free(tmp); return rc; }
    uint32_t hinv = bb_inv(bb_to_monty((uint32_t)height));
    #pragma omp parallel for schedule(static) if (height >= 4096)
    for (size_t r = 0; r < height; r++) {
        size_t src = r ? height - r : 0;
        for (size_t c = 0; c < width; c++) out[r * width + c] = bb_mul(tmp[src * width + c], hinv);
    }
    free(tmp);
    return 0;
}

/* [UPSTREAM coset_dft_batch]: scale row i by shift^i, then dft_batch. */
int p3o_coset_dft_batch(const uint32_t *in, uint32_t *out, size_t height, size_t width,
                        uint32_t shift) {
    if (!height || !width) return 0;
    uint32_t *tmp = malloc(height * width * 4);
    size_t nblk = (height + P3O_BLK - 1) / P3O_BLK;
    #pragma omp parallel for schedule(static) if (nblk > 1)
    for (size_t b = 0; b < nblk; b++) {
        size_t lo = b * P3O_BLK, hi = lo + P3O_BLK < height ? lo + P3O_BLK : height;
        uint32_t wgt = bb_pow(shift, lo);
        for (size_t r = lo; r < hi; r++) {
            for (size_t c = 0; c < width; c++) tmp[r * width + c] = bb_mul(in[r * width + c], wgt);
            wgt = bb_mul(wgt, shift);
        }
    }
    int rc = p3o_dft_batch(tmp, out, height, width);
    free(tmp);
    return rc;
}

/* [UPSTREAM coset_lde_batch]: coeffs = idft(mat); zero-pad to h<<added_bits rows;
 * coset_dft_batch(coeffs, shift).  The PCS caller (TwoAdicFriPcs::commit) then takes
 * .bit_reverse_rows(); bit_reversed_out=1 applies that too. out: (h<<added_bits) x w. */
int p3o_coset_lde_batch(const uint32_t *in, uint32_t *out, size_t height, size_t width,
                        unsigned added_bits, uint32_t shift, int bit_reversed_out) {
    if (!height || !width) return 0;
    if (!is_pow2(height)) return -1;
    size_t big = height << added_bits;
    uint32_t *coeffs = calloc(big * width, 4);
    int rc = p3o_idft_batch(in, coeffs, height, width);
    if (rc) { free(coeffs); return rc; }
    if (!bit_reversed_out) {
        rc = p3o_coset_dft_batch(coeffs, out, big, width, shift);
    } else {
        uint32_t *nat = malloc(big * width * 4);
        rc = p3o_coset_dft_batch(coeffs, nat, big, width, shift);
        p3o_bit_reverse_rows(out, nat, big, width);
        free(nat);
    }
    free(coeffs);
    return rc;
}
