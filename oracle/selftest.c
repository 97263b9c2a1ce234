/* ORACLE — TEST INFRASTRUCTURE ONLY.  Sanitizer driver: the oracle's prover/verifier pairs (both hash
 * configurations), trees with injected matrices and the transforms, run under AddressSanitizer +
 * UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on the pool):
 *   make -C oracle sanitize                                                                       */
#include "p3_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t lcg(uint64_t *s) { *s = *s * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(*s >> 33) % 0x78000001u; }

int main(void) {
    uint64_t seed = 1;
    for (int hash = 0; hash < 2; hash++) {
        for (unsigned log_n = 1; log_n <= 9; log_n += 2) {
            unsigned params[3][4] = {{1, 0, 6, 3}, {2, 1, 4, 2}, {1, 2, 5, 4}};
            for (int p = 0; p < 3; p++) {
                if (params[p][1] >= log_n && params[p][1] > 0) continue;
                uint8_t *pf = NULL; size_t len = 0;
                if (p3o_prove_fib_air_hash(hash, 2, 3, log_n, params[p][0], params[p][1], params[p][2], params[p][3], &pf, &len)) { printf("prove failed\n"); return 1; }
                uint64_t l = 2, r = 3;
                for (uint64_t i = 1; i < (1ull << log_n); i++) { uint64_t t = (l + r) % 0x78000001u; l = r; r = t; }
                if (p3o_verify_fib_air_hash(hash, pf, len, 2, 3, r, log_n, params[p][0], params[p][1], params[p][2], params[p][3])) { printf("verify failed\n"); return 2; }
                /* every truncation and a few corruptions must be rejected without touching memory out of bounds */
                for (size_t cut = 0; cut < len; cut += 1 + len / 97)
                    if (!p3o_verify_fib_air_hash(hash, pf, cut, 2, 3, r, log_n, params[p][0], params[p][1], params[p][2], params[p][3])) { printf("truncated proof accepted\n"); return 3; }
                for (int k = 0; k < 64; k++) {
                    size_t pos = lcg(&seed) % len;
                    pf[pos] ^= 0x5a;
                    (void)p3o_verify_fib_air_hash(hash, pf, len, 2, 3, r, log_n, params[p][0], params[p][1], params[p][2], params[p][3]);
                    pf[pos] ^= 0x5a;
                }
                p3o_free(pf);
            }
        }
        /* the hiding configuration: prove, verify, truncations, corruptions */
        for (unsigned log_n = 1; log_n <= 7; log_n += 3) {
            uint8_t *pf = NULL; size_t len = 0;
            if (p3o_prove_fib_air_hiding(hash, 2, 3, log_n, 1, 0, 5, 3, 1, &pf, &len)) { printf("hiding prove failed\n"); return 8; }
            uint64_t l = 2, r = 3;
            for (uint64_t i = 1; i < (1ull << log_n); i++) { uint64_t t = (l + r) % 0x78000001u; l = r; r = t; }
            if (p3o_verify_fib_air_hiding(hash, pf, len, 2, 3, r, log_n, 1, 0, 5, 3)) { printf("hiding verify failed\n"); return 9; }
            for (size_t cut = 0; cut < len; cut += 1 + len / 61)
                if (!p3o_verify_fib_air_hiding(hash, pf, cut, 2, 3, r, log_n, 1, 0, 5, 3)) { printf("truncated hiding proof accepted\n"); return 10; }
            for (int k = 0; k < 64; k++) {
                size_t pos = lcg(&seed) % len;
                pf[pos] ^= 0x5a;
                (void)p3o_verify_fib_air_hiding(hash, pf, len, 2, 3, r, log_n, 1, 0, 5, 3);
                pf[pos] ^= 0x5a;
            }
            p3o_free(pf);
        }
        /* a tree with three height classes and odd widths */
        size_t hs[4] = {64, 64, 16, 1}, ws[4] = {3, 36, 5, 9};
        uint32_t *mats[4];
        for (int m = 0; m < 4; m++) { mats[m] = malloc(hs[m] * ws[m] * 4); for (size_t i = 0; i < hs[m] * ws[m]; i++) mats[m][i] = lcg(&seed); }
        uint32_t root[8], rows[64], path[8 * 8];
        p3o_tree_t *t = p3o_mmcs_commit_kind(hash, (const uint32_t *const *)mats, hs, ws, 4, root);
        for (size_t idx = 0; idx < 64; idx += 7) {
            if (p3o_mmcs_open_batch(t, idx, rows, path)) return 4;
            if (p3o_mmcs_verify_batch_kind(hash, root, hs, ws, 4, idx, rows, path, 6)) { printf("opening rejected\n"); return 5; }
        }
        p3o_mmcs_free(t);
        for (int m = 0; m < 4; m++) free(mats[m]);
    }
    /* transforms: LDE then check one known relation (blowup 2 with shift 1 extends the input itself) */
    { size_t h = 256, w = 3; uint32_t *x = malloc(h * w * 4), *y = malloc(2 * h * w * 4);
      for (size_t i = 0; i < h * w; i++) x[i] = lcg(&seed);
      if (p3o_coset_lde_batch(x, y, h, w, 1, p3o_to_monty(1), 0)) return 6;
      for (size_t i = 0; i < h; i++) for (size_t c = 0; c < w; c++) if (y[2 * i * w + c] != x[i * w + c]) { printf("lde mismatch\n"); return 7; }
      free(x); free(y); }
    printf("oracle selftest ok\n");
    return 0;
}
