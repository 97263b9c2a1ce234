/* ORACLE — TEST INFRASTRUCTURE ONLY (see bb31.h).
 * Poseidon2 Merkle-tree MMCS.  The reference wires the Keccak analogues at
 * native/src/fib_air.rs:31-51 (PaddingFreeSponge / CompressionFunctionFromHasher /
 * MerkleTreeHidingMmcs); north_star asks for the Poseidon2 instantiation
 *   hash     = PaddingFreeSponge<Perm16, WIDTH 16, RATE 8, OUT 8>
 *   compress = TruncatedPermutation<Perm16, N 2, CHUNK 8, WIDTH 16>
 *   mmcs     = MerkleTreeMmcs<.., DIGEST_ELEMS 8>
 * whose code is in the ABSENT crates p3-symmetric / p3-merkle-tree 0.4.2 [UPSTREAM-RECALL,
 * SURVEY.md §8a R11-R12].  PARITY UNPINNED beyond the permutation: no fixture in the
 * reference covers sponge / compress / tree layout. */
#include "p3_oracle.h"
#include "bb31.h"
#include <stdlib.h>
#include <string.h>

/* PaddingFreeSponge::hash_iter: overwrite-mode absorb of RATE=8 items, permute after every
 * full chunk and after a non-empty partial chunk; empty input -> all-zero digest. */
void p3o_hash_row(const uint32_t *items, size_t n, uint32_t out[8]) {
    uint32_t st[16] = {0};
    size_t i = 0;
    while (i < n) {
        size_t take = n - i < 8 ? n - i : 8;
        memcpy(st, items + i, take * 4);
        p3o_poseidon2_permute(st);
        i += take;
    }
    memcpy(out, st, 32);
}
/* TruncatedPermutation::compress: state = left || right, permute, first 8. */
void p3o_compress(const uint32_t l[8], const uint32_t r[8], uint32_t out[8]) {
    uint32_t st[16];
    memcpy(st, l, 32); memcpy(st + 8, r, 32);
    p3o_poseidon2_permute(st);
    memcpy(out, st, 32);
}

/* hash configuration of a tree: 0 = Poseidon2 (north_star), 1 = Keccak (keccak.c, the reference's own) */
typedef void (*hash_row_fn)(const uint32_t *, size_t, uint32_t[8]);
typedef void (*compress_fn)(const uint32_t[8], const uint32_t[8], uint32_t[8]);
static hash_row_fn hash_of(int kind) { return kind ? p3o_keccak_hash_row : p3o_hash_row; }
static compress_fn compress_of(int kind) { return kind ? p3o_keccak_compress : p3o_compress; }

struct p3o_tree {
    int kind;
    size_t n_mats;
    const uint32_t **mats; size_t *heights, *widths; /* borrowed matrix pointers */
    size_t n_layers; size_t *layer_len; uint32_t **layers;
    size_t log_max_height;
};

static int is_pow2(size_t n) { return n && !(n & (n - 1)); }
static unsigned log2_exact(size_t n) { unsigned l = 0; while (((size_t)1 << l) < n) l++; return l; }

/* hash of the concatenated i-th rows of every matrix of height h (original order) */
static void hash_rows_of_height(const p3o_tree_t *t, size_t h, size_t row, uint32_t out[8]) {
    size_t tot = 0;
    for (size_t m = 0; m < t->n_mats; m++) if (t->heights[m] == h) tot += t->widths[m];
    uint32_t *buf = malloc((tot ? tot : 1) * 4);
    size_t off = 0;
    for (size_t m = 0; m < t->n_mats; m++) if (t->heights[m] == h) {
        memcpy(buf + off, t->mats[m] + row * t->widths[m], t->widths[m] * 4);
        off += t->widths[m];
    }
    hash_of(t->kind)(buf, tot, out);
    free(buf);
}
static int has_height(const p3o_tree_t *t, size_t h) {
    for (size_t m = 0; m < t->n_mats; m++) if (t->heights[m] == h) return 1;
    return 0;
}

/* MerkleTree::new: first_digest_layer over the tallest matrices, then per layer
 * compress pairs and, where matrices of that height exist, compress in their row hash
 * (compress_and_inject).  Only power-of-two heights are accepted here. */
p3o_tree_t *p3o_mmcs_commit_kind(int kind, const uint32_t *const *mats, const size_t *heights,
                                 const size_t *widths, size_t n, uint32_t root_out[8]) {
    if (!n) return NULL;
    compress_fn compress = compress_of(kind);
    size_t maxh = 0;
    for (size_t i = 0; i < n; i++) { if (!is_pow2(heights[i])) return NULL; if (heights[i] > maxh) maxh = heights[i]; }
    p3o_tree_t *t = calloc(1, sizeof *t);
    t->kind = kind;
    t->n_mats = n;
    t->mats = malloc(n * sizeof *t->mats); t->heights = malloc(n * sizeof(size_t)); t->widths = malloc(n * sizeof(size_t));
    for (size_t i = 0; i < n; i++) { t->mats[i] = mats[i]; t->heights[i] = heights[i]; t->widths[i] = widths[i]; }
    t->log_max_height = log2_exact(maxh);
    t->n_layers = t->log_max_height + 1;
    t->layer_len = malloc(t->n_layers * sizeof(size_t));
    t->layers = malloc(t->n_layers * sizeof(uint32_t *));
    t->layer_len[0] = maxh;
    t->layers[0] = malloc(maxh * 32);
    /* rows are independent: optional OpenMP for the multi-core CPU baseline (p3o_set_threads) */
    #pragma omp parallel for schedule(static)
    for (size_t r = 0; r < maxh; r++) hash_rows_of_height(t, maxh, r, t->layers[0] + r * 8);
    for (size_t l = 1; l < t->n_layers; l++) {
        size_t len = t->layer_len[l - 1] / 2;
        t->layer_len[l] = len;
        t->layers[l] = malloc(len * 32);
        int inject = has_height(t, len);
        #pragma omp parallel for schedule(static) if (len >= 256)
        for (size_t i = 0; i < len; i++) {
            uint32_t d[8];
            compress(t->layers[l - 1] + 2 * i * 8, t->layers[l - 1] + (2 * i + 1) * 8, d);
            if (inject) {
                uint32_t rh[8];
                hash_rows_of_height(t, len, i, rh);
                compress(d, rh, t->layers[l] + i * 8);
            } else memcpy(t->layers[l] + i * 8, d, 32);
        }
    }
    memcpy(root_out, t->layers[t->n_layers - 1], 32);
    return t;
}
p3o_tree_t *p3o_mmcs_commit(const uint32_t *const *mats, const size_t *heights,
                            const size_t *widths, size_t n, uint32_t root_out[8]) {
    return p3o_mmcs_commit_kind(0, mats, heights, widths, n, root_out);
}
size_t p3o_tree_num_layers(const p3o_tree_t *t) { return t->n_layers; }
size_t p3o_tree_layer_len(const p3o_tree_t *t, size_t l) { return t->layer_len[l]; }
const uint32_t *p3o_tree_layer(const p3o_tree_t *t, size_t l) { return t->layers[l]; }
size_t p3o_tree_log_max_height(const p3o_tree_t *t) { return t->log_max_height; }

/* MerkleTreeMmcs::open_batch: row (index >> (log_max - log_h)) of every matrix, and the
 * sibling digest_layers[i][(index >> i) ^ 1] for i in 0..log_max_height. */
int p3o_mmcs_open_batch(const p3o_tree_t *t, size_t index, uint32_t *rows_out, uint32_t *path_out) {
    if (index >= ((size_t)1 << t->log_max_height)) return -1;
    size_t off = 0;
    for (size_t m = 0; m < t->n_mats; m++) {
        size_t r = index >> (t->log_max_height - log2_exact(t->heights[m]));
        memcpy(rows_out + off, t->mats[m] + r * t->widths[m], t->widths[m] * 4);
        off += t->widths[m];
    }
    for (size_t i = 0; i < t->log_max_height; i++)
        memcpy(path_out + i * 8, t->layers[i] + (((index >> i) ^ 1) * 8), 32);
    return 0;
}

/* MerkleTreeMmcs::verify_batch for power-of-two heights; rows = opened rows in matrix order. */
int p3o_mmcs_verify_batch_kind(int kind, const uint32_t root[8], const size_t *heights, const size_t *widths,
                               size_t n, size_t index, const uint32_t *rows, const uint32_t *path,
                               size_t path_len) {
    hash_row_fn hash_row = hash_of(kind);
    compress_fn compress = compress_of(kind);
    size_t maxh = 0;
    for (size_t i = 0; i < n; i++) if (heights[i] > maxh) maxh = heights[i];
    if (!n || path_len != log2_exact(maxh)) return -1;
    size_t *offs = malloc(n * sizeof(size_t)), tot = 0;
    for (size_t i = 0; i < n; i++) { offs[i] = tot; tot += widths[i]; }
    uint32_t *buf = malloc((tot ? tot : 1) * 4);
    uint32_t cur[8];
    size_t h = maxh;
    for (size_t level = 0;; level++) {
        size_t k = 0; int any = 0;
        for (size_t m = 0; m < n; m++) if (heights[m] == h) { memcpy(buf + k, rows + offs[m], widths[m] * 4); k += widths[m]; any = 1; }
        if (level == 0) hash_row(buf, k, cur);
        else if (any) { uint32_t rh[8], d[8]; hash_row(buf, k, rh); compress(cur, rh, d); memcpy(cur, d, 32); }
        if (h == 1) break;
        uint32_t d[8];
        const uint32_t *sib = path + level * 8;
        if ((index >> level) & 1) compress(sib, cur, d); else compress(cur, sib, d);
        memcpy(cur, d, 32);
        h >>= 1;
    }
    free(buf); free(offs);
    return memcmp(cur, root, 32) == 0 ? 0 : 1;
}
int p3o_mmcs_verify_batch(const uint32_t root[8], const size_t *heights, const size_t *widths,
                          size_t n, size_t index, const uint32_t *rows, const uint32_t *path,
                          size_t path_len) {
    return p3o_mmcs_verify_batch_kind(0, root, heights, widths, n, index, rows, path, path_len);
}
void p3o_mmcs_free(p3o_tree_t *t) {
    if (!t) return;
    for (size_t l = 0; l < t->n_layers; l++) free(t->layers[l]);
    free(t->layers); free(t->layer_len); free(t->mats); free(t->heights); free(t->widths); free(t);
}
