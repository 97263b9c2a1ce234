/* ORACLE — TEST INFRASTRUCTURE ONLY (see bb31.h).  Plain-C CPU restatement of
 * the fib_air NTT/LDE + Poseidon2-MMCS (+ FRI/STARK glue) path.  Each function
 * cites the reference file:line (or the absent upstream crate) it follows. */
#ifndef P3_ORACLE_H
#define P3_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* OpenMP thread count of the hashing / transform loops (default: all cores; 1 = the reference's serial build) */
void p3o_set_threads(int n);
int p3o_max_threads(void);

/* ---- field helpers exported for python tests ---- */
uint32_t p3o_to_monty(uint32_t canon);
uint32_t p3o_from_monty(uint32_t monty);
uint32_t p3o_add(uint32_t a, uint32_t b);
uint32_t p3o_sub(uint32_t a, uint32_t b);
uint32_t p3o_mul(uint32_t a, uint32_t b);
uint32_t p3o_inv(uint32_t a);
uint32_t p3o_pow(uint32_t a, uint64_t e);
uint32_t p3o_two_adic_generator(unsigned bits);
void p3o_ext_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]);
void p3o_ext_inv(const uint32_t a[4], uint32_t out[4]);

/* ---- dft.c ---- */
void p3o_twiddle_table(unsigned log_n, uint32_t *out /* (1<<log_n)-1 words */);
void p3o_bit_reverse_rows(uint32_t *dst, const uint32_t *src, size_t height, size_t width);
void p3o_stage_in_place(uint32_t *data, size_t width, size_t height, unsigned stage,
                        const uint32_t *stage_twiddles);
void p3o_naive_dft(const uint32_t *in, uint32_t *out, size_t height, size_t width);
int p3o_dft_batch(const uint32_t *in, uint32_t *out, size_t height, size_t width);
int p3o_idft_batch(const uint32_t *in, uint32_t *out, size_t height, size_t width);
int p3o_coset_dft_batch(const uint32_t *in, uint32_t *out, size_t height, size_t width,
                        uint32_t shift_monty);
int p3o_coset_lde_batch(const uint32_t *in, uint32_t *out, size_t height, size_t width,
                        unsigned added_bits, uint32_t shift_monty, int bit_reversed_out);

/* ---- poseidon2.c ---- */
void p3o_poseidon2_permute(uint32_t state[16]);
/* same permutation with caller-supplied (Montgomery-form) constants: KAT pinning */
void p3o_poseidon2_permute_rc(uint32_t state[16], const uint32_t ext_init[4][16],
                              const uint32_t internal[13], const uint32_t ext_final[4][16]);

/* ---- mmcs.c ---- */
void p3o_hash_row(const uint32_t *items, size_t n, uint32_t out[8]);
void p3o_compress(const uint32_t left[8], const uint32_t right[8], uint32_t out[8]);
typedef struct p3o_tree p3o_tree_t;
/* mats[i]: row-major heights[i] x widths[i], heights powers of two. Returns NULL on bad input. */
p3o_tree_t *p3o_mmcs_commit(const uint32_t *const *mats, const size_t *heights,
                            const size_t *widths, size_t n_mats, uint32_t root_out[8]);
size_t p3o_tree_num_layers(const p3o_tree_t *t);
size_t p3o_tree_layer_len(const p3o_tree_t *t, size_t layer);
const uint32_t *p3o_tree_layer(const p3o_tree_t *t, size_t layer); /* len*8 words */
size_t p3o_tree_log_max_height(const p3o_tree_t *t);
/* rows_out: concatenated opened rows (sum of widths words); path_out: log_max_height*8 words */
int p3o_mmcs_open_batch(const p3o_tree_t *t, size_t index, uint32_t *rows_out, uint32_t *path_out);
int p3o_mmcs_verify_batch(const uint32_t root[8], const size_t *heights, const size_t *widths,
                          size_t n_mats, size_t index, const uint32_t *rows,
                          const uint32_t *path, size_t path_len);
void p3o_mmcs_free(p3o_tree_t *t);
/* kind 0 = Poseidon2 (as above), 1 = the reference's Keccak configuration (keccak.c) */
p3o_tree_t *p3o_mmcs_commit_kind(int kind, const uint32_t *const *mats, const size_t *heights,
                                 const size_t *widths, size_t n_mats, uint32_t root_out[8]);
int p3o_mmcs_verify_batch_kind(int kind, const uint32_t root[8], const size_t *heights, const size_t *widths,
                               size_t n_mats, size_t index, const uint32_t *rows,
                               const uint32_t *path, size_t path_len);

/* ---- keccak.c: Keccak-f[1600], PaddingFreeSponge<KeccakF,25,17,4>, SerializingHasher, CompressionFunctionFromHasher ---- */
void p3o_keccak_f(uint64_t state[25]);
void p3o_keccak_sponge_u64(const uint64_t *items, size_t n, uint64_t out[4]);
void p3o_keccak_hash_row(const uint32_t *items, size_t n, uint32_t out[8]);
void p3o_keccak_compress(const uint32_t left[8], const uint32_t right[8], uint32_t out[8]);
void p3o_keccak256(const uint8_t *in, size_t n, uint8_t out[32]);

/* ---- rng.c: SmallRng::seed_from_u64 (xoshiro256++ / SplitMix64, rand 0.9.2) drawing BabyBear elements ---- */
void p3o_rng_seed_from_u64(uint64_t s[4], uint64_t seed);
uint64_t p3o_rng_next_u64(uint64_t s[4]);
void p3o_rng_fill_field(uint64_t s[4], uint32_t *out, size_t n);

/* ---- stark.c: fib_air prover / verifier (uni-stark + two-adic FRI PCS + duplex challenger) ---- */
int p3o_prove_fib_air(uint64_t a, uint64_t b, unsigned log_n, unsigned log_blowup, unsigned log_final_poly_len,
                      unsigned num_queries, unsigned pow_bits, uint8_t **out, size_t *out_len);
int p3o_verify_fib_air(const uint8_t *proof, size_t len, uint64_t a, uint64_t b, uint64_t x_pub, unsigned log_n,
                       unsigned log_blowup, unsigned log_final_poly_len, unsigned num_queries, unsigned pow_bits);
/* hash = 0: the Poseidon2 configuration above; 1: the reference's own hashes (fib_air.rs:28-53, non-hiding):
 * Keccak MMCS (keccak.c) + SerializingChallenger32<BabyBear, HashChallenger<u8, Keccak256Hash, 32>> */
int p3o_prove_fib_air_hash(int hash, uint64_t a, uint64_t b, unsigned log_n, unsigned log_blowup, unsigned log_final_poly_len,
                           unsigned num_queries, unsigned pow_bits, uint8_t **out, size_t *out_len);
int p3o_verify_fib_air_hash(int hash, const uint8_t *proof, size_t len, uint64_t a, uint64_t b, uint64_t x_pub, unsigned log_n,
                            unsigned log_blowup, unsigned log_final_poly_len, unsigned num_queries, unsigned pow_bits);
/* ---- stark_hiding.c: the HIDING half of the reference's configuration (fib_air.rs:40-65: MerkleTreeHidingMmcs +
 * HidingFriPcs, SmallRng::seed_from_u64(seed)), for either hash configuration; wire format version 2 ---- */
int p3o_prove_fib_air_hiding(int hash, uint64_t a, uint64_t b, unsigned log_n, unsigned log_blowup, unsigned log_final_poly_len,
                             unsigned num_queries, unsigned pow_bits, uint64_t seed, uint8_t **out, size_t *out_len);
int p3o_verify_fib_air_hiding(int hash, const uint8_t *proof, size_t len, uint64_t a, uint64_t b, uint64_t x_pub, unsigned log_n,
                              unsigned log_blowup, unsigned log_final_poly_len, unsigned num_queries, unsigned pow_bits);
void p3o_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
