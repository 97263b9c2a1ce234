/* libp3hip — C ABI of the MI355X (gfx950) backend for the fib_air NTT/LDE + Poseidon2-MMCS path.
 *
 * The reference (miha-stopar/Plonky3-mobile) has no C ABI: its plug points are Rust traits and five
 * JNI exports.  Every entry point below names the reference interface it stands in
 * for, so a Rust `backend_hip.rs` (same shape as native/src/backend_metal.rs:5-10) and a
 * `HipMmcs: Mmcs<BabyBear>` can bind them with `extern "C"`; INTEGRATION.md shows those stubs.
 *
 * Conventions (carried over from the reference):
 *   - values are u32 BabyBear Montgomery words in [0, P), P = 0x78000001 — `to_unique_u32`
 *     (native/src/backend_vulkan.rs:2002-2005); matrices are row-major height x width;
 *   - every function returns 0 on success or a negative P3HIP_ERR_* code and NEVER aborts; the message
 *     goes to a per-thread take-and-clear mailbox (native/src/gpu_dft.rs:42,65-68);
 *   - device state (cached tables, scratch) belongs to the (calling thread, current device) pair and is created on
 *     first use (native/src/backend_vulkan.rs:100-124: the reference's runtime is thread-local too).  A thread that
 *     switches device (hipSetDevice) simply gets that device's own context; objects that own HBM (trees, provers)
 *     remember their device and return P3HIP_ERR_BAD_ARG when used with another one current;
 *   - there is NO CPU fallback inside this library: on error the caller decides (the Rust GpuDft keeps
 *     its own Radix2DitParallel fallback, native/src/gpu_dft.rs:100-112).
 *   - `*_dev` variants take HBM pointers (hipMalloc / p3hip_malloc / a torch tensor's data_ptr) and a
 *     hipStream_t passed as void*; they enqueue and return without synchronising.  STREAM CONTRACT: any number of
 *     streams may be used from one thread, also interleaved — intermediates live in scratch keyed by (thread, stream),
 *     tables built at first use are filled by a kernel on the calling stream and guarded by an event for the others.
 *     The exceptions to "enqueue only": the very first call of a thread on a device builds its context (blocking),
 *     a scratch slab that has to GROW is reallocated after a device synchronise, and so is the bounded table cache
 *     once a thread has used more than 256 distinct (shift, height) pairs.  A given stream must not be used from two
 *     threads at once with buffers that alias.
 */
#ifndef P3HIP_H
#define P3HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3HIP_OK 0
#define P3HIP_ERR_BAD_ARG (-1)  /* null pointer, non power-of-two height (backend_vulkan.rs:1992-1995), ... */
#define P3HIP_ERR_HIP (-2)      /* HIP runtime failure; text via p3hip_take_last_error */
#define P3HIP_ERR_BACKEND (-3)  /* unknown backend name (gpu_dft.rs:59) */
#define P3HIP_ERR_INTERNAL (-4)

/* BackendKind codes: gpu_dft.rs:14-40 (Cpu=0, Vulkan=1, Metal=2, WebGpu=3) plus the new Hip=4. */
#define P3HIP_BACKEND_CPU 0
#define P3HIP_BACKEND_VULKAN 1
#define P3HIP_BACKEND_METAL 2
#define P3HIP_BACKEND_WEBGPU 3
#define P3HIP_BACKEND_HIP 4

/* ---- selector + diagnostics -------------------------------------------------------------------- */
/* set_backend_kind_from_str (gpu_dft.rs:53-63) behind JNI setBackend (lib.rs:133-146): case-insensitive
 * "cpu" | "vulkan" | "metal" | "webgpu" | "hip"; unknown -> P3HIP_ERR_BACKEND, message "unknown backend '<x>'".
 * Process-global relaxed atomic, default P3HIP_BACKEND_HIP. */
int p3hip_set_backend(const char *name);
/* get_backend_kind (gpu_dft.rs:49-51) */
int p3hip_get_backend(void);
/* is_vulkan_available (backend_vulkan.rs:726-731) behind JNI isVulkanAvailable (lib.rs:167-179): creates
 * the context; writes "HIP available: <device>" or "HIP unavailable: <error>" into msg. Returns 0 if usable. */
int p3hip_is_available(char *msg, size_t cap);
/* take_last_vulkan_error (gpu_dft.rs:65-68): returns the calling thread's pending message and clears it;
 * NULL when there is none.  The pointer stays valid until the next call on the same thread. */
const char *p3hip_take_last_error(void);

/* ---- device memory helpers for FFI callers that do not link HIP themselves ---------------------- */
int p3hip_malloc(void **dev_ptr, size_t bytes);
int p3hip_free(void *dev_ptr);
int p3hip_upload(void *dev_dst, const void *host_src, size_t bytes);
int p3hip_download(void *host_dst, const void *dev_src, size_t bytes);
int p3hip_sync(void *stream);
/* Frees the calling thread's device state (tables, scratch) on every device it used, after synchronising them.
 * Optional: worker threads that are about to exit call it so that nothing is left behind in HBM. */
void p3hip_release_thread_context(void);

/* ---- TwoAdicSubgroupDft<BabyBear> --------------------------------------------------------------- */
/* backend_vulkan::dft_batch (backend_vulkan.rs:1988-2063) / setup_vulkan_pipeline_plan (:1028-1031):
 * natural row order in, natural row order out, out[k][c] = sum_i in[i][c] w^(ik).  Host pointers:
 * upload, kernels, download, synchronise — the reference's "e2e" view (fib_air.rs:148-157). */
int p3hip_dft_batch_bb31(const uint32_t *in, uint32_t *out, size_t height, size_t width);
/* TwoAdicSubgroupDft::idft_batch [upstream provided method; SURVEY.md §8a R9] */
int p3hip_idft_batch_bb31(const uint32_t *in, uint32_t *out, size_t height, size_t width);
/* TwoAdicSubgroupDft::coset_dft_batch: coefficients -> evaluations over shift*<g>, natural order */
int p3hip_coset_dft_batch_bb31(const uint32_t *in, uint32_t *out, size_t height, size_t width,
                               uint32_t shift_monty);
/* TwoAdicSubgroupDft::coset_lde_batch (+ the `.bit_reverse_rows()` TwoAdicFriPcs::commit applies when
 * bit_reversed_out != 0).  out has (height << added_bits) rows. */
int p3hip_coset_lde_batch_bb31(const uint32_t *in, uint32_t *out, size_t height, size_t width,
                               unsigned added_bits, uint32_t shift_monty, int bit_reversed_out);
/* The reference logs one line per DFT call (backend_vulkan.rs:1385-1423: upload / stages / readback / total, plus the GPU
 * timestamps).  The host-pointer entry points above keep the same line — "hip dft: op=.. h=.. w=.. stages=..
 * upload=..ms stages=..ms readback=..ms total=..ms gpu(stage=..ms copy_back=..ms total=..ms)" — for the calling thread's
 * last call (NULL before the first); with P3HIP_LOG_TIMING=1 it is also written to stderr like the reference's. */
const char *p3hip_last_timing_line(void);
/* device-resident forms — the reference's "kernel-only" view (backend_vulkan.rs:1428-1693) */
int p3hip_dft_batch_bb31_dev(const uint32_t *d_in, uint32_t *d_out, size_t height, size_t width, void *stream);
int p3hip_idft_batch_bb31_dev(const uint32_t *d_in, uint32_t *d_out, size_t height, size_t width, void *stream);
int p3hip_coset_dft_batch_bb31_dev(const uint32_t *d_in, uint32_t *d_out, size_t height, size_t width,
                                   uint32_t shift_monty, void *stream);
int p3hip_coset_lde_batch_bb31_dev(const uint32_t *d_in, uint32_t *d_out, size_t height, size_t width,
                                   unsigned added_bits, uint32_t shift_monty, int bit_reversed_out,
                                   void *stream);
/* The same extension when the caller already holds COEFFICIENTS (natural order, `height` rows = the degree bound) instead of
 * evaluations: rows of d_out = evaluations over shift*<g_{height << added_bits}> in bit-reversed order, i.e.
 * TwoAdicSubgroupDft::coset_dft_batch of the zero-padded coefficient matrix followed by bit_reverse_rows, without transforming
 * to the subgroup and back (HidingFriPcs commits its blinded quotient chunks from coefficients, fib_air.rs:64-65). */
int p3hip_coset_lde_from_coeffs_bb31_dev(const uint32_t *d_coeffs, uint32_t *d_out, size_t height, size_t width,
                                         unsigned added_bits, uint32_t shift_monty, void *stream);
/* prepare_compute_plan (backend_vulkan.rs:959-975): the reference's VulkanComputePlan carries, next to the stage parameters, the
 * LAUNCH GEOMETRY of its plan (`dispatch`: workgroup counts of one stage; the host then loops log2(height) such dispatches,
 * :1182-1294).  The hip backend's plan for dft_batch / idft_batch of a height x width matrix is a handful of LDS-tiled passes:
 * *n_passes = kernel launches, stages_per_pass[i] = radix-2 stages pass i performs (their sum is log2 height; at most `cap` entries
 * are written).  Host-only: no GPU is touched. */
int p3hip_dft_plan_bb31(size_t height, size_t width, uint32_t *stages_per_pass, size_t cap, size_t *n_passes);
/* write_bit_reversed_rows_u32 (backend_vulkan.rs:1005-1026) on device */
int p3hip_bit_reverse_rows_dev(const uint32_t *d_in, uint32_t *d_out, size_t height, size_t width, void *stream);

/* ---- FibonacciAir workload (native/src/fib_air.rs:224-306) --------------------------------------- */
/* generate_trace_rows (fib_air.rs:266-284): n x 2 trace, row 0 = (a, b), row i = (right, left + right);
 * n must be a power of two (fib_air.rs:267). */
int p3hip_fib_trace_dev(uint64_t a, uint64_t b, size_t n, uint32_t *d_out, void *stream);

/* ---- Poseidon2-BabyBear-16 (default_babybear_poseidon2_16, native/src/poseidon_cpu.rs:17-18) ----- */
/* n independent width-16 states, in place. */
int p3hip_poseidon2_permute_dev(uint32_t *d_states, size_t n, void *stream);
int p3hip_poseidon2_permute(uint32_t *states, size_t n);
/* The permutation exists in two arithmetic forms that must agree word for word: int32 Montgomery (variant 0) and exact
 * integer arithmetic in fp64 (variant 1: what the large tree layers run).  Diagnostics / tests:
 *   _variant_dev  the chosen form on n states in place;
 *   _f64_probe    the fp64 form on integer-valued DOUBLES (16 per state) of any magnitude its contract allows, canonical
 *                 Montgomery words out — mode 0: whole permutation (|v| <= 2^33), 1: the 13 internal rounds (|v| <= 2^37),
 *                 2: the modular reduction alone. */
int p3hip_poseidon2_permute_variant_dev(uint32_t *d_states, size_t n, int variant, void *stream);
int p3hip_poseidon2_f64_probe_dev(const double *d_in, uint32_t *d_out, size_t n, int mode, void *stream);

/* ---- Mmcs<BabyBear>: MerkleTreeMmcs<Poseidon2 sponge 16/8/8, TruncatedPermutation 2/8/16, digest 8>
 *      (the Poseidon2 analogue of the Keccak MMCS wired at native/src/fib_air.rs:31-51) ------------ */
typedef struct p3hip_tree p3hip_tree_t;
/* Mmcs::commit: matrices are device pointers (row-major, power-of-two heights); the tree keeps every
 * digest layer in HBM and BORROWS the matrices (they must outlive the tree).  root_out is a host buffer;
 * the call synchronises the stream before returning the root. */
int p3hip_mmcs_commit_dev(const uint32_t *const *d_mats, const size_t *heights, const size_t *widths,
                          size_t n_mats, uint32_t root_out[8], p3hip_tree_t **tree_out, void *stream);
/* as above without the root download/synchronise (root readable later via p3hip_mmcs_root) */
int p3hip_mmcs_commit_async_dev(const uint32_t *const *d_mats, const size_t *heights, const size_t *widths,
                                size_t n_mats, p3hip_tree_t **tree_out, void *stream);
int p3hip_mmcs_root(const p3hip_tree_t *tree, uint32_t root_out[8], void *stream);
size_t p3hip_mmcs_log_max_height(const p3hip_tree_t *tree);
size_t p3hip_mmcs_num_layers(const p3hip_tree_t *tree);
/* device pointer to digest layer `layer` (layer 0 = leaf digests), 8 words per digest */
const uint32_t *p3hip_mmcs_layer_dev(const p3hip_tree_t *tree, size_t layer, size_t *len_out);
/* Mmcs::open_batch: opened rows of every matrix (concatenated, host buffer of sum(widths) words) and the
 * sibling path (host buffer of log_max_height*8 words). */
int p3hip_mmcs_open_batch(const p3hip_tree_t *tree, size_t index, uint32_t *rows_out, uint32_t *path_out,
                          void *stream);
void p3hip_mmcs_free(p3hip_tree_t *tree);
/* host-pointer convenience: uploads the matrices, commits, keeps its own device copies inside the tree */
int p3hip_mmcs_commit(const uint32_t *const *mats, const size_t *heights, const size_t *widths,
                      size_t n_mats, uint32_t root_out[8], p3hip_tree_t **tree_out);

/* ---- the reference's own hash configuration (native/src/fib_air.rs:28-38): U64Hash = PaddingFreeSponge<KeccakF, 25,
 *      17, 4>, FieldHash = SerializingHasher<U64Hash>, MyCompress = CompressionFunctionFromHasher<U64Hash, 2, 4>.
 *      Digests are [u64; 4], stored as 8 little-endian u32 words, so trees of both configurations share
 *      p3hip_mmcs_root / open_batch / layer_dev / free.  Non-hiding (MerkleTreeMmcs, not MerkleTreeHidingMmcs). ---- */
#define P3HIP_HASH_POSEIDON2 0
#define P3HIP_HASH_KECCAK 1
int p3hip_mmcs_commit_hash_dev(int hash, const uint32_t *const *d_mats, const size_t *heights, const size_t *widths,
                               size_t n_mats, uint32_t root_out[8], p3hip_tree_t **tree_out, void *stream);
/* host-pointer convenience, as p3hip_mmcs_commit */
int p3hip_mmcs_commit_hash(int hash, const uint32_t *const *mats, const size_t *heights, const size_t *widths,
                           size_t n_mats, uint32_t root_out[8], p3hip_tree_t **tree_out);
/* Mmcs::commit into CALLER-PROVIDED digest-layer storage of p3hip_mmcs_layer_words(max height) 32-bit words (leaf layer
 * first, 8 words per digest): nothing is allocated and nothing synchronises — the call only enqueues.  The tree borrows the
 * storage and the matrices; read the root with p3hip_mmcs_root. */
size_t p3hip_mmcs_layer_words(size_t max_height);
int p3hip_mmcs_commit_into_dev(int hash, const uint32_t *const *d_mats, const size_t *heights, const size_t *widths,
                               size_t n_mats, uint32_t *d_layers, p3hip_tree_t **tree_out, void *stream);
/* KeccakF::permute_mut on n independent [u64; 25] states in device memory (p3-keccak's KeccakF, fib_air.rs:32) */
int p3hip_keccak_f_dev(uint64_t *d_states, size_t n, void *stream);

/* ---- fib_air prover: p3_uni_stark::prove(&config, &FibonacciAir{}, trace, &pis) as called at
 *      native/src/fib_air.rs:70, for StarkConfig<TwoAdicFriPcs<BabyBear, Dft, Poseidon2 Mmcs, ExtensionMmcs>,
 *      BinomialExtensionField<BabyBear,4>, DuplexChallenger<BabyBear, Poseidon2-16, 16, 8>> ---------------- */
typedef struct {
    uint32_t log_blowup;          /* p3_fri::FriParameters::log_blowup */
    uint32_t log_final_poly_len;  /* ::log_final_poly_len (create_test_fri_params(mmcs, 2) in fib_air.rs:62) */
    uint32_t num_queries;
    uint32_t proof_of_work_bits;
} p3hip_fri_params_t;
typedef struct p3hip_fib_prover p3hip_fib_prover_t;
/* PROFILES — chosen when an object is created, as the reference chooses its backend when GpuDft is constructed
 * (native/src/gpu_dft.rs:85-92 `with_backend`); nothing is read from the environment.  LATENCY: one proof at a time, which is what
 * the reference does (app/src/main/java/com/plonky3/android/MainActivity.kt:29-33, native/src/fib_air.rs:56-72): layers of
 * 2^10..2^15 digests use the forms that shorten a lone proof's chain of dependent launches (one Poseidon2 state per DPP quad,
 * cooperative Keccak up to 2^12 digests), the hiding prover commits its randomization polynomial on a second side stream, the FRI
 * rounds of at most 2^7 rows run in one single-workgroup launch, and a hiding proof whose LDE domain has at most 2^8 points — the
 * reference's own instance, n = 8 (fib_air.rs:56-57) — is ONE kernel launch of one workgroup.  THROUGHPUT: several provers share the chip and VALU issue is what
 * is short: the per-lane forms.  Proof bytes, digests and every intermediate are the same under both.
 * Defaults: p3hip_fib_prover_create* and p3hip_run_fib_air_zk = LATENCY; p3hip_fib_batch_create* with more than one prover =
 * THROUGHPUT; free functions (p3hip_mmcs_commit*) = the calling thread's profile (LATENCY until p3hip_set_thread_profile). */
#define P3HIP_PROFILE_THROUGHPUT 1
#define P3HIP_PROFILE_LATENCY 2
int p3hip_set_thread_profile(int profile);
int p3hip_get_thread_profile(void);  /* -1 when the thread has no device context (no HIP device) */
/* The general creation entry: any hash (P3HIP_HASH_*), hiding != 0 for the reference's MerkleTreeHidingMmcs + HidingFriPcs
 * (then `seed` seeds its SmallRng streams), any profile.  The create functions below are this one with P3HIP_PROFILE_LATENCY. */
int p3hip_fib_prover_create_profile(int profile, int hash, int hiding, uint64_t seed, unsigned log_n, const p3hip_fri_params_t *params,
                                    void *stream, int own_stream, p3hip_fib_prover_t **out);
/* Allocates the prover's HBM arena for 2^log_n-row traces.  stream: hipStream_t to enqueue on, or pass
 * own_stream != 0 to let the prover create (and own) a non-blocking stream — one prover per host thread. */
int p3hip_fib_prover_create(unsigned log_n, const p3hip_fri_params_t *params, void *stream, int own_stream,
                            p3hip_fib_prover_t **out);
/* Proves the instance whose first trace row is (a, b) (generate_trace_rows(a, b, 2^log_n), public values
 * [a, b, last right value], fib_air.rs:61,68).  The returned bytes stay valid until the next prove/destroy.
 * Wire format: DESIGN.md "proof bytes". */
int p3hip_fib_prover_prove(p3hip_fib_prover_t *prover, uint64_t a, uint64_t b, const uint8_t **proof_out,
                           size_t *proof_len);
/* The same with the proof handed over in the CALLER's buffer (e.g. the pinned staging row of a gather): the prover serialises
 * into its own buffer and copies ONCE into `out` (what is saved are the caller-side copies — a Python bytes object and its copy
 * into the staging row).  Returns ERR_BAD_ARG when cap is too small, the needed size in *proof_len: checked BEFORE proving once
 * the prover has produced a proof (proofs of one prover have one length); on a first call that fails this way the proof has been
 * computed and discarded.  A retry returns the same bytes (the hiding prover restarts its streams from the seed for every proof). */
int p3hip_fib_prover_prove_into(p3hip_fib_prover_t *prover, uint64_t a, uint64_t b, uint8_t *out, size_t cap, size_t *proof_len);
/* The same in two halves, to keep the prover's stream busy across proofs: enqueue returns as soon as the proof's launches
 * are queued (nothing is waited for), finish waits for the OLDEST enqueued proof and returns its bytes (valid until the next
 * prove / finish / destroy).  At most two proofs may be in flight; the second one's kernels queue behind the first on the
 * prover's stream (the arena is reused in stream order, the results land in two pinned buffers), so the host's turnaround
 * between proofs — wake-up, serialisation, the next enqueue: 0.2-0.3 ms at 2^20 — no longer leaves the stream empty.
 * Not available for the hiding prover. */
int p3hip_fib_prover_enqueue(p3hip_fib_prover_t *prover, uint64_t a, uint64_t b);
int p3hip_fib_prover_finish(p3hip_fib_prover_t *prover, const uint8_t **proof_out, size_t *proof_len);
/* host wall-clock per stage [trace commit, quotient commit, open, FRI commit phase, grind, queries] in ms,
 * accumulated over *proofs proofs */
int p3hip_fib_prover_stage_times(p3hip_fib_prover_t *prover, double out_ms[6], uint64_t *proofs, int reset);
/* Diagnostics of the proof-of-work continuation path (no reference counterpart: p3_fri grinds on the host).  *misses = proofs
 * of this prover whose first device search range held no witness; indices_out[0..min(cap, *n_out)) = the device's query-index
 * buffer as it stood when the host learnt of the LAST miss, before the search continued: all zero by construction, because
 * the gather kernel queued behind the query kernel reads that buffer whatever the search returned. */
int p3hip_fib_prover_grind_miss_probe(p3hip_fib_prover_t *prover, uint64_t *misses, uint32_t *indices_out, size_t cap,
                                      size_t *n_out);
void p3hip_fib_prover_destroy(p3hip_fib_prover_t *prover);
/* verify(&config, &FibonacciAir{}, &proof, &pis) (native/src/fib_air.rs:71-72) with pis = [a, b, x]: host-side, a few
 * thousand permutations.  Returns 0 to accept, a positive code naming the failed check otherwise (message via
 * p3hip_take_last_error, e.g. "fib_air verification failed: OodEvaluationMismatch"). */
int p3hip_verify_fib_air(const uint8_t *proof, size_t len, uint64_t a, uint64_t b, uint64_t x, unsigned log_n,
                         const p3hip_fri_params_t *params);

/* The same prover / verifier under the reference's own hashes (hash = P3HIP_HASH_KECCAK; native/src/fib_air.rs:28-53):
 * Keccak MMCS + SerializingChallenger32<BabyBear, HashChallenger<u8, Keccak256Hash, 32>>, non-hiding.  Same wire
 * format; digests are [u64; 4] as 8 little-endian u32 words. */
int p3hip_fib_prover_create_hash(int hash, unsigned log_n, const p3hip_fri_params_t *params, void *stream, int own_stream,
                                 p3hip_fib_prover_t **out);
int p3hip_verify_fib_air_hash(int hash, const uint8_t *proof, size_t len, uint64_t a, uint64_t b, uint64_t x,
                              unsigned log_n, const p3hip_fri_params_t *params);

/* ---- the HIDING half of the reference's configuration (native/src/fib_air.rs:40-65):
 *      MerkleTreeHidingMmcs<.., SmallRng, .., SALT_ELEMS 4> (rng = SmallRng::seed_from_u64(1)) and
 *      HidingFriPcs::new(dft, val_mmcs, fri_params, num_random_codewords 4, SmallRng::seed_from_u64(1)), p3_uni_stark with
 *      SC::Pcs::ZK.  Randomized trace, blinded quotient chunks, randomization polynomial, salted leaves; the random
 *      streams are generated on the device.  Wire format version 2.  The protocol details are recalled from
 *      the absent upstream crates: parity unpinned. ---- */
int p3hip_fib_prover_create_hiding(int hash, unsigned log_n, const p3hip_fri_params_t *params, uint64_t seed, void *stream,
                                   int own_stream, p3hip_fib_prover_t **out);
int p3hip_verify_fib_air_hiding(int hash, const uint8_t *proof, size_t len, uint64_t a, uint64_t b, uint64_t x,
                                unsigned log_n, const p3hip_fri_params_t *params);
/* rand 0.9.2 `SmallRng::seed_from_u64(seed)` (xoshiro256++ behind SplitMix64) as a device-resident stream of BabyBear
 * elements (Montgomery words), exactly the sequence a host loop over `rng.random::<BabyBear>()` yields.  One stream is
 * used from one HIP stream at a time. */
typedef struct p3hip_rng p3hip_rng_t;
int p3hip_rng_create(uint64_t seed, p3hip_rng_t **out);
int p3hip_rng_fill_field_dev(p3hip_rng_t *rng, uint32_t *d_out, size_t n, void *stream);
/* synchronises `stream` and returns the generator state (s[0..4)) after everything enqueued so far */
int p3hip_rng_state(p3hip_rng_t *rng, uint64_t state_out[4], void *stream);
void p3hip_rng_destroy(p3hip_rng_t *rng);
/* MerkleTreeHidingMmcs::commit (native/src/fib_air.rs:40-51, SALT_ELEMS = 4): every matrix is paired with a
 * height x 4 matrix of draws from `rng` (in input order), leaf rows are m0 || s0 || m1 || s1 ...; the tree owns the salt
 * matrices.  p3hip_mmcs_open_batch on such a tree returns the rows in that interleaved order (2 n_mats entries: the
 * caller splits values from salts, as MerkleTreeHidingMmcs::open_batch does); root / layers / free are shared. */
int p3hip_mmcs_commit_hiding_dev(int hash, const uint32_t *const *d_mats, const size_t *heights, const size_t *widths,
                                 size_t n_mats, p3hip_rng_t *rng, uint32_t root_out[8], p3hip_tree_t **tree_out, void *stream);

/* ---- batches of independent proofs (BASELINE configs[3]; SURVEY.md §8e: instance i is self-contained) ------
 * A pool of n_provers provers, each on its own host thread (thread-local context, as the reference's runtime,
 * backend_vulkan.rs:100-102) and its own stream, so the transcript round trips of one proof hide behind the
 * kernels of the others.  p3hip_fib_batch_prove proves instances (a[i], b[i]) and returns pointers to the proof
 * bytes, valid until the next call on the same batch. */
typedef struct p3hip_fib_batch p3hip_fib_batch_t;
/* Profile of the pool's provers: THROUGHPUT when n_provers > 1, LATENCY for a pool of one. */
int p3hip_fib_batch_create(unsigned log_n, const p3hip_fri_params_t *params, unsigned n_provers,
                           p3hip_fib_batch_t **out);
/* the pool under either hash configuration (P3HIP_HASH_POSEIDON2 / P3HIP_HASH_KECCAK) */
int p3hip_fib_batch_create_hash(int hash, unsigned log_n, const p3hip_fri_params_t *params, unsigned n_provers,
                                p3hip_fib_batch_t **out);
/* the pool in the reference's hiding configuration (fib_air.rs:40-65): every prover's SmallRng streams start from `seed`
 * for every proof, as a freshly built config would */
int p3hip_fib_batch_create_hiding(int hash, unsigned log_n, const p3hip_fri_params_t *params, uint64_t seed, unsigned n_provers,
                                  p3hip_fib_batch_t **out);
int p3hip_fib_batch_prove(p3hip_fib_batch_t *batch, size_t n, const uint64_t *a, const uint64_t *b,
                          const uint8_t **proofs_out, size_t *lens_out);
/* The same in two halves, for callers that keep the pool busy: submit copies the instance list, queues the batch behind
 * the ones already submitted and returns a ticket at once; collect waits for that batch (any order) and hands out the
 * proof pointers, valid until the next collect / prove / destroy on this pool.  A prover that has finished its share of
 * one batch starts on the next without waiting for the batch to complete — joining the provers after every batch costs
 * ~10 % at 64 proofs per batch (464 against ~520 proofs/s at 2^20).  At most 8 batches may be in flight.  A failed proof
 * fails its own batch's collect, not the others. */
int p3hip_fib_batch_submit(p3hip_fib_batch_t *batch, size_t n, const uint64_t *a, const uint64_t *b, uint64_t *ticket_out);
int p3hip_fib_batch_collect(p3hip_fib_batch_t *batch, uint64_t ticket, const uint8_t **proofs_out, size_t *lens_out);
void p3hip_fib_batch_destroy(p3hip_fib_batch_t *batch);

/* ---- The reference's report-returning entry points (native/src/lib.rs:37-131 call fib_air::run_fib_air_zk /
 * fib_air::run_dft_benchmark and hand the returned String to Java).  Both write a NUL-terminated text of at most cap - 1 bytes
 * to out and return the length of the WHOLE text (snprintf convention); they never fail by status: a failure is text that
 * contains "failed", as in the JNI wrappers (lib.rs:48,104), and a message waiting in the error mailbox is appended as
 * "\nHIP error: ..." (lib.rs:62-65 appends "\nVulkan error: ..."). ---- */
/* run_fib_air_zk (native/src/fib_air.rs:27-75): the reference's own instance and configuration — n = 8, x = 21, Keccak hashes,
 * MerkleTreeHidingMmcs + HidingFriPcs seeded with 1, create_test_fri_params(_, 2) — proved on the device and verified on the
 * host: "fib_air zk ok (n=8, x=21)" (fib_air.rs:74) or "fib_air zk failed: <check>".  Honours the selector: with a backend other
 * than "hip" selected nothing is run and the text says so (fib_air.rs:60 hard-codes Vulkan; see integration/native/src/fib_air.rs.patch). */
int p3hip_run_fib_air_zk(char *out, size_t cap);
/* The CPU column of the benchmark is the caller's: the reference times Plonky3's Radix2DitParallel (fib_air.rs:101,137-141),
 * which libp3hip does not contain (no CPU path in the product).  Returns 0 on success; Montgomery words, natural row order. */
typedef int (*p3hip_cpu_dft_fn)(void *user, const uint32_t *in, uint32_t *out, size_t height, size_t width);
/* run_dft_benchmark (native/src/fib_air.rs:98-222): the reference's 11 shapes, warmup 1, repeats 10, batches of 4; per shape
 * avg/median/p95 of hip_e2e (host matrix in and out), hip_e2e_batched (4 transforms per synchronisation) and hip_kernel (device
 * resident, HIP events), the speedups over the caller's CPU transform and the reference's equality check (fib_air.rs:193-196:
 * "dft benchmark failed: dft benchmark mismatch at h=.., w=.."); cpu_dft == NULL leaves the CPU column and the check out. */
int p3hip_run_dft_benchmark(p3hip_cpu_dft_fn cpu_dft, void *user, char *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* P3HIP_H */
