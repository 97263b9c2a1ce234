// NEVER COMPILED HERE (no Rust toolchain in the build image; p3-* 0.4.2 path dependencies absent).
//
// native/src/hip_mmcs.rs — `HipMmcs`: `Mmcs<BabyBear>` whose Merkle trees are built and kept in HBM by libp3hip.
// The reference hands Plonky3's own MMCS to the PCS (native/src/fib_air.rs:40-51); this type takes its place:
//
//     let val_mmcs = HipMmcs::keccak(field_hash, compress);          // the reference's hashes (fib_air.rs:28-38)
//     let val_mmcs = HipMmcs::poseidon2(sponge, truncated_perm);     // north_star's configuration
//
// `commit` / `open_batch` run on the device (p3hip_mmcs_commit_hash / p3hip_mmcs_open_batch); `verify_batch` is the
// verifier's job and stays Plonky3's own CPU code (the wrapped `MerkleTreeMmcs`), exactly as in the reference.
// Trait shapes are those of p3-commit 0.4.2 as recalled (the crate is not in this container) — [UPSTREAM-RECALL].
use core::ffi::c_void;
use core::marker::PhantomData;

use p3_baby_bear::BabyBear;
use p3_commit::{BatchOpening, BatchOpeningRef, Mmcs};
use p3_field::PackedValue;
use p3_matrix::dense::RowMajorMatrix;
use p3_matrix::{Dimensions, Matrix};
use p3_merkle_tree::{MerkleTreeError, MerkleTreeMmcs};
use p3_symmetric::{CryptographicHasher, Hash, PseudoCompressionFunction};
use serde::{Deserialize, Serialize};

pub const P3HIP_HASH_POSEIDON2: i32 = 0;
pub const P3HIP_HASH_KECCAK: i32 = 1;

#[repr(C)]
pub struct p3hip_tree_t {
    _private: [u8; 0],
}
#[repr(C)]
pub struct p3hip_rng_t {
    _private: [u8; 0],
}

// include/p3hip.h
extern "C" {
    fn p3hip_take_last_error() -> *const core::ffi::c_char;
    fn p3hip_mmcs_commit_hash(
        hash: i32,
        mats: *const *const u32,
        heights: *const usize,
        widths: *const usize,
        n_mats: usize,
        root_out: *mut u32,
        tree_out: *mut *mut p3hip_tree_t,
    ) -> i32;
    fn p3hip_mmcs_commit_hash_dev(
        hash: i32,
        d_mats: *const *const u32,
        heights: *const usize,
        widths: *const usize,
        n_mats: usize,
        root_out: *mut u32,
        tree_out: *mut *mut p3hip_tree_t,
        stream: *mut c_void,
    ) -> i32;
    fn p3hip_malloc(dev_ptr: *mut *mut c_void, bytes: usize) -> i32;
    fn p3hip_free(dev_ptr: *mut c_void) -> i32;
    fn p3hip_upload(dev_dst: *mut c_void, host_src: *const c_void, bytes: usize) -> i32;
    fn p3hip_rng_create(seed: u64, out: *mut *mut p3hip_rng_t) -> i32;
    fn p3hip_rng_destroy(rng: *mut p3hip_rng_t);
    fn p3hip_mmcs_commit_hiding_dev(
        hash: i32,
        d_mats: *const *const u32,
        heights: *const usize,
        widths: *const usize,
        n_mats: usize,
        rng: *mut p3hip_rng_t,
        root_out: *mut u32,
        tree_out: *mut *mut p3hip_tree_t,
        stream: *mut c_void,
    ) -> i32;
    fn p3hip_mmcs_open_batch(tree: *const p3hip_tree_t, index: usize, rows_out: *mut u32, path_out: *mut u32, stream: *mut c_void) -> i32;
    fn p3hip_mmcs_log_max_height(tree: *const p3hip_tree_t) -> usize;
    fn p3hip_mmcs_free(tree: *mut p3hip_tree_t);
}

fn last_error() -> String {
    unsafe {
        let p = p3hip_take_last_error();
        if p.is_null() {
            "hip backend error".to_string()
        } else {
            std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}

/// Prover data: the committed matrices (host copies, for `get_matrices`) and the device tree (digest layers + device
/// copies of the matrices, owned by libp3hip).  Freed with the handle.
pub struct HipTree<M> {
    handle: *mut p3hip_tree_t,
    mats: Vec<M>,
    widths: Vec<usize>,
    /// device copies the tree was built over when every input was already resident (hip_matrix.rs): kept alive with the tree
    resident: Vec<crate::hip_matrix::ResidentWords>,
}
unsafe impl<M: Send> Send for HipTree<M> {}
unsafe impl<M: Sync> Sync for HipTree<M> {}
impl<M> Drop for HipTree<M> {
    fn drop(&mut self) {
        unsafe { p3hip_mmcs_free(self.handle) }
    }
}

/// A digest word: `BabyBear` (Poseidon2: 8 field elements) or `u64` (Keccak: `[u64; 4]`).  On the C side every digest
/// is 8 little-endian `u32` words; for the Keccak configuration two of them make one `u64`, low half first.
pub trait DigestWord: Copy + Default + Send + Sync + Serialize + for<'de> Deserialize<'de> + 'static {
    const HASH_KIND: i32;
    const DIGEST_ELEMS: usize;
    fn from_words(words: &[u32; 8]) -> Vec<Self>;
}
impl DigestWord for BabyBear {
    const HASH_KIND: i32 = P3HIP_HASH_POSEIDON2;
    const DIGEST_ELEMS: usize = 8;
    fn from_words(words: &[u32; 8]) -> Vec<Self> {
        // Montgomery words, the representation BabyBear is repr(transparent) over (backend_vulkan.rs:2002-2005)
        words.iter().map(|w| unsafe { core::mem::transmute::<u32, BabyBear>(*w) }).collect()
    }
}
impl DigestWord for u64 {
    const HASH_KIND: i32 = P3HIP_HASH_KECCAK;
    const DIGEST_ELEMS: usize = 4;
    fn from_words(words: &[u32; 8]) -> Vec<Self> {
        (0..4).map(|i| words[2 * i] as u64 | ((words[2 * i + 1] as u64) << 32)).collect()
    }
}

/// `Mmcs<BabyBear>` on the hip backend.  `W` is the digest word, `N` the digest length: `HipMmcs<.., BabyBear, 8>` for
/// Poseidon2 (PaddingFreeSponge<Perm, 16, 8, 8> + TruncatedPermutation<Perm, 2, 8, 16>), `HipMmcs<.., u64, 4>` for the
/// reference's Keccak configuration (SerializingHasher<PaddingFreeSponge<KeccakF, 25, 17, 4>> +
/// CompressionFunctionFromHasher<_, 2, 4>).  `inner` is Plonky3's own tree over the SAME hasher and compressor: it
/// verifies openings (CPU, verifier side) and pins the hash configuration the device tree must reproduce.
#[derive(Clone, Debug)]
pub struct HipMmcs<P, PW, H, C, W, const N: usize> {
    inner: MerkleTreeMmcs<P, PW, H, C, N>,
    _w: PhantomData<W>,
}

impl<P, PW, H, C, W, const N: usize> HipMmcs<P, PW, H, C, W, N> {
    pub fn new(hash: H, compress: C) -> Self {
        Self { inner: MerkleTreeMmcs::new(hash, compress), _w: PhantomData }
    }
}

impl<P, PW, H, C, W, const N: usize> Mmcs<BabyBear> for HipMmcs<P, PW, H, C, W, N>
where
    P: PackedValue<Value = BabyBear>,
    PW: PackedValue<Value = W>,
    W: DigestWord + Eq,
    H: CryptographicHasher<BabyBear, [W; N]> + CryptographicHasher<P, [PW; N]> + Sync + Clone,
    C: PseudoCompressionFunction<[W; N], 2> + PseudoCompressionFunction<[PW; N], 2> + Sync + Clone,
    [W; N]: Serialize + for<'de> Deserialize<'de>,
{
    type ProverData<M> = HipTree<M>;
    type Commitment = Hash<BabyBear, W, N>;
    type Proof = Vec<[W; N]>;
    type Error = MerkleTreeError;

    fn commit<M: Matrix<BabyBear>>(&self, inputs: Vec<M>) -> (Self::Commitment, Self::ProverData<M>) {
        assert_eq!(N, W::DIGEST_ELEMS);
        let heights: Vec<usize> = inputs.iter().map(|m| m.height()).collect();
        let widths: Vec<usize> = inputs.iter().map(|m| m.width()).collect();
        let mut root = [0u32; 8];
        let mut handle: *mut p3hip_tree_t = core::ptr::null_mut();
        // Resident inputs: a dense host matrix that `HipMatrix::to_row_major_matrix` downloaded (what TwoAdicFriPcs::commit hands
        // over after GpuDft's hip arm produced the LDE) still has its device copy registered under the address of its words.
        // `row_slice(0)` of a dense matrix derefs into those words; of any other matrix type into a temporary, which is simply
        // not in the registry.  When EVERY input is resident the tree is built over the device copies: no upload at all.
        // A hit is VERIFIED — address, length and a sampled fingerprint read through the Matrix trait (hip_matrix.rs `fingerprint_of`),
        // plus a full compare against the device words in debug builds — and the registry is only touched when EVERY input matched
        // (`take_residents`): a stale entry at a reused address, or a mix of resident and foreign inputs, takes the upload path below.
        let queries: Vec<crate::hip_matrix::ResidentQuery> = inputs
            .iter()
            .filter(|m| m.height() > 0)
            .map(|m| {
                let first = m.row_slice(0);
                crate::hip_matrix::ResidentQuery {
                    host_addr: first.as_ptr() as usize,
                    len: m.height() * m.width(),
                    fingerprint: crate::hip_matrix::fingerprint_of(m),
                }
            })
            .collect();
        let mut resident: Vec<crate::hip_matrix::ResidentWords> = if queries.len() == inputs.len() {
            crate::hip_matrix::take_residents(&queries).unwrap_or_default()
        } else {
            Vec::new()
        };
        #[cfg(debug_assertions)]
        {
            if resident.len() == inputs.len() && !resident.iter().zip(inputs.iter()).all(|(r, m)| r.debug_same_words(m)) {
                resident.clear(); // fingerprint collision or a mutated host copy: commit what the caller actually handed over
            }
        }
        if resident.len() == inputs.len() {
            let ptrs: Vec<*const u32> = resident.iter().map(|r| r.device_ptr()).collect();
            let rc = unsafe {
                p3hip_mmcs_commit_hash_dev(W::HASH_KIND, ptrs.as_ptr(), heights.as_ptr(), widths.as_ptr(), inputs.len(), root.as_mut_ptr(), &mut handle, core::ptr::null_mut())
            };
            assert!(rc == 0, "hip mmcs commit failed: {}", last_error());
            let digest: [W; N] = W::from_words(&root).try_into().ok().expect("digest length");
            return (Hash::from(digest), HipTree { handle, mats: inputs, widths, resident });
        }
        // otherwise: row-major Montgomery words of every matrix (BabyBear is repr(transparent) over u32), uploaded by the library
        let dense: Vec<RowMajorMatrix<BabyBear>> = inputs.iter().map(|m| RowMajorMatrix::new(m.rows().flatten().collect(), m.width())).collect();
        let ptrs: Vec<*const u32> = dense.iter().map(|m| m.values.as_ptr() as *const u32).collect();
        let rc = unsafe {
            p3hip_mmcs_commit_hash(W::HASH_KIND, ptrs.as_ptr(), heights.as_ptr(), widths.as_ptr(), dense.len(), root.as_mut_ptr(), &mut handle)
        };
        // Mmcs::commit is infallible in Plonky3: a device failure is a panic here, caught by the JNI wrapper like any
        // other prover panic (native/src/lib.rs:45-59)
        assert!(rc == 0, "hip mmcs commit failed: {}", last_error());
        let digest: [W; N] = W::from_words(&root).try_into().ok().expect("digest length");
        (Hash::from(digest), HipTree { handle, mats: inputs, widths, resident: Vec::new() })
    }

    fn open_batch<M: Matrix<BabyBear>>(&self, index: usize, prover_data: &Self::ProverData<M>) -> BatchOpening<BabyBear, Self> {
        let total: usize = prover_data.widths.iter().sum();
        let depth = unsafe { p3hip_mmcs_log_max_height(prover_data.handle) };
        let mut rows = vec![0u32; total.max(1)];
        let mut path = vec![0u32; (depth * 8).max(8)];
        let rc = unsafe { p3hip_mmcs_open_batch(prover_data.handle, index, rows.as_mut_ptr(), path.as_mut_ptr(), core::ptr::null_mut()) };
        assert!(rc == 0, "hip mmcs open_batch failed: {}", last_error());
        let mut opened_values = Vec::with_capacity(prover_data.widths.len());
        let mut off = 0;
        for &w in &prover_data.widths {
            opened_values.push(rows[off..off + w].iter().map(|v| unsafe { core::mem::transmute::<u32, BabyBear>(*v) }).collect());
            off += w;
        }
        let opening_proof = (0..depth)
            .map(|l| {
                let words: [u32; 8] = path[8 * l..8 * l + 8].try_into().unwrap();
                let d: [W; N] = W::from_words(&words).try_into().ok().expect("digest length");
                d
            })
            .collect();
        BatchOpening::new(opened_values, opening_proof)
    }

    fn get_matrices<'a, M: Matrix<BabyBear>>(&self, prover_data: &'a Self::ProverData<M>) -> Vec<&'a M> {
        prover_data.mats.iter().collect()
    }

    fn verify_batch(
        &self,
        commit: &Self::Commitment,
        dimensions: &[Dimensions],
        index: usize,
        batch_opening: BatchOpeningRef<'_, BabyBear, Self>,
    ) -> Result<(), Self::Error> {
        // the verifier side: Plonky3's own CPU code over the same hasher / compressor
        let (opened_values, opening_proof) = batch_opening.unpack();
        self.inner.verify_batch(commit, dimensions, index, BatchOpeningRef::new(opened_values, opening_proof))
    }
}

/// The reference's hash configuration (native/src/fib_air.rs:28-38), non-hiding: digests `[u64; 4]`.
pub type HipKeccakMmcs<FieldHash, MyCompress> =
    HipMmcs<[BabyBear; p3_keccak::VECTOR_LEN], [u64; p3_keccak::VECTOR_LEN], FieldHash, MyCompress, u64, 4>;
/// north_star's configuration: Poseidon2 sponge 16/8/8 + TruncatedPermutation 2/8/16, digests of 8 field elements.
pub type HipPoseidon2Mmcs<Sponge, Compress> =
    HipMmcs<<BabyBear as p3_field::Field>::Packing, <BabyBear as p3_field::Field>::Packing, Sponge, Compress, BabyBear, 8>;

// ---------------------------------------------------------------------------------------------------------------
// MerkleTreeHidingMmcs<.., SmallRng, DIGEST, SALT_ELEMS = 4> (native/src/fib_air.rs:40-51) on the hip backend.
// The reference seeds the MMCS's rng with SmallRng::seed_from_u64(1) (fib_air.rs:50); here the stream lives in HBM
// (p3hip_rng_create: xoshiro256++ behind SplitMix64, the same sequence) and `commit` draws the height x 4 salt matrix
// of every input from it, in input order.  Proof = (salts per matrix, sibling digests), as upstream.
pub const SALT_ELEMS: usize = 4;

pub struct HipHidingTree<M> {
    handle: *mut p3hip_tree_t,
    device_mats: Vec<*mut c_void>,
    mats: Vec<M>,
    widths: Vec<usize>,
}
impl<M> Drop for HipHidingTree<M> {
    fn drop(&mut self) {
        unsafe {
            p3hip_mmcs_free(self.handle);
            for p in &self.device_mats {
                p3hip_free(*p);
            }
        }
    }
}

pub struct HipHidingMmcs<P, PW, H, C, W, const N: usize> {
    inner: p3_merkle_tree::MerkleTreeHidingMmcs<P, PW, H, C, rand::rngs::SmallRng, N, SALT_ELEMS>,
    rng: *mut p3hip_rng_t,
    seed: u64,
    _w: PhantomData<W>,
}
impl<P, PW, H: Clone, C: Clone, W, const N: usize> HipHidingMmcs<P, PW, H, C, W, N> {
    /// `ValHidingMmcs::new(field_hash, compress, SmallRng::seed_from_u64(seed))` (fib_air.rs:50-51)
    pub fn new(hash: H, compress: C, seed: u64) -> Self {
        use rand::SeedableRng;
        let mut rng: *mut p3hip_rng_t = core::ptr::null_mut();
        let rc = unsafe { p3hip_rng_create(seed, &mut rng) };
        assert!(rc == 0, "hip rng create failed: {}", last_error());
        Self {
            inner: p3_merkle_tree::MerkleTreeHidingMmcs::new(hash, compress, rand::rngs::SmallRng::seed_from_u64(seed)),
            rng,
            seed,
            _w: PhantomData,
        }
    }
}
// `challenge_mmcs = ExtensionMmcs::new(val_mmcs.clone())` (fib_air.rs:59): a clone restarts the stream from the seed
impl<P, PW, H: Clone, C: Clone, W, const N: usize> Clone for HipHidingMmcs<P, PW, H, C, W, N>
where
    p3_merkle_tree::MerkleTreeHidingMmcs<P, PW, H, C, rand::rngs::SmallRng, N, SALT_ELEMS>: Clone,
{
    fn clone(&self) -> Self {
        let mut rng: *mut p3hip_rng_t = core::ptr::null_mut();
        let rc = unsafe { p3hip_rng_create(self.seed, &mut rng) };
        assert!(rc == 0, "hip rng create failed: {}", last_error());
        Self { inner: self.inner.clone(), rng, seed: self.seed, _w: PhantomData }
    }
}
impl<P, PW, H, C, W, const N: usize> Drop for HipHidingMmcs<P, PW, H, C, W, N> {
    fn drop(&mut self) {
        unsafe { p3hip_rng_destroy(self.rng) }
    }
}

impl<P, PW, H, C, W, const N: usize> Mmcs<BabyBear> for HipHidingMmcs<P, PW, H, C, W, N>
where
    P: PackedValue<Value = BabyBear>,
    PW: PackedValue<Value = W>,
    W: DigestWord + Eq,
    H: CryptographicHasher<BabyBear, [W; N]> + CryptographicHasher<P, [PW; N]> + Sync + Clone,
    C: PseudoCompressionFunction<[W; N], 2> + PseudoCompressionFunction<[PW; N], 2> + Sync + Clone,
    [W; N]: Serialize + for<'de> Deserialize<'de>,
    p3_merkle_tree::MerkleTreeHidingMmcs<P, PW, H, C, rand::rngs::SmallRng, N, SALT_ELEMS>: Mmcs<
        BabyBear,
        Commitment = Hash<BabyBear, W, N>,
        Proof = (Vec<Vec<BabyBear>>, Vec<[W; N]>),
        Error = MerkleTreeError,
    >,
{
    type ProverData<M> = HipHidingTree<M>;
    type Commitment = Hash<BabyBear, W, N>;
    /// The first item is salts; the second is the usual Merkle proof (sibling digests).
    type Proof = (Vec<Vec<BabyBear>>, Vec<[W; N]>);
    type Error = MerkleTreeError;

    fn commit<M: Matrix<BabyBear>>(&self, inputs: Vec<M>) -> (Self::Commitment, Self::ProverData<M>) {
        let dense: Vec<RowMajorMatrix<BabyBear>> = inputs.iter().map(|m| m.to_row_major_matrix()).collect();
        let heights: Vec<usize> = dense.iter().map(|m| m.height()).collect();
        let widths: Vec<usize> = dense.iter().map(|m| m.width()).collect();
        let mut device_mats: Vec<*mut c_void> = Vec::with_capacity(dense.len());
        for m in &dense {
            let bytes = m.values.len() * 4;
            let mut d: *mut c_void = core::ptr::null_mut();
            let rc = unsafe { p3hip_malloc(&mut d, bytes) };
            assert!(rc == 0, "hip malloc failed: {}", last_error());
            let rc = unsafe { p3hip_upload(d, m.values.as_ptr() as *const c_void, bytes) };
            assert!(rc == 0, "hip upload failed: {}", last_error());
            device_mats.push(d);
        }
        let ptrs: Vec<*const u32> = device_mats.iter().map(|p| *p as *const u32).collect();
        let mut root = [0u32; 8];
        let mut handle: *mut p3hip_tree_t = core::ptr::null_mut();
        let rc = unsafe {
            p3hip_mmcs_commit_hiding_dev(
                W::HASH_KIND,
                ptrs.as_ptr(),
                heights.as_ptr(),
                widths.as_ptr(),
                dense.len(),
                self.rng,
                root.as_mut_ptr(),
                &mut handle,
                core::ptr::null_mut(),
            )
        };
        assert!(rc == 0, "hip hiding mmcs commit failed: {}", last_error());
        let digest: [W; N] = W::from_words(&root).try_into().ok().expect("digest length");
        (Hash::from(digest), HipHidingTree { handle, device_mats, mats: inputs, widths })
    }

    fn open_batch<M: Matrix<BabyBear>>(&self, index: usize, prover_data: &Self::ProverData<M>) -> BatchOpening<BabyBear, Self> {
        // the device tree's rows are m0 || s0 || m1 || s1 ...: split values from salts (MerkleTreeHidingMmcs::open_batch)
        let total: usize = prover_data.widths.iter().map(|w| w + SALT_ELEMS).sum();
        let depth = unsafe { p3hip_mmcs_log_max_height(prover_data.handle) };
        let mut rows = vec![0u32; total.max(1)];
        let mut path = vec![0u32; (depth * 8).max(8)];
        let rc = unsafe { p3hip_mmcs_open_batch(prover_data.handle, index, rows.as_mut_ptr(), path.as_mut_ptr(), core::ptr::null_mut()) };
        assert!(rc == 0, "hip mmcs open_batch failed: {}", last_error());
        let felt = |v: &u32| unsafe { core::mem::transmute::<u32, BabyBear>(*v) };
        let (mut opened_values, mut salts) = (Vec::new(), Vec::new());
        let mut off = 0;
        for &w in &prover_data.widths {
            opened_values.push(rows[off..off + w].iter().map(felt).collect::<Vec<_>>());
            salts.push(rows[off + w..off + w + SALT_ELEMS].iter().map(felt).collect::<Vec<_>>());
            off += w + SALT_ELEMS;
        }
        let siblings = (0..depth)
            .map(|l| {
                let words: [u32; 8] = path[8 * l..8 * l + 8].try_into().unwrap();
                let d: [W; N] = W::from_words(&words).try_into().ok().expect("digest length");
                d
            })
            .collect();
        BatchOpening::new(opened_values, (salts, siblings))
    }

    fn get_matrices<'a, M: Matrix<BabyBear>>(&self, prover_data: &'a Self::ProverData<M>) -> Vec<&'a M> {
        prover_data.mats.iter().collect()
    }

    fn verify_batch(
        &self,
        commit: &Self::Commitment,
        dimensions: &[Dimensions],
        index: usize,
        batch_opening: BatchOpeningRef<'_, BabyBear, Self>,
    ) -> Result<(), Self::Error> {
        let (opened_values, opening_proof) = batch_opening.unpack();
        self.inner.verify_batch(commit, dimensions, index, BatchOpeningRef::new(opened_values, opening_proof))
    }
}

/// The reference's own `ValHidingMmcs` (native/src/fib_air.rs:40-48) on the hip backend.
pub type HipKeccakHidingMmcs<FieldHash, MyCompress> =
    HipHidingMmcs<[BabyBear; p3_keccak::VECTOR_LEN], [u64; p3_keccak::VECTOR_LEN], FieldHash, MyCompress, u64, 4>;
