// NEVER COMPILED HERE (no Rust toolchain in the build image; p3-* 0.4.2 path dependencies absent).
//
// native/src/hip_matrix.rs — `HipMatrix<F>` (a matrix that LIVES IN HBM), `GpuEvaluations<F>` (what `GpuDft` returns once the
// hip arm exists) and the RESIDENCY REGISTRY that lets `HipMmcs::commit` find the device copy of a matrix it is handed.
//
// Why: the reference fixes `type Evaluations = RowMajorMatrix<F>` (native/src/gpu_dft.rs:95), so on the trait path every LDE is
// downloaded by `coset_lde_batch` and uploaded again by `Mmcs::commit` — two PCIe crossings of the largest object of the proof
// per commitment (2^21 x 2 words = 16 MiB at the benchmark size, 2^26 x 2 = 512 MiB at BASELINE configs[2]).
// `Evaluations` is an associated type (`BitReversibleMatrix<F>`), so the hip arm can return a device matrix instead; but
// upstream's `TwoAdicFriPcs::commit` ends its LDE chain with `.bit_reverse_rows().to_row_major_matrix()` and fixes
// `ProverData = InputMmcs::ProverData<RowMajorMatrix<Val>>` [UPSTREAM-RECALL, p3-fri 0.4.2 two_adic_pcs.rs], i.e. the matrix
// that reaches `Mmcs::commit` is a host `RowMajorMatrix` whatever `Evaluations` was — and upstream's prover reads that host
// copy afterwards (quotient values, openings), so ONE download per commitment belongs to the trait path.  What does not is the
// upload: `HipMatrix::to_row_major_matrix` downloads once and REGISTERS (host address of the words -> the device allocation);
// `HipMmcs::commit` looks its inputs up there and passes the resident pointer to `p3hip_mmcs_commit_hash_dev`, for ANY AIR
// proved through `p3_uni_stark::prove` (native/src/fib_air.rs:70).  Were upstream's PCS to drop the `.to_row_major_matrix()`
// the same types keep LDE -> commit entirely in HBM (commit takes `GpuEvaluations::Device` directly, rows download lazily).
// INTEGRATION.md section 5 states what each path costs per commit.
//
// Trait shapes are those of p3-matrix 0.4.2 as recalled (the crate is not in this container) — [UPSTREAM-RECALL]:
//   trait Matrix<T>: Send + Sync { fn width(&self) -> usize; fn height(&self) -> usize;
//                                  unsafe fn row_subseq_unchecked(..) / fn row(&self, r) -> impl Iterator<Item = T>; .. }
//   trait BitReversibleMatrix<T>: Matrix<T> { type BitRev: BitReversibleMatrix<T>; fn bit_reverse_rows(self) -> Self::BitRev; }
use core::ffi::c_void;
use core::marker::PhantomData;
use std::collections::VecDeque;
use std::sync::{Arc, Mutex, OnceLock};

use p3_field::Field;
use p3_matrix::bitrev::BitReversibleMatrix;
use p3_matrix::dense::RowMajorMatrix;
use p3_matrix::Matrix;

// include/p3hip.h
extern "C" {
    fn p3hip_take_last_error() -> *const core::ffi::c_char;
    fn p3hip_malloc(dev_ptr: *mut *mut c_void, bytes: usize) -> i32;
    fn p3hip_free(dev_ptr: *mut c_void) -> i32;
    fn p3hip_upload(dev_dst: *mut c_void, host_src: *const c_void, bytes: usize) -> i32;
    fn p3hip_download(host_dst: *mut c_void, dev_src: *const c_void, bytes: usize) -> i32;
    fn p3hip_bit_reverse_rows_dev(d_in: *const u32, d_out: *mut u32, height: usize, width: usize, stream: *mut c_void) -> i32;
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

/// One HBM allocation, freed when the last matrix that views it is dropped.
struct DeviceBuf {
    ptr: *mut c_void,
    bytes: usize,
}
unsafe impl Send for DeviceBuf {}
unsafe impl Sync for DeviceBuf {}
impl Drop for DeviceBuf {
    fn drop(&mut self) {
        unsafe {
            p3hip_free(self.ptr);
        }
    }
}

/// Row order of the words in HBM relative to the matrix the type presents.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum RowOrder {
    Natural,
    /// row r of the matrix is stored at index bitrev(r): what `p3hip_coset_lde_batch_bb31_dev(.., bit_reversed_out = 1)`
    /// writes and what `TwoAdicFriPcs::commit` commits
    BitReversed,
}

/// `height x width` Montgomery words of `F = BabyBear` (repr(transparent) over u32, backend_vulkan.rs:2002-2005) in HBM.
/// `Clone` is cheap (shared allocation).  Host access goes through ONE lazy download of the whole matrix, kept for the
/// lifetime of the value; nothing is downloaded if no host code reads a row (commit -> open on the device does not).
#[derive(Clone)]
pub struct HipMatrix<F> {
    buf: Arc<DeviceBuf>,
    height: usize,
    width: usize,
    order: RowOrder,
    host: Arc<OnceLock<Vec<F>>>,
    _f: PhantomData<F>,
}

impl<F: Field> core::fmt::Debug for HipMatrix<F> {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "HipMatrix({} x {}, {:?}, device {:p})", self.height, self.width, self.order, self.buf.ptr)
    }
}

impl<F: Field> HipMatrix<F> {
    /// Uninitialised device matrix for a kernel to fill.
    pub fn alloc(height: usize, width: usize, order: RowOrder) -> Result<Self, String> {
        let bytes = height * width * 4;
        let mut p: *mut c_void = core::ptr::null_mut();
        if unsafe { p3hip_malloc(&mut p, bytes.max(4)) } != 0 {
            return Err(last_error());
        }
        Ok(Self { buf: Arc::new(DeviceBuf { ptr: p, bytes }), height, width, order, host: Arc::new(OnceLock::new()), _f: PhantomData })
    }

    /// Upload of a host matrix (the trace: `prove()` hands a RowMajorMatrix in, native/src/fib_air.rs:69-70).
    pub fn from_host(mat: &RowMajorMatrix<F>) -> Result<Self, String> {
        let m = Self::alloc(mat.height(), mat.width(), RowOrder::Natural)?;
        if unsafe { p3hip_upload(m.buf.ptr, mat.values.as_ptr() as *const c_void, m.buf.bytes) } != 0 {
            return Err(last_error());
        }
        Ok(m)
    }

    pub fn device_ptr(&self) -> *const u32 {
        self.buf.ptr as *const u32
    }
    pub fn device_ptr_mut(&self) -> *mut u32 {
        self.buf.ptr as *mut u32
    }
    pub fn order(&self) -> RowOrder {
        self.order
    }

    /// The words as they lie in HBM, downloaded once.
    fn host_words(&self) -> &Vec<F> {
        self.host.get_or_init(|| {
            let mut v: Vec<F> = Vec::with_capacity(self.height * self.width);
            let rc = unsafe { p3hip_download(v.as_mut_ptr() as *mut c_void, self.buf.ptr, self.buf.bytes) };
            // Matrix::row is infallible upstream: a device failure is a panic here, caught by the JNI wrapper like any other
            // prover panic (native/src/lib.rs:45-59)
            assert!(rc == 0, "hip matrix download failed: {}", last_error());
            unsafe { v.set_len(self.height * self.width) };
            v
        })
    }

    fn stored_row(&self, r: usize) -> usize {
        match self.order {
            RowOrder::Natural => r,
            RowOrder::BitReversed => {
                let bits = self.height.trailing_zeros();
                if bits == 0 { 0 } else { r.reverse_bits() >> (usize::BITS - bits) }
            }
        }
    }

    /// Natural-order dense copy on the host (what `to_row_major_matrix` returns).
    pub fn download(&self) -> RowMajorMatrix<F> {
        let words = self.host_words();
        let mut out = Vec::with_capacity(words.len());
        for r in 0..self.height {
            let s = self.stored_row(r) * self.width;
            out.extend_from_slice(&words[s..s + self.width]);
        }
        RowMajorMatrix::new(out, self.width)
    }
}

impl<F: Field> Matrix<F> for HipMatrix<F> {
    fn width(&self) -> usize {
        self.width
    }
    fn height(&self) -> usize {
        self.height
    }
    unsafe fn get_unchecked(&self, r: usize, c: usize) -> F {
        self.host_words()[self.stored_row(r) * self.width + c]
    }
    unsafe fn row_subseq_unchecked(&self, r: usize, start: usize, end: usize) -> impl IntoIterator<Item = F, IntoIter = impl Iterator<Item = F> + Send + Sync> {
        let s = self.stored_row(r) * self.width;
        self.host_words()[s + start..s + end].iter().copied()
    }
    fn to_row_major_matrix(self) -> RowMajorMatrix<F>
    where
        Self: Sized,
        F: Clone,
    {
        // the PCS's `.bit_reverse_rows().to_row_major_matrix()` lands here with `order == Natural` over bit-reversed storage:
        // the host copy has the layout of the device words, and the pair is remembered for `HipMmcs::commit`
        let host = self.download();
        if self.order == RowOrder::Natural {
            register_resident(host.values.as_ptr() as usize, host.values.len(), fingerprint_of(&host), self.buf.clone());
        }
        host
    }
}

// ---- residency registry: host address of a dense matrix's words -> the HBM allocation holding the same words ----
// Bounded (the last 8 downloads; a prover commits each LDE right after producing it).  An address alone proves nothing: if the
// RowMajorMatrix that `to_row_major_matrix` returned is dropped or mutated before the commit, the allocator readily hands the same
// address to the next Vec of the same size, and a later, unrelated matrix would hit the stale entry — the tree would be built over
// the OLD device words while the tree's host rows are the new ones (round-4 advisor finding).  So a hit must be VERIFIED:
//   * every entry carries a 64-bit fingerprint of the matrix it was registered with (`fingerprint_of`: dimensions, the first and
//     last 16 elements and 64 strided samples, FNV-1a over their canonical values, read through the `Matrix` trait — so it can be
//     recomputed for ANY matrix type the commit is handed, without assuming its memory layout); a reused address with other
//     contents misses;
//   * under `cfg(debug_assertions)` `HipMmcs::commit` downloads the device copy and compares it in full (`debug_same_words`),
//     like the dft self-check of backend_hip.rs;
//   * looking up does not consume: `take_residents` removes entries only when EVERY input of the commit matched, so a commit with a
//     mix of resident and foreign inputs leaves the registry as it was.
pub struct ResidentQuery {
    pub host_addr: usize,
    pub len: usize,
    pub fingerprint: u64,
}
struct Resident {
    host_addr: usize,
    len: usize,
    fingerprint: u64,
    buf: Arc<DeviceBuf>,
}
/// Sampled content hash of a matrix, through the trait's accessors only.
pub fn fingerprint_of<F: Field, M: Matrix<F>>(m: &M) -> u64 {
    const OFFSET: u64 = 0xcbf2_9ce4_8422_2325;
    const PRIME: u64 = 0x0000_0100_0000_01b3;
    let (h, w) = (m.height(), m.width());
    let n = h * w;
    let mut acc = OFFSET ^ (h as u64) ^ ((w as u64) << 40);
    let mut eat = |i: usize| {
        // element i of the row-major order; hashed through its Debug-independent bytes: BabyBear is repr(transparent) over its
        // Montgomery u32 (backend_vulkan.rs:2002-2005), read here without assuming how the MATRIX stores it
        let e: F = unsafe { m.get_unchecked(i / w, i % w) };
        let word: u32 = unsafe { core::mem::transmute_copy::<F, u32>(&e) };
        for b in word.to_le_bytes() {
            acc = (acc ^ b as u64).wrapping_mul(PRIME);
        }
    };
    if n == 0 {
        return acc;
    }
    for i in 0..n.min(16) {
        eat(i);
    }
    for i in n.saturating_sub(16)..n {
        eat(i);
    }
    if n > 32 {
        let step = (n / 64).max(1);
        let mut i = step / 2;
        while i < n {
            eat(i);
            i += step;
        }
    }
    acc
}
fn registry() -> &'static Mutex<VecDeque<Resident>> {
    static R: OnceLock<Mutex<VecDeque<Resident>>> = OnceLock::new();
    R.get_or_init(|| Mutex::new(VecDeque::new()))
}
fn register_resident(host_addr: usize, len: usize, fingerprint: u64, buf: Arc<DeviceBuf>) {
    if let Ok(mut q) = registry().lock() {
        // one entry per address: a newer matrix at the same address replaces the older one
        q.retain(|r| r.host_addr != host_addr);
        if q.len() == 8 {
            q.pop_front();
        }
        q.push_back(Resident { host_addr, len, fingerprint, buf });
    }
}
/// The device copy of a dense host matrix that `HipMatrix::to_row_major_matrix` produced.
/// The guard keeps the allocation alive (hand it to the tree that is built over it).
pub struct ResidentWords(Arc<DeviceBuf>);
impl ResidentWords {
    pub fn device_ptr(&self) -> *const u32 {
        self.0.ptr as *const u32
    }
    /// Debug builds: the device words equal the matrix's, element by element (one extra download per commitment).
    pub fn debug_same_words<F: Field, M: Matrix<F>>(&self, m: &M) -> bool {
        let n = m.height() * m.width();
        let mut dev = vec![0u32; n];
        if unsafe { p3hip_download(dev.as_mut_ptr() as *mut c_void, self.0.ptr, n * 4) } != 0 {
            return false;
        }
        let w = m.width().max(1);
        (0..n).all(|i| {
            let e: F = unsafe { m.get_unchecked(i / w, i % w) };
            dev[i] == unsafe { core::mem::transmute_copy::<F, u32>(&e) }
        })
    }
}
/// Takes the entries of ALL the queried matrices, or none: `None` unless every query has an entry with the same address, length
/// and fingerprint (and no two queries name the same entry).
pub fn take_residents(all: &[ResidentQuery]) -> Option<Vec<ResidentWords>> {
    let mut q = registry().lock().ok()?;
    let mut idx = Vec::with_capacity(all.len());
    for want in all {
        let i = q.iter().position(|r| r.host_addr == want.host_addr && r.len == want.len && r.fingerprint == want.fingerprint)?;
        if idx.contains(&i) {
            return None;
        }
        idx.push(i);
    }
    let out: Vec<ResidentWords> = idx.iter().map(|&i| ResidentWords(q[i].buf.clone())).collect();
    let mut sorted = idx;
    sorted.sort_unstable_by(|a, b| b.cmp(a)); // highest index first: the others keep their places
    for i in sorted {
        q.remove(i);
    }
    Some(out)
}

// ---- what GpuDft returns: host matrices from the cpu / vulkan / metal / webgpu arms, device matrices from the hip arm ----
#[derive(Clone, Debug)]
pub enum GpuEvaluations<F> {
    Host(RowMajorMatrix<F>),
    Device(HipMatrix<F>),
}
impl<F: Field> Matrix<F> for GpuEvaluations<F> {
    fn width(&self) -> usize {
        match self {
            GpuEvaluations::Host(m) => m.width(),
            GpuEvaluations::Device(m) => m.width(),
        }
    }
    fn height(&self) -> usize {
        match self {
            GpuEvaluations::Host(m) => m.height(),
            GpuEvaluations::Device(m) => m.height(),
        }
    }
    unsafe fn get_unchecked(&self, r: usize, c: usize) -> F {
        match self {
            GpuEvaluations::Host(m) => m.get_unchecked(r, c),
            GpuEvaluations::Device(m) => m.get_unchecked(r, c),
        }
    }
    unsafe fn row_subseq_unchecked(&self, r: usize, start: usize, end: usize) -> impl IntoIterator<Item = F, IntoIter = impl Iterator<Item = F> + Send + Sync> {
        // one concrete iterator type for both arms
        let v: Vec<F> = match self {
            GpuEvaluations::Host(m) => m.row_subseq_unchecked(r, start, end).into_iter().collect(),
            GpuEvaluations::Device(m) => m.row_subseq_unchecked(r, start, end).into_iter().collect(),
        };
        v
    }
    fn to_row_major_matrix(self) -> RowMajorMatrix<F>
    where
        Self: Sized,
        F: Clone,
    {
        match self {
            GpuEvaluations::Host(m) => m,
            GpuEvaluations::Device(m) => m.to_row_major_matrix(),
        }
    }
}
/// Host arm: Plonky3's own lazy view, densified (one pass on the host, as `RowMajorMatrix::bit_reverse_rows` users pay today);
/// device arm: a relabelling.
impl<F: Field> BitReversibleMatrix<F> for GpuEvaluations<F> {
    type BitRev = GpuEvaluations<F>;
    fn bit_reverse_rows(self) -> Self::BitRev {
        match self {
            GpuEvaluations::Host(m) => GpuEvaluations::Host(m.bit_reverse_rows().to_row_major_matrix()),
            GpuEvaluations::Device(m) => GpuEvaluations::Device(m.bit_reverse_rows()),
        }
    }
}

/// `TwoAdicFriPcs::commit` calls `.bit_reverse_rows()` on every LDE before `mmcs.commit` (p3-fri two_adic_pcs.rs,
/// [UPSTREAM-RECALL]).  For a device matrix that is a relabelling: the SAME words, the other `RowOrder` — no pass over
/// the data, no copy.  (`GpuDft::coset_lde_batch` on the hip arm asks the kernel for bit-reversed STORAGE and returns the
/// matrix labelled `BitReversed`, i.e. presenting natural order as the trait demands; the PCS's `bit_reverse_rows()` then
/// yields the natural labelling of the bit-reversed rows, which is exactly the storage order.)
impl<F: Field> BitReversibleMatrix<F> for HipMatrix<F> {
    type BitRev = HipMatrix<F>;
    fn bit_reverse_rows(self) -> Self::BitRev {
        let order = match self.order {
            RowOrder::Natural => RowOrder::BitReversed,
            RowOrder::BitReversed => RowOrder::Natural,
        };
        // the cached host copy (if any) is of the stored words and stays valid under either labelling
        HipMatrix { order, ..self }
    }
}

impl<F: Field> HipMatrix<F> {
    /// A copy whose STORAGE is in the order the type presents (natural labelling): one device pass
    /// (`p3hip_bit_reverse_rows_dev`), for consumers that need dense natural rows in HBM.
    pub fn materialize_natural(&self) -> Result<Self, String> {
        if self.order == RowOrder::Natural {
            return Ok(self.clone());
        }
        let out = Self::alloc(self.height, self.width, RowOrder::Natural)?;
        let rc = unsafe { p3hip_bit_reverse_rows_dev(self.device_ptr(), out.device_ptr_mut(), self.height, self.width, core::ptr::null_mut()) };
        if rc != 0 {
            return Err(last_error());
        }
        Ok(out)
    }
}
