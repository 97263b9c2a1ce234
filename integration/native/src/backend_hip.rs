// NEVER COMPILED HERE (no Rust toolchain in the build image; p3-* 0.4.2 path dependencies absent).
//
// native/src/backend_hip.rs — the MI355X backend module of the reference's selector.  Same shape as the stub
// backends (native/src/backend_metal.rs:5-10) and as backend_vulkan::dft_batch (native/src/backend_vulkan.rs:1988-2063):
//     pub fn dft_batch<F: TwoAdicField>(cpu: &Radix2DitParallel<F>, mat: RowMajorMatrix<F>) -> Result<RowMajorMatrix<F>, String>
// Err(String) makes GpuDft fall back to Plonky3's CPU DFT (native/src/gpu_dft.rs:100-112); libp3hip itself has no fallback.
use core::any::TypeId;
use core::ffi::c_char;
use std::ffi::CStr;

use p3_baby_bear::BabyBear;
use p3_dft::Radix2DitParallel;
use p3_field::TwoAdicField;
use p3_matrix::dense::RowMajorMatrix;
use p3_matrix::Matrix;

// include/p3hip.h
extern "C" {
    fn p3hip_is_available(msg: *mut c_char, cap: usize) -> i32;
    fn p3hip_take_last_error() -> *const c_char;
    fn p3hip_dft_batch_bb31(input: *const u32, out: *mut u32, height: usize, width: usize) -> i32;
    fn p3hip_idft_batch_bb31(input: *const u32, out: *mut u32, height: usize, width: usize) -> i32;
    fn p3hip_coset_dft_batch_bb31(input: *const u32, out: *mut u32, height: usize, width: usize, shift_monty: u32) -> i32;
    fn p3hip_coset_lde_batch_bb31(
        input: *const u32,
        out: *mut u32,
        height: usize,
        width: usize,
        added_bits: u32,
        shift_monty: u32,
        bit_reversed_out: i32,
    ) -> i32;
    fn p3hip_coset_lde_batch_bb31_dev(
        d_in: *const u32,
        d_out: *mut u32,
        height: usize,
        width: usize,
        added_bits: u32,
        shift_monty: u32,
        bit_reversed_out: i32,
        stream: *mut core::ffi::c_void,
    ) -> i32;
}

fn last_error() -> String {
    unsafe {
        let p = p3hip_take_last_error();
        if p.is_null() {
            "hip backend error".to_string()
        } else {
            CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}

/// backend_vulkan::is_vulkan_available (native/src/backend_vulkan.rs:726-731) for the hip backend.
pub fn is_hip_available() -> Result<String, String> {
    let mut buf = [0 as c_char; 256];
    let rc = unsafe { p3hip_is_available(buf.as_mut_ptr(), buf.len()) };
    let text = unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();
    if rc == 0 {
        Ok(text)
    } else {
        let _ = last_error();
        Err(text)
    }
}

fn require_babybear<F: 'static>() -> Result<(), String> {
    // native/src/backend_vulkan.rs:1999-2001
    if TypeId::of::<F>() != TypeId::of::<BabyBear>() {
        return Err("hip backend currently only supports BabyBear".to_string());
    }
    Ok(())
}

/// BabyBear is `#[repr(transparent)]` over its Montgomery `u32` — the word `to_unique_u32` yields
/// (native/src/backend_vulkan.rs:2002-2005) — so the matrix is handed over as it lies in memory and no per-element
/// map is needed in either direction (the reference does two: backend_vulkan.rs:2002-2005 and :2021-2025).
fn call<F: TwoAdicField>(
    mat: &RowMajorMatrix<F>,
    out_rows: usize,
    f: impl FnOnce(*const u32, *mut u32, usize, usize) -> i32,
) -> Result<RowMajorMatrix<F>, String> {
    require_babybear::<F>()?;
    let (h, w) = (mat.height(), mat.width());
    if !h.is_power_of_two() {
        return Err(format!("hip backend requires power-of-two height, got {h}")); // backend_vulkan.rs:1992-1995
    }
    let mut out: Vec<F> = Vec::with_capacity(out_rows * w);
    let rc = f(mat.values.as_ptr() as *const u32, out.as_mut_ptr() as *mut u32, h, w);
    if rc != 0 {
        return Err(last_error());
    }
    unsafe { out.set_len(out_rows * w) };
    Ok(RowMajorMatrix::new(out, w))
}

pub fn dft_batch<F: TwoAdicField + Ord>(cpu: &Radix2DitParallel<F>, mat: RowMajorMatrix<F>) -> Result<RowMajorMatrix<F>, String> {
    let h = mat.height();
    // the reference's debug self-check (native/src/backend_vulkan.rs:2008-2057): debug builds keep the prover's live input,
    // run Plonky3's CPU DFT beside the device and compare — tolerant, like the reference, of a result whose ROWS are
    // bit-reversed (backend_vulkan.rs:2034-2050) — and turn a mismatch into Err, i.e. CPU fallback + mailbox message
    #[cfg(debug_assertions)]
    let input = mat.clone();
    let out = call(&mat, h, |i, o, h, w| unsafe { p3hip_dft_batch_bb31(i, o, h, w) })?;
    #[cfg(debug_assertions)]
    {
        use p3_dft::TwoAdicSubgroupDft;
        use p3_matrix::bitrev::BitReversibleMatrix;
        let expected = cpu.dft_batch(input).to_row_major_matrix();
        if out.values != expected.values {
            let rev = expected.clone().bit_reverse_rows().to_row_major_matrix();
            let first = out.values.iter().zip(expected.values.iter()).position(|(a, b)| a != b).unwrap_or(0);
            return Err(if out.values == rev.values {
                format!("hip dft_batch: output rows are bit-reversed relative to Radix2DitParallel (h={h}, w={})", out.width())
            } else {
                format!("hip dft_batch: mismatch against Radix2DitParallel at word {first} (h={h}, w={})", out.width())
            });
        }
    }
    #[cfg(not(debug_assertions))]
    let _ = cpu;
    Ok(out)
}

pub fn idft_batch<F: TwoAdicField>(mat: RowMajorMatrix<F>) -> Result<RowMajorMatrix<F>, String> {
    let h = mat.height();
    call(&mat, h, |i, o, h, w| unsafe { p3hip_idft_batch_bb31(i, o, h, w) })
}

pub fn coset_dft_batch<F: TwoAdicField>(mat: RowMajorMatrix<F>, shift: F) -> Result<RowMajorMatrix<F>, String> {
    require_babybear::<F>()?;
    let h = mat.height();
    let s = unsafe { *(&shift as *const F as *const u32) };
    call(&mat, h, |i, o, h, w| unsafe { p3hip_coset_dft_batch_bb31(i, o, h, w, s) })
}

/// TwoAdicSubgroupDft::coset_lde_batch in ONE device round trip (the provided method costs two dft_batch calls plus
/// three CPU passes).  Natural row order out, as the trait specifies; TwoAdicFriPcs::commit
/// applies `.bit_reverse_rows()` itself.
pub fn coset_lde_batch<F: TwoAdicField>(mat: RowMajorMatrix<F>, added_bits: usize, shift: F) -> Result<RowMajorMatrix<F>, String> {
    require_babybear::<F>()?;
    let h = mat.height();
    let s = unsafe { *(&shift as *const F as *const u32) };
    call(&mat, h << added_bits, |i, o, h, w| unsafe { p3hip_coset_lde_batch_bb31(i, o, h, w, added_bits as u32, s, 0) })
}

/// The same LDE with the result LEFT IN HBM (hip_matrix.rs): upload the evaluations, run the transform with bit-reversed STORAGE
/// (what `TwoAdicFriPcs::commit` commits), return a `HipMatrix` labelled `BitReversed`, i.e. presenting the natural order the
/// trait specifies.  The PCS's `.bit_reverse_rows()` is then a relabelling, and `.to_row_major_matrix()` is the one download of
/// the trait path, which registers the device copy for `HipMmcs::commit`.
pub fn coset_lde_batch_resident<F: TwoAdicField>(
    mat: RowMajorMatrix<F>,
    added_bits: usize,
    shift: F,
) -> Result<crate::hip_matrix::HipMatrix<F>, String> {
    use crate::hip_matrix::{HipMatrix, RowOrder};
    require_babybear::<F>()?;
    let (h, w) = (mat.height(), mat.width());
    if !h.is_power_of_two() {
        return Err(format!("hip backend requires power-of-two height, got {h}"));
    }
    let s = unsafe { *(&shift as *const F as *const u32) };
    let input = HipMatrix::from_host(&mat)?;
    let out = HipMatrix::<F>::alloc(h << added_bits, w, RowOrder::BitReversed)?;
    let rc = unsafe { p3hip_coset_lde_batch_bb31_dev(input.device_ptr(), out.device_ptr_mut(), h, w, added_bits as u32, s, 1, core::ptr::null_mut()) };
    if rc != 0 {
        return Err(last_error());
    }
    Ok(out)
}
