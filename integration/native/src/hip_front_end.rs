// NEVER COMPILED HERE (no Rust toolchain in the build image; p3-* 0.4.2 path dependencies absent).
//
// native/src/hip_front_end.rs — what native/src/fib_air.rs calls when the selector says "hip" (fib_air.rs.patch):
// run_fib_air_zk and run_dft_benchmark as ONE call each into libp3hip, returning the same Result<String, String> the
// reference's functions return (fib_air.rs:27,98), so lib.rs:37-131 needs no change.
use core::ffi::{c_char, c_void};

use p3_baby_bear::BabyBear;
use p3_dft::TwoAdicSubgroupDft;
use p3_matrix::dense::RowMajorMatrix;

use crate::gpu_dft::{BackendKind, GpuDft};

type Val = BabyBear;

/// include/p3hip.h `p3hip_cpu_dft_fn`
type CpuDftFn = unsafe extern "C" fn(*mut c_void, *const u32, *mut u32, usize, usize) -> i32;

// include/p3hip.h, "report-returning entry points"
extern "C" {
    fn p3hip_run_fib_air_zk(out: *mut c_char, cap: usize) -> i32;
    fn p3hip_run_dft_benchmark(cpu_dft: Option<CpuDftFn>, user: *mut c_void, out: *mut c_char, cap: usize) -> i32;
}

fn text_of(buf: &[c_char]) -> String {
    unsafe { std::ffi::CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned()
}

/// The reference reports failures as `Err(String)` and lib.rs prefixes them ("fib_air zk failed: {err}", lib.rs:48): strip
/// the prefix libp3hip already wrote so the Java side sees the same text either way.
fn into_result(text: String, prefix: &str) -> Result<String, String> {
    match text.strip_prefix(prefix) {
        Some(err) => Err(err.to_string()),
        // an otherwise-ok report may carry a trailing "\nHIP error: .." (a message left in the mailbox by a call that then fell
        // back): it stays part of the Ok text, exactly as lib.rs:60-75 appends "\nVulkan error: .." to a SUCCESSFUL report
        None => Ok(text),
    }
}

pub fn run_fib_air_zk_hip() -> Result<String, String> {
    let mut buf = vec![0 as c_char; 1024];
    unsafe { p3hip_run_fib_air_zk(buf.as_mut_ptr(), buf.len()) };
    into_result(text_of(&buf), "fib_air zk failed: ")
}

/// CPU column of the benchmark: Plonky3's Radix2DitParallel on the words libp3hip hands over (BabyBear is
/// `#[repr(transparent)]` over its Montgomery u32, backend_hip.rs).
unsafe extern "C" fn cpu_dft_cb(user: *mut c_void, input: *const u32, out: *mut u32, height: usize, width: usize) -> i32 {
    let cpu = &*(user as *const GpuDft<Val>);
    let src = core::slice::from_raw_parts(input as *const Val, height * width);
    // GpuDft holds no interior state a panic could leave half-updated; the closure borrows it and a fresh Vec
    let res = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| cpu.dft_batch(RowMajorMatrix::new(src.to_vec(), width)).to_row_major_matrix()));
    match res {
        Ok(m) => {
            core::ptr::copy_nonoverlapping(m.values.as_ptr() as *const u32, out, height * width);
            0
        }
        Err(_) => 1,
    }
}

pub fn run_dft_benchmark_hip() -> Result<String, String> {
    let cpu = GpuDft::<Val>::with_backend(BackendKind::Cpu);
    let mut buf = vec![0 as c_char; 1 << 14];
    unsafe {
        p3hip_run_dft_benchmark(
            Some(cpu_dft_cb),
            &cpu as *const GpuDft<Val> as *mut c_void,
            buf.as_mut_ptr(),
            buf.len(),
        )
    };
    into_result(text_of(&buf), "dft benchmark failed: ")
}
