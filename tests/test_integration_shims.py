"""integration/ holds the Rust side of the drop-in (never compiled here: no Rust toolchain).  What CAN be checked
offline: every `extern "C"` function the shims declare exists in include/p3hip.h with the same number of parameters,
every file carries the "NEVER COMPILED HERE" banner, and the selector patch names the reference lines it changes."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTEG = os.path.join(ROOT, "integration")


def _header_functions():
    text = open(os.path.join(ROOT, "include", "p3hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    funcs = {}
    for m in re.finditer(r"\b(p3hip_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        funcs[m.group(1)] = n
    return funcs


def _rust_externs(path):
    text = open(path).read()
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S):
        for m in re.finditer(r"fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", blk.group(1), flags=re.S):
            args = m.group(2).strip().rstrip(",")
            n = 0 if not args else len([a for a in args.split(",") if a.strip()])
            out[m.group(1)] = n
    return out


def test_rust_shims_bind_the_declared_c_abi():
    header = _header_functions()
    assert "p3hip_dft_batch_bb31" in header and header["p3hip_coset_lde_batch_bb31"] == 7
    seen = set()
    for dirpath, _, files in os.walk(INTEG):
        for f in files:
            if not f.endswith(".rs"):
                continue
            for name, n in _rust_externs(os.path.join(dirpath, f)).items():
                assert name in header, "%s: %s is not declared in include/p3hip.h" % (f, name)
                assert header[name] == n, "%s: %s takes %d parameters in include/p3hip.h, %d in the shim" % (f, name, header[name], n)
                seen.add(name)
    # the shims cover the boundary the reference needs: the DFT entry, the LDE, the MMCS and the diagnostics
    for must in ("p3hip_dft_batch_bb31", "p3hip_coset_lde_batch_bb31", "p3hip_mmcs_commit_hash", "p3hip_mmcs_open_batch",
                 "p3hip_mmcs_free", "p3hip_take_last_error", "p3hip_is_available", "p3hip_run_fib_air_zk", "p3hip_run_dft_benchmark",
                 # round 4: the LDE left in HBM and found again by the commit (hip_matrix.rs, backend_hip.rs, hip_mmcs.rs)
                 "p3hip_coset_lde_batch_bb31_dev", "p3hip_mmcs_commit_hash_dev", "p3hip_download", "p3hip_upload", "p3hip_malloc",
                 "p3hip_free", "p3hip_bit_reverse_rows_dev"):
        assert must in seen, must


def test_shim_files_say_they_were_never_compiled():
    n = 0
    for dirpath, _, files in os.walk(INTEG):
        for f in files:
            if f.endswith((".rs", ".patch")):
                assert "NEVER COMPILED HERE" in open(os.path.join(dirpath, f)).read(), f
                n += 1
    assert n >= 5


def test_selector_patch_follows_the_reference_enum():
    patch = open(os.path.join(INTEG, "native", "src", "gpu_dft.rs.patch")).read()
    for needle in ("BackendKind::Hip => 4", '"hip" => BackendKind::Hip', "4 => BackendKind::Hip", "backend_hip::dft_batch",
                   "LAST_VULKAN_ERROR", "self.cpu.dft_batch(mat).to_row_major_matrix()"):
        assert needle in patch, needle
    hdr = open(os.path.join(ROOT, "include", "p3hip.h")).read()
    assert "#define P3HIP_BACKEND_HIP 4" in hdr


def test_fib_air_patch_makes_the_prover_honour_the_selector():
    """Round 2's shim set left native/src/fib_air.rs untouched: line 60 hard-codes Vulkan, so setBackend("hip") +
    runFibAirZk() still ran Vulkan.  The patch must remove exactly that line, route both report functions through
    hip_front_end.rs when the selector says hip, and offer the hiding MMCS swap."""
    patch = open(os.path.join(INTEG, "native", "src", "fib_air.rs.patch")).read()
    ref_line = "let dft = GpuDft::<Val>::with_backend(BackendKind::Vulkan);"
    assert "-    " + ref_line in patch and "+    let dft = GpuDft::<Val>::default();" in patch
    for needle in ("run_fib_air_zk_hip()", "run_dft_benchmark_hip()", "BackendKind::Hip", "HipKeccakHidingMmcs<FieldHash, MyCompress>",
                   "ValHidingMmcs::new(field_hash, compress, 1)", "PCIe-bound"):
        assert needle in patch, needle
    front = open(os.path.join(INTEG, "native", "src", "hip_front_end.rs")).read()
    for needle in ("pub fn run_fib_air_zk_hip() -> Result<String, String>", "pub fn run_dft_benchmark_hip() -> Result<String, String>",
                   "Radix2DitParallel", 'strip_prefix(prefix)'):
        assert needle in front, needle
    assert "mod hip_front_end;" in open(os.path.join(INTEG, "native", "src", "lib.rs.patch")).read()


def test_trait_path_keeps_the_lde_resident_and_carries_the_debug_self_check():
    """Round-3 review, item 8: `GpuDft::Evaluations` on the hip arm is a device matrix, the PCS's download registers the device copy,
    `HipMmcs::commit` takes it instead of uploading; and backend_hip.rs has the reference's debug compare (backend_vulkan.rs:2008-2057)."""
    src = os.path.join(INTEG, "native", "src")
    mat = open(os.path.join(src, "hip_matrix.rs")).read()
    for needle in ("pub struct HipMatrix<F>", "impl<F: Field> Matrix<F> for HipMatrix<F>", "impl<F: Field> BitReversibleMatrix<F> for HipMatrix<F>",
                   "pub enum GpuEvaluations<F>", "pub fn take_residents(", "register_resident(", "RowOrder::BitReversed", "p3hip_download(",
                   # round-4 advisor finding: a registry hit is verified (fingerprint through the Matrix trait, full compare in debug builds)
                   "pub fn fingerprint_of<", "r.fingerprint == want.fingerprint", "pub fn debug_same_words<"):
        assert needle in mat, needle
    patch = open(os.path.join(src, "gpu_dft.rs.patch")).read()
    assert "type Evaluations = GpuEvaluations<F>;" in patch and "-    type Evaluations = RowMajorMatrix<F>;" in patch
    assert "backend_hip::coset_lde_batch_resident(" in patch and "GpuEvaluations::Device(result)" in patch
    mmcs = open(os.path.join(src, "hip_mmcs.rs")).read()
    assert "crate::hip_matrix::take_residents(" in mmcs and "p3hip_mmcs_commit_hash_dev(" in mmcs
    assert "crate::hip_matrix::fingerprint_of(m)" in mmcs and "debug_same_words(m)" in mmcs  # entries leave the registry only when ALL inputs matched
    hip = open(os.path.join(src, "backend_hip.rs")).read()
    assert "#[cfg(debug_assertions)]" in hip and "cpu.dft_batch(input)" in hip and "bit_reverse_rows()" in hip
    assert "p3hip_coset_lde_batch_bb31_dev(" in hip and "RowOrder::BitReversed" in hip
    assert "mod hip_matrix;" in open(os.path.join(src, "lib.rs.patch")).read()
